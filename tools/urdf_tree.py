"""Minimal URDF reader: links, joints, and Bullet's joint/link index order.

Bullet numbers a URDF's links depth-first (pre-order) from the root link, visiting
child joints in the order they appear in the XML; link i's parent joint is joint i and
the root is index -1.  SURVEY.md App. D reproduces the reference notebook's joint-name
table (testing_bullet_ik.ipynb cell 2) with exactly this rule; tests/test_golden_assets.py
pins it against the committed copy of that table.
"""
import xml.etree.ElementTree as ET

import numpy as np


def _floats(s, n=3, default=0.0):
    if s is None:
        return [default] * n
    return [float(x) for x in s.split()]


def rpy_to_mat(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _origin(el):
    o = el.find('origin') if el is not None else None
    if o is None:
        return np.zeros(3), np.eye(3)
    return np.array(_floats(o.get('xyz'))), rpy_to_mat(_floats(o.get('rpy')))


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links = {}
    for l in root.findall('link'):
        name = l.get('name')
        inertial = l.find('inertial')
        info = {'name': name, 'has_inertial': inertial is not None, 'collisions': [], 'contact': {}}
        if inertial is not None:
            info['mass'] = float(inertial.find('mass').get('value'))
            info['com_xyz'], info['com_R'] = _origin(inertial)
        for c in l.findall('collision'):
            xyz, R = _origin(c)
            g = c.find('geometry')
            shape = None
            if g.find('box') is not None:
                shape = {'type': 'box', 'size': _floats(g.find('box').get('size'))}
            elif g.find('cylinder') is not None:
                shape = {'type': 'cylinder', 'radius': float(g.find('cylinder').get('radius')),
                         'length': float(g.find('cylinder').get('length'))}
            elif g.find('sphere') is not None:
                shape = {'type': 'sphere', 'radius': float(g.find('sphere').get('radius'))}
            elif g.find('mesh') is not None:
                m = g.find('mesh')
                shape = {'type': 'mesh', 'filename': m.get('filename'),
                         'scale': _floats(m.get('scale'), default=1.0) if m.get('scale') else [1.0, 1.0, 1.0]}
            shape['xyz'], shape['R'] = xyz, R
            info['collisions'].append(shape)
        ct = l.find('contact')
        if ct is not None:
            for e in ct:
                if e.get('value') is not None:
                    info['contact'][e.tag] = float(e.get('value'))
                else:
                    info['contact'][e.tag] = True
        links[name] = info
    joints = []
    for j in root.findall('joint'):
        xyz, R = _origin(j)
        ax = j.find('axis')
        lim = j.find('limit')
        joints.append({
            'name': j.get('name'), 'type': j.get('type'),
            'parent': j.find('parent').get('link'), 'child': j.find('child').get('link'),
            'xyz': xyz, 'R': R,
            'axis': np.array(_floats(ax.get('xyz'))) if ax is not None else np.array([1.0, 0, 0]),
            'lower': float(lim.get('lower', 0)) if lim is not None else 0.0,
            'upper': float(lim.get('upper', 0)) if lim is not None else 0.0,
            'effort': float(lim.get('effort', 0)) if lim is not None else 0.0,
        })
    children = {j['child'] for j in joints}
    roots = [n for n in links if n not in children]
    assert len(roots) == 1, roots
    order = []  # (joint dict, parent index)

    def visit(link_name, parent_index):
        for j in joints:
            if j['parent'] == link_name:
                idx = len(order)
                order.append((j, parent_index))
                visit(j['child'], idx)

    visit(roots[0], -1)
    return {'root': roots[0], 'links': links, 'joints_in_order': order}


BULLET_JOINT_TYPE = {'revolute': 0, 'prismatic': 1, 'fixed': 4, 'continuous': 0}


def joint_types(tree):
    return [BULLET_JOINT_TYPE[j['type']] for j, _ in tree['joints_in_order']]


def joint_names(tree):
    return [j['name'] for j, _ in tree['joints_in_order']]
