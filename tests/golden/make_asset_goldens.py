#!/usr/bin/env python3
"""Second, independent reading of the reference's model files -> tests/golden/assets_independent.json (DATA: numbers only).

    python tests/golden/make_asset_goldens.py            (in the build container: reads /root/reference, which does not travel)

tools/bake_assets.py (with tools/urdf_tree.py) turns the reference's URDFs, collision meshes and scene calls into the tables that the CPU
oracle, the frozen reference step AND the HIP library consume - a wrong frame, mass or collider there is common-mode and no parity test
can see it.  This script shares no code with the bake: its own XML walk, its own rpy convention, its own STL / OBJ vertex readers.  It
records only what the files SAY (joint origins / axes / limits in XML order, link inertials, collision geometries with the axis-aligned
bounds of their mesh vertices); everything derived (Bullet's link order, merging of fixed links, inertias from collision bounds, forward
kinematics, mass matrices) is recomputed by tests/test_bake_independent.py in numpy and compared with what the oracle holds.
"""
import json
import os
import struct
import xml.etree.ElementTree as ET

REF = '/root/reference/roboticsPlayroomPybullet/envs'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets_independent.json')


def nums(text, n, default):
    return [float(v) for v in text.split()] if text else [default] * n


def stl_vertices(path):
    raw = open(path, 'rb').read()
    ntri = struct.unpack_from('<I', raw, 80)[0] if len(raw) >= 84 else -1
    if ntri >= 0 and 84 + 50 * ntri == len(raw):                 # binary STL
        out = []
        for t in range(ntri):
            f = struct.unpack_from('<12f', raw, 84 + 50 * t)
            out += [f[3:6], f[6:9], f[9:12]]
        return out
    return [tuple(float(v) for v in ln.split()[1:4]) for ln in raw.decode('ascii', 'ignore').splitlines() if ln.strip().startswith('vertex')]


def obj_vertices(path):
    return [tuple(float(v) for v in ln.split()[1:4]) for ln in open(path, errors='ignore') if ln.startswith('v ')]


def mesh_bounds(path, scale):
    v = stl_vertices(path) if path.lower().endswith('.stl') else obj_vertices(path)
    assert v, path
    lo = [min(p[k] for p in v) * scale[k] for k in range(3)]
    hi = [max(p[k] for p in v) * scale[k] for k in range(3)]
    return [min(a, b) for a, b in zip(lo, hi)], [max(a, b) for a, b in zip(lo, hi)], len(v)


def origin_of(el):
    o = el.find('origin') if el is not None else None
    if o is None:
        return [0.0] * 3, [0.0] * 3
    return nums(o.get('xyz'), 3, 0.0), nums(o.get('rpy'), 3, 0.0)


def read_urdf(path):
    root = ET.parse(path).getroot()
    base = os.path.dirname(path)
    links, joints = {}, []
    for ln in root.findall('link'):
        rec = {'mass': None, 'inertial_xyz': [0.0] * 3, 'inertial_rpy': [0.0] * 3, 'collisions': [], 'contact': {}}
        ine = ln.find('inertial')
        if ine is not None:
            rec['mass'] = float(ine.find('mass').get('value'))
            rec['inertial_xyz'], rec['inertial_rpy'] = origin_of(ine)
        for col in ln.findall('collision'):
            xyz, rpy = origin_of(col)
            g = list(col.find('geometry'))[0]
            c = {'xyz': xyz, 'rpy': rpy, 'kind': g.tag}
            if g.tag == 'box':
                c['size'] = nums(g.get('size'), 3, 0.0)
            elif g.tag == 'cylinder':
                c['radius'], c['length'] = float(g.get('radius')), float(g.get('length'))
            elif g.tag == 'sphere':
                c['radius'] = float(g.get('radius'))
            elif g.tag == 'mesh':
                fn = g.get('filename')
                scale = nums(g.get('scale'), 3, 1.0)
                rel = fn[len('package://'):] if fn.startswith('package://') else fn      # Bullet resolves package:// against the URDF's directory
                c['file'] = os.path.basename(rel)
                c['lo'], c['hi'], c['n_vertices'] = mesh_bounds(os.path.join(base, rel), scale)
            rec['collisions'].append(c)
        ct = ln.find('contact')
        if ct is not None:
            for ch in ct:
                rec['contact'][ch.tag] = float(ch.get('value', 1.0)) if ch.get('value') is not None else 1.0
        links[ln.get('name')] = rec
    for j in root.findall('joint'):
        xyz, rpy = origin_of(j)
        ax = j.find('axis')
        lim = j.find('limit')
        joints.append({'name': j.get('name'), 'type': j.get('type'), 'parent': j.find('parent').get('link'), 'child': j.find('child').get('link'),
                       'xyz': xyz, 'rpy': rpy, 'axis': nums(ax.get('xyz'), 3, 0.0) if ax is not None else [1.0, 0.0, 0.0],
                       'lower': float(lim.get('lower', 0.0)) if lim is not None else 0.0, 'upper': float(lim.get('upper', 0.0)) if lim is not None else 0.0})
    children = {j['child'] for j in joints}
    roots = [n for n in links if n not in children]
    assert len(roots) == 1, roots
    return {'root': roots[0], 'links': links, 'joints': joints}


def main():
    out = {'ur5': read_urdf(os.path.join(REF, 'ur_e_description', 'ur5e2.urdf')),
           'panda': read_urdf(os.path.join(REF, 'franka_panda', 'panda.urdf')),
           'scene_meshes': {}}
    for name, scale in (('door.obj', 0.0015), ('drawer2.obj', 1.25)):       # scenes.py:117-182 (meshScale 0.0015), 319-333 (1.25)
        lo, hi, n = mesh_bounds(os.path.join(REF, 'env_meshes', name), [scale] * 3)
        out['scene_meshes'][name] = {'scale': scale, 'lo': lo, 'hi': hi, 'n_vertices': n}
    json.dump(out, open(OUT, 'w'), indent=0, sort_keys=True)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
