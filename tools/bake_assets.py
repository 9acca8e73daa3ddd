#!/usr/bin/env python3
"""Bake the reference's robot/scene ASSETS (URDF, STL/OBJ, scene call log) into model tables.

Inputs (data only, read in the build container):
  * /root/reference/.../ur5e2.urdf, panda.urdf and their collision meshes   
  * tests/golden/scenes.json  — the reference's scenes.py executed against the recording fake
    (SURVEY.md App. C), so scene geometry here is *captured from the reference*, not retyped.
  * env_meshes/drawer2.obj, door.obj — concave trimeshes, decomposed EXACTLY into boxes
    (every face is axis-aligned); the decomposition is verified against the mesh vertices below.

Outputs (committed; they are what the GPU box receives):
  * roboticsplayroompybullet_amd/assets/models.json         human/py readable
  * roboticsplayroompybullet_amd/csrc/generated/rp_models_gen.h   C tables for oracle/ and the HIP library

Modelling hypotheses about Bullet (cannot be checked here, PyBullet absent; DESIGN.md §H lists them):
  H1 link index order = DFS pre-order, children in XML joint order           (pinned by NB cell 2)
  H2 links without <inertial>: mass 1, identity inertial frame
  H3 URDF <inertia> ignored; inertia recomputed from the collision shape AABB in the inertial
     frame (box formula; hull margin 0.001 counted as Bullet does); no collision shape => 0
  H4 fixed joints are rigid attachments: merged into the parent movable link (exactly equivalent)
  H5 concave-trimesh free body (drawer) has zero inertia => no angular response (translates only)
  H6 mesh colliders are approximated by the oriented bounding box of their convex hull
     (Bullet uses the hull itself through GJK); cylinders by the equal-area square prism.
"""
import json
import os
import struct
import sys

import numpy as np
from scipy.spatial import ConvexHull

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import urdf_tree  # noqa: E402

ENVS = '/root/reference/roboticsPlayroomPybullet/envs'
HULL_MARGIN = 0.001      # gUrdfDefaultCollisionMargin (H3)


# ------------------------------------------------------------------ mesh io
def load_stl(path):
    data = open(path, 'rb').read()
    n = struct.unpack('<I', data[80:84])[0]
    if 84 + 50 * n == len(data):
        arr = np.frombuffer(data, dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', 9), ('a', '<u2')]), count=n, offset=84)
        return arr['v'].reshape(-1, 3).astype(np.float64)
    verts = []
    for line in data.decode('ascii', 'ignore').splitlines():
        t = line.split()
        if t and t[0] == 'vertex':
            verts.append([float(x) for x in t[1:4]])
    return np.array(verts)


def load_obj(path):
    verts = []
    for line in open(path):
        t = line.split()
        if t and t[0] == 'v':
            verts.append([float(x) for x in t[1:4]])
    return np.array(verts)


def load_mesh(path):
    return load_stl(path) if path.lower().endswith('.stl') else load_obj(path)


def hull_points(P):
    return P[ConvexHull(P).vertices]


def obb_of(P):
    """Smallest of {frame-aligned box, PCA-aligned box} around points P. Returns center, R (cols = axes), he."""
    best = None
    C = np.cov((P - P.mean(0)).T)
    _, vecs = np.linalg.eigh(C)
    if np.linalg.det(vecs) < 0:
        vecs[:, 2] *= -1
    for R in (np.eye(3), vecs):
        L = P @ R
        lo, hi = L.min(0), L.max(0)
        he = 0.5 * (hi - lo)
        c = R @ (0.5 * (hi + lo))
        vol = np.prod(he)
        if best is None or vol < best[0] * 0.97:      # prefer the link-aligned box unless PCA is clearly tighter
            best = (vol, c, R.copy(), he)
    return best[1], best[2], best[3]


# ------------------------------------------------------------------ URDF -> merged arm model
def resolve_mesh(urdf_dir, fn):
    fn = fn.replace('package://', '')
    return os.path.join(urdf_dir, fn)


def link_colliders_and_inertia(link, urdf_dir):
    """Per-URDF-link: collider boxes in the LINK frame and the (H3) inertia diag in the inertial frame."""
    cols = []
    com = link.get('com_xyz', np.zeros(3)) if link['has_inertial'] else np.zeros(3)
    Rc = link.get('com_R', np.eye(3)) if link['has_inertial'] else np.eye(3)
    mass = link['mass'] if link['has_inertial'] else 1.0          # H2
    lo = np.full(3, np.inf)
    hi = np.full(3, -np.inf)
    single_identity = False
    friction = link['contact'].get('lateral_friction', 0.5)
    for c in link['collisions']:
        if c['type'] == 'mesh':
            P = hull_points(load_mesh(resolve_mesh(urdf_dir, c['filename'])) * np.array(c['scale']))
            ctr, R, he = obb_of(P)
            pts = P
            margin = HULL_MARGIN
        elif c['type'] == 'box':
            he = 0.5 * np.array(c['size'])
            ctr, R = np.zeros(3), np.eye(3)
            pts = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * he
            margin = 0.0
        elif c['type'] == 'cylinder':
            s = c['radius'] * np.sqrt(np.pi) / 2.0                  # H6 equal-area square prism
            he = np.array([s, s, 0.5 * c['length']])
            ctr, R = np.zeros(3), np.eye(3)
            r, h = c['radius'], 0.5 * c['length']
            pts = np.array([[sx * r, sy * r, sz * h] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
            margin = 0.0
        else:
            raise NotImplementedError(c['type'])
        # collider pose in link frame; the exact shape rides along for the frozen Bullet-like oracle (oracle/rp_bullet_ref.c):
        # hull vertices (link frame) for meshes, radius / half length for cylinders (the collider frame is the cylinder's frame)
        extra = {}
        if c['type'] == 'mesh':
            extra['hull'] = (c['R'] @ P.T).T + c['xyz']
        elif c['type'] == 'cylinder':
            extra['cyl'] = (c['radius'], 0.5 * c['length'])
        cols.append(dict({'type': 'box', 'he': he, 'pos': c['xyz'] + c['R'] @ ctr, 'rot': c['R'] @ R, 'friction': friction,
                          'contact': dict(link['contact'])}, **extra))
        # AABB of the shape in the inertial frame (H3)
        Pl = (c['R'] @ pts.T).T + c['xyz']
        Pi = (Rc.T @ (Pl - com).T).T
        lo = np.minimum(lo, Pi.min(0) - margin)
        hi = np.maximum(hi, Pi.max(0) + margin)
        single_identity = (len(link['collisions']) == 1 and np.allclose(c['xyz'], com) and np.allclose(c['R'], Rc))
        last_margin = margin
    # btCollisionShape::getAngularMotionDisc of the link's collision shape (children in the inertial frame): |AABB centre| + half
    # diagonal; times gContactBreakingThreshold = 0.02 it is the pair's contact breaking threshold (relative thresholds are the
    # dispatcher's default: CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD)
    disc = float(np.linalg.norm(0.5 * (lo + hi)) + 0.5 * np.linalg.norm(hi - lo)) if link['collisions'] else 0.0
    for cc in cols:
        cc['disc'] = disc
    if not link['collisions'] or mass == 0.0:
        idiag = np.zeros(3)
    else:
        ext = hi - lo
        if single_identity:
            ext = ext + 2 * last_margin     # btPolyhedralConvexShape adds the margin once more
        idiag = mass / 12.0 * np.array([ext[1] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[1] ** 2])
    return cols, mass, com, Rc, idiag


def build_arm(urdf_path):
    tree = urdf_tree.parse_urdf(urdf_path)
    urdf_dir = os.path.dirname(urdf_path)
    order = tree['joints_in_order']
    links = tree['links']
    n = len(order)
    # frame of every link relative to its nearest movable ancestor body
    body_of = [None] * n          # movable body index owning link i
    T_rel = [None] * n            # (R, p): link frame in owning body's frame
    bodies = []                   # movable bodies
    base = {'cols': []}
    cols0, _, _, _, _ = link_colliders_and_inertia(links[tree['root']], urdf_dir)
    base['cols'] = cols0
    for i, (j, parent) in enumerate(order):
        if parent < 0:
            pb, (Rp, pp) = -1, (np.eye(3), np.zeros(3))
        else:
            pb, (Rp, pp) = body_of[parent], T_rel[parent]
        R_j, p_j = Rp @ j['R'], pp + Rp @ j['xyz']      # joint frame in parent *body* frame
        if j['type'] in ('revolute', 'prismatic', 'continuous'):
            b = len(bodies)
            ax = j['axis'] / np.linalg.norm(j['axis'])
            bodies.append({'parent': pb, 'jtype': 0 if j['type'] != 'prismatic' else 1, 'jpos': p_j, 'jrot': R_j,
                           'axis': ax, 'lower': j['lower'], 'upper': j['upper'], 'bullet_index': i,
                           'name': j['name'], 'parts': [], 'cols': []})
            body_of[i], T_rel[i] = b, (np.eye(3), np.zeros(3))
        else:
            body_of[i], T_rel[i] = pb, (R_j, p_j)
        link = links[j['child']]
        cols, mass, com, Rc, idiag = link_colliders_and_inertia(link, urdf_dir)
        R_l, p_l = T_rel[i]
        tgt = bodies[body_of[i]] if body_of[i] >= 0 else base
        for c in cols:
            tc = {'type': 'box', 'he': c['he'], 'pos': p_l + R_l @ c['pos'], 'rot': R_l @ c['rot'],
                  'friction': c['friction'], 'link': i, 'disc': c['disc'], 'contact': c['contact']}
            if 'hull' in c:
                tc['hull'] = (R_l @ c['hull'].T).T + p_l
            if 'cyl' in c:
                tc['cyl'] = c['cyl']
            tgt['cols'].append(tc)
        if body_of[i] >= 0 and mass > 0:
            Rw = R_l @ Rc
            tgt['parts'].append({'mass': mass, 'com': p_l + R_l @ com, 'I': Rw @ np.diag(idiag) @ Rw.T})
    for b in bodies:
        m = sum(p['mass'] for p in b['parts'])
        com = sum(p['mass'] * p['com'] for p in b['parts']) / m
        I = np.zeros((3, 3))
        for p in b['parts']:
            d = p['com'] - com
            I += p['I'] + p['mass'] * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        b['mass'], b['com'], b['inertia'] = m, com, I
        del b['parts']

    def site(link_index):
        j, _ = order[link_index]
        link = links[j['child']]
        R_l, p_l = T_rel[link_index]
        if link['has_inertial']:
            return body_of[link_index], p_l + R_l @ link['com_xyz'], R_l @ link['com_R']
        return body_of[link_index], p_l, R_l

    return {'bodies': bodies, 'base_cols': base['cols'], 'site': site, 'n_links': n,
            'names': urdf_tree.joint_names(tree)}


# ------------------------------------------------------------------ scene meshes -> boxes
def box_from_bounds(lo, hi):
    lo, hi = np.array(lo, float), np.array(hi, float)
    return {'he': 0.5 * (hi - lo), 'pos': 0.5 * (hi + lo), 'rot': np.eye(3)}


def drawer_boxes(scale):
    """Exact solid decomposition of env_meshes/drawer2.obj (units of the OBJ, then * scale)."""
    X0, X1, Y0, Y1, Z0, Z1 = -0.0994058, 0.0505942, -0.065, 0.0, -0.1, 0.185
    cx0, cx1, cyf, cz0, cz1 = -0.0863326, 0.0370187, -0.045, -0.0885405, 0.0858919     # main cavity (open at y=0)
    hx0, hx1, hz0, hz1 = -0.0841239, 0.035037, 0.0985757, 0.165435                    # handle through-hole
    B = [
        ((X0, Y0, Z0), (X1, cyf, hz0)),          # floor slab under the cavity, back wall to handle slab
        ((X0, cyf, Z0), (X1, Y1, cz0)),          # back wall above floor level
        ((X0, cyf, cz0), (cx0, Y1, cz1)),        # left wall of cavity
        ((cx1, cyf, cz0), (X1, Y1, cz1)),        # right wall of cavity
        ((X0, cyf, cz1), (X1, Y1, hz0)),         # wall between cavity and handle hole
        ((X0, Y0, hz0), (hx0, Y1, hz1)),         # left of handle hole
        ((hx1, Y0, hz0), (X1, Y1, hz1)),         # right of handle hole
        ((X0, Y0, hz1), (X1, Y1, Z1)),           # front bar of the handle
        ((X1, -0.06, -0.09), (0.0905942, -0.01, -0.04)),     # right rail
        ((-0.1394058, -0.06, -0.09), (X0, -0.01, -0.04)),    # left rail
    ]
    V = load_obj(os.path.join(ENVS, 'env_meshes', 'drawer2.obj'))
    _check_on_surface(V, B, 'drawer2.obj')
    return [box_from_bounds(np.array(lo) * scale, np.array(hi) * scale) for lo, hi in B]


def door_boxes(scale):
    """Exact solid decomposition of env_meshes/door.obj: a slab plus a rectangular loop handle."""
    B = [
        ((-99.4058, -5.0, -100.0), (50.5942, 0.0, 100.0)),            # slab
        ((-58.636, -55.0, -5.0), (-50.9872, -5.0, 5.0)),              # handle left bar
        ((15.1423, -55.0, -5.0), (21.364, -5.0, 5.0)),                # handle right bar
        ((-50.9872, -55.0, -5.0), (15.1423, -48.5251, 5.0)),          # handle far bar
        ((-50.9872, -10.8335, -5.0), (15.1423, -5.0, 5.0)),           # handle near bar (against the slab)
    ]
    V = load_obj(os.path.join(ENVS, 'env_meshes', 'door.obj'))
    _check_on_surface(V, B, 'door.obj')
    return [box_from_bounds(np.array(lo) * scale, np.array(hi) * scale) for lo, hi in B]


def _check_on_surface(V, B, name):
    """Every mesh vertex must lie on the boundary of the union of boxes (inside no box's open interior)."""
    eps = 1e-6 * max(1.0, np.abs(V).max())
    for v in V:
        on = False
        for lo, hi in B:
            lo, hi = np.array(lo), np.array(hi)
            inside = np.all(v >= lo - eps) and np.all(v <= hi + eps)
            strictly = np.all(v > lo + eps) and np.all(v < hi - eps)
            assert not strictly, (name, 'vertex inside a box', v, lo, hi)
            on = on or inside
        assert on, (name, 'vertex not on any box', v)


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def parse_scene(log):
    """Turn the recorded createCollisionShape/createMultiBody calls into bodies."""
    shapes = {}
    visuals = {}
    bodies = []
    extras = {}
    for e in log:
        if e['fn'] == 'createVisualShape':
            visuals[e['ret']] = e['kwargs'].get('rgbaColor', [1, 1, 1, 1])[:3]
        if e['fn'] == 'createCollisionShape':
            shapes[e['ret']] = (e['args'][0], e['kwargs'])
        elif e['fn'] == 'createMultiBody':
            a, k = e['args'], e['kwargs']
            b = {'id': e['ret'], 'mass': a[0], 'shape': shapes.get(a[1]), 'pos': np.array(a[3], float),
                 'rot': quat_to_mat(a[4]) if len(a) > 4 else quat_to_mat(k.get('baseOrientation', [0, 0, 0, 1])),
                 'friction': 0.5, 'link': None, 'rgb': visuals.get(a[2] if len(a) > 2 else -1, [1, 1, 1])}
            if k.get('linkMasses'):
                b['link_rgb'] = visuals.get((k.get('linkVisualShapeIndices') or [-1])[0], None)
                b['link'] = {'mass': k['linkMasses'][0], 'shape': shapes[k['linkCollisionShapeIndices'][0]],
                             'pos': np.array(k['linkPositions'][0], float), 'rot': quat_to_mat(k['linkOrientations'][0]),
                             'jtype': k['linkJointTypes'][0], 'axis': np.array(k['linkJointAxis'][0], float)}
            bodies.append(b)
        elif e['fn'] == 'loadURDF':
            bodies.append({'id': e['ret'], 'urdf': e['args'][0], 'pos': np.array(e['args'][1], float),
                           'rot': quat_to_mat(e['args'][2])})
        elif e['fn'] == 'changeDynamics' and 'lateralFriction' in e['kwargs']:
            bodies[e['args'][0]]['friction'] = e['kwargs']['lateralFriction']
        elif e['fn'] == 'setJointMotorControl2':
            extras[e['args'][0]] = {'target': e['kwargs']['targetPosition'], 'force': e['kwargs']['force']}
    return bodies, extras


def _set_disc(boxes):
    """angular-motion disc of one scene collision shape in its own frame (see link_colliders_and_inertia): a box |he|, a sphere
    r sqrt(3) (Bullet takes the half diagonal of the AABB), a mesh the AABB of its (exact) box decomposition"""
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for b in boxes:
        ext = np.abs(b['rot']) @ b['he'] if b['type'] == 'box' else b['he']
        lo, hi = np.minimum(lo, b['pos'] - ext), np.maximum(hi, b['pos'] + ext)
    disc = float(np.linalg.norm(0.5 * (lo + hi)) + 0.5 * np.linalg.norm(hi - lo))
    for b in boxes:
        b['disc'] = disc


def shape_boxes(shape):
    typ, kw = shape
    if typ == 3:
        return [{'type': 'box', 'he': np.array(kw['halfExtents'], float), 'pos': np.zeros(3), 'rot': np.eye(3)}]
    if typ == 2:
        return [{'type': 'sphere', 'he': np.array([kw['radius']] * 3, float), 'pos': np.zeros(3), 'rot': np.eye(3)}]
    if typ == 5:
        fn = kw['fileName'].split('/')[-1]
        boxes = drawer_boxes(kw['meshScale'][0]) if 'drawer' in fn else door_boxes(kw['meshScale'][0])
        return [dict(b, type='box') for b in boxes]
    raise NotImplementedError(typ)


# ------------------------------------------------------------------ assemble a model per env kind
WORLD = 0


def make_model(kind, arm, scene_log, arm_base_pos, arm_base_rot, ee_index, rest, arm_type='UR5', scene='complex_scene'):
    nb = len(arm['bodies'])
    M = {'kind': kind, 'arm_type': arm_type, 'scene': scene, 'n_arm': nb, 'arm': arm['bodies'], 'base_pos': arm_base_pos, 'base_rot': arm_base_rot,
         'rest': list(rest), 'free': [], 'joint1': [], 'col': [], 'pair': []}
    col = M['col']

    def add_col(body, c, friction, world_pose=None, tag=''):
        pos, rot = c['pos'], c['rot']
        if world_pose is not None:
            Rw, pw = world_pose
            pos, rot = pw + Rw @ pos, Rw @ rot
        e = {'body': body, 'type': 0 if c['type'] == 'box' else 1, 'he': np.array(c['he'], float),
             'pos': np.array(pos, float), 'rot': np.array(rot, float), 'friction': float(friction), 'tag': tag,
             'link': int(c.get('link', -1)), 'disc': float(c.get('disc', 0.0)), 'contact': c.get('contact', {}),
             'rgb': [float(v) for v in c.get('rgb', [0.7, 0.7, 0.7] if arm_type == 'UR5' else [0.9, 0.9, 0.9])], 'toggle': int(c.get('toggle', 0))}
        if 'hull' in c:
            H = np.array(c['hull'], float)
            e['hull'] = (world_pose[0] @ H.T).T + world_pose[1] if world_pose is not None else H
        if 'cyl' in c:
            e['cyl'] = c['cyl']
        col.append(e)

    for c in arm['base_cols']:
        add_col(WORLD, c, c['friction'], (arm_base_rot, arm_base_pos), 'arm_base')
    for i, b in enumerate(arm['bodies']):
        for c in b['cols']:
            add_col(1 + i, c, c['friction'], None, 'arm')
    bodies, extras = parse_scene(scene_log)
    excluded = set()     # (col index, col index) never collide (same multibody)
    for b in bodies:
        if 'urdf' in b:      # tray/traybox.urdf lives in pybullet_data (absent): geometry from memory, UNVERIFIED
            tray = [((.6, .6, .02), (0, 0, .005), (0, 0, 0)), ((.02, .6, .15), (.25, 0, .059), (0, .575469961, 0)),
                    ((.02, .6, .15), (-.25, 0, .059), (0, -.575469961, 0)), ((.6, .02, .15), (0, -.25, .059), (.575469961, 0, 0)),
                    ((.6, .02, .15), (0, .25, .059), (-.575469961, 0, 0))]
            tb = [{'type': 'box', 'he': 0.5 * np.array(size), 'pos': np.array(xyz), 'rot': urdf_tree.rpy_to_mat(rpy)} for size, xyz, rpy in tray]
            _set_disc(tb)
            for c in tb:
                add_col(WORLD, c, 0.5, (b['rot'], b['pos']), 'tray')
            continue
        boxes = shape_boxes(b['shape'])
        _set_disc(boxes)
        if b['mass'] == 0:
            base_cols = []
            tiny = max(boxes[0]['he']) < 1e-3
            if not tiny:
                for c in boxes:
                    base_cols.append(len(col))
                    # colours of the visual shapes (scenes.py); the two toggles are recoloured by updateToggles (environments.py:469-483):
                    # 1 = the globe over the button (body 10), 2 = the grill over the dial (body 8)
                    tog = (1 if b['id'] == 10 else (2 if b['id'] == 8 else 0)) if scene == 'complex_scene' else 0
                    add_col(WORLD, dict(c, rgb=b['rgb'], toggle=tog), b['friction'], (b['rot'], b['pos']), 'static%d' % b['id'])
            if b['link'] is not None:
                L = b['link']
                jb = 1 + nb + 100 + len(M['joint1'])       # provisional id, fixed below
                Rl = b['rot'] @ L['rot']
                pl = b['pos'] + b['rot'] @ L['pos']
                lb = shape_boxes(L['shape'])
                _set_disc(lb)
                typ, kw = L['shape']
                if typ == 3:
                    he = np.array(kw['halfExtents'], float)
                    I = L['mass'] / 12.0 * np.array([(2 * he[1]) ** 2 + (2 * he[2]) ** 2, (2 * he[0]) ** 2 + (2 * he[2]) ** 2,
                                                      (2 * he[0]) ** 2 + (2 * he[1]) ** 2])
                else:
                    I = np.zeros(3)         # concave trimesh: btTriangleMeshShape::calculateLocalInertia -> 0
                ax = L['axis'] / np.linalg.norm(L['axis'])
                j1 = {'jtype': 0 if L['jtype'] == 0 else 1, 'pos': pl, 'rot': Rl, 'axis': ax, 'mass': L['mass'],
                      'inertia_axis': float(ax @ np.diag(I) @ ax), 'cols': [],
                      'motor_pos': extras.get(b['id'], {}).get('target'), 'motor_force': extras.get(b['id'], {}).get('force'),
                      'scene_id': b['id']}
                for c in lb:
                    j1['cols'].append(len(col))
                    # a link without a visual shape is drawn from its collision shape; the dial link takes the colour of its (1e-5) base
                    add_col(jb, dict(c, rgb=b.get('link_rgb') or b['rgb']), b['friction'], None, 'joint1_%d' % b['id'])
                for a in base_cols:
                    for c2 in j1['cols']:
                        excluded.add((a, c2))
                M['joint1'].append(j1)
        else:
            typ, kw = b['shape']
            if typ == 3:
                he = np.array(kw['halfExtents'], float)
                I = b['mass'] / 12.0 * np.array([(2 * he[1]) ** 2 + (2 * he[2]) ** 2, (2 * he[0]) ** 2 + (2 * he[2]) ** 2,
                                                  (2 * he[0]) ** 2 + (2 * he[1]) ** 2])
            else:
                I = np.zeros(3)             # H5
            fb = {'mass': b['mass'], 'inertia': I, 'pos0': b['pos'], 'rot0': b['rot'], 'cols': [], 'scene_id': b['id'],
                  'rot_locked': int(np.all(I == 0))}
            for c in boxes:
                fb['cols'].append(len(col))
                add_col(1 + nb + 200 + len(M['free']), dict(c, rgb=b['rgb']), b['friction'], None, 'free%d' % b['id'])
            M['free'].append(fb)
    # the reference lists objects (blocks) first in obs; scene creation order puts the drawer before the block.
    # canonical order here: free[0] = block (if any), free[1] = drawer.
    M['free'].sort(key=lambda f: f['rot_locked'])
    # environments.py:432, 454: the objects are recoloured green, blue
    for k, f in enumerate(x for x in M['free'] if not x['rot_locked']):
        for ci in f['cols']:
            col[ci]['rgb'] = [[0.0, 1.0, 0.0], [0.0, 0.0, 1.0]][k % 2]
    # final body ids
    remap = {}
    for k, f in enumerate(M['free']):
        for ci in f['cols']:
            remap[ci] = 1 + nb + k
    for k, j in enumerate(M['joint1']):
        for ci in j['cols']:
            remap[ci] = 1 + nb + len(M['free']) + k
    for ci, b in remap.items():
        col[ci]['body'] = b
    # reorder joint1 to the reference's obs order [door, button, dial] (complex_scene returns [door, button, dial])
    if len(M['joint1']) == 3:
        ids = [j['scene_id'] for j in M['joint1']]       # creation order: door(1), dial(7), button(9)
        order = [ids.index(1), ids.index(9), ids.index(7)]
        M['joint1'] = [M['joint1'][i] for i in order]
        for k, j in enumerate(M['joint1']):
            for ci in j['cols']:
                col[ci]['body'] = 1 + nb + len(M['free']) + k

    # sites
    sb, sp, sr = arm['site'](ee_index)
    M['sites'] = [{'body': 1 + sb, 'pos': sp, 'rot': sr}]
    if arm_type != 'Panda':
        for li in (ee_index - 1, 18, 20):
            sb, sp, sr = arm['site'](li)
            M['sites'].append({'body': 1 + sb, 'pos': sp, 'rot': sr})

    # candidate pairs
    def is_arm(c):
        return c['tag'] in ('arm', 'arm_base')

    # collision objects (one Bullet manifold per object pair): statics per scene body, arm per URDF link, others per body
    keys = {}
    for c in col:
        key = (c['tag'], c['link']) if c['tag'] == 'arm' else (c['tag'],)
        c['obj'] = keys.setdefault(key, len(keys))
    reach = [reach_aabb(M, c) for c in col]
    for i in range(len(col)):
        for j in range(i + 1, len(col)):
            a, b = col[i], col[j]
            if a['body'] == b['body']:
                continue
            if is_arm(a) and is_arm(b):
                continue
            if (i, j) in excluded or (j, i) in excluded:
                continue
            (alo, ahi), (blo, bhi) = reach[i], reach[j]
            if np.any(alo > bhi + 0.03) or np.any(blo > ahi + 0.03):
                continue
            # order so that the first collider belongs to the higher body id (dynamic one first)
            M['pair'].append((j, i) if b['body'] > a['body'] else (i, j))
    # pairs of one manifold (object pair) contiguous, in a canonical order
    # dynamic objects with the highest ids first (block, drawer, scene joints, then arm links): they survive the contact cap
    M['pair'].sort(key=lambda p: (-col[p[0]]['obj'], col[p[1]]['obj'], p[0], p[1]))
    return M


def reach_aabb(M, c):
    """Conservative world AABB a collider can ever occupy (statics: exact; 1-DoF bodies: swept; others: infinite)."""
    nb = M['n_arm']
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * c['he']
    P = (c['rot'] @ corners.T).T + c['pos']
    if c['body'] == WORLD:
        return P.min(0), P.max(0)
    k = c['body'] - 1 - nb - len(M['free'])
    if k >= 0:
        j = M['joint1'][k]
        if j['jtype'] == 1:
            ax = j['rot'] @ j['axis']
            lo, hi = (-0.35, 0.35) if abs(ax[0]) > 0.5 else (-0.05, 0.08)    # door travel / button travel (generous)
            W = (j['rot'] @ P.T).T + j['pos']
            A = np.vstack([W + lo * ax, W + hi * ax])
            return A.min(0), A.max(0)
        r = np.linalg.norm(P, axis=1).max()
        return j['pos'] - r, j['pos'] + r
    return np.full(3, -np.inf), np.full(3, np.inf)


# ------------------------------------------------------------------ emit
def to_jsonable(o):
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    if isinstance(o, dict):
        return {k: to_jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [to_jsonable(v) for v in o]
    return o


def c_array(name, arr, ctype='double'):
    flat = np.asarray(arr).reshape(-1)
    if ctype == 'double':
        body = ','.join(repr(float(v)) for v in flat)
    else:
        body = ','.join(str(int(v)) for v in flat)
    return '  static const %s %s[%d] = {%s};\n' % (ctype, name, max(1, len(flat)), body if len(flat) else '0')


def emit_header(models, path):
    out = ['/* GENERATED by tools/bake_assets.py from the reference assets (URDF, meshes, captured scene log).\n'
           ' * Do not edit.  Consumed by oracle/rp_oracle.c and roboticsplayroompybullet_amd/csrc (rp_model.h). */\n'
           '#ifndef RP_MODELS_GEN_H\n#define RP_MODELS_GEN_H\n#include <string.h>\n#include "../rp_model.h"\n\n']
    for M in models:
        k = M['kind']
        out.append('static inline void rp_fill_model_%s(rp_model* m) {\n' % k)
        out.append('  memset(m, 0, sizeof(*m));\n')
        nb = M['n_arm']
        A = M['arm']
        out.append('  m->kind = %d; m->n_arm = %d; m->n_free = %d; m->n_joint1 = %d; m->n_col = %d; m->n_pair = %d; m->n_site = %d;\n'
                   % ('URPQVW'.index(k), nb, len(M['free']), len(M['joint1']), len(M['col']), len(M['pair']), len(M['sites'])))

        drawer = [i for i, f in enumerate(M['free']) if f['rot_locked']]
        out.append('  m->drawer_free = %d;\n' % (drawer[0] if drawer else -1))
        # the rotation-locked drawer shares the arm's half of the solver's velocity layout (lanes n_arm .. : three translation lanes in the
        # default build, six in the wide one), so that its resting contacts are solved beside the objects' instead of after them
        assert not drawer or nb + 3 <= 16
        out.append('  m->free_row0 = %d;\n' % ((1 << drawer[0]) if drawer else 0))
        out.append('  m->arm_type = %d; m->scene = %d;\n' % (['UR5', 'Panda'].index(M['arm_type']),
                                                               ['complex_scene', 'default_scene', 'push_scene'].index(M['scene'])))

        def put(field, arr, ctype='double'):
            name = 't_' + field
            out.append(c_array(name, arr, ctype))
            out.append('  memcpy(m->%s, %s, sizeof(%s));\n' % (field, name, name))

        put('arm_parent', [b['parent'] for b in A], 'int')
        put('arm_jtype', [b['jtype'] for b in A], 'int')
        put('arm_bullet_index', [b['bullet_index'] for b in A], 'int')
        put('arm_jpos', [b['jpos'] for b in A])
        put('arm_jrot', [b['jrot'] for b in A])
        put('arm_axis', [b['axis'] for b in A])
        put('arm_mass', [b['mass'] for b in A])
        put('arm_com', [b['com'] for b in A])
        put('arm_inertia', [b['inertia'] for b in A])
        put('arm_lower', [b['lower'] for b in A])
        put('arm_upper', [b['upper'] for b in A])
        put('base_pos', M['base_pos'])
        put('base_rot', M['base_rot'])
        put('rest', M['rest'] + [0.0] * (nb - len(M['rest'])))
        put('site_body', [s['body'] for s in M['sites']], 'int')
        put('site_pos', [s['pos'] for s in M['sites']])
        put('site_rot', [s['rot'] for s in M['sites']])
        if M['free']:
            put('free_mass', [f['mass'] for f in M['free']])
            put('free_inertia', [f['inertia'] for f in M['free']])
            put('free_pos0', [f['pos0'] for f in M['free']])
            put('free_rot0', [f['rot0'] for f in M['free']])
            put('free_rot_locked', [f['rot_locked'] for f in M['free']], 'int')
        if M['joint1']:
            J = M['joint1']
            put('j1_type', [j['jtype'] for j in J], 'int')
            put('j1_pos', [j['pos'] for j in J])
            put('j1_rot', [j['rot'] for j in J])
            put('j1_axis', [j['axis'] for j in J])
            put('j1_mass', [j['mass'] for j in J])
            put('j1_inertia_axis', [j['inertia_axis'] for j in J])
            put('j1_has_pos_motor', [int(j['motor_pos'] is not None) for j in J], 'int')
            put('j1_motor_target', [j['motor_pos'] or 0.0 for j in J])
            put('j1_motor_force', [j['motor_force'] or 0.0 for j in J])
        C = M['col']
        put('col_body', [c['body'] for c in C], 'int')
        put('col_type', [c['type'] for c in C], 'int')
        put('col_he', [c['he'] for c in C])
        put('col_pos', [c['pos'] for c in C])
        put('col_rot', [c['rot'] for c in C])
        put('col_friction', [c['friction'] for c in C])
        put('col_link', [c['link'] for c in C], 'int')
        put('col_obj', [c['obj'] for c in C], 'int')
        put('col_thr', [0.02 * c['disc'] for c in C])
        # URDF <contact> stiffness / damping (gripper links; ur5e2.urdf:306-312, panda.urdf:256-262); 0 = none (rigid contact)
        put('col_rgb', [c['rgb'] for c in C])
        put('col_toggle', [c['toggle'] for c in C], 'int')
        put('col_stiffness', [float(c.get('contact', {}).get('stiffness', 0.0)) for c in C])
        put('col_damping', [float(c.get('contact', {}).get('damping', 0.0)) for c in C])
        put('col_spin', [float(c.get('contact', {}).get('spinning_friction', 0.0)) for c in C])
        put('pair', M['pair'], 'unsigned char')
        out.append('}\n\n')
    out.append('#endif\n')
    open(path, 'w').write(''.join(out))


def emit_hulls(models, path):
    """exact collision shapes of the arm colliders for oracle/rp_bullet_ref.c (the frozen Bullet-like model): convex-hull vertices in
    the owning body's frame for mesh colliders, radius / half length for cylinders (the collider pose of rp_model is the cylinder's
    frame), and the URDF <contact> block (stiffness, damping, spinning friction, friction anchor) per collider"""
    out = ['/* GENERATED by tools/bake_assets.py from the reference URDFs and collision meshes.  Do not edit.\n'
           ' * Consumed by oracle/rp_bullet_ref.c only (test infrastructure); the product never includes it. */\n'
           '#ifndef RP_HULLS_GEN_H\n#define RP_HULLS_GEN_H\n\n'
           'typedef struct { int kind; int col; int shape; int n; const double* v; double radius, halflen; double stiffness, damping, spinning; int anchor; } rpb_shape;\n'
           '/* shape: 2 = convex hull (n vertices v, body frame, Bullet margin 0.001), 3 = cylinder along the collider z axis */\n\n']
    entries = []
    for M in models:
        k = 'URPQVW'.index(M['kind'])
        for ci, c in enumerate(M['col']):
            ct = c.get('contact', {})
            stiff, damp = float(ct.get('stiffness', -1.0)), float(ct.get('damping', -1.0))
            spin, anchor = float(ct.get('spinning_friction', 0.0)), int(bool(ct.get('friction_anchor', False)))
            if 'hull' in c:
                name = 'rpb_v_%s_%d' % (M['kind'], ci)
                H = np.asarray(c['hull'], float)
                out.append('static const double %s[%d] = {%s};\n' % (name, H.size, ','.join('%.9g' % v for v in H.reshape(-1))))
                entries.append('{%d, %d, 2, %d, %s, 0, 0, %r, %r, %r, %d}' % (k, ci, len(H), name, stiff, damp, spin, anchor))
            elif 'cyl' in c:
                entries.append('{%d, %d, 3, 0, 0, %r, %r, %r, %r, %r, %d}' % (k, ci, c['cyl'][0], c['cyl'][1], stiff, damp, spin, anchor))
            elif ct:
                entries.append('{%d, %d, %d, 0, 0, 0, 0, %r, %r, %r, %d}' % (k, ci, c['type'], stiff, damp, spin, anchor))
    out.append('\nstatic const rpb_shape rpb_shapes[%d] = {\n  %s};\n' % (len(entries), ',\n  '.join(entries)))
    out.append('static const int rpb_n_shapes = %d;\n\n#endif\n' % len(entries))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, 'w').write(''.join(out))


def cfloat(x):
    t = '%.9g' % x
    return t + ('f' if ('.' in t or 'e' in t or 'n' in t) else '.f')


def emit_hull_verts(models, path):
    """convex-hull vertices of the arm colliders for the PRODUCT (HIP library) and the fast model's oracle: float, owning body's frame, one table per arm
    (the kinds of one arm share it), per kind the first vertex and the count of every collider (count 0 = no hull: boxes, cylinders, the scene).  Used
    for the vertex-against-face contacts of arm links with static boxes (rp_kernels.cuh hull_face, rp_oracle.c hull_face)."""
    out = ['/* GENERATED by tools/bake_assets.py from the reference URDFs and collision meshes.  Do not edit.\n'
           ' * Convex-hull vertices of the arm links\' collision meshes (body frame; Bullet keeps a 0.001 margin around them), shared by the HIP library and\n'
           ' * oracle/rp_oracle.c. */\n#ifndef RP_HULLVERTS_GEN_H\n#define RP_HULLVERTS_GEN_H\n\n#define RP_HULL_MARGIN 0.001f\n\n']
    tables = {}          # arm name -> (table name, flat list of vertices)
    per_kind = []
    for M in models:
        arm = M['arm_type']
        name = 'rp_hullv_%s' % arm.upper()
        verts, off, cnt = [], [0] * 64, [0] * 64
        for ci, c in enumerate(M['col']):
            if 'hull' in c:
                H = np.asarray(c['hull'], np.float32)
                off[ci], cnt[ci] = len(verts), len(H)
                verts += [tuple(float(x) for x in v) for v in H]
        if arm in tables:
            assert tables[arm][1] == verts, 'the kinds of one arm share their hulls'
        else:
            tables[arm] = (name, verts)
        per_kind.append((M['kind'], name, len(verts), off, cnt))
    for arm, (name, verts) in tables.items():
        out.append('static const float %s[%d][4] = {\n%s};\n\n' % (name, len(verts), ',\n'.join(
            ','.join('{%s,%s,%s,0.f}' % tuple(cfloat(x) for x in v) for v in verts[i:i + 4]) for i in range(0, len(verts), 4))))
    for kind, name, n, off, cnt in per_kind:
        out.append('static const int rp_hull_off_%s[64] = {%s};\nstatic const int rp_hull_cnt_%s[64] = {%s};\n' % (
            kind, ','.join(str(v) for v in off), kind, ','.join(str(v) for v in cnt)))
    out.append('\n/* kind (RP_KIND_*) -> vertex table, number of vertices in it, per-collider first vertex and count */\n'
               'static inline int rp_hull_tables(int kind, const float (**v)[4], const int** off, const int** cnt) {\n  switch (kind) {\n')
    for kind, name, n, off, cnt in per_kind:
        out.append('    case %d: *v = %s; *off = rp_hull_off_%s; *cnt = rp_hull_cnt_%s; return %d;\n' % ('URPQVW'.index(kind), name, kind, kind, n))
    out.append('  }\n  *v = 0; *off = 0; *cnt = 0; return 0;\n}\n\n#endif\n')
    open(path, 'w').write(''.join(out))


def main():
    gold = json.load(open(os.path.join(REPO, 'tests', 'golden', 'scenes.json')))
    ur5 = build_arm(os.path.join(ENVS, 'ur_e_description', 'ur5e2.urdf'))
    panda = build_arm(os.path.join(ENVS, 'franka_panda', 'panda.urdf'))
    models = []
    # Q, V: the Panda (instance_init_P: base pose, EE index and rest pose are per arm type, environments.py:356-363) in the
    # other two scenes
    gold2 = json.load(open(os.path.join(REPO, 'tests', 'golden', 'two_object_ids.json')))
    for kind, arm, scene, ini in (('U', ur5, 'complex_scene', 'U'), ('R', ur5, 'default_scene', 'R'), ('P', panda, 'push_scene', 'P'),
                                  ('Q', panda, 'default_scene', 'P'), ('V', panda, 'complex_scene', 'P'), ('W', panda, 'complex_scene', 'P')):
        init = gold['instance_init_' + ini]
        M = make_model(kind, arm, gold2['complex_scene_2obj']['log'] if kind == 'W' else gold[scene]['log'], np.array(init['base_pos'], float), quat_to_mat(init['base_orn']),
                       init['ee_index'], init['rest'], 'UR5' if arm is ur5 else 'Panda', scene)
        print(kind, 'arm dofs', M['n_arm'], 'free', len(M['free']), 'joint1', len(M['joint1']), 'colliders', len(M['col']),
              'pairs', len(M['pair']))
        models.append(M)
    os.makedirs(os.path.join(REPO, 'roboticsplayroompybullet_amd', 'csrc', 'generated'), exist_ok=True)
    emit_header(models, os.path.join(REPO, 'roboticsplayroompybullet_amd', 'csrc', 'generated', 'rp_models_gen.h'))
    emit_hulls(models, os.path.join(REPO, 'oracle', 'generated', 'rp_hulls_gen.h'))
    emit_hull_verts(models, os.path.join(REPO, 'roboticsplayroompybullet_amd', 'csrc', 'generated', 'rp_hullverts_gen.h'))
    slim = []
    for M in models:
        m = {k: v for k, v in M.items()}
        slim.append(to_jsonable(m))
    json.dump({'models': slim, 'ur5_joint_names': ur5['names'], 'panda_joint_names': panda['names']},
              open(os.path.join(REPO, 'roboticsplayroompybullet_amd', 'assets', 'models.json'), 'w'), indent=1)
    return models


if __name__ == '__main__':
    main()
