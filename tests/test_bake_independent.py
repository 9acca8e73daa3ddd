"""SURVEY.md row a17 (scene / model construction): the baked tables against a SECOND, independent reading of the reference's files.

Everything the CPU oracle, the frozen reference step and the HIP library know about the models comes from ONE script (tools/bake_assets.py +
tools/urdf_tree.py), so an error there is common-mode.  This test shares no code with it:
  * tests/golden/assets_independent.json - what the URDF / STL / OBJ files say, read by tests/golden/make_asset_goldens.py (its own XML walk
    and mesh readers; numbers only);
  * tests/golden/scenes.json - the scene calls of scenes.py:8-426 as the reference's own Python issued them to a recording client;
  * numpy below - Bullet's link order (depth-first, children in XML order: the reference notebook's table), forward kinematics over the FULL
    tree (fixed joints included, nothing merged), composite bodies, the link inertias of hypotheses H2 / H3 (no <inertial> =>
    mass 1 and identity frame; inertia = box formula on the bounds of the link's collision shapes in its inertial frame), the joint-space mass
    matrix from per-link Jacobians.
Compared with what the oracle HOLDS (rpo_arm_table, rpo_site_pose, rpo_mass_matrix_inv, rpo_collider_*): joint types / limits / Bullet indices /
body masses, EE / wrist / pad link poses at random joint angles, M^-1 at 5 random poses, every arm collider's link and place, every scene
collider's shape, pose, friction and mass, the mesh bodies' bounds.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import OracleEnv

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, 'golden', 'assets_independent.json')))
SCN = json.load(open(os.path.join(HERE, 'golden', 'scenes.json')))
MARGIN = 0.001                       # gUrdfDefaultCollisionMargin: convex hull shapes of URDF meshes (H3)
KINDS = {'U': ('ur5', 'instance_init_U', 'complex_scene'), 'R': ('ur5', 'instance_init_R', 'default_scene'), 'P': ('panda', 'instance_init_P', 'push_scene')}


def rot_rpy(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def rot_quat(q):
    x, y, z, w = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_axis(a, t):
    a = np.asarray(a, float) / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * K @ K


class Arm:
    """the arm as the URDF says it, nothing merged"""

    def __init__(self, model, base_pos, base_quat):
        self.m = model
        self.base = (rot_quat(base_quat), np.asarray(base_pos, float))
        self.order = []                                     # Bullet's joint / link order: depth-first pre-order, children in XML order

        def visit(link, parent_index):
            for j in model['joints']:
                if j['parent'] == link:
                    self.order.append((j, parent_index))
                    visit(j['child'], len(self.order) - 1)
        visit(model['root'], -1)
        self.movable = [i for i, (j, _) in enumerate(self.order) if j['type'] in ('revolute', 'prismatic', 'continuous')]

    def frames(self, q):
        """world (R, p) of every link frame (index = Bullet link index) and of every joint frame before its motion"""
        qi = dict(zip(self.movable, q))
        link, joint = [], []
        for i, (j, par) in enumerate(self.order):
            Rp, pp = self.base if par < 0 else link[par]
            Rj, pj = Rp @ rot_rpy(j['rpy']), pp + Rp @ np.asarray(j['xyz'])
            joint.append((Rj, pj))
            R, p = Rj, pj
            if i in qi:
                ax = np.asarray(j['axis'], float) / np.linalg.norm(j['axis'])
                if j['type'] == 'prismatic':
                    p = pj + Rj @ ax * qi[i]
                else:
                    R = Rj @ rot_axis(ax, qi[i])
            link.append((R, p))
        return link, joint

    def link_inertial(self, name):
        """(mass, com in link frame, rotation of the inertial frame, principal inertias): H2 / H3"""
        ln = self.m['links'][name]
        has = ln['mass'] is not None
        mass = ln['mass'] if has else 1.0
        com = np.asarray(ln['inertial_xyz']) if has else np.zeros(3)
        Rc = rot_rpy(ln['inertial_rpy']) if has else np.eye(3)
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        last_margin, single_identity = 0.0, False
        for c in ln['collisions']:
            if c['kind'] == 'mesh':
                blo, bhi, margin = np.asarray(c['lo']), np.asarray(c['hi']), MARGIN
            elif c['kind'] == 'box':
                bhi = 0.5 * np.asarray(c['size']); blo = -bhi; margin = 0.0
            else:
                bhi = np.array([c['radius'], c['radius'], 0.5 * c['length']]); blo = -bhi; margin = 0.0
            corners = np.array([[x, y, z] for x in (blo[0], bhi[0]) for y in (blo[1], bhi[1]) for z in (blo[2], bhi[2])])
            Rs = rot_rpy(c['rpy'])
            in_link = corners @ Rs.T + np.asarray(c['xyz'])
            in_inertial = (in_link - com) @ Rc
            lo, hi = np.minimum(lo, in_inertial.min(0) - margin), np.maximum(hi, in_inertial.max(0) + margin)
            last_margin = margin
            single_identity = len(ln['collisions']) == 1 and np.allclose(c['xyz'], com) and np.allclose(Rs, Rc)
        if not ln['collisions'] or mass == 0.0:
            return mass, com, Rc, np.zeros(3)
        ext = hi - lo + (2 * last_margin if single_identity else 0.0)       # btPolyhedralConvexShape adds its margin once more
        return mass, com, Rc, mass / 12.0 * np.array([ext[1] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[2] ** 2, ext[0] ** 2 + ext[1] ** 2])

    def mass_matrix(self, q):
        link, joint = self.frames(q)
        n = len(self.movable)
        M = np.zeros((n, n))
        col = {i: k for k, i in enumerate(self.movable)}
        for i, (j, par) in enumerate(self.order):
            mass, com, Rc, I = self.link_inertial(j['child'])
            if mass == 0.0:
                continue
            R, p = link[i]
            c = p + R @ com
            Iw = (R @ Rc) @ np.diag(I) @ (R @ Rc).T
            Jv, Jw = np.zeros((3, n)), np.zeros((3, n))
            a = i
            while a >= 0:
                if a in col:
                    ja = self.order[a][0]
                    Rj, pj = joint[a]
                    ax = Rj @ (np.asarray(ja['axis'], float) / np.linalg.norm(ja['axis']))
                    if ja['type'] == 'prismatic':
                        Jv[:, col[a]] = ax
                    else:
                        Jw[:, col[a]] = ax
                        Jv[:, col[a]] = np.cross(ax, c - pj)
                a = self.order[a][1]
            M += mass * Jv.T @ Jv + Jw.T @ Iw @ Jw
        return M

    def body_masses(self):
        """mass of the composite body of every movable joint: its child link plus everything hanging on it by fixed joints"""
        out = {i: 0.0 for i in self.movable}
        for i, (j, par) in enumerate(self.order):
            a = i
            while a >= 0 and a not in out:
                a = self.order[a][1]
            if a >= 0:
                out[a] += self.link_inertial(j['child'])[0]
        return [out[i] for i in self.movable]


def make(kind):
    arm_name, init, _ = KINDS[kind]
    ini = SCN[init]
    return Arm(GOLD[arm_name], ini['base_pos'], ini['base_orn']), OracleEnv(kind, seed=0), ini


@pytest.mark.parametrize('kind', ['U', 'R', 'P'])
def test_joint_table_and_body_masses(kind):
    arm, o, ini = make(kind)
    t = o.arm_table()
    assert len(arm.movable) == o.n_arm
    for k, i in enumerate(arm.movable):
        j = arm.order[i][0]
        assert int(t[k, 4]) == i                                                  # Bullet joint index of dof k
        assert int(t[k, 0]) == (1 if j['type'] == 'prismatic' else 0)
        assert t[k, 1] == pytest.approx(j['lower'], abs=1e-12) and t[k, 2] == pytest.approx(j['upper'], abs=1e-12)
        par = arm.order[i][1]
        while par >= 0 and par not in arm.movable:
            par = arm.order[par][1]
        assert int(t[k, 5]) == (arm.movable.index(par) if par >= 0 else -1)       # movable parent
    np.testing.assert_allclose(t[:, 3], arm.body_masses(), rtol=1e-12)
    assert ini['ee_index'] == {'ur5': 7, 'panda': 11}[KINDS[kind][0]]


@pytest.mark.parametrize('kind', ['U', 'R', 'P'])
def test_link_poses_at_random_joint_angles(kind):
    """getLinkState(arm, i)[0:2] = the link's COM frame: EE link (environments.py:746-764), and for the UR5 the wrist link (ee index - 1) and the two
    pad links 18 / 20 of gripper_proprioception (environments.py:720-743)"""
    arm, o, ini = make(kind)
    rng = np.random.default_rng(1)
    t = o.arm_table()
    sites = {0: ini['ee_index']}
    if KINDS[kind][0] == 'ur5':
        sites.update({1: ini['ee_index'] - 1, 2: 18, 3: 20})
    for trial in range(5):
        q = t[:, 1] + (t[:, 2] - t[:, 1]) * rng.random(o.n_arm)
        q[:6] = rng.uniform(-2.5, 2.5, 6)
        o.set_arm_q(q)
        link, _ = arm.frames(q)
        for site, li in sites.items():
            ln = arm.m['links'][arm.order[li][0]['child']]
            R, p = link[li]
            if ln['mass'] is not None:
                p, R = p + R @ np.asarray(ln['inertial_xyz']), R @ rot_rpy(ln['inertial_rpy'])
            pos, quat, _, _ = o.site_pose(site)
            np.testing.assert_allclose(pos, p, atol=1e-9, err_msg='site %d link %d' % (site, li))
            np.testing.assert_allclose(rot_quat(quat), R, atol=1e-9, err_msg='site %d link %d' % (site, li))


@pytest.mark.parametrize('kind', ['U', 'P'])
def test_mass_matrix_at_random_poses(kind):
    """rpo_mass_matrix_inv (ABA unit-impulse responses over the baked, merged bodies) against the inverse of a joint-space mass matrix summed over
    the URDF's links one by one"""
    arm, o, _ = make(kind)
    rng = np.random.default_rng(2)
    t = o.arm_table()
    for trial in range(5):
        q = t[:, 1] + (t[:, 2] - t[:, 1]) * rng.random(o.n_arm)
        q[:6] = rng.uniform(-2.5, 2.5, 6)
        o.set_arm_q(q)
        M = arm.mass_matrix(q)
        Minv = o.mass_matrix_inv()
        np.testing.assert_allclose(Minv @ M, np.eye(o.n_arm), atol=2e-7)
        np.testing.assert_allclose(np.linalg.inv(Minv), M, rtol=1e-6, atol=1e-9 * np.abs(M).max())


@pytest.mark.parametrize('kind', ['U', 'P'])
def test_arm_colliders_sit_on_their_links(kind):
    """every <collision> of the URDF has a collider with that Bullet link index: boxes with their half extents at their place; cylinders as the
    equal-area prism (H6) at their place; meshes as an OBB whose centre lies inside the mesh's bounds, whose diagonal is about theirs and whose
    volume is not larger than theirs"""
    arm, o, _ = make(kind)
    q = np.array([arm.order[i][0]['lower'] + 0.3 * (arm.order[i][0]['upper'] - arm.order[i][0]['lower']) for i in arm.movable])
    q[:6] = [-1.5, -1.6, -1.9, -1.2, 1.57, 0.07]
    o.set_arm_q(q)
    link, _ = arm.frames(q)
    cols = [c for c in o.collider_list() if c['link'] >= 0 or (c['body'] == 0 and c['link'] == -1 and False)]
    seen = 0
    for li, (j, par) in enumerate(arm.order):
        ln = arm.m['links'][j['child']]
        mine = [c for c in cols if c['link'] == li]
        assert len(mine) == len(ln['collisions']), (li, j['child'], len(mine), len(ln['collisions']))
        R, p = link[li]
        for c in ln['collisions']:
            Rs, ps = R @ rot_rpy(c['rpy']), p + R @ np.asarray(c['xyz'])
            if c['kind'] == 'mesh':
                lo, hi = np.asarray(c['lo']), np.asarray(c['hi'])
                half_diag = 0.5 * np.linalg.norm(hi - lo)          # a tight OBB of the hull: centre inside the mesh's bounds, about as long as they are, never larger
                ok = [m for m in mine if (np.abs((m['p'] - ps) @ Rs - 0.5 * (lo + hi)) <= 0.5 * (hi - lo) + 1e-9).all()
                      and 0.55 * half_diag < np.linalg.norm(m["he"]) < 1.15 * half_diag and np.prod(2 * m['he']) < 1.05 * np.prod(hi - lo)]
            elif c['kind'] == 'box':
                ok = [m for m in mine if np.allclose(m['p'], ps, atol=1e-9) and np.allclose(np.sort(m['he']), np.sort(0.5 * np.asarray(c['size'])), atol=1e-9)]
            else:
                s = c['radius'] * np.sqrt(np.pi) / 2
                ok = [m for m in mine if np.allclose(m['p'], ps, atol=1e-9) and np.allclose(m['he'], [s, s, 0.5 * c['length']], atol=1e-9) and np.allclose(m['R'], Rs, atol=1e-9)]
            assert ok, (li, j['child'], c)
            contact = ln['contact']
            for m in ok[:1]:
                assert m['friction'] == pytest.approx(contact.get('lateral_friction', 0.5))
                assert m['stiffness'] == pytest.approx(contact.get('stiffness', 0.0) if 'stiffness' in contact else 0.0)
            seen += 1
    assert seen >= 10


def scene_bodies(log):
    """createMultiBody calls of a scene log -> [(mass, pos, R, shape, links)] with shape / link shapes resolved from the createCollisionShape calls"""
    shapes, out = {}, []
    for c in log:
        if c['fn'] == 'createCollisionShape':
            shapes[c['ret']] = (c['args'][0], c['kwargs'])
        elif c['fn'] == 'createMultiBody':
            a, kw = c['args'], c['kwargs']
            a = list(a) + [None] * (5 - len(a))
            mass = kw.get('baseMass', a[0])
            col = kw.get('baseCollisionShapeIndex', a[1])
            pos = kw.get('basePosition', a[3]) or [0, 0, 0]
            orn = kw.get('baseOrientation', a[4]) or [0, 0, 0, 1]
            links = []
            for k, ci in enumerate(kw.get('linkCollisionShapeIndices', [])):
                links.append({'mass': kw['linkMasses'][k], 'shape': shapes.get(ci), 'pos': kw['linkPositions'][k], 'orn': kw['linkOrientations'][k],
                              'joint': kw['linkJointTypes'][k], 'axis': kw['linkJointAxis'][k]})
            out.append({'id': c['ret'], 'mass': mass, 'pos': np.asarray(pos, float), 'R': rot_quat(orn), 'shape': shapes.get(col), 'links': links})
    return out


GEOM_SPHERE, GEOM_BOX, GEOM_MESH = 2, 3, 5


@pytest.mark.parametrize('kind', ['U', 'R', 'P'])
def test_scene_colliders_against_the_reference_scene_calls(kind):
    """scenes.py as the reference issued it (tests/golden/scenes.json) against the oracle's collider table at its creation state: every box / sphere
    body and link with its half extents, world pose and mass; the concave meshes (door link, drawer) by the bounds of their box decompositions;
    lateral friction from changeDynamics (block 1.5, scenes.py:77-81)"""
    _, o, _ = make(kind)
    log = SCN[KINDS[kind][2]]['log']
    bodies = scene_bodies(log)
    fric = {c['args'][0]: c['kwargs']['lateralFriction'] for c in log if c['fn'] == 'changeDynamics' and 'lateralFriction' in c['kwargs']}
    cols = [c for c in o.collider_list() if c['link'] < 0]                # everything that is not an arm link (the arm's base link has link -1 too: excluded by place below)
    used = set()

    def find(shape, R, p, mass, what):
        t, kw = shape
        for k, c in enumerate(cols):
            if k in used or not np.allclose(c['p'], p, atol=1e-9):
                continue
            if t == GEOM_BOX and c['type'] == 0 and np.allclose(c['he'], kw['halfExtents'], atol=1e-12) and np.allclose(c['R'], R, atol=1e-9):
                pass
            elif t == GEOM_SPHERE and c['type'] == 1 and c['he'][0] == pytest.approx(kw['radius']):
                pass
            else:
                continue
            assert c['mass'] == pytest.approx(mass), (what, c)
            used.add(k)
            return c
        raise AssertionError('no collider for %s: %s at %s' % (what, shape, p))

    n_checked = 0
    for b in bodies:
        if b['shape'] is not None and b['shape'][0] in (GEOM_BOX, GEOM_SPHERE):
            tiny = b['shape'][0] == GEOM_BOX and max(b['shape'][1]['halfExtents']) < 1e-4      # the dial's 1e-5 base box: nothing can touch it
            if not tiny:
                c = find(b['shape'], b['R'], b['pos'], b['mass'], 'body %d' % b['id'])
                if b['id'] in fric:
                    assert c['friction'] == pytest.approx(fric[b['id']])
                n_checked += 1
        for ln in b['links']:
            Rl, pl = b['R'] @ rot_quat(ln['orn']), b['pos'] + b['R'] @ np.asarray(ln['pos'], float)
            if ln['shape'][0] in (GEOM_BOX, GEOM_SPHERE):
                find(ln['shape'], Rl, pl, ln['mass'], 'link of body %d' % b['id'])
                n_checked += 1
            else:                                                     # concave mesh: the box decomposition fills the mesh's bounds
                g = GOLD['scene_meshes'][os.path.basename(ln['shape'][1]['fileName'])]
                assert ln['shape'][1]['meshScale'][0] == pytest.approx(g['scale'])
                check_mesh_body(cols, used, Rl, pl, g, ln['mass'])
                n_checked += 1
        if b['shape'] is not None and b['shape'][0] == GEOM_MESH:
            g = GOLD['scene_meshes'][os.path.basename(b['shape'][1]['fileName'])]
            assert b['shape'][1]['meshScale'][0] == pytest.approx(g['scale'])
            check_mesh_body(cols, used, b['R'], b['pos'], g, b['mass'])
            n_checked += 1
    assert n_checked >= {'U': 17, 'R': 1, 'P': 2}[kind], n_checked
    if kind != 'P':            # (push_scene adds pybullet_data's tray, which is not in the reference repo: H10)
        left = [c for k, c in enumerate(cols) if k not in used and c['body'] != 0]
        assert not left, left


def check_mesh_body(cols, used, R, p, g, mass):
    """the colliders of one movable body whose boxes lie inside the mesh's bounds and together reach all six faces of them"""
    lo, hi = np.asarray(g['lo']), np.asarray(g['hi'])
    mine = []
    for k, c in enumerate(cols):
        if k in used or c['type'] != 0 or c['body'] == 0:
            continue
        corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]) * c['he'] @ c['R'].T + c['p']
        local = (corners - p) @ R
        if (local >= lo - 1e-6).all() and (local <= hi + 1e-6).all():
            mine.append((k, local))
    assert mine, 'no boxes inside the mesh bounds'
    body = {cols[k]['body'] for k, _ in mine}
    assert len(body) == 1
    allp = np.concatenate([l for _, l in mine])
    np.testing.assert_allclose(allp.min(0), lo, atol=2e-3)
    np.testing.assert_allclose(allp.max(0), hi, atol=2e-3)
    for k, _ in mine:
        assert cols[k]['mass'] == pytest.approx(mass)
        used.add(k)
