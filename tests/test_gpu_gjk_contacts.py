"""The hull contacts of the narrowphase, pose by pose: device (rp_debug_substep: the contact list of one substep) against the fp64 and fp32 CPU oracles at the SAME state, in
the poses that rollouts of the literal random-action distribution pass through - the arm slews into the table, the cabinet and the drawer, so arm links lie beside box
faces and GJK's distance phase (oracle hull_box_gjk; narrowphase_coop) decides their contacts.  Every state is history-free on both sides (records only: no contact cache),
so the lists are the narrowphase of the pose and nothing else.  Matches environments.py:397, 409-411 (mesh colliders, for which Bullet runs GJK) through the oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

IDS = {'U': 'UR5PlayAbsRPY1Obj-v0', 'P': 'pandaPick-v0', 'V': 'pandaPlayAbsRPY1Obj-v0'}


@pytest.mark.parametrize('kind', ['U', 'P', 'V'])
def test_contact_lists_pose_by_pose(kind):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from gpu_debug import record_from_oracle
    import oracle
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    lib = oracle.load(f32=True)
    lib.rpo_gjk_stats.argtypes = [C.c_void_p, C.c_int]
    lib.rpo_epa_stats.argtypes = [C.c_void_p, C.c_int]
    st = (C.c_long * 8)()
    se = (C.c_long * 4)()
    epa_contacts = 0
    env = VecPlayEnv(IDS[kind], 2, seed=7)
    rng = np.random.default_rng(11)
    poses = gjk_contacts = mismatched = 0
    worst = 0.0
    for e in range(4):
        o = OracleEnv(kind, seed=7, env_index=e, f32=True)
        o.reset()
        for t in range(60):
            if kind == 'P':
                a = np.concatenate([np.array([-0.3, -0.3, -0.1]) + np.array([0.6, 0.6, 0.3]) * rng.random(3), (2 * rng.random(3) - 1) * 3, 2 * rng.random(1) - 1])
            else:
                a = np.concatenate([(2 * rng.random(6) - 1) * 6, 2 * rng.random(1) - 1])
            o.step(a)
            rec = record_from_oracle(o)
            env.set_state(torch.tensor(np.tile(rec, (2, 1))))
            dbg = env.debug_substep(0).numpy()
            o.set_state(o.get_state())                       # (empties the oracle's contact cache: both sides start this substep without history)
            lib.rpo_gjk_stats(st, 1)
            lib.rpo_epa_stats(se, 1)
            oc = o.contacts()
            lib.rpo_gjk_stats(st, 0)
            lib.rpo_epa_stats(se, 0)
            epa_contacts += se[2]
            o.set_state(o.get_state())
            poses += 1
            gjk_contacts += st[3]
            ncon = int(dbg[0])
            gc = dbg[16:16 + 9 * ncon].reshape(ncon, 9)
            ok = ncon == len(oc) and np.array_equal(gc[:, :2], oc[:, :2])
            if ok:
                err = max(float(np.abs(gc[:, 2:5] - oc[:, 2:5]).max(initial=0)), float(np.abs(gc[:, 8] - oc[:, 8]).max(initial=0)), 0.1 * float(np.abs(gc[:, 5:8] - oc[:, 5:8]).max(initial=0)))
                worst = max(worst, err)
                ok = err <= 5e-5
            if not ok:
                mismatched += 1
                if mismatched <= 3:
                    print('pose %d of env %d: device %d contacts, oracle %d' % (t, e, ncon, len(oc)))
                    print(np.round(gc, 5)); print(np.round(oc, 5))
    print('%s: %d poses, %d GJK contacts and %d EPA contacts (overlapping cores, round 5) in the oracle, %d poses with another list on the device, worst point / distance error of the rest %.1e' % (kind, poses, gjk_contacts, epa_contacts, mismatched, worst))
    assert gjk_contacts + epa_contacts >= 5, 'the rollouts no longer pass through GJK / EPA contacts'
    assert mismatched == 0


@pytest.mark.parametrize('kind', ['U', 'P', 'V'])
def test_block_against_the_robots_static_base_pose_by_pose(kind):
    """the robot's STATIC links (the Panda's link0 and its mount, the UR5's base: body 0, meshes of the robot's URDF all the same) meet a movable box with their convex
    hulls on both sides (oracle hull_link since round 6; the device's narrowphase asks hull_cnt > 0 and nothing about the body).  Until then the oracle gave those pairs to
    the box-box detector: a block thrown at the robot's base met a box there and a hull here, found by the lock-step test.  The block - resting,
    rolled, tumbled - on, over the edge of and beside the base's hull, history-free on both sides: same contact lists, points and distances."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from gpu_debug import record_from_oracle
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    env = VecPlayEnv(IDS[kind], 2, seed=7)
    o = OracleEnv(kind, seed=7, env_index=0, f32=True)
    o.reset()
    na = o.n_arm
    cols = o.collider_list()
    base = [c for c, d in enumerate(cols) if d['body'] == 0 and o.hull_vertices(c) is not None]
    assert base, 'the model has a static link with a hull'
    rng = np.random.default_rng(5)
    s0 = o.get_state()
    poses = with_base = mismatched = 0
    worst = 0.0
    for c in base:
        ctr, he = cols[c]['p'], cols[c]['he']
        for t in range(60):
            s = s0.copy()
            # the block's centre: over the top face, over its edges, beside the box - within a block's reach of the surface; orientation: flat, rolled, random
            off = (2 * rng.random(3) - 1) * (he + 0.03)
            off[2] = he[2] + 0.025 * (2 * rng.random() - 0.6)
            if t % 3 == 2:
                off = (2 * rng.random(3) - 1) * (he + 0.03)
            q = np.array([0.0, 0.0, 0.0, 1.0]) if t % 4 == 0 else rng.normal(size=4)
            q /= np.linalg.norm(q)
            s[2 * na:2 * na + 3] = ctr + off
            s[2 * na + 3:2 * na + 7] = q
            s[2 * na + 7:2 * na + 13] = 0.0
            o.set_state(s)
            rec = record_from_oracle(o)
            env.set_state(torch.tensor(np.tile(rec, (2, 1))))
            dbg = env.debug_substep(0).numpy()
            o.set_state(s)
            oc = o.contacts()
            o.set_state(s)
            ncon = int(dbg[0])
            gc = dbg[16:16 + 9 * ncon].reshape(ncon, 9)
            poses += 1
            with_base += int(any(int(r[1]) == c or int(r[0]) == c for r in oc))
            ok = ncon == len(oc) and np.array_equal(gc[:, :2], oc[:, :2])
            if ok and ncon:
                err = max(float(np.abs(gc[:, 2:5] - oc[:, 2:5]).max()), float(np.abs(gc[:, 8] - oc[:, 8]).max()), 0.1 * float(np.abs(gc[:, 5:8] - oc[:, 5:8]).max()))
                worst = max(worst, err)
                ok = err <= 5e-5
            if not ok:
                mismatched += 1
                if mismatched <= 3:
                    print('pose %d at collider %d: device %d contacts, oracle %d' % (t, c, ncon, len(oc)))
                    print(np.round(gc, 5)); print(np.round(oc, 5))
    print('%s: %d poses of the block at the robot\'s static hull(s) %s, %d with a contact against it in the oracle, %d with another list on the device, worst point / distance error %.1e'
          % (kind, poses, base, with_base, mismatched, worst))
    assert with_base >= 10, 'the poses no longer touch the base'
    assert mismatched == 0
