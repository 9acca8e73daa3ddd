"""The frozen Bullet-like reference step (oracle/rp_bullet_ref.c, "mode B"): its narrowphase pieces against independent checks,
its manifold life cycle, and - with every difference switched off - its agreement with the fast model's oracle.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle
from oracle import OracleEnv

DP = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(DP)


@pytest.fixture(scope='module')
def lib():
    lb = oracle.load(bullet_ref=True)
    lb.rpo_ref_gjk_epa_boxes.argtypes = [DP] * 6 + [C.c_double] + [DP] * 4
    lb.rpo_ref_box_box.argtypes = [DP] * 8
    lb.rpo_ref_collider_distance.argtypes = [C.c_void_p, C.c_int, C.c_int, DP, DP, DP, DP]
    lb.rpo_ref_manifolds.argtypes = [C.c_void_p, DP, C.c_int]
    return lb


def rot(rng):
    q = rng.normal(size=4)
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def corners(c, R, h):
    s = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float)
    return (s * h) @ R.T + c


def random_boxes(rng):
    ha, hb = 0.02 + 0.1 * rng.random(3), 0.02 + 0.1 * rng.random(3)
    Ra, Rb = rot(rng), rot(rng)
    cb = rng.normal(size=3)
    cb *= (0.05 + 0.3 * rng.random()) / np.linalg.norm(cb)
    return [np.ascontiguousarray(x, dtype=np.float64) for x in (np.zeros(3), Ra, ha, cb, Rb, hb)]


def test_gjk_and_epa_on_random_boxes(lib):
    """witness points lie on the boxes, (pA - pB).n is the reported distance, and n is a separating direction when apart (the
    boxes' corners are rounded by the 1 mm margin, hence the 1 mm allowance); when the boxes overlap the EPA depth agrees with
    the least-penetration axis of the SAT box-box test to the same allowance"""
    rng = np.random.default_rng(0)
    seen = [0, 0, 0]
    for _ in range(1500):
        b = random_boxes(rng)
        dist, pa, pb, n = np.zeros(1), np.zeros(3), np.zeros(3), np.zeros(3)
        r = lib.rpo_ref_gjk_epa_boxes(*[_p(x) for x in b], 0.001, _p(dist), _p(pa), _p(pb), _p(n))
        seen[r] += 1
        assert r in (1, 2)
        ca, Ra, ha, cb, Rb, hb = b
        assert np.all(np.abs(Ra.T @ (pa - ca)) <= ha + 1e-7) and np.all(np.abs(Rb.T @ (pb - cb)) <= hb + 1e-7)
        assert abs((pa - pb) @ n - dist[0]) < 1e-7 and abs(np.linalg.norm(n) - 1) < 1e-9
        VA, VB = corners(ca, Ra, ha), corners(cb, Rb, hb)
        nrm, out = np.zeros(3), np.zeros(16)
        k = lib.rpo_ref_box_box(*[_p(x) for x in b], _p(nrm), _p(out))
        if r == 1:                 # apart: n separates, up to the rounding of the corners
            assert (VA @ n).min() - pa @ n > -1.1e-3 and pb @ n - (VB @ n).max() > -1.1e-3
            if dist[0] > 7.5e-4:       # (the rounded boxes can be apart while the sharp ones overlap at an edge or corner: < (sqrt(3) - 1) mm)
                assert k == 0
        else:
            assert k >= 1
            # EPA depth of the rounded boxes vs the deepest SAT point (the detector prefers face axes over edge axes by 5 %)
            assert abs(-dist[0] - out[3::4][:k].max()) < 2.5e-3 + 0.06 * out[3::4][:k].max()
    assert seen[1] > 200 and seen[2] > 200


def test_box_box_detector_against_the_fast_models_sat(lib):
    """the dBoxBox2 restatement and the fast model's own box_box (at margin 0) are two independent SAT + clipping codes: same
    normal, same deepest penetration, every point inside both boxes' slabs along the normal"""
    la = oracle.load()
    rng = np.random.default_rng(1)
    hits = ordered = 0
    for _ in range(1500):
        b = random_boxes(rng)
        nrm, out, outA = np.zeros(3), np.zeros(16), np.zeros(28)
        k = lib.rpo_ref_box_box(*[_p(x) for x in b], _p(nrm), _p(out))
        kA = la.rpo_box_box(*[_p(x) for x in b], 0.0, _p(outA))
        assert (k > 0) == (kA > 0) or (k > 0 and kA == 0 and out[3::4][:k].max() < 1e-6) or (k == 0 and kA > 0 and -outA[6::7][:kA].min() < 1e-6)
        if k == 0 or kA == 0:
            continue
        hits += 1
        assert 1 <= k <= 4
        nA = outA[3:6]
        dA = -outA[6::7][:kA].min()
        d = out[3::4][:k].max()
        assert (out[3::4][:k] >= 0).all()
        if abs(d - dA) > 1e-6:          # the two codes may pick different axes only when two axes tie within the 5 % edge preference
            assert abs(d - dA) < 0.06 * max(d, dA) + 1e-5
        else:
            assert nrm @ nA > 0.999
            # ... and under RPO_RULE_ODEORDER the fast model emits a face contact's points in the detector's ORDER (the order of the solver's rows):
            # same count and same depths (so nothing was thinned differently) -> the same points, one by one (the fast model's point is the midpoint,
            # the detector's lies on B: half a depth apart along the normal)
            if k == kA and np.allclose(np.sort(out[3::4][:k]), np.sort(-outA[6::7][:kA]), atol=1e-9):
                for i in range(k):
                    pB = outA[7 * i:7 * i + 3] - 0.5 * outA[7 * i + 6] * nA
                    assert np.abs(pB - out[4 * i:4 * i + 3]).max() < 1e-8, (i, k)
                    ordered += 1
    assert hits > 200 and ordered > 150


def test_hull_and_sphere_colliders_distance(lib):
    """support-mapped colliders of the UR5 play scene: the globe (sphere) against the block matches the closed form; an arm link's hull
    is never farther from the table than its bounding box and never closer than its hull vertices allow"""
    env = OracleEnv('U', seed=1, env_index=0, bullet_ref=True)
    env.reset()
    s = env.get_state()
    na = env.n_arm
    s[2 * na:2 * na + 3] = [-0.25, 0.45, 0.32]              # block 2.5 cm above the globe (sphere r 0.03 at z 0.24), axis aligned
    s[2 * na + 3:2 * na + 7] = [0, 0, 0, 1]
    env.set_state(s)
    dist, pa, pb, n = np.zeros(1), np.zeros(3), np.zeros(3), np.zeros(3)
    r = lib.rpo_ref_collider_distance(env.h, 51, 45, _p(dist), _p(pa), _p(pb), _p(n))      # block (51) vs globe (45)
    assert r == 1 and abs(dist[0] - (0.32 - 0.025 - 0.27)) < 1e-8 and abs(n[2] - 1) < 1e-9
    # forearm hull (collider 3) against the table top (46) at the reset pose
    r = lib.rpo_ref_collider_distance(env.h, 3, 46, _p(dist), _p(pa), _p(pb), _p(n))
    assert r == 1 and dist[0] > 0.05
    flags = env.lib.rpo_get_ref_flags(env.h)
    env.lib.rpo_set_ref_flags(env.h, flags & ~1)             # the same pair with the link as its bounding box
    d_obb = np.zeros(1)
    lib.rpo_ref_collider_distance(env.h, 3, 46, _p(d_obb), _p(pa), _p(pb), _p(n))
    assert d_obb[0] <= dist[0] + 2e-3                        # the box encloses the hull (up to the hull's 1 mm margin)


def manifold_points(lib, env):
    buf = np.zeros(17 * 256)
    n = lib.rpo_ref_manifolds(env.h, _p(buf), 256)
    return buf[:17 * n].reshape(n, 17)


def test_manifold_life_cycle(lib):
    """block at rest on the table: four cached points that age from step to step and carry their impulses; lifted by less than the
    pair's breaking threshold (1.22 mm: the block's relative threshold) they stay, now at a positive distance, although the box-box
    detector adds no points to boxes that are apart; lifted by more they are dropped"""
    env = OracleEnv('U', seed=2, env_index=0, bullet_ref=True)
    env.reset()
    for _ in range(5):
        env.lib.rpo_substep(env.h)
    pts = manifold_points(lib, env)
    blk = pts[(pts[:, 2] == 51) & (pts[:, 3] == 46)]
    assert len(blk) == 4 and (blk[:, 15] > 50).all() and abs(blk[0, 16] - 0.02 * np.linalg.norm([0.05, 0.025, 0.025])) < 1e-9
    assert (blk[:, 13] < 1e-4).all() and blk[:, 14].sum() == pytest.approx(0.3 * 9.8 / 300, rel=0.05)     # impulses carry the weight
    s = env.get_state()
    na = env.n_arm
    z = s[2 * na + 2]
    for lift, keep in ((0.0010, True), (0.0020, False)):
        s2 = s.copy()
        s2[2 * na + 2] = z + lift
        s2[2 * na + 7:2 * na + 13] = 0
        env.set_state(s2)
        env.lib.rpo_substep(env.h)
        pts = manifold_points(lib, env)
        blk = pts[(pts[:, 2] == 51) & (pts[:, 3] == 46)]
        assert (len(blk) == 4) == keep, (lift, len(blk))
        if keep:
            assert (blk[:, 13] > 5e-4).all()


def test_with_the_remaining_differences_off_it_is_the_fast_model():
    """the fast model shares the soft gripper contacts, the friction skip and (since round 3) the row order in alternating direction and the
    violated-only limit rule with the reference step; with the differences that remain switched off (no hulls, no persistence, midpoint lever
    arms, no anchors, no torsional friction) the reference step against the fast model's oracle at contact margin 0: no contacts at all (UR5Reach) - identical; a block on a plane
    (pandaPick) - the same trajectories up to what is left (GJK instead of closed forms, no cap, the detector's own point culling)"""
    rng = np.random.default_rng(3)
    for kind, tol in (('R', 1e-10), ('P', 2e-4)):
        a = OracleEnv(kind, seed=5, env_index=1, margin=0.0)
        b = OracleEnv(kind, seed=5, env_index=1, bullet_ref=True, ref_flags=sum(oracle.REF_FLAGS[k] for k in ('soft', 'fricskip', 'order', 'limit')))
        oa, ob = a.reset(), b.reset()
        np.testing.assert_allclose(ob['obs_quat'], oa['obs_quat'], atol=tol, rtol=0)
        for t in range(40):
            act = np.concatenate([[-0.1, -0.1, 0.05] + np.array([0.2, 0.2, 0.15]) * rng.random(3), rng.random(3) - 0.5, [2 * rng.random() - 1]])
            oa, ob = a.step(act)[0], b.step(act)[0]
            np.testing.assert_allclose(ob['obs_quat'][:3], oa['obs_quat'][:3], atol=max(tol, 1e-9), rtol=0, err_msg='%s step %d' % (kind, t))
            np.testing.assert_allclose(a.get_state()[:a.n_arm], b.get_state()[:a.n_arm], atol=tol * 5, rtol=0)


def test_on_ur5reach_the_fast_model_is_the_reference_step():
    """what the shipped model took from the reference step in round 3 - row order and limit rule, hull vertices against static boxes, the box-box detector's
    point order, per-body lever arms, torsional friction, persistent manifolds - leaves nothing on UR5Reach-v0: the frozen reference step (default flags) and
    the fast model walk the same 200-step trajectories to 1e-5, gripper-on-table contacts included; and so do the two without their contact caches (reference
    step with persistence switched off, fast model with RPO_RULE_PERSIST off) to 1e-6: the cache is one difference, switched on both sides alike
    (tools/model_divergence.py: rows 'A default', 'B -persist' / 'A -persist')"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import model_divergence as md
    contact_substeps = 0
    for e in (0, 4, 7, 9):
        acts = md.random_actions('R', 200, np.random.default_rng(1000 + e))
        s0 = None
        for flags, rule, tol in ((oracle.REF_DEFAULT, None, 1e-5), (oracle.REF_DEFAULT & ~oracle.REF_FLAGS['persist'], 503 & ~256, 1e-6)):
            ref = OracleEnv('R', seed=77, env_index=e, bullet_ref=True, ref_flags=flags)
            if s0 is None:
                ref.reset()
                s0 = ref.get_state()
                ref = OracleEnv('R', seed=77, env_index=e, bullet_ref=True, ref_flags=flags)      # (both sides start without contact history)
            qb, _ = md.rollout(ref, 'R', 'random', 200, acts, s0)
            a = OracleEnv('R', seed=77, env_index=e, **({} if rule is None else dict(rule=rule)))
            qa, _ = md.rollout(a, 'R', 'random', 200, acts, s0)
            assert np.abs(qa - qb).max() < tol, (e, flags, np.abs(qa - qb).max())
            contact_substeps += a.lib.rpo_contact_substeps(a.h)
    assert contact_substeps > 200


def test_default_flags_are_everything_but_warm_starting():
    env = OracleEnv('U', seed=0, env_index=0, bullet_ref=True)
    assert env.lib.rpo_get_ref_flags(env.h) == oracle.REF_DEFAULT == sum(v for k, v in oracle.REF_FLAGS.items() if k != 'warm')
    assert oracle.REF_DEFAULT == 255 + 512


FROZEN_SHA256 = {       # recorded in DESIGN.md section 2 ("the freeze"); the file last changed in commit 2bd34cf (round 2, before any of that round's kernel work)
    'oracle/rp_bullet_ref.c': '35ea2de6f7c903a6ff344558d48a6fc691e433c5366499d361ccf2e274548d0c',
    'oracle/generated/rp_hulls_gen.h': '1c2e6a61327da163d5c7e7c90b8ce7a1043309625e33e29caf8787881fc01a96',
}


def test_the_reference_step_is_frozen():
    """kernel work may not touch the frozen reference step: an edit of rp_bullet_ref.c (or of the hull tables it includes) fails here and has to be a
    visible, justified event - correct a recollection of Bullet, say so in DESIGN.md section 2, record the new hash there and here"""
    import hashlib
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel, want in FROZEN_SHA256.items():
        got = hashlib.sha256(open(os.path.join(repo, rel), 'rb').read()).hexdigest()
        assert got == want, '%s changed (sha256 %s): the reference step is frozen, see DESIGN.md section 2' % (rel, got)
