"""HIP path (through the C ABI) vs the FROZEN Bullet-like reference step (oracle/rp_bullet_ref.c, librp_oracle_bullet.so, default flags).

The other GPU parity tests hold the device to the fast model's own oracle - the model the kernels implement.  This file holds it to the
independent restatement of Bullet's step instead (PARITY UNPINNED like all physics here: PyBullet itself is absent; rp_bullet_ref.c is the only
independent evidence there is).  Measure: north_star's, max over steps and the ARM'S OWN joints (6 UR5 / 7 Panda) of |q_hip - q_ref| / max(1, |q_ref|)
per env, 200 steps, identical post-reset states and actions, tolerance 1e-3.

What the shipped model shares with the reference step since round 3: the non-contact row order walked in alternating direction, joint-limit rows
only while violated (erp 0.2), soft gripper contacts, the friction skip, hull vertices of arm links against static boxes, the box-box detector's point
order, per-body lever arms, torsional friction, persistent manifolds.  What it does not: GJK / EPA beyond vertex-on-face, one manifold per collider pair
 - neither matters on these two ids: the fp64 oracles of the two models agree to 6e-6 on UR5Reach.  Hence
  * both ids must meet 1e-3 in every env for as long as the rollout is free motion - until the first substep in which either the reference step
    or the fast model (an fp64 CPU follower of the same env) has a contact row: the fingers of both grippers reach the ground plate at
    z = -0.07 in many rollouts;
  * after that the measured figure is reported and bounded: median over the envs 2e-4 (measured: R 4e-6, Q 3e-6; this round's first model 5e-4), every
    env 1e-2 (measured max: R 2.4e-3 - what fp32 makes of a pad's impact -, Q 1.5e-4).
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

IDS = {'R': 'UR5Reach-v0', 'Q': 'pandaReach-v0'}
TOL = 1e-3


def reach_actions(steps, n, seed):
    rng = np.random.default_rng(seed)
    a = np.array([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0]) + np.array([0.36, 0.3, 0.25, 1.0, 1.0, 1.0, 2.0]) * rng.random((steps, n, 7))
    a[..., 0:3] = np.array([-0.18, -0.18, 0.0]) + np.array([0.36, 0.36, 0.2]) * rng.random((steps, n, 3))
    return a


@pytest.mark.parametrize('kind', ['R', 'Q'])
def test_hip_vs_frozen_reference_step(kind):
    import os
    import sys
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from gpu_debug import record_from_oracle
    n, steps = 16, 200
    n_main = 6 if kind == 'R' else 7
    env = VecPlayEnv(IDS[kind], n, seed=21)
    env.reset()
    refs = [OracleEnv(kind, seed=21, env_index=e, bullet_ref=True) for e in range(n)]
    fast = [OracleEnv(kind, seed=21, env_index=e) for e in range(n)]
    # ... and four fp32 runs of the fast model per env, started 1e-5 .. 2e-5 off in the arm joints (tests/tolerances.py): where the IK's position-only stopping test leaves the
    # orientation unconverged, the joint targets depend on the iteration it stopped at, the next step's IK starts from other measured joints, and a difference of 1e-5
    # grows fivefold per step for a few steps (round 4: env 6 of UR5Reach, steps 89 - 93, tools/dbg_ref_R.py - the device, another evaluation order, drew 2e-3 where the fp32
    # CPU run stayed at 6e-7).  Such an env is held to three times what those runs make of it
    nudged = [[OracleEnv(kind, seed=21, env_index=e, f32=True) for _ in range(4)] for e in range(n)]
    for o, f, nd in zip(refs, fast, nudged):
        o.reset()
        f.reset()
        f.set_state(o.get_state())
        for k, x in enumerate(nd):
            x.reset()
            s0 = o.get_state().copy()
            s0[:x.n_arm] *= 1.0 + (1e-5 if k % 2 == 0 else -1e-5) * (1 + k // 2)
            x.set_state(s0)
    c0 = [(o.lib.rpo_contact_substeps(o.h), f.lib.rpo_contact_substeps(f.h)) for o, f in zip(refs, fast)]
    env.set_state(torch.tensor(np.stack([record_from_oracle(o) for o in refs])))     # both start from the reference step's post-reset state
    acts = reach_actions(steps, n, 3)
    d_free, d_all, g_free = np.zeros(n), np.zeros(n), np.zeros(n)
    touched = np.zeros(n, bool)
    first = np.full(n, steps)
    for t in range(steps):
        obs, r, done, info = env.step(torch.tensor(acts[t], dtype=torch.float32))
        q = env.get_state()[:, :n_main].cpu().numpy()
        for e, o in enumerate(refs):
            a = acts[t, e].astype(np.float32).astype(np.float64)
            o.step(a)
            fast[e].step(a)
            for x in nudged[e]:
                x.step(a)
            if not touched[e] and (o.lib.rpo_contact_substeps(o.h), fast[e].lib.rpo_contact_substeps(fast[e].h)) != c0[e]:
                touched[e] = True
                first[e] = t
            qo = o.get_state()[:n_main]
            d = float((np.abs(q[e] - qo) / np.maximum(1.0, np.abs(qo))).max())
            d_all[e] = max(d_all[e], d)
            if not touched[e]:
                d_free[e] = max(d_free[e], d)
                g_free[e] = max(g_free[e], max(float((np.abs(x.get_state()[:n_main] - qo) / np.maximum(1.0, np.abs(qo))).max()) for x in nudged[e]))
        assert int((info['status'] & 1).sum()) == 0
    bound = np.maximum(TOL, 3 * g_free)
    print('%s vs the frozen reference step, %d envs x %d steps, arm joints: contact-free part max %.2e median %.2e (%d envs within 1e-3; the nudged fp32 CPU runs: max %.2e); '
          'whole rollout max %.2e median %.2e; %d envs touched something (first at step %s)' % (kind, n, steps, d_free.max(), np.median(d_free), int((d_free <= TOL).sum()), g_free.max(),
                                                                                                d_all.max(), np.median(d_all), int(touched.sum()), int(first.min()) if touched.any() else '-'))
    assert (d_free <= bound).all() and (d_free <= TOL).sum() >= n - 1, (d_free, g_free)
    assert (~touched).sum() >= 2 and (d_all[~touched] <= bound[~touched]).all()      # some envs stay free for all 200 steps
    assert np.median(d_all) <= 2e-4 and (d_all <= 1e-2).all(), d_all       # fp32 against fp64 across a pad's impacts: DESIGN.md section 2


# ---- the ids whose arm touches the scene (round 4): the headline playroom id and pandaPick.  Both models always have contact rows there (the block rests on the table,
# the drawer on its rails), so "free motion" is judged by geometry: the prefix of the rollout in which every arm collider's AABB stays clear of every other collider's
# by 5 mm in the reference step's state.
IDS_CONTACT = {'U': 'UR5PlayAbsRPY1Obj-v0', 'P': 'pandaPick-v0'}
# bounds one notch above the measured values (round 4, GJK on: printed by the test)
BOUNDS_CONTACT = {'U': dict(median=6e-4, within=8), 'P': dict(median=3e-5, within=15)}      # measured: U median 4.2e-4, 9 of 16; P median 5.4e-6, 16 of 16, max 5.6e-5 (round 5, the expanding polytope; round 4: 1.2e-5, 15 of 16)


def arm_clear(o, pairs=None, gap=0.005):
    """no baked candidate pair with an arm collider in it has its AABBs within `gap` of each other"""
    cols = o.collider_list()
    n_arm = o.n_arm
    lo, hi, arm = [], [], []
    for c in cols:
        R, he = np.abs(c['R']), c['he']
        e = R @ he if c['type'] == 0 else np.full(3, he[0])
        lo.append(c['p'] - e); hi.append(c['p'] + e); arm.append(1 <= c['body'] <= n_arm)
    lo, hi, arm = np.array(lo), np.array(hi), np.array(arm)
    pr = np.array([(a, b) for a, b in (pairs if pairs is not None else o.pair_list()) if arm[a] or arm[b]])
    if pr.size == 0:
        return True
    sep = np.maximum(lo[pr[:, 0]] - hi[pr[:, 1]], lo[pr[:, 1]] - hi[pr[:, 0]]).max(axis=1)      # > 0: the AABBs are apart along some axis
    return bool((sep > gap).all())


@pytest.mark.parametrize('kind', ['U', 'P'])
def test_hip_vs_frozen_reference_step_with_arm_contacts(kind):
    """Device (default model) against the frozen reference step where the arm does touch the scene: 16 envs x 200 steps from the reference step's post-reset
    state, tools/model_divergence.py's random actions (environments.py:485-490).  Strict (1e-3, every env) while the arm is clear of everything; over the whole
    rollout the median and the number of envs within 1e-3 are bounded one notch above what was measured - the CPU twin of this test (tests/test_fidelity_table.py)
    shows that each of the contact model's features (hull vertices, contact cache, GJK beside the face) is needed to stay inside."""
    import os
    import sys
    from oracle import OracleEnv
    from roboticsplayroompybullet_amd import VecPlayEnv
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from gpu_debug import record_from_oracle
    import model_divergence as md
    n, steps = 16, 200
    n_main = 6 if kind == 'U' else 7
    env = VecPlayEnv(IDS_CONTACT[kind], n, seed=77)
    env.reset()
    refs = [OracleEnv(kind, seed=77, env_index=e, bullet_ref=True) for e in range(n)]
    for o in refs:
        o.reset()
    env.set_state(torch.tensor(np.stack([record_from_oracle(o) for o in refs])))
    # the fast model's own fp32 CPU oracle on the same rollout: what the MODEL parts from the reference step by, in one more evaluation order - the device's counts are held
    # against its counts (thresholds relative to the CPU runs, not to measured constants alone)
    twins = [OracleEnv(kind, seed=77, env_index=e, f32=True) for e in range(n)]
    for o, r in zip(twins, refs):
        o.set_state(r.get_state())
    acts = np.stack([md.random_actions(kind, steps, np.random.default_rng(1000 + e)) for e in range(n)], axis=1)
    d_free, d_all, d_twin = np.zeros(n), np.zeros(n), np.zeros(n)
    pairs = refs[0].pair_list()
    clear = np.array([arm_clear(o, pairs) for o in refs])
    first = np.where(clear, steps, 0)
    for t in range(steps):
        obs, r, done, info = env.step(torch.tensor(acts[t], dtype=torch.float32))
        q = env.get_state()[:, :n_main].cpu().numpy()
        for e, o in enumerate(refs):
            o.step(acts[t, e].astype(np.float32).astype(np.float64))
            if clear[e] and not arm_clear(o, pairs):
                clear[e] = False
                first[e] = t
            qo = o.get_state()[:n_main]
            d = float((np.abs(q[e] - qo) / np.maximum(1.0, np.abs(qo))).max())
            d_all[e] = max(d_all[e], d)
            twins[e].step(acts[t, e].astype(np.float32).astype(np.float64))
            d_twin[e] = max(d_twin[e], float((np.abs(twins[e].get_state()[:n_main] - qo) / np.maximum(1.0, np.abs(qo))).max()))
            if clear[e]:
                d_free[e] = max(d_free[e], d)
        assert int((info['status'] & 1).sum()) == 0
    within = int((d_all <= TOL).sum())
    print('%s vs the frozen reference step, %d envs x %d steps, arm joints: while the arm is clear of the scene (first approach at steps p50 %d, min %d) max %.2e; whole rollout '
          'median %.2e p75 %.2e max %.2e, %d envs within 1e-3' % (kind, n, steps, int(np.median(first)), int(first.min()), d_free.max(), np.median(d_all), np.percentile(d_all, 75), d_all.max(), within))
    assert (d_free <= TOL).all(), d_free
    b = BOUNDS_CONTACT[kind]
    within_twin = int((d_twin <= TOL).sum())
    print('    the fast model\'s fp32 CPU oracle on the same rollout: median %.2e, %d envs within 1e-3' % (np.median(d_twin), within_twin))
    assert np.median(d_all) <= b['median'] and within >= b['within'], (np.median(d_all), within, d_all)
    assert within >= within_twin - 2 and np.median(d_all) <= max(3 * np.median(d_twin), 1e-5), (within, within_twin, np.median(d_all), np.median(d_twin))
