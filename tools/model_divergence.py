#!/usr/bin/env python3
"""How far is the fast model (the one the HIP kernels implement; oracle/rp_oracle.c "mode A") from the frozen Bullet-like reference
step (oracle/rp_bullet_ref.c "mode B")?  CPU only, fp64 both.

For every BASELINE config's env kind and two action scenarios - `random` (bench.py's distribution B) and `grasp` (drive the open gripper
onto the block, close, lift: the contact-rich case) - N envs are reset once by mode B, both models start from that state, get the same
200 actions, and the divergence is reported as
    arm     max over steps and the arm's own joints (6 UR5 / 7 Panda) of |q_A - q_B| / max(1, |q_B|)
    joints  the same over all dofs, the gripper's auxiliary joints included (north_star's "relative joint-state divergence")
    block   max over steps of |block position A - B| in metres
as median / 90th percentile / max over the envs.  Rows:
    A default      the shipped model: Bullet's row order and limit rule, hull vertices against static and movable boxes, GJK's distance phase where the deepest vertex lies
                   beside the face (since round 4), arm boxes overlap-only, box-box points in the detector's order, per-body lever arms, torsional friction rows, persistent manifolds (a uniform margin `A m=...` switches those off: it is a study
                   of the stateless contacts)
    A round 2      last round's model (rule 0); A -x: the shipped model with one of its round-3 features off
    A m=...        the same model with one uniform contact margin (0, 5 mm = round 1's choice, 20 mm = gContactBreakingThreshold taken absolute)
    B -flag        mode B with one of its differences switched off (what each Bullet feature is worth, measured inside mode B)
    B +warm        mode B with warm starting on
Writes a markdown table to stdout (DESIGN.md section 2 quotes it) and profiles/<tag>_model_divergence.json.

    python tools/model_divergence.py [--envs 12] [--steps 200] [--tag r02]
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
import oracle  # noqa: E402
from oracle import OracleEnv  # noqa: E402

LO = np.array([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0])
HI = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])


def random_actions(kind, steps, rng):
    a = LO + (HI - LO) * rng.random((steps, 7))
    if kind in ('R', 'P'):
        a[:, 0:3] = np.array([-0.18, -0.18, 0.0]) + np.array([0.36, 0.36, 0.2]) * rng.random((steps, 3))
    return a


def grasp_action(kind, obs, t):
    a = np.zeros(7)
    blk = obs['achieved_goal'][:3]
    a[0:3] = blk
    if kind == 'U':
        a[2] = 0.02 if t < 60 else 0.15
        a[6] = -1.0 if t < 30 else 1.0
    else:
        a[2] = blk[2] + (0.0 if t < 60 else 0.15)
        a[6] = -1.0 if t < 30 else 1.0
    return a


def rollout(env, kind, scenario, steps, acts, state0):
    env.set_state(state0)
    obs = env.calc_state()
    na = env.n_arm
    q, blk = [], []
    for t in range(steps):
        a = acts[t] if scenario == 'random' else grasp_action(kind, obs, t)
        obs = env.step(a)[0]
        s = env.get_state()
        q.append(s[:na].copy())
        blk.append(s[2 * na:2 * na + 3].copy() if env.nv > na else np.zeros(3))
    return np.array(q), np.array(blk)


def divergence(qa, ba, qb, bb, n_main):
    """(arm joints proper: 6 UR5 / 7 Panda, all dofs incl. the gripper's, block) - max over the rollout"""
    rel = np.abs(qa - qb) / np.maximum(1.0, np.abs(qb))
    return float(rel[:, :n_main].max()), float(rel.max()), float(np.linalg.norm(ba - bb, axis=1).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=12)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--tag', default='r03')
    ap.add_argument('--kinds', default='R,Q,U,P')
    args = ap.parse_args()
    D = oracle.REF_DEFAULT
    # the shipped model's rule bits (rp_oracle.c RPO_RULE_*): 1 Bullet's row order, 2 violated-only limits, 4 hull vertices against static boxes, 16 arm boxes overlap-only,
    # 32 box-box points in the detector's order, 64 per-body lever arms, 128 torsional friction rows, 256 persistent manifolds, 512 hull vertices against movable boxes too, 1024 GJK's distance phase beside the face
    ALL = 503 | 512 | 1024
    variants = [('A default (shipped)', dict()), ('A round 2 (rule 0)', dict(rule=0)), ('A -order -limit', dict(rule=ALL & ~3)), ('A -hull', dict(rule=ALL & ~4)),
                ('A -boxorder', dict(rule=ALL & ~32)), ('A -lever', dict(rule=ALL & ~64)), ('A -spin', dict(rule=ALL & ~128)), ('A -persist', dict(rule=ALL & ~256)), ('A -hullmov', dict(rule=ALL & ~512)), ('A -gjk (= RP_CFG_OBB_EDGES, round 3\'s default)', dict(rule=ALL & ~1024)), ('A -limit (= RP_CFG_SPECULATIVE_LIMITS)', dict(rule=ALL & ~2)),
                ('A -persist -hullmov -boxorder -lever -spin -gjk (round 3\'s first model)', dict(rule=23)), ('A -persist -boxoverlap', dict(rule=ALL & ~256 & ~16)),
                ('A m=0', dict(margin=0.0)), ('A m=5mm', dict(margin=0.005)), ('A m=20mm', dict(margin=0.02))]
    for name, bit in oracle.REF_FLAGS.items():
        if name == 'warm':
            variants.append(('B +warm', dict(bullet_ref=True, ref_flags=D | bit)))
        else:
            variants.append(('B -%s' % name, dict(bullet_ref=True, ref_flags=D & ~bit)))
    results = {}
    for kind in args.kinds.split(','):
        for scenario in (('random',) if kind in ('R', 'Q') else ('random', 'grasp')):
            rows = {v[0]: [] for v in variants}
            for e in range(args.envs):
                ref = OracleEnv(kind, seed=77, env_index=e, bullet_ref=True)
                ref.reset()
                state0 = ref.get_state()
                goal = ref.calc_state()['desired_goal']
                acts = random_actions('R' if kind == 'Q' else kind, args.steps, np.random.default_rng(1000 + e))
                qb, bb = rollout(ref, kind, scenario, args.steps, acts, state0)
                for name, kw in variants:
                    env = OracleEnv(kind, seed=77, env_index=e, **kw)
                    env.lib.rpo_set_goal(env.h, oracle._d(goal)[1]) if hasattr(env.lib, 'rpo_set_goal') else None
                    qa, ba = rollout(env, kind, scenario, args.steps, acts, state0)
                    rows[name].append(divergence(qa, ba, qb, bb, 6 if kind in ('R', 'U') else 7))
            results['%s/%s' % (kind, scenario)] = {k: {'arm': [r[0] for r in v], 'joints': [r[1] for r in v], 'block': [r[2] for r in v]} for k, v in rows.items()}
    os.makedirs(os.path.join(REPO, 'profiles'), exist_ok=True)
    out = {'envs': args.envs, 'steps': args.steps, 'reference': 'oracle/rp_bullet_ref.c, default flags %d' % D, 'results': results}
    json.dump(out, open(os.path.join(REPO, 'profiles', '%s_model_divergence.json' % args.tag), 'w'), indent=1)

    def stat(v):
        v = np.array(v)
        return '%.1e / %.1e / %.1e' % (np.median(v), np.percentile(v, 90), v.max())

    print('| config / scenario | model | arm joints: median / p90 / max | all dofs (with the gripper\'s): median / p90 / max | block [m]: median / p90 / max |')
    print('|---|---|---|---|---|')
    for key, rows in results.items():
        for name, r in rows.items():
            print('| %s | %s | %s | %s | %s |' % (key, name, stat(r['arm']), stat(r['joints']), stat(r['block'])))


if __name__ == '__main__':
    main()
