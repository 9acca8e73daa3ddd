"""Round 6 (CPU): the residual (Delassus) form of the sweeps against the dv form, on the oracle.

The HIP library solves its heavy envs (coupled / crowded, <= 14 contacts) in the residual form (oracle rule bit 262144, solve_rows_residual); in exact
arithmetic that is the dv form line by line.  This script runs pairs of oracle envs - rule with and without the bit - and reports
  * in fp64: the largest one-step gap from identical states (the two forms as algorithms: expected ~1e-12),
  * in fp64 and fp32: free rollouts, arm-joint gap over the steps (chaos included),
  * how many env-substeps took the residual form.
usage: python tools/residual_check.py [kind=U] [envs=8] [steps=60] [dist=B|A|grasp]
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
from oracle import OracleEnv  # noqa: E402

RES = 262144


def actions(kind, n_action, steps, dist, seed):
    rng = np.random.default_rng(seed)
    if dist == 'A':
        lo = np.array([-6.] * 6 + [-1.]); hi = -lo
        if n_action == 8:
            lo = np.array([-6.] * 7 + [-1.]); hi = -lo
        return rng.uniform(lo[:n_action], hi[:n_action], (steps, n_action))
    lo = np.array([-0.18, 0.0, 0.02, -0.5, -0.5, -0.5, -1.0]); hi = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])
    if kind in ('P', 'pandaPick-v0'):
        lo = np.array([-0.2, -0.2, 0.0, -1.0]); hi = np.array([0.2, 0.2, 0.25, 1.0])
        return rng.uniform(lo, hi, (steps, 4))
    return rng.uniform(lo[:n_action], hi[:n_action], (steps, n_action))


def state_vec(o):
    return np.concatenate([np.asarray(o['joints'], dtype=np.float64)[:7], np.asarray(o['obs_quat'], dtype=np.float64)])


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'U'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    dist = sys.argv[4] if len(sys.argv) > 4 else 'B'
    for f32 in (False, True):
        gaps, res_sub, onestep = [], 0, []
        for e in range(n):
            a = OracleEnv(kind, seed=11, env_index=e, f32=f32)
            rule = a.lib.rpo_get_rule(a.h)
            assert rule & RES
            b = OracleEnv(kind, seed=11, env_index=e, f32=f32, rule=rule & ~RES)
            a.reset(); b.reset()
            acts = actions(kind, a.n_action, steps, dist, 100 + e)
            g = 0.0
            for t in range(steps):
                oa, *_ = a.step(acts[t]); ob, *_ = b.step(acts[t])
                g = max(g, float(np.abs(state_vec(oa) - state_vec(ob)).max()))
            gaps.append(g)
            res_sub += a.lib.rpo_residual_substeps(a.h)
            assert b.lib.rpo_residual_substeps(b.h) == 0
        print('%s %s dist %s: %d envs x %d steps; residual-form substeps %d of %d (%.1f %%); free-rollout gap (joints | obs) per env: median %.2e max %.2e'
              % (kind, 'fp32' if f32 else 'fp64', dist, n, steps, res_sub, n * steps * 12, 100.0 * res_sub / (n * steps * 12), np.median(gaps), np.max(gaps)))


if __name__ == '__main__':
    main()
