#!/usr/bin/env python3
"""CPU (verdict round 5, item 3): how many passes of the damped-least-squares loop does an env's IK take per step?  The device's k_action runs the oracle's loop, four envs per
wave (one per DPP row), and a wave lasts as long as its slowest env; the launch as long as its slowest wave.  UR5 ids: calc_angles = 4 chained calls of at most 20 passes
(inverseKinematics.py:44-50); Panda ids: one call of at most 200 (environments.py:995-997).  A pass that ends at the residual test counts.
    python tools/ik_histogram.py [envs=64] [steps=60]   ->   profiles/r06_ik_histogram.txt"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
from oracle import OracleEnv  # noqa: E402


def actions(kind, n_action, steps, dist, rng):
    if dist == 'A':
        hi = np.array([6.0] * (n_action - 1) + [1.0])
        return rng.uniform(-hi, hi, (steps, n_action))
    lo = np.array([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0]); hi = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])      # bench.py's distribution B
    a = lo + (hi - lo) * rng.random((steps, 7))
    if kind == 'P':
        a[:, 0:3] = np.array([-0.18, -0.18, 0.0]) + np.array([0.36, 0.36, 0.2]) * rng.random((steps, 3))
    return a[:, :n_action] if n_action < 7 else a


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    out = []
    for kind, cap in (('U', 80), ('P', 200)):
        for dist in ('B', 'A'):
            def run(e):
                o = OracleEnv(kind, seed=21, env_index=e, f32=True)
                o.reset()
                acts = actions(kind, o.n_action, steps, dist, np.random.default_rng(500 + e))
                its = []
                for t in range(steps):
                    o.lib.rpo_ik_iterations(1)
                    o.step(acts[t])
                    its.append(o.lib.rpo_ik_iterations(1))
                return its
            with ThreadPoolExecutor(8) as ex:
                its = np.array(list(ex.map(run, range(n))))      # [env, step]
            flat = its.reshape(-1)
            waves = np.sort(its, axis=0)[::-1].reshape(n // 4, 4, steps) if False else its.reshape(n // 4, 4, steps).max(axis=1)      # four envs per wave (in index order)
            line = ('%s, distribution %s, %d envs x %d steps (fp32 oracle): passes per env-step p10 %d p50 %d p90 %d p99 %d max %d (cap %d: reached by %.1f %% of the env-steps); '
                    'per WAVE of four envs p50 %d p90 %d; a launch of 4096 envs has an env at the cap with probability %.3f per step'
                    % (kind, dist, n, steps, *np.percentile(flat, [10, 50, 90, 99]).astype(int), flat.max(), cap, 100.0 * (flat >= cap).mean(),
                       *np.percentile(waves.reshape(-1), [50, 90]).astype(int), 1.0 - (1.0 - (flat >= cap).mean()) ** 4096))
            print(line, flush=True)
            out.append(line)
    with open(os.path.join(REPO, 'profiles', 'r06_ik_histogram.txt'), 'w') as f:
        f.write('\n'.join(out) + '\n')


if __name__ == '__main__':
    main()
