#!/usr/bin/env python3
"""CPU: does the FROZEN REFERENCE STEP also lose pandaPick blocks through the reference's 0.2 mm ground plate (scenes.py:8-21)?

The scenario of tests/test_gpu_parity.py::test_config_panda_pick_4096_envs (closed fingers driven onto the block lying on the plate, then lifted),
on `--envs` sampled env indices of that test's seed, once with the fast model's fp64 oracle (mode A) and once with the frozen reference step (mode B).
An env counts as lost when its block ends below the plate (z < -0.07 - 0.03).
    python tools/plate_tunnelling.py [--envs 256] [--threads 8]"""
import argparse
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))


def run(args):
    e, ref = args
    from oracle import OracleEnv
    o = OracleEnv('P', seed=12, env_index=e, bullet_ref=ref)
    obs = o.reset()
    zmin = 1.0
    for t in range(60):
        a = np.zeros(7)
        a[0:3] = obs['achieved_goal'][0:3]
        a[2] += 0.0 if t < 30 else 0.15
        a[6] = -1.0 if t < 15 else 1.0
        obs = o.step(a)[0]
        zmin = min(zmin, obs['achieved_goal'][2])
    return e, ref, float(obs['achieved_goal'][2]), float(zmin)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=256)
    ap.add_argument('--threads', type=int, default=8)
    a = ap.parse_args()
    idx = np.unique(np.linspace(0, 4095, a.envs).astype(int))
    jobs = [(int(e), ref) for ref in (False, True) for e in idx]
    with ProcessPoolExecutor(a.threads) as ex:
        res = list(ex.map(run, jobs, chunksize=8))
    for ref in (False, True):
        z = np.array([r[2] for r in res if r[1] == ref])
        zm = np.array([r[3] for r in res if r[1] == ref])
        lost = z < -0.10
        print('%s: %d envs, lost through the plate %d (%.2f %%), lowest block z over the rollout %.4f, lifted > 5 cm: %d' % (
            'frozen reference step (mode B)' if ref else 'fast model, fp64 oracle (mode A)', len(z), int(lost.sum()), 100.0 * lost.mean(), zm.min(), int((z > 0.0).sum())))
        if lost.any():
            print('   lost env indices:', [r[0] for r in res if r[1] == ref and r[2] < -0.10][:20])


if __name__ == '__main__':
    main()
