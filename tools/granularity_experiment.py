#!/usr/bin/env python3
"""Verdict item 6 of round 3: what is the reference step's MANIFOLD GRANULARITY worth on the objects?  CPU only, fp64.

The shipped model keeps one manifold per OBJECT pair (<= 4 points) and, of a rotation-locked body against the static world (the drawer on its rails), the deepest point
alone; the frozen reference step keeps one manifold per COLLIDER pair (the drawer on its two rails: 8 points, the block across two table colliders: 8) and every point.
Rule bit 32768 (RPO_RULE_XGRAN, oracle only) switches the shipped model's contact cache to the reference step's granularity; the oracle is rebuilt here with room for
the rows (MAX_CONTACTS 40, PM_MAX 24).  Reported against the frozen reference step: arm, BLOCK and DRAWER divergence over 200 steps of the playroom id (12 envs, random
actions), and the drawer-pull scenario of tests/test_gpu_fixtures.py (how far the gripper drags the drawer out and pushes it back in).
    python tools/granularity_experiment.py [--envs 12]"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=12)
    ap.add_argument('--steps', type=int, default=200)
    args = ap.parse_args()
    scratch = tempfile.mkdtemp(prefix='rp_gran_')
    for f in os.listdir(os.path.join(REPO, 'oracle')):
        if f.endswith('.py') or f.endswith('.so'):
            shutil.copy(os.path.join(REPO, 'oracle', f), scratch)
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-std=gnu11', '-fno-fast-math', '-ffp-contract=off', '-DMAX_CONTACTS=40', '-DPM_MAX=24', '-w', '-shared', '-o',
                           os.path.join(scratch, 'librp_oracle.so'), 'rp_oracle.c', '-lm', '-lpthread'], cwd=os.path.join(REPO, 'oracle'))
    sys.path.insert(0, scratch)
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import oracle
    from oracle import OracleEnv
    import model_divergence as md
    variants = [('shipped model (rule 2039), room for 40 contacts / 24 manifolds', 2039, True), ('+ one manifold per collider pair, all points of the drawer (bit 32768)', 2039 | 32768, True)]
    print('## playroom id, random actions, %d envs x %d steps, against the frozen reference step' % (args.envs, args.steps))
    for name, rule, big in variants:
        rows = []
        for e in range(args.envs):
            ref = OracleEnv('U', seed=77, env_index=e, bullet_ref=True)
            ref.reset()
            s0 = ref.get_state()
            acts = md.random_actions('U', args.steps, np.random.default_rng(1000 + e))
            a = OracleEnv('U', seed=77, env_index=e, rule=rule)
            a.set_state(s0); ref.set_state(s0)
            na = a.n_arm
            arm = blk = drw = 0.0
            ncon_max = 0
            for t in range(args.steps):
                a.step(acts[t]); ref.step(acts[t])
                sa, sb = a.get_state(), ref.get_state()
                arm = max(arm, float((np.abs(sa[:6] - sb[:6]) / np.maximum(1.0, np.abs(sb[:6]))).max()))
                blk = max(blk, float(np.linalg.norm(sa[2 * na:2 * na + 3] - sb[2 * na:2 * na + 3])))
                drw = max(drw, float(np.linalg.norm(sa[2 * na + 13:2 * na + 16] - sb[2 * na + 13:2 * na + 16])))
            rows.append((arm, blk, drw))
        r = np.array(rows)
        print('%-74s arm median %.1e p75 %.1e | block [m] median %.1e p75 %.1e max %.1e | drawer [m] median %.1e max %.1e' % (
            name, np.median(r[:, 0]), np.percentile(r[:, 0], 75), np.median(r[:, 1]), np.percentile(r[:, 1], 75), r[:, 1].max(), np.median(r[:, 2]), r[:, 2].max()))
    # the drawer-pull scenario (tests/test_gpu_fixtures.py): gripper into the handle, out, back in
    print('## drawer scenario: how far the drawer is pulled out and pushed back in (y of the drawer body, m)')
    script = [((-0.13, -0.165, 0.10), 1.0, 40), ((-0.13, -0.165, -0.05), 1.0, 40), ((-0.13, -0.30, -0.05), 1.0, 60), ((-0.13, -0.02, -0.05), 1.0, 80)]
    for name, kw in [('frozen reference step', dict(bullet_ref=True)), ('shipped model (rule 2039)', dict(rule=2039)), ('+ bit 32768', dict(rule=2039 | 32768))]:
        o = OracleEnv('U', seed=6, env_index=0, **kw)
        o.reset()
        na = o.n_arm
        ys = []
        for target, grip, steps in script:
            for _ in range(steps):
                o.step(np.array(list(target) + [0, 0, 0, grip]))
            ys.append(o.get_state()[2 * na + 13 + 1])
        print('%-30s after the pull %+.4f, after the push %+.4f' % (name, ys[2], ys[3]))
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
