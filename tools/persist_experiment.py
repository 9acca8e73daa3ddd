#!/usr/bin/env python3
"""What are PERSISTENT MANIFOLDS in the fast model worth?  CPU only.  (This experiment came first; the shipped model has them since: RPO_RULE_PERSIST.)

`oracle/rp_oracle.c -DRPO_ABX` (built here into a scratch directory) adds rule bit 2048 to mode A: its own narrowphase (hull vertices, box_box in the
detector's order - overlap only for every pair -, sphere_box) feeds the frozen reference step's manifold upkeep (`rpb_find_manifold`, `rpb_add_point`,
`rpb_refresh`: points in the two bodies' frames, refreshed every substep, dropped beyond the breaking threshold, a new point within the threshold of a
cached one replaces it).  Bit 1024 on top: manifolds keyed by OBJECT pair and the deepest point alone for a rotation-locked body against the static world -
the fast model's own manifold rules, i.e. the same row counts as today's model.  Bit 16384: rows in manifold order instead of the four-tier partition.

Prints the 200-step arm divergence from the frozen reference step (default flags) per env and median / p90 / max, like tools/model_divergence.py.
    python tools/persist_experiment.py [--kinds R,U,P] [--envs 12]
Measured (round 3): R 3.0e-4 max -> 1.2e-8; U median 3.6e-3 -> 1.5e-3 (max 9e-2 either way: the chaotic envs); P unchanged (1e-14 median, one env 2e-2:
a finger's OBB on the block).  DESIGN.md section 2 quotes this."""
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
    ap.add_argument('--kinds', default='R,U,P')
    ap.add_argument('--envs', type=int, default=12)
    ap.add_argument('--steps', type=int, default=200)
    args = ap.parse_args()
    scratch = tempfile.mkdtemp(prefix='rp_persist_')
    for f in os.listdir(os.path.join(REPO, 'oracle')):
        if f.endswith('.py') or f.endswith('.so'):
            shutil.copy(os.path.join(REPO, 'oracle', f), scratch)
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-std=gnu11', '-fno-fast-math', '-ffp-contract=off', '-DRPO_ABX', '-w', '-shared', '-o',
                           os.path.join(scratch, 'librp_oracle.so'), 'rp_oracle.c', '-lm', '-lpthread'], cwd=os.path.join(REPO, 'oracle'))
    sys.path.insert(0, scratch)
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import oracle
    from oracle import OracleEnv
    import model_divergence as md
    base = 247
    variants = [('stateless contacts (RP_CFG_STATELESS_CONTACTS)', base), ('persistent manifolds (RPO_RULE_PERSIST)', base | 256), ('+ the reference step\'s own manifold upkeep', base | 2048), ('+ the same, the fast model\'s manifold rules', base | 2048 | 1024),
                ('+ the same, manifold order', base | 2048 | 16384), ('+ hull vertices against movable boxes (RPO_RULE_HULLMOV)', base | 256 | 512),
                ('shipped model: + GJK distance phase beside the face (RPO_RULE_GJK)', base | 256 | 512 | 1024), ('shipped + GJK / EPA where the vertex lies beside the face (experiment)', base | 256 | 512 | 4096), ('shipped + GJK distance only there, OBB when the cores overlap (experiment)', base | 256 | 512 | 4096 | 8192)]
    for kind in args.kinds.split(','):
        res = {v[0]: [] for v in variants}
        for e in range(args.envs):
            ref = OracleEnv(kind, seed=77, env_index=e, bullet_ref=True)
            ref.reset()
            s0 = ref.get_state()
            acts = md.random_actions('R' if kind == 'Q' else kind, args.steps, np.random.default_rng(1000 + e))
            b = OracleEnv(kind, seed=77, env_index=e, bullet_ref=True)
            qb, bb = md.rollout(b, kind, 'random', args.steps, acts, s0)
            for name, rule in variants:
                a = OracleEnv(kind, seed=77, env_index=e, rule=rule)
                qa, ba = md.rollout(a, kind, 'random', args.steps, acts, s0)
                res[name].append(md.divergence(qa, ba, qb, bb, 6 if kind in ('R', 'U') else 7)[0])
        for name, v in res.items():
            v = np.array(v)
            print('%s/random  %-52s arm median %.1e p90 %.1e max %.1e' % (kind, name, np.median(v), np.percentile(v, 90), v.max()))
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
