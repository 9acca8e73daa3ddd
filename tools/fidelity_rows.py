#!/usr/bin/env python3
"""Selected rows of tools/model_divergence.py's table, quickly: the fast model (rule bits given on the command line) against the frozen reference step,
12 envs x 200 steps, fp64.  Prints arm median / p75 / max, envs within 1e-3, block median / max per (kind, scenario).
    python tools/fidelity_rows.py [--rules 1015,2039] [--kinds U,P] [--envs 12]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import oracle  # noqa: E402
from oracle import OracleEnv  # noqa: E402
import model_divergence as md  # noqa: E402


def rows(kind, rule, envs=12, steps=200, scenario='random', **kw):
    out = []
    for e in range(envs):
        ref = OracleEnv(kind, seed=77, env_index=e, bullet_ref=True)
        ref.reset()
        state0 = ref.get_state()
        acts = md.random_actions('R' if kind == 'Q' else kind, steps, np.random.default_rng(1000 + e))
        qb, bb = md.rollout(ref, kind, scenario, steps, acts, state0)
        env = OracleEnv(kind, seed=77, env_index=e, rule=rule, **kw)
        qa, ba = md.rollout(env, kind, scenario, steps, acts, state0)
        out.append(md.divergence(qa, ba, qb, bb, 6 if kind in ('R', 'U') else 7))
    a = np.array(out)
    return dict(arm_median=float(np.median(a[:, 0])), arm_p75=float(np.percentile(a[:, 0], 75)), arm_max=float(a[:, 0].max()), within_1e3=int((a[:, 0] <= 1e-3).sum()),
                block_median=float(np.median(a[:, 2])), block_max=float(a[:, 2].max()), envs=envs)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--rules', default='1015,2039,133111')
    ap.add_argument('--kinds', default='U,P')
    ap.add_argument('--envs', type=int, default=12)
    args = ap.parse_args()
    for kind in args.kinds.split(','):
        for rule in [int(r) for r in args.rules.split(',')]:
            r = rows(kind, rule, args.envs)
            print('%s/random rule %5d: arm median %.1e p75 %.1e max %.1e, %d/%d <= 1e-3; block median %.1e max %.1e' % (
                kind, rule, r['arm_median'], r['arm_p75'], r['arm_max'], r['within_1e3'], r['envs'], r['block_median'], r['block_max']))
    lib = oracle.load()
    st = (C.c_long * 8)()
    lib.rpo_gjk_stats.argtypes = [C.c_void_p, C.c_int]
    lib.rpo_gjk_stats(st, 0)
    print('GJK (oracle, all runs above): calls %d, rounds %d (%.2f per call), two-point seeds %d, results contact %d / apart %d / cores overlap %d, tetrahedra %d, rounds spent in early apart exits %d' % (
        st[0], st[1], st[1] / max(1, st[0]), st[2], st[3], st[4], st[5], st[6], st[7]))
