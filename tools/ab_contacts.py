#!/usr/bin/env python3
"""Where do the fast model (mode A) and the frozen reference step (mode B) part?  CPU only.
One env of `kind` runs `steps` random-action steps on mode B; before every step mode A is set to mode B's state and takes the same step (lock-step), and
the per-step difference of the arm's joints is recorded.  The steps of largest difference are then replayed and the contact lists of both models at
the start of the step printed side by side (collider names from the oracle's collider table: body / link).
    python tools/ab_contacts.py R 9 [--steps 200] [--flags-off persist,lever,spin] [--top 3]"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import oracle  # noqa: E402
from oracle import OracleEnv  # noqa: E402
import model_divergence as md  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('kind')
    ap.add_argument('env', type=int)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--flags-off', default='')
    ap.add_argument('--top', type=int, default=3)
    ap.add_argument('--rule', type=int, default=None)
    ap.add_argument('--substeps', action='store_true', help='lock-step per SUBSTEP (mode A is set to mode B\'s state before every substep)')
    args = ap.parse_args()
    fl = oracle.REF_DEFAULT
    for f in filter(None, args.flags_off.split(',')):
        fl &= ~oracle.REF_FLAGS[f]
    kind = args.kind
    b = OracleEnv(kind, seed=77, env_index=args.env, bullet_ref=True, ref_flags=fl)
    b.reset()
    akw = {} if args.rule is None else dict(rule=args.rule)
    a = OracleEnv(kind, seed=77, env_index=args.env, **akw)
    acts = md.random_actions('R' if kind == 'Q' else kind, args.steps, np.random.default_rng(1000 + args.env))
    na = b.n_arm
    nm = 6 if kind in ('R', 'U') else 7
    states, gaps = [], []
    for t in range(args.steps):
        s = b.get_state()
        states.append(s.copy())
        a.set_state(s)
        if not args.substeps:
            a.step(acts[t]); b.step(acts[t])
            gaps.append(np.abs(a.get_state()[:na] - b.get_state()[:na]))
            continue
        a.perform_action(acts[t]); b.perform_action(acts[t])
        states.pop()
        for k in range(12):
            s = b.get_state()
            states.append(s.copy())
            a.set_state(s)
            a.substep(); b.substep()
            gaps.append(np.abs(a.get_state()[na:2 * na] - b.get_state()[na:2 * na]))      # joint velocities after one substep
    gaps = np.array(gaps)
    print('lock-step gap per step (|dq|) or substep (|dqd|): arm max %.2e (step %d), all dofs max %.2e' % (gaps[:, :nm].max(), gaps[:, :nm].max(1).argmax(), gaps.max()))
    cols = a.collider_list()
    name = lambda c: 'c%d(b%d l%d t%d)' % (c, cols[c]['body'], cols[c]['link'], cols[c]['type'])   # noqa: E731
    for t in np.argsort(-gaps[:, :nm].max(1))[:args.top]:
        print('--- step %d: arm gap %s' % (t, np.array2string(gaps[t, :nm], precision=2)))
        a.set_state(states[t])
        scratch = OracleEnv(kind, seed=77, env_index=args.env, bullet_ref=True, ref_flags=fl & ~oracle.REF_FLAGS['persist'])
        scratch.set_state(states[t])
        for tag, env in (('A', a), ('B (fresh manifolds)', scratch)):
            con = env.contacts()
            print('  %s: %d contacts' % (tag, len(con)))
            for c in con:
                print('     %-18s %-18s p %s n %s d %+.5f' % (name(int(c[0])), name(int(c[1])), np.array2string(c[2:5], precision=4), np.array2string(c[5:8], precision=3), c[8]))


if __name__ == '__main__':
    main()
