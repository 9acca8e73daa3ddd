#!/usr/bin/env python3
"""Round 6 (verdict item 5): which model difference owns the headline id's p75 against the frozen reference step?  U / random (distribution B), 24 envs x 200 steps, fp64,
everything from the reference step's post-reset state on the same actions, arm joints (relative), against the reference step with ALL its flags (B):
  * A: the shipped fast model (with and without the residual form of round 6: the same model in another rounding);
  * B -x: the reference step with ONE of its flags switched off - for `anchor` that IS the fast model's choice (the only flag on which the two differ); the others show what
    each feature is worth on this workload and how far a 1e-14-level change (anchor) is from a real one;
  * A with the reference step's own pieces (experiment build librp_oracle_abx.so): its GJK / EPA behind the hull contacts (rule bit 4096), its manifold upkeep and solver
    behind the fast model's narrowphase (rule bit 2048) - what is NOT a flag: the narrowphase and the manifolds themselves.
Writes profiles/r06_flag_table.md.
    python tools/fidelity_r06.py [--envs 24] [--steps 200]"""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import oracle  # noqa: E402
from oracle import OracleEnv  # noqa: E402
import model_divergence as md  # noqa: E402

RES = 262144


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=24)
    ap.add_argument('--steps', type=int, default=200)
    args = ap.parse_args()
    D = oracle.REF_DEFAULT
    variants = [('A, shipped (rule 2039 | residual form)', dict(rule=2039 | RES)), ('A without the residual form (rule 2039: round 5\'s arithmetic)', dict(rule=2039))]
    for name, bit in oracle.REF_FLAGS.items():
        if D & bit:
            variants.append(('B -%s%s' % (name, ' (= the fast model\'s choice)' if name == 'anchor' else ''), dict(bullet_ref=True, ref_flags=D & ~bit)))
    try:
        oracle.load(False, False, True)
        variants += [('A + the reference step\'s own GJK / EPA behind the hull contacts (abx, rule bit 4096)', dict(rule=2039 | 4096, abx=True)),
                     ('A\'s narrowphase + the reference step\'s manifold upkeep and solver (abx, rule bit 2048)', dict(rule=2039 | 2048, abx=True))]
    except Exception as ex:      # the experiment build is optional
        print('no abx build:', ex)

    def one(e):
        ref = OracleEnv('U', seed=77, env_index=e, bullet_ref=True)
        ref.reset()
        state0 = ref.get_state()
        acts = md.random_actions('U', args.steps, np.random.default_rng(1000 + e))
        qb, bb = md.rollout(ref, 'U', 'random', args.steps, acts, state0)
        out = {}
        for name, kw in variants:
            env = OracleEnv('U', seed=77, env_index=e, **kw)
            qa, ba = md.rollout(env, 'U', 'random', args.steps, acts, state0)
            out[name] = md.divergence(qa, ba, qb, bb, 6)
        return out
    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(one, range(args.envs)))
    lines = ['| model | arm joints: median / p75 / p90 / max | envs <= 1e-3 | block [m]: median / p90 |', '|---|---|---|---|']
    for name, _ in variants:
        a = np.array([o[name] for o in outs])
        lines.append('| %s | %.1e / %.1e / %.1e / %.1e | %d of %d | %.1e / %.1e |' % (name, np.median(a[:, 0]), np.percentile(a[:, 0], 75), np.percentile(a[:, 0], 90), a[:, 0].max(),
                                                                                  int((a[:, 0] <= 1e-3).sum()), len(a), np.median(a[:, 2]), np.percentile(a[:, 2], 90)))
    head = ('# U / random against the frozen reference step with all its flags, one difference at a time (tools/fidelity_r06.py: %d envs x %d steps, fp64)\n\n' % (args.envs, args.steps))
    open(os.path.join(REPO, 'profiles', 'r06_flag_table.md'), 'w').write(head + '\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
