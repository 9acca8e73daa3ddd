#!/usr/bin/env python3
"""Round 5's rows of the fidelity table: the fast model (oracle/rp_oracle.c, the model the HIP kernels implement) against the frozen Bullet-like reference step
(oracle/rp_bullet_ref.c), fp64 both, 12 envs x 200 steps from the reference step's post-reset state, same actions.  New against tools/model_divergence.py:
  * action distribution A - the literal U(action_space) rollout BASELINE.json's metric names (environments.py:104-110) - beside distribution B ("random");
  * `A +epa`: the fast model with the reference step's own GJK / EPA where a hull's deepest vertex lies beside the box face (the OBB path's case for overlapping cores;
    experiment build librp_oracle_abx.so, rule bit 4096): what EPA on the device would buy;
  * `A +creation-order`: contacts solved in the manifolds' creation order (Bullet's) instead of the four-tier partition of the two-stream solver (rule bit 65536);
  * p75 and the number of envs within 1e-3 in every row; `B -anchor` (the reference step against itself with friction anchors off: a 1e-14-level change) as the
    yardstick of what chaos alone does to each column.
Writes profiles/<tag>_model_divergence.md and .json.
    python tools/fidelity_r05.py [--envs 12] [--steps 200] [--tag r05]"""
import argparse
import json
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

EPA = 131072


def shipped(kind):      # rp_oracle.c rpo_create: the expanding polytope is in the Panda kinds' default, not in the UR5 kinds'
    return 2039 | (0 if kind in 'UR' else EPA)


def actions(kind, dist, steps, rng):
    if dist == 'random':
        return md.random_actions('R' if kind == 'Q' else kind, steps, rng)
    hi = OracleEnv(kind).action_high()
    return (2 * rng.random((steps, len(hi))) - 1) * hi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=12)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--tag', default='r05')
    args = ap.parse_args()
    D = oracle.REF_DEFAULT
    variants = [('A -epa (rule 2039: overlapping cores on the OBB path)', dict(rule=2039)), ('A +epa (rule 133111: the expanding polytope for overlapping cores)', dict(rule=2039 | EPA)),
                ('A -epa -gjk (= RP_CFG_OBB_EDGES, rule 1015)', dict(rule=1015)),
                ('A -epa + the reference step\'s own GJK / EPA (experiment build)', dict(rule=2039 | 4096, abx=True)), ('A +epa +creation-order (experiment)', dict(rule=2039 | EPA | 65536)),
                ('B -anchor (the reference step against itself)', dict(bullet_ref=True, ref_flags=D & ~oracle.REF_FLAGS['anchor']))]
    ship_name = {k: [n for n, kw in variants if kw.get('rule') == shipped(k) and not kw.get('abx')][0] for k in 'URPQVW'}
    cases = [('R', 'random'), ('Q', 'random'), ('U', 'random'), ('U', 'A-dist'), ('P', 'random'), ('P', 'A-dist'), ('V', 'random'), ('V', 'A-dist')]
    results = {}

    def one(job):
        kind, dist, e = job
        ref = OracleEnv(kind, seed=77, env_index=e, bullet_ref=True)
        ref.reset()
        state0 = ref.get_state()
        acts = actions(kind, dist, args.steps, np.random.default_rng(1000 + e))
        qb, bb = md.rollout(ref, kind, 'random', args.steps, acts, state0)
        out = {}
        for name, kw in variants:
            env = OracleEnv(kind, seed=77, env_index=e, **kw)
            qa, ba = md.rollout(env, kind, 'random', args.steps, acts, state0)
            out[name] = md.divergence(qa, ba, qb, bb, 6 if kind in ('R', 'U') else 7)
        return out
    jobs = [(k, d, e) for k, d in cases for e in range(args.envs)]
    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(one, jobs))
    for (k, d, e), o in zip(jobs, outs):
        for name, v in o.items():
            results.setdefault('%s/%s' % (k, d), {}).setdefault(name, []).append(v)
    lines = ['| config / actions | model | arm joints: median / p75 / p90 / max | envs <= 1e-3 | block [m]: median / p90 / max |', '|---|---|---|---|---|']
    js = {}
    for key, rows in results.items():
        for name, v in rows.items():
            a = np.array(v)
            js.setdefault(key, {})[name] = {'arm': a[:, 0].tolist(), 'joints': a[:, 1].tolist(), 'block': a[:, 2].tolist()}
            blk = '-' if key[0] in 'RQ' else '%.1e / %.1e / %.1e' % (np.median(a[:, 2]), np.percentile(a[:, 2], 90), a[:, 2].max())
            lines.append('| %s | %s | %.1e / %.1e / %.1e / %.1e | %d of %d | %s |' % (key, name + (' **<- shipped for this id**' if name == ship_name[key[0]] else ''), np.median(a[:, 0]), np.percentile(a[:, 0], 75), np.percentile(a[:, 0], 90), a[:, 0].max(),
                                                                                   int((a[:, 0] <= 1e-3).sum()), len(a), blk))
    os.makedirs(os.path.join(REPO, 'profiles'), exist_ok=True)
    json.dump({'envs': args.envs, 'steps': args.steps, 'reference': 'oracle/rp_bullet_ref.c, default flags %d' % D, 'results': js},
              open(os.path.join(REPO, 'profiles', '%s_model_divergence.json' % args.tag), 'w'), indent=1)
    head = ('# Fast model vs the frozen reference step, round 5 (tools/fidelity_r05.py: %d envs x %d steps, fp64, from the reference step\'s post-reset state)\n\n'
            '`random` = bench.py\'s distribution B (workspace-uniform targets); `A-dist` = the literal U(action_space) rollout.  Under A-dist every pair of runs parts within tens\n'
            'of steps - the `B -anchor` row is the reference step against ITSELF with a 1e-14-level change - so its columns measure chaos, not models.\n'
            'The expanding polytope (`+epa`) is in the default of the Panda ids (P, Q, V, W), whose rows it moves, and not of the UR5 ids (U, R), whose rows it does not (DESIGN.md).\n\n' % (args.envs, args.steps))
    open(os.path.join(REPO, 'profiles', '%s_model_divergence.md' % args.tag), 'w').write(head + '\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
