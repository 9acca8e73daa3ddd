#!/usr/bin/env python3
"""CPU: the `moved` cases of tests/test_gpu_dist_a.py::test_distribution_a_lockstep_with_contact_history (RP_LOCKSTEP_DUMP=<dir>: env-steps in which device and fp32
oracle hold the same manifolds after ONE step from identical inputs and a free body or scene joint still differs by more than 1e-4) replayed on the fp32 oracle and on
eight nudged copies of it (arm joints +-1, +-2, +-4, +-8 ulp): which coordinate moved, how far the device is from the oracle, how far the nudged oracles are from it.
A case is a ROUNDING-DECIDED EVENT when the nudged CPU runs spread as far as the device is off; anything else is a finding.
    python tools/lockstep_moved.py gpurun_out/r06/dump/lockstep_moved_U.npz U"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tests')); sys.path.insert(0, os.path.join(REPO, 'tools'))
from oracle import OracleEnv  # noqa: E402
from lockstep_replay import oracle_state_from_record  # noqa: E402


def names(o):
    n, nf = o.n_arm, (2 if o.kind in (0, 4) else (1 if o.kind == 2 else 0))
    nj = 3 if o.kind in (0, 4) else 0
    out = []
    for k in range(nf):
        out += ['free%d.%s' % (k, c) for c in ('x', 'y', 'z', 'qx', 'qy', 'qz', 'qw')]
    out += ['joint%d' % k for k in range(nj)]
    idx = [2 * n + 13 * k + i for k in range(nf) for i in range(7)] + [2 * n + 13 * nf + i for i in range(nj)]
    return out, idx


def main():
    path, kind = sys.argv[1], sys.argv[2]
    d = np.load(path)
    cases = sorted({int(k.rsplit('_', 1)[1]) for k in d.files})
    inside = 0
    for i in cases:
        pre, act, tgt, post = d['pre_%d' % i], d['action_%d' % i], d['targets_%d' % i], d['post_device_%d' % i]
        env = int(d['env_%d' % i])
        runs = []
        for ulp in (0, 1, -1, 2, -2, 4, -4, 8, -8):
            o = OracleEnv(kind, seed=31, env_index=env, f32=True)
            o.reset(); o.step(act.astype(np.float64))
            s = oracle_state_from_record(o, pre)
            q = s[:o.n_arm].astype(np.float32)
            for _ in range(abs(ulp)):
                q = np.nextafter(q, np.float32(np.inf if ulp > 0 else -np.inf))
            s[:o.n_arm] = q.astype(np.float64)
            o.set_state(s); o.set_cache_row(pre[128:])
            o.perform_action(act.astype(np.float64)); o.goto_joint_poses(tgt, gripper=float(act[-1]))
            o.run_simulation()
            runs.append(o.get_state())
        nm, idx = names(o)
        sd = oracle_state_from_record(o, post)
        dev = np.abs(sd[idx] - runs[0][idx])
        spread = np.max([np.abs(r[idx] - runs[0][idx]) for r in runs[1:]], axis=0)
        k = int(np.argmax(dev))
        ok = dev.max() <= 3.0 * max(spread.max(), 1e-6)
        inside += ok
        print('case %2d: step %3d env %2d  device off the oracle by %.1e in %s; the eight nudged oracles spread %.1e there (%.1e at the most, in %s)  -> %s'
              % (i, int(d['step_%d' % i]), env, dev[k], nm[k], spread[k], spread.max(), nm[int(np.argmax(spread))], 'rounding-decided' if ok else 'LOOK'))
    print('%d of %d cases: the device is within three times the spread of the nudged CPU runs' % (inside, len(cases)))


if __name__ == '__main__':
    main()
