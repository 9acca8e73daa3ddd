#!/usr/bin/env python3
"""CPU: replay the cases tests/test_gpu_dist_a.py::test_distribution_a_lockstep_with_contact_history dumped (RP_LOCKSTEP_DUMP=<dir>: env-steps in which the device's
contact cache differs from the oracle's although the arm agrees to 1e-6) on the fp32 oracle, substep by substep, printing the cache after every substep beside the
device's post-step cache.
    python tools/lockstep_replay.py gpurun_out/r05/lockstep_U.npz U [case]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tests')); sys.path.insert(0, os.path.join(REPO, 'tools'))
import cache_rows  # noqa: E402
from oracle import OracleEnv  # noqa: E402


def oracle_state_from_record(o, r):
    n, nf = o.n_arm, (2 if o.kind in (0, 4) else (1 if o.kind == 2 else 0))
    nj = 3 if o.kind in (0, 4) else 0
    s = list(r[0:n]) + list(r[12:12 + n])
    for k in range(nf):
        s += list(r[24 + 13 * k:24 + 13 * k + 13])
    s += list(r[50:50 + nj]) + list(r[53:53 + nj])
    return np.array(s, dtype=np.float64)


def main():
    path, kind = sys.argv[1], sys.argv[2]
    d = np.load(path)
    cases = sorted({int(k.rsplit('_', 1)[1]) for k in d.files})
    if len(sys.argv) > 3:
        cases = [int(sys.argv[3])]
    for i in cases:
        pre, act, tgt, post = d['pre_%d' % i], d['action_%d' % i], d['targets_%d' % i], d['post_device_%d' % i]
        print('=== case %d: step %d env %d' % (i, int(d['step_%d' % i]), int(d['env_%d' % i])))
        o = OracleEnv(kind, seed=31, env_index=int(d['env_%d' % i]), f32=True)
        o.reset(); o.step(act.astype(np.float64))
        o.set_state(oracle_state_from_record(o, pre)); o.set_cache_row(pre[128:])
        o.perform_action(act.astype(np.float64)); o.goto_joint_poses(tgt, gripper=float(act[-1]))
        print('  pre   :', cache_rows.describe(pre[128:]))
        for sub in range(12):
            o.substep()
            print('  sub %2d:' % sub, cache_rows.describe(o.get_cache_row()).split(' | gjk')[0])
        print('  oracle:', cache_rows.describe(o.get_cache_row()))
        print('  device:', cache_rows.describe(post[128:]))
        dd, do = cache_rows.decode(post[128:]), cache_rows.decode(o.get_cache_row())
        for md, mo in zip(dd['manifolds'], do['manifolds']):
            if md['ab'] != mo['ab'] and md['key'] == mo['key']:
                print('  manifold %d: device lA / dist' % md['key']); print(np.round(np.c_[md['lA'], md['dist']], 5))
                print('  manifold %d: oracle lA / dist' % mo['key']); print(np.round(np.c_[mo['lA'], mo['dist']], 5))


if __name__ == '__main__':
    main()
