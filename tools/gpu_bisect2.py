#!/usr/bin/env python3
"""Lock-step bisect (GPU box): oracle (fp32) drives; before every substep the device gets the oracle's state, runs ONE fused substep
(rp_debug_substep) and the resulting velocities / contact lists are compared.  Prints the first substep that differs.
    python tools/gpu_bisect2.py [margin]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tools'))
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
from gpu_debug import record_from_oracle
np.set_printoptions(precision=7, suppress=True, linewidth=220)
LO = np.array([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0]); HI = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])
margin = float(sys.argv[1]) if len(sys.argv) > 1 else None
n = 6
rng = np.random.default_rng(3)
a = LO + (HI - LO) * rng.random((6, n, 7)); a[:, :, 2] = 0.03
env = VecPlayEnv('UR5PlayAbsRPY1Obj-v0', 2, seed=8, contact_margin=margin)
hi = np.array([6] * 6 + [1.0])
for e in range(n):
    o = OracleEnv('U', seed=8, env_index=e, f32=True, margin=margin)
    o.reset()
    na = o.n_arm
    for t in range(6):
        o.perform_action(np.clip(a[t, e], -hi, hi))
        for sub in range(12):
            rec = record_from_oracle(o)
            env.set_state(torch.tensor(np.tile(rec, (2, 1))))
            dbg = env.debug_substep(0).numpy()
            ncon = int(dbg[0]); gc = dbg[16:16 + 9 * ncon].reshape(ncon, 9)
            oc = o.contacts()
            o.substep()
            s1 = o.get_state()
            vg = dbg[480:480 + 27] + dbg[544:544 + 27]
            vo = np.concatenate([s1[na:2*na], s1[2*na+7:2*na+13], s1[2*na+13+7:2*na+26], s1[2*na+26+3:2*na+26+6]])
            dvel = np.abs(vg - vo)
            same = ncon == len(oc) and np.allclose(gc, oc, atol=1e-4)
            if not same or dvel.max() > 1e-3:
                print('env', e, 't', t, 'sub', sub, 'ncon', ncon, len(oc), 'contacts_same', same, 'max dvel', dvel.max(), 'argmax', dvel.argmax(), 'nsmall', int(dbg[1]), 'oracle rows', o.num_rows())
                print(' gpu contacts\n', gc); print(' cpu contacts\n', oc)
                print(' v gpu', vg); print(' v cpu', vo); print(' diff', vg - vo)
                print(' q', s1[:na])
                sys.exit(0)
    print('env', e, 'ok')
