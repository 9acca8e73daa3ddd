#!/usr/bin/env python3
"""debug: the pandaPick grasp scenario of tests/test_gpu_fixtures.py, device vs fp32 / fp64 oracle, per step"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle'))
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
n, seed = 6, 3
env = VecPlayEnv('pandaPick-v0', n, seed=seed); env.reset()
o32 = [OracleEnv('P', seed=seed, env_index=e, f32=True) for e in range(n)]
o64 = [OracleEnv('P', seed=seed, env_index=e) for e in range(n)]
ob32 = [o.reset() for o in o32]
[o.reset() for o in o64]
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for t in range(45):
    a = np.zeros((n, 7))
    for e in range(n):
        blk = ob32[e]['achieved_goal'][:3]
        a[e, 0:3] = blk; a[e, 2] = blk[2] if t < 60 else 0.15; a[e, 6] = -1.0 if t < 30 else 1.0
    obs, r, _, info = env.step(torch.tensor(a, dtype=torch.float32))
    got = obs['obs_quat'].cpu().numpy()
    rc = env.debug_row_counts()
    for e in range(n):
        ob32[e] = o32[e].step(a[e])[0]
        b64 = o64[e].step(a[e])[0]['obs_quat']
        if e == E and t >= 27:
            err = np.abs(got[e] - ob32[e]['obs_quat']); gap = np.abs(ob32[e]['obs_quat'] - b64)
            i = int(err.argmax())
            print('t %2d err max %.2e at %d (gap %.2e) dev %.5f o32 %.5f o64 %.5f | ncon dev %s o32 %d | gripper obs dev %s' % (t, err[i], i, gap[i], got[e][i], ob32[e]['obs_quat'][i], b64[i], rc[e], len(o32[e].contacts()), np.round(got[e][7:], 4)))
