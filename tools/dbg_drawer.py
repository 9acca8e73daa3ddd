#!/usr/bin/env python3
"""debug: the drawer scenario of tests/test_gpu_fixtures.py, device vs fp32 / fp64 oracle, per step"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle'))
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
U = 'UR5PlayAbsRPY1Obj-v0'
n, seed = 1, 5
env = VecPlayEnv(U, n, seed=seed); env.reset()
o32 = OracleEnv('U', seed=seed, env_index=0, f32=True); o32.reset()
o64 = OracleEnv('U', seed=seed, env_index=0); o64.reset()
script = [((-0.13, -0.165, 0.10), 1.0, 40), ((-0.13, -0.165, -0.05), 1.0, 40), ((-0.13, -0.30, -0.05), 1.0, 60), ((-0.13, -0.02, -0.05), 1.0, 80)]
t = 0
for target, grip, steps in script:
    a = np.array(list(target) + [0, 0, 0, grip], dtype=np.float64)
    for _ in range(steps):
        obs, r, _, info = env.step(torch.tensor(np.tile(a, (n, 1)), dtype=torch.float32))
        got = obs['obs_quat'].cpu().numpy()[0]
        a32 = o32.step(a)[0]['obs_quat']; a64 = o64.step(a)[0]['obs_quat']
        err = np.abs(got - a32); gap = np.abs(a32 - a64)
        i = int(err.argmax())
        print('t %3d  max err %.2e at %2d (o32-o64 there %.2e)  drawer y dev %.5f o32 %.5f o64 %.5f  ncon dev %d o32 %d' % (
            t, err[i], i, gap[i], got[15], a32[15], a64[15], int(env.debug_row_counts()[0, 1]) % 1000, len(o32.contacts())))
        t += 1
