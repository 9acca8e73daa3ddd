#!/usr/bin/env python3
"""GPU box: the door scenario of tests/test_gpu_fixtures.py step by step - device against the fp32 / fp64 oracles: joint-target gap (the IK), arm gap, door gap, status bits."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tests')); sys.path.insert(0, os.path.join(REPO, 'tools'))
from test_gpu_fixtures import make
np.set_printoptions(precision=6, suppress=True, linewidth=220)
env, o32, o64 = make(3, 6)
script = [((-0.06, 0.322, 0.15), 1.0, 40), ((-0.06, 0.322, 0.10), 1.0, 30), ((0.15, 0.322, 0.10), 1.0, 80),
          ((0.15, 0.322, 0.2), 1.0, 20), ((0.28, 0.322, 0.2), 1.0, 20), ((0.28, 0.322, 0.10), 1.0, 30), ((0.05, 0.322, 0.10), 1.0, 70)]
t = 0
for target, grip, steps in script:
    a = np.array(list(target) + [0, 0, 0, grip])
    for _ in range(steps):
        obs, r, _, info = env.step(torch.tensor(np.tile(a, (3, 1)), dtype=torch.float32))
        q = env.get_state()[:, :6].cpu().numpy()
        tp = info['target_poses'].cpu().numpy(); st = info['status'].cpu().numpy()
        line = 't %3d' % t
        for e in range(3):
            r32 = o32[e].step(a); r64 = o64[e].step(a)
            s32 = o32[e].get_state()
            line += ' | e%d st %2d tp gap %.1e (64: %.1e) arm %.1e door %.1e (32-64: %.1e)' % (e, st[e], np.abs(tp[e] - r32[3]['target_poses']).max(), np.abs(r64[3]['target_poses'] - r32[3]['target_poses']).max(),
                                                                          np.abs(q[e] - s32[:6]).max(), abs(obs['obs_quat'][e, 16].item() - r32[0]['obs_quat'][16]), abs(r64[0]['obs_quat'][16] - r32[0]['obs_quat'][16]))
        if t % 5 == 0 or t > 215:
            print(line)
        t += 1
