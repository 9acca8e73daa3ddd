#!/usr/bin/env python3
"""Per-launch kernel times of the bench workload under distribution A (a ~ U(action_space): the literal random-action rollout) beside distribution B's.
    python tools/dist_a_profile.py            (GPU)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticsplayroompybullet_amd import VecPlayEnv  # noqa: E402
import bench  # noqa: E402

n = 4096
for name in ('B', 'A'):
    env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
    env.reset()
    if name == 'B':
        acts = bench.make_actions(n, 300, env.device, 1234)
    else:
        g = torch.Generator(device=env.device).manual_seed(4321)
        acts = (2 * torch.rand((300, n, 7), generator=g, device=env.device) - 1) * env.action_high
    for k in range(250):
        _, _, _, info = env.step(acts[k])
    st = info['status']
    env.enable_timers(50)
    for k in range(250, 300):
        env.step(acts[k])
    torch.cuda.synchronize()
    tm = env.timers()
    env.enable_timers(0)
    print('distribution %s: %s; IK out of iterations in %.1f %% of the envs (status bit 8)' % (name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items()}, 100.0 * float(((st & 8) != 0).float().mean())))
    env.close()
