#!/usr/bin/env python3
"""GPU: 2 500 steps x 4096 envs of distribution A (a ~ U(action_space), the arm-collision-heavy rollout in which k_prep2's two waves share hull pairs) with masked
resets of 5 % of the envs every 500 steps, four ids: a hang / non-finite check of the LDS hand-over under load.  Round 4: no non-finite record, no hang."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from roboticsplayroompybullet_amd import VecPlayEnv
for gid in ('UR5PlayAbsRPY1Obj-v0', 'pandaPlayAbsRPY1Obj-v0', 'pandaPick-v0', 'UR5Reach-v0'):
    n, steps = 4096, 2500
    env = VecPlayEnv(gid, n, seed=321); env.reset()
    g = torch.Generator(device=env.device).manual_seed(7)
    t0 = time.perf_counter(); bad = 0
    for k in range(steps):
        a = (2 * torch.rand((n, env.action_high.numel()), generator=g, device=env.device) - 1) * env.action_high
        obs, r, d, info = env.step(a)
        if k % 500 == 499:
            bad = int((info['status'] & 1).sum()); fell = int(((info['status'] & 2) != 0).sum())
            m = torch.rand(n, device=env.device) < 0.05
            env.reset(mask=m.to(torch.uint8))
    torch.cuda.synchronize()
    print('%-26s %d steps x %d envs of distribution A with masked resets: non-finite %d, fallen %d, %.2f M env-steps/s' % (gid, steps, n, bad, fell, n * steps / (time.perf_counter() - t0) / 1e6))
    env.close()
