#!/usr/bin/env python3
"""Distribution A (a ~ U(action_space.low, high), resampled every step) beside distribution B: throughput in the default mode and the per-launch kernel times
(hipEvent timers, env groups off) - where the literal random-action rollout spends its step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from roboticsplayroompybullet_amd import VecPlayEnv
n, steps, warm = 4096, 100, 40
for name in ('B', 'A'):
    env = VecPlayEnv(bench.ENV_ID, n, seed=1234); env.reset()
    if name == 'B':
        acts = bench.make_actions(n, steps + warm, env.device, 1234)
    else:
        g = torch.Generator(device=env.device).manual_seed(4321)
        acts = (2 * torch.rand((steps + warm, n, 7), generator=g, device=env.device) - 1) * env.action_high
    for k in range(warm): env.step(acts[k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): env.step(acts[warm + k])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    env.enable_timers(50)
    for k in range(50): env.step(acts[warm + k])
    torch.cuda.synchronize()
    tm = env.timers(); env.enable_timers(0)
    print('distribution %s: %.3fM env-steps/s (%.3f ms per step); per launch, groups off: %s' % (
        name, n * steps / dt / 1e6, 1e3 * dt / steps, {k: round(v, 4) for k, v in tm.items() if isinstance(v, float) and v > 0}))
    env.close()
