#!/usr/bin/env python3
"""Which kernel stretches in the slow steps?  The bench workload (distribution B), per step: wall time in the default mode and - from a second env driven with the same
actions, env groups off - the per-launch averages of that step's kernels (hipEvent timers)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from roboticsplayroompybullet_amd import VecPlayEnv
n, warm, steps = 4096, 205, 40
a = VecPlayEnv(bench.ENV_ID, n, seed=1234); a.reset()
b = VecPlayEnv(bench.ENV_ID, n, seed=1234); b.reset()
acts = bench.make_actions(n, warm + steps, a.device, 1234)
for k in range(warm): a.step(acts[k]); b.step(acts[k])
torch.cuda.synchronize()
rows = []
for k in range(steps):
    t0 = time.perf_counter(); a.step(acts[warm + k]); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    b.enable_timers(1); b.step(acts[warm + k]); torch.cuda.synchronize(); tm = b.timers(); b.enable_timers(0)
    rows.append((1e3 * dt, tm['avg_action_ms'], tm['avg_prep_ms'], tm['avg_solve_ms']))
print('step: wall ms (default mode, one step at a time) | groups off: k_action_prep, k_prep2 avg of 12, k_solve2 avg of 12 [ms]')
for k, r in enumerate(rows): print('%3d: %.2f | %.3f %.4f %.4f' % ((k,) + r))
