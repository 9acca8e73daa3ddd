#!/usr/bin/env python3
"""The bench workload with a host-resident hand-off: actions come from pinned host memory every step and the whole observation dict, reward,
success and status go back to (pinned) host memory before the next action is sent - what a CPU policy loop would see.  bench.py's `value`
keeps everything in HBM; DESIGN.md section 6 quotes this number beside it.   python tools/pcie_rate.py [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from roboticsplayroompybullet_amd import VecPlayEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
obs = env.reset()
acts = bench.make_actions(n, 220, env.device, 1234).cpu().pin_memory()
host = {k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in obs.items() if v is not None}
hr = torch.empty(n).pin_memory(); hs = torch.empty(n, dtype=torch.int32).pin_memory()
nbytes = sum(v.numel() * v.element_size() for v in host.values()) + hr.numel() * 4 + hs.numel() * 4 + acts[0].numel() * 4
def step(k):
    a = acts[k].to(env.device, non_blocking=True)
    o, r, d, info = env.step(a)
    for key in host:
        host[key].copy_(o[key], non_blocking=True)
    hr.copy_(r, non_blocking=True); hs.copy_(info['status'], non_blocking=True)
    torch.cuda.synchronize()
for k in range(20):
    step(k)
t0 = time.perf_counter()
for k in range(200):
    step(20 + k)
dt = time.perf_counter() - t0
print('N %d: host hand-off every step: %.3f ms/step = %.0f env-steps/s; %.1f KB per env-step over PCIe (%.2f MB per step)' % (n, 1e3 * dt / 200, n * 200 / dt, nbytes / n / 1e3, nbytes / 1e6))
