import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from roboticsplayroompybullet_amd import VecPlayEnv
for gid, n in (('pandaPlay-v0', 4096), ('pandaPlay-v0', 1024)):
    env = VecPlayEnv(gid, n, seed=1)
    t0 = time.perf_counter(); env.reset(); torch.cuda.synchronize(); tr = time.perf_counter() - t0
    g = torch.Generator(device='cuda').manual_seed(1)
    lo = torch.tensor([-0.18, 0.0, 0.05, -0.1, -0.1, -0.1, 0.9, -1.0], device='cuda'); hi = torch.tensor([0.18, 0.3, 0.3, 0.1, 0.1, 0.1, 1.0, 1.0], device='cuda')
    acts = lo + (hi - lo) * torch.rand((30, n, 8), generator=g, device='cuda')
    for t in range(5): env.step(acts[t])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(5, 30): o, r, d, info = env.step(acts[t])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%s N=%d: reset %.0f ms, %.1f ms/step, %.0f env-steps/s, flagged %d' % (gid, n, 1e3 * tr, 1e3 * dt / 25, n * 25 / dt, int(info['status'].sum())))
