"""step / reset cost of the two-object ids (wide build) next to the one-object id of the same arm and scene"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticsplayroompybullet_amd import VecPlayEnv
cfgs = [(a.split(':')[0], int(a.split(':')[1])) for a in sys.argv[1:]] or [('pandaPlay-v0', 4096), ('pandaPlay1Obj-v0', 4096), ('pandaPlay-v0', 1024), ('pandaPlay1Obj-v0', 1024)]
for gid, n in cfgs:
    env = VecPlayEnv(gid, n, seed=1)
    t0 = time.perf_counter(); env.reset(); torch.cuda.synchronize(); tr = time.perf_counter() - t0
    g = torch.Generator(device='cuda').manual_seed(1)
    lo = torch.tensor([-0.18, 0.0, 0.05, -0.1, -0.1, -0.1, 0.9, -1.0], device='cuda'); hi = torch.tensor([0.18, 0.3, 0.3, 0.1, 0.1, 0.1, 1.0, 1.0], device='cuda')
    acts = lo + (hi - lo) * torch.rand((120, n, 8), generator=g, device='cuda')
    for t in range(20): env.step(acts[t])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(20, 120): o, r, d, info = env.step(acts[t])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('%s N=%d: reset %.0f ms, %.2f ms/step, %.0f env-steps/s, flagged %d' % (gid, n, 1e3 * tr, 1e3 * dt / 100, n * 100 / dt, int((info['status'] & 7).sum())))
    env.close()
