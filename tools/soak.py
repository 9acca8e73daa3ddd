"""Long rollouts with the bench's action distribution (B): flagged envs, non-finite records, success counts, throughput"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticsplayroompybullet_amd import VecPlayEnv
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for gid in sys.argv[2:] or ['UR5PlayAbsRPY1Obj-v0', 'pandaPlayAbsRPY1Obj-v0', 'pandaPick-v0']:
    n = 4096
    env = VecPlayEnv(gid, n, seed=11)
    env.reset()
    g = torch.Generator(device='cuda').manual_seed(3)
    lo = torch.tensor([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0], device='cuda'); hi = torch.tensor([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0], device='cuda')
    if gid == 'pandaPick-v0':
        lo[:3] = torch.tensor([-0.18, -0.18, -0.05]); hi[:3] = torch.tensor([0.18, 0.18, 0.2])
    bad = torch.zeros(n, dtype=torch.int64, device='cuda'); succ = torch.zeros(n, dtype=torch.int64, device='cuda'); fell = torch.zeros(n, dtype=torch.bool, device='cuda')
    t0 = time.perf_counter()
    for t in range(steps):
        a = lo + (hi - lo) * torch.rand((n, 7), generator=g, device='cuda')
        obs, r, d, info = env.step(a)
        bad += info['status'] & 7; succ += info['is_success']; fell |= (info['status'] & 2) != 0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = env.get_state(); L = env.state_layout
    blk = s[:, L['free0'][0]:L['free0'][0] + 3]
    off = fell & ((blk[:, 0].abs() > 0.36) | (blk[:, 1] < -0.04) | (blk[:, 1] > 0.54))      # x, y beyond the table top's edges: pushed off, not pressed through
    print('   objects below the scene: %d (of them beyond the table top\'s edges in x / y: %d)' % (int(fell.sum()), int(off.sum())))
    print('%-24s %d steps x %d envs: flagged %d, non-finite %d, successes %d, block z min %.3f max %.3f, max |qd| %.1f, %.2f M env-steps/s'
          % (gid, steps, n, int((bad > 0).sum()), int((~torch.isfinite(s).all(dim=1)).sum()), int(succ.sum()),
             float(s[:, L['free0'][0] + 2].min()), float(s[:, L['free0'][0] + 2].max()), float(s[:, L['qd'][0]:L['qd'][1]].abs().max()), n * steps / dt / 1e6))
    env.close()
