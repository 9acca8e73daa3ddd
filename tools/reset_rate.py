"""reset() cost: full reset of N envs and masked resets, split pipeline vs the fused one-kernel reset (run on the GPU box)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticsplayroompybullet_amd import VecPlayEnv

for gid in sys.argv[1:] or ['UR5PlayAbsRPY1Obj-v0', 'pandaPick-v0', 'UR5Reach-v0']:
    for n in (4096, 256):
        for fused in (0, 1):
            env = VecPlayEnv(gid, n, seed=1)
            env.set_fused(fused)
            env.reset(); torch.cuda.synchronize()
            t0 = time.perf_counter(); env.reset(); torch.cuda.synchronize(); full = time.perf_counter() - t0
            mask = torch.zeros(n, dtype=torch.uint8); mask[::64] = 1
            t0 = time.perf_counter(); env.reset(mask=mask); torch.cuda.synchronize(); part = time.perf_counter() - t0
            print('%-24s N=%5d %s: full reset %7.1f ms, masked (%d envs) %7.1f ms' % (gid, n, 'fused' if fused else 'split', 1e3 * full, int(mask.sum()), 1e3 * part))
            env.close()
