"""ms per step of the one-kernel path (rp_set_fused(1): the test-only reference pipeline) on the bench workload - 14.8 ms at N = 4096 against 1.9 ms for the split pipeline"""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from roboticsplayroompybullet_amd import VecPlayEnv
n = 4096
env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
env.set_fused(1)
env.reset()
acts = bench.make_actions(n, 70, env.device, 1234)
for k in range(10): env.step(acts[k])
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(50): env.step(acts[10 + k])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('fused: %.3f ms/step' % (dt * 1e3 / 50))
