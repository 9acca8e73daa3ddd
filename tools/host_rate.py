import sys, time, os, torch
sys.path.insert(0, '/root/repo')
import bench
from roboticsplayroompybullet_amd import VecPlayEnv
n = 4096
env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 3
env.set_groups(g)
env.reset()
acts = bench.make_actions(n, 220, env.device, 1234)
for k in range(20): env.step(acts[k])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(200): env.step(acts[20 + k])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('groups', g, 'host enqueue %.3f ms/step, total %.3f ms/step' % ((t1 - t0) * 1e3 / 200, (t2 - t0) * 1e3 / 200))
