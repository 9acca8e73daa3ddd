"""pandaPush-v0: per-step divergence of the device vs the fp32 oracle (diagnostic)"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'oracle')); sys.path.insert(0, os.path.join(R, 'tests'))
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
from test_gpu_parity import actions
n = 6
env = VecPlayEnv('pandaPush-v0', n, seed=31); env.reset()
orc = [OracleEnv('pandaPush-v0', seed=31, env_index=e, f32=True) for e in range(n)]
for o in orc: o.reset()
acts = actions('P', 8, n, 4); acts[..., 2] = -0.03 + 0.05 * acts[..., 2]
for t in range(8):
    obs, r, d, info = env.step(torch.tensor(acts[t], dtype=torch.float32))
    rc = env.debug_row_counts().numpy()
    for e, o in enumerate(orc):
        oo = o.step(acts[t, e])[0]
        g, w = obs['obs_quat'][e].cpu().numpy(), oo['obs_quat']
        dj = np.abs(obs['joints'][e].cpu().numpy() - oo['joints']).max() if 'joints' in oo else -1
        print(t, e, 'dmax pos %.2e vel %.2e joints %.2e rows %s' % (np.abs(g - w)[[0,1,2,6,7,8,9]].max(), np.abs(g - w)[[3,4,5,10,11,12]].max(), dj, rc[e]))
