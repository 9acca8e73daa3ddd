import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
ID = 'UR5PlayAbsRPY1Obj-v0'
n, steps = 6, 40
env = VecPlayEnv(ID, n, seed=21); obs = env.reset()
ors = [OracleEnv('U', seed=21, env_index=e, f32=True) for e in range(n)]
for o in ors: o.reset()
span_seen = 0; arm_seen = 0
maxd = []
for t in range(steps):
    blk = obs['achieved_goal'][:, 0:3].cpu().numpy().copy()
    a = np.zeros((n, 7)); a[:, 0:3] = blk; a[:, 2] = 0.02 if t < 25 else 0.15; a[:, 6] = -1.0 if t < 12 else 1.0
    obs, r, d, info = env.step(torch.tensor(a, dtype=torch.float32))
    rc = env.debug_row_counts()
    span_seen += int((rc[:, 3] > 0).sum()); arm_seen += int((rc[:, 2] > 0).sum())
    dd = 0
    for e, o in enumerate(ors):
        oo, _, _, _ = o.step(a[e])
        dd = max(dd, float(np.abs(obs['obs_quat'][e].cpu().numpy() - oo['obs_quat']).max()))
    maxd.append(dd)
print('arm-contact env-steps', arm_seen, 'spanning', span_seen)
print('max |obs_quat diff| per step:', ' '.join('%.1e' % x for x in maxd))
