import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo/tests')
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
from gpu_debug import record_from_oracle
from test_gpu_reference_step import reach_actions
kind='R'; n=16; steps=200
env = VecPlayEnv('UR5Reach-v0', n, seed=21); env.reset()
refs = [OracleEnv(kind, seed=21, env_index=e, bullet_ref=True) for e in range(n)]
fast = [OracleEnv(kind, seed=21, env_index=e) for e in range(n)]
f32 = [OracleEnv(kind, seed=21, env_index=e, f32=True) for e in range(n)]
for o, f, g in zip(refs, fast, f32):
    o.reset(); f.reset(); g.reset(); f.set_state(o.get_state()); g.set_state(o.get_state())
env.set_state(torch.tensor(np.stack([record_from_oracle(o) for o in refs])))
acts = reach_actions(steps, n, 3)
e=6
for t in range(steps):
    obs, r, done, info = env.step(torch.tensor(acts[t], dtype=torch.float32))
    q = env.get_state()[:, :6].cpu().numpy()
    a = acts[t, e].astype(np.float32).astype(np.float64)
    refs[e].step(a); fast[e].step(a); r32 = f32[e].step(a)
    qo = refs[e].get_state()[:6]
    d = np.abs(q[e]-qo).max(); df=np.abs(fast[e].get_state()[:6]-qo).max(); d32=np.abs(f32[e].get_state()[:6]-qo).max()
    if t >= 86 and t <= 93:
        print('   tp f32 oracle', np.round(r32[3]['target_poses'][:6], 5), 'dev - f32', np.round(info['target_poses'][e].cpu().numpy()[:6] - r32[3]['target_poses'][:6], 6))
    if d > 2e-5 or t % 25 == 0:
        print('t %3d device-ref %.2e  fast64-ref %.2e  fast32-ref %.2e status %d tp dev %s' % (t, d, df, d32, int(info['status'][e]), np.round(info['target_poses'][e].cpu().numpy()[:6],5)))
    if d > 2e-3: break
