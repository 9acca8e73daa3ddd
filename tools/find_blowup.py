"""Find the first env / step whose record goes non-finite under stress actions; dump the record before that step and the action
to gpurun_out/blowup.npz for a CPU post-mortem with the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticsplayroompybullet_amd import VecPlayEnv
gid = sys.argv[1] if len(sys.argv) > 1 else 'UR5PlayRelJoints1Obj-v0'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = 4096
env = VecPlayEnv(gid, n, seed=3)
env.reset()
hi = env.action_high
g = torch.Generator(device='cuda').manual_seed(1)
hist = []
for t in range(steps):
    a = (2 * torch.rand((n, hi.numel()), generator=g, device='cuda') - 1) * hi
    prev = env.get_state().clone()
    hist.append((prev, a.clone()))
    hist = hist[-6:]
    obs, r, d, info = env.step(a)
    s = env.get_state()
    badm = ~torch.isfinite(s).all(dim=1) | (info['status'] != 0) | (s[:, 12:24].abs().max(dim=1).values > 200)
    if badm.any():
        e = int(badm.nonzero()[0])
        print('step', t, 'env', e, 'status', int(info['status'][e]), 'record finite', bool(torch.isfinite(s[e]).all()))
        os.makedirs('gpurun_out', exist_ok=True)
        np.savez('gpurun_out/blowup.npz', gid=gid, step=t, env=e, recs=np.stack([h[0][e].cpu().numpy() for h in hist]),
                 acts=np.stack([h[1][e].cpu().numpy() for h in hist]), after=s[e].cpu().numpy())
        print('qd before', prev[e, 12:24].cpu().numpy())
        print('after    ', s[e, 0:24].cpu().numpy())
        break
else:
    print('no blow-up in', steps, 'steps')
