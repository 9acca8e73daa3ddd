#!/usr/bin/env python3
"""Histogram of motor/limit rows and contact points per env-substep in the bench workload (GPU box)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticsplayroompybullet_amd import VecPlayEnv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_actions
n = 4096
env = VecPlayEnv('UR5PlayAbsRPY1Obj-v0', n, seed=1234); env.reset()
acts = make_actions(n, 40, env.device, 1234)
env.lib.rp_debug_row_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
buf = (C.c_int32 * (2 * n))()
allc = []
for t in range(40):
    env.step(acts[t])
    if t >= 10 and t % 5 == 0:
        env.lib.rp_debug_row_counts(env.h, buf)
        a = np.frombuffer(buf, dtype=np.int32).reshape(n, 2).copy()
        allc.append(a)
a = np.concatenate(allc)
print('nsmall: mean %.1f max %d' % (a[:, 0].mean(), a[:, 0].max()), np.bincount(a[:, 0])[12:])
span = a[:, 1] // 100000; a[:, 1] = a[:, 1] % 100000
cp = a[:, 1] >= 1000; a[:, 1] = a[:, 1] % 1000
print('coupled envs: %d; of them with spanning contacts (arm + other body in one row): %.3f; spanning contacts per coupled env: mean %.2f; contacts per coupled env: mean %.2f' % (cp.sum(), (span[cp] > 0).mean(), span[cp].mean(), a[cp, 1].mean()))
hv = cp & (a[:, 1] >= 12)
print('heavy coupled envs (>=12 contacts): %d; spanning contacts mean %.2f of %.2f; with none spanning: %.3f' % (hv.sum(), span[hv].mean(), a[hv, 1].mean(), (span[hv] == 0).mean()))
print('coupled fraction (env-substeps with an arm contact): %.3f; pairs with any coupled: %.3f' % (cp.mean(), (cp[0::2] | cp[1::2]).mean()))
print('ncon: mean %.2f max %d' % (a[:, 1].mean(), a[:, 1].max()))
print('ncon hist', np.bincount(a[:, 1], minlength=22))
print('cum frac <=k', np.round(np.cumsum(np.bincount(a[:, 1], minlength=22)) / len(a), 3))
pair = np.maximum(a[0::2, 1], a[1::2, 1])
print('pair-max ncon cum', np.round(np.cumsum(np.bincount(pair, minlength=22)) / len(pair), 3))
print('env-substeps with contacts that touch both halves of the velocity layout (nC > 0: the two-env solve path): %.4f; with more than 16 row-1 or 8 row-0 contacts: n/a' % (span > 0).mean())
q = (span > 0).reshape(-1, 4).any(axis=1) if len(span) % 4 == 0 else None
if q is not None:
    print('quads (4 consecutive envs in index order - the launch pairs by load class, so this is an upper bound) with a coupled env: %.4f' % q.mean())
