#!/usr/bin/env python3
"""Where the two-env path of k_solve2 spends its prologue (profiling build -DRP_PROLOGUE_CLOCKS, RP_PLAYROOM_LIB=tools/prologue_clocks.so): shader clocks between marks of the last launch"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from roboticsplayroompybullet_amd import VecPlayEnv
n = 4096
env = VecPlayEnv(bench.ENV_ID, n, seed=1234); env.set_groups(1); env.reset()
acts = bench.make_actions(n, 40, env.device, 1234)
buf = (C.c_uint64 * (8 * 2048))()
env.lib.rp_debug_prologue_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
C.memset(buf, 0, 8 * 8 * 2048)
for k in range(40): env.step(acts[k])
torch.cuda.synchronize()
assert env.lib.rp_debug_prologue_clocks(env.h, buf, 2048) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8).astype(np.int64)
d = np.diff(a[:, :7], axis=1)
# marks are only written by two-env waves of the LAST launch; stale rows (older launches) are still two-env rows: keep rows whose total is sane
ok = (a[:, 0] > 0) & (d > 0).all(axis=1) & (d.sum(axis=1) < 200000)
recent = a[:, 0] >= a[ok, 0].max() - 400000 if ok.any() else ok
ok &= recent
names = ['pair table + header', 'issue unit-row / plane loads', 'wait for the staged rows', 'issue contact plane loads', 'LDS gathers of the row registers', 'wait for every load']
print('two-env waves of the last launch:', int(ok.sum()))
for i, nm in enumerate(names):
    print('%-36s cycles p50 %6d p90 %6d' % (nm, np.median(d[ok, i]), np.percentile(d[ok, i], 90)))
print('%-36s cycles p50 %6d' % ('total', np.median(d[ok].sum(axis=1))))
if (a[ok, 7] > 0).any():
    print('gathers, first pass %d cycles, second pass (same code again) %d cycles' % (np.median(a[ok, 7] - a[ok, 4]), np.median(a[ok, 5] - a[ok, 7])))
