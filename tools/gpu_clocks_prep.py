#!/usr/bin/env python3
"""Per-wave phase timestamps of the last k_prep2 launch (profiling build: hipcc -DRP_CLOCKS=2, RP_PLAYROOM_LIB=clocks2.so).
columns: 0 start, 1 after FK/AABB, 2 after collide, 3 after dynamics, 4 after rows, 5 end (shader clock); 6/7 start/end wall clock (100 MHz)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticsplayroompybullet_amd import VecPlayEnv  # noqa: E402
import bench  # noqa: E402

n = int(os.environ.get('N_ENVS', '4096'))
steps = int(os.environ.get('STEPS', '30'))
env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
env.set_groups(1)
env.reset()
acts = bench.make_actions(n, steps, env.device, 1234)
if os.environ.get('DIST') == 'A':      # the literal random-action rollout: a ~ U(action_space)
    acts = (2 * torch.rand((steps, n, 7), generator=torch.Generator(device=env.device).manual_seed(4321), device=env.device) - 1) * env.action_high
for k in range(steps):
    env.step(acts[k])
torch.cuda.synchronize()
buf = (C.c_uint64 * (32 * n))()
env.lib.rp_debug_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
assert env.lib.rp_debug_clocks(env.h, buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 32).astype(np.int64)
w0, w1 = a[:, 6] - a[:, 6].min(), a[:, 7] - a[:, 6].min()
print('waves', n, 'kernel span %.1f us' % (w1.max() / 100.0))
print('start offsets us: p10 %.1f p50 %.1f p90 %.1f max %.1f' % tuple(np.percentile(w0, [10, 50, 90, 100]) / 100.0))
print('end offsets us:   p10 %.1f p50 %.1f p90 %.1f max %.1f' % tuple(np.percentile(w1, [10, 50, 90, 100]) / 100.0))
names = ['load+FK+AABB', 'collide', 'dynamics', 'rows', 'output']
for i, nm in enumerate(names):
    d = a[:, i + 1] - a[:, i]
    print('%-14s cycles: p10 %d p50 %d p90 %d max %d' % ((nm,) + tuple(np.percentile(d, [10, 50, 90, 100]))))
tot = a[:, 5] - a[:, 0]
print('%-14s cycles: p10 %d p50 %d p90 %d max %d' % (('total',) + tuple(np.percentile(tot, [10, 50, 90, 100]))))
for nm, i0, i1 in (('  load state', 0, 16), ('  FK', 16, 17), ('   FK local', 16, 19), ('   FK chain', 19, 20), ('   FK rest', 20, 17), ('  subspaces', 17, 18), ('  AABBs', 18, 1), ('  broadphase', 1, 8), ('  narrowphase', 8, 9), ('  manifolds', 9, 10), ('   cache load', 9, 21), ('   (a) (b)', 21, 22), ('   (c) add', 22, 23), ('   (d) (e)', 23, 24), ('   cache store', 24, 25), ('  collide tail', 10, 2)):
    d = a[:, i1] - a[:, i0]
    print('%-14s cycles: p10 %d p50 %d p90 %d max %d' % ((nm,) + tuple(np.percentile(d, [10, 50, 90, 100]))))
for nm, i0, i1 in (('  inertia+comp', 2, 11), ('  M, bias, tau', 11, 13), ('  chol+inverse', 13, 14), ('  vstar etc', 14, 3)):
    d = a[:, i1] - a[:, i0]
    print('%-14s cycles: p10 %d p50 %d p90 %d max %d' % ((nm,) + tuple(np.percentile(d, [10, 50, 90, 100]))))
na = a[:, 12]
print('active pairs: p10 %d p50 %d p90 %d max %d' % tuple(np.percentile(na, [10, 50, 90, 100])))
npd = (a[:, 9] - a[:, 8]).astype(float)
print('narrowphase cycles vs active pairs: corr %.2f' % np.corrcoef(na, npd)[0, 1])
late = w0 > np.percentile(w0, 50)
print('first-round waves: total p50 %d; second-round waves: total p50 %d' % (np.median(tot[~late]), np.median(tot[late])))
hp, ha, hpr, hob = a[:, 15] & 0xFFFF, (a[:, 15] >> 16) & 0xFFFF, (a[:, 15] >> 32) & 0xFFFF, (a[:, 15] >> 48) & 0xFFFF
print('hull pairs per env: stopped by the 15-axis OBB test %.2f, scanned %.2f, of those apart along a box axis or the probe direction %.2f (by the probe alone %.2f); GJK calls %.2f, GJK rounds %.2f' % (hob.mean(), hp.mean(), ha.mean(), hpr.mean(), (a[:, 27] & 0xFFFF).mean(), a[:, 26].mean()))
print('GJK outcomes per env: apart %.3f, contact %.3f, cores overlap (OBB path) %.3f' % (((a[:, 27] >> 16) & 0xFFFF).mean(), ((a[:, 27] >> 32) & 0xFFFF).mean(), ((a[:, 27] >> 48) & 0xFFFF).mean()))
a[:, 27] &= 0xFFFF
print('hull pairs scanned per env: mean %.2f p90 %d max %d; of them apart (two scans for nothing): mean %.2f' % (hp.mean(), np.percentile(hp, 90), hp.max(), ha.mean()))
g = a[:, 27] > 0
if g.any():
    r = a[g, 26].astype(float)
    pass
    print('GJK: waves with a call %d of %d; calls per such wave p50 %d max %d; rounds per call p50 %.1f max %.1f; cycles per call p50 %d max %d' % (
        g.sum(), n, np.median(a[g, 27]), a[g, 27].max(), np.median(a[g, 26] / a[g, 27]), (a[g, 26] / a[g, 27]).max(), np.median(a[g, 28] / a[g, 27]), (a[g, 28] / a[g, 27]).max()))

hull = a[:, 19] - a[:, 8]
batches = a[:, 9] - a[:, 19]
print('narrowphase = hull phase (decisions, classes, hull_item16: four pairs at a time) p50 %d p90 %d max %d + the eight-lanes-per-pair batches p50 %d p90 %d max %d cycles' % (
    np.median(hull), np.percentile(hull, 90), hull.max(), np.median(batches), np.percentile(batches, 90), batches.max()))
print('hull_item16, row 0\'s items, cycles per env with any: set-up + 15-axis test p50 %d p90 %d max %d | face scan p50 %d p90 %d max %d | GJK p50 %d p90 %d max %d' % (
    tuple(np.percentile(a[hp > 0, 29], [50, 90, 100])) + tuple(np.percentile(a[hp > 0, 30], [50, 90, 100])) + tuple(np.percentile(a[hp > 0, 31], [50, 90, 100]))))
# the launch lasts as long as its slowest blocks: what are they made of?
order = np.argsort(-tot)[:12]
print('the 12 slowest blocks: total | load+FK+AABB, broadphase, hull phase, batches, manifolds, wait for the other wave, rows after the join [k cycles] | active pairs, hull pairs scanned, GJK calls, GJK rounds | row 0: set-up, face scan, GJK [k cycles]')
for i in order:
    join = max(a[i, 2], a[i, 3])
    print('  %6.1f | %5.1f %5.1f %6.1f %6.1f %5.1f %5.1f %5.1f | %d %d %d %d' % (tot[i] / 1e3, (a[i, 1] - a[i, 0]) / 1e3, (a[i, 8] - a[i, 1]) / 1e3, (a[i, 19] - a[i, 8]) / 1e3, (a[i, 9] - a[i, 19]) / 1e3,
                                                                          (a[i, 2] - a[i, 9]) / 1e3, (join - a[i, 2]) / 1e3, (a[i, 4] - join) / 1e3, a[i, 12], hp[i], a[i, 27], a[i, 26]) + ' | %.1f %.1f %.1f' % (a[i, 29] / 1e3, a[i, 30] / 1e3, a[i, 31] / 1e3))
