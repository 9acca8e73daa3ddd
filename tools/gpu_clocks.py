#!/usr/bin/env python3
"""Per-wave phase timestamps of the last k_solve2 launch (profiling build: hipcc -DRP_CLOCKS, RP_PLAYROOM_LIB=clocks.so).

columns of g_clk per block: 0 start, 1 rows loaded, 2 sweeps done, 3 end (shader clock), 4/5 start/end (100 MHz wall
clock), 6 na|nj<<8|nc<<16|par<<24, 7 xcc<<32|hw_id
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticsplayroompybullet_amd import VecPlayEnv  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(os.environ.get('N_ENVS', '4096'))
steps = int(os.environ.get('STEPS', '30'))
env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
env.set_groups(int(os.environ.get('GROUPS', '1')))
env.reset()
acts = bench.make_actions(n, steps, env.device, 1234)
if os.environ.get('DIST') == 'A':      # the literal random-action rollout: a ~ U(action_space)
    acts = (2 * torch.rand((steps, n, 7), generator=torch.Generator(device=env.device).manual_seed(4321), device=env.device) - 1) * env.action_high
for k in range(steps):
    env.step(acts[k])
torch.cuda.synchronize()
nb = (n + 1) // 2
buf = (C.c_uint64 * (8 * nb))()
env.lib.rp_debug_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
rc = env.lib.rp_debug_clocks(env.h, buf, nb)
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
na, nj, nc, par = a[:, 6] & 255, (a[:, 6] >> 8) & 255, (a[:, 6] >> 16) & 255, (a[:, 6] >> 24) & 1
xcc = a[:, 7] >> 32
hw = a[:, 7] & 0xffffffff
t_load, t_sweep, t_tail = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
w0, w1 = a[:, 4] - a[:, 4].min(), a[:, 5] - a[:, 4].min()
print('blocks', nb, 'kernel span (wall 100MHz ticks)', w1.max(), '= %.1f us' % (w1.max() / 100.0))
print('start offsets us: p50 %.1f p90 %.1f max %.1f' % tuple(np.percentile(w0, [50, 90, 100]) / 100.0))
print('end offsets us:   p50 %.1f p90 %.1f max %.1f' % tuple(np.percentile(w1, [50, 90, 100]) / 100.0))
steps_par = np.maximum(na, nj + 3 * nc)      # approx (chunking ignored)
steps_seq = na + nj + 3 * nc
stp = np.where(par == 1, steps_par, steps_seq)
clk_per_wall = (a[:, 3] - a[:, 0]).astype(np.float64) / np.maximum(a[:, 5] - a[:, 4], 1)
print('shader clocks per 100MHz tick: median %.2f' % np.median(clk_per_wall))
two = ((a[:, 6] >> 30) & 1) == 0
print('load   cycles (two-env path waves): p50 %d p90 %d max %d' % tuple(np.percentile(t_load[two], [50, 90, 100])) if two.any() else 'no two-env waves')
print('sweeps cycles: p50 %d p90 %d max %d' % tuple(np.percentile(t_sweep, [50, 90, 100])))
print('tail   cycles: p50 %d p90 %d max %d' % tuple(np.percentile(t_tail, [50, 90, 100])))
for name, sel in (('PAR', par == 1), ('SEQ', par == 0)):
    if sel.sum() == 0:
        continue
    cps = t_sweep[sel] / (50.0 * np.maximum(stp[sel], 1))
    print('%s waves %d: rows/sweep p50 %d max %d; cycles per row-step p50 %.1f p10 %.1f p90 %.1f' % (
        name, sel.sum(), np.median(stp[sel]), stp[sel].max(), np.median(cps), np.percentile(cps, 10), np.percentile(cps, 90)))
for name, sel in (('PAR', par == 1), ('SEQ', par == 0)):
    if sel.sum() < 8:
        continue
    A = np.stack([np.ones(sel.sum()), (na[sel] - 12).astype(float), nc[sel].astype(float)], 1)
    y = t_sweep[sel] / 50.0
    coef, res, _, _ = np.linalg.lstsq(A, y, rcond=None)
    print('%s sweep cycles ~ %.0f + %.0f * n_limit_rows + %.0f * n_contacts   (rms resid %.0f)' % (name, coef[0], coef[1], coef[2], np.sqrt(np.mean((A @ coef - y) ** 2))))
order = np.argsort(-w1)[:12]
print('last finishers: blk start_us end_us load sweep tail na nj nc par xcc hw')
for b in order:
    print(b, '%.1f %.1f' % (w0[b] / 100.0, w1[b] / 100.0), t_load[b], t_sweep[b], t_tail[b], na[b], nj[b], nc[b], par[b], xcc[b], hex(hw[b]))
# resident waves per (xcc, cu): hw_id bits: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3
key = xcc * 100000 + se * 1000 + sh * 100 + cu
u, cnt = np.unique(key, return_counts=True)
print('distinct CUs used', len(u), 'waves per CU: min %d p50 %d max %d' % (cnt.min(), np.median(cnt), cnt.max()))
key2 = key * 10 + simd
u2, cnt2 = np.unique(key2, return_counts=True)
print('distinct SIMDs used', len(u2), 'waves per SIMD: min %d p50 %d max %d' % (cnt2.min(), np.median(cnt2), cnt2.max()))
four = (a[:, 6] >> 30) & 1
if four.any():
    ld4 = (a[four == 1, 1] - a[four == 1, 4]) / 100.0
    print('four-env waves: rows loaded after p10 %.1f p50 %.1f p90 %.1f max %.1f us of their p50 %.1f us' % (tuple(np.percentile(ld4, [10, 50, 90, 100])) + (np.median((a[four == 1, 5] - a[four == 1, 4]) / 100.0),)))
    d4 = (a[four == 1, 5] - a[four == 1, 4]) / 100.0
    d2 = (a[four == 0, 5] - a[four == 0, 4]) / 100.0
    print('waves on the four-env path: %d of %d; their duration us: p50 %.1f p90 %.1f max %.1f; two-env path waves: %d, duration us p50 %.1f max %.1f' % (
        int(four.sum()), nb, np.median(d4), np.percentile(d4, 90), d4.max(), int((four == 0).sum()), np.median(d2) if len(d2) else 0, d2.max() if len(d2) else 0))
    e4 = (a[four == 1, 5] - a[:, 4].min()) / 100.0
    e2 = (a[four == 0, 5] - a[:, 4].min()) / 100.0
    print('end offsets us: four-env path max %.1f; two-env path max %.1f' % (e4.max(), e2.max() if len(e2) else 0))
