#!/usr/bin/env python3
"""Round 6: per-wave timestamps of the last k_solve2 launch with the heavy path's worker blocks (profiling build: tools/build_profiling_libs.sh, RP_PLAYROOM_LIB=tools/clocks1.so).

g_clk words per wave: 0 start, 1 rows / columns ready, 2 sweeps done, 3 end (shader clock), 4 / 5 start / end (100 MHz wall clock), 6 flags, 7 xcc << 32 | hw_id
flags: bit 30 = four-env wave; bit 28 = worker wave (bit 29: it solved a heavy env: unit lanes | contacts << 8 | torsional rows << 16); else a two-env wave (solve2_body).
env: N_ENVS, STEPS, GROUPS (default 1), DIST=A, RP_HV_WAVES
"""
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
env.set_groups(int(os.environ.get('GROUPS', '1')))
env.reset()
acts = bench.make_actions(n, steps, env.device, 1234)
if os.environ.get('DIST') == 'A':
    acts = (2 * torch.rand((steps, n, 7), generator=torch.Generator(device=env.device).manual_seed(4321), device=env.device) - 1) * env.action_high
for k in range(steps):
    env.step(acts[k])
torch.cuda.synchronize()
nw = 4096
buf = (C.c_uint64 * (8 * nw))()
env.lib.rp_debug_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
rc = env.lib.rp_debug_clocks(env.h, buf, nw)
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 8).astype(np.int64)
fl = a[:, 6]
four = ((fl >> 30) & 1) == 1
worker = ((fl >> 28) & 1) == 1
heavy = ((fl >> 29) & 1) == 1
live = (a[:, 4] > 0) & (a[:, 5] > 0)
two = live & ~four & ~worker & ~heavy
t0 = a[live, 4].min()
us = lambda x: x / 100.0


def pct(x, ps=(50, 90, 100)):
    return tuple(np.percentile(x, ps)) if len(x) else tuple(0 for _ in ps)


print('waves with marks: %d; worker waves %d (heavy list length %s), of them solved an env: %d; four-env waves %d; two-env waves %d' % (
    live.sum(), worker.sum(), np.unique(fl[worker & ~heavy] & 0xffff).tolist()[:4], (heavy & live).sum(), (four & live).sum(), two.sum()))
print('launch span: %.1f us' % us(a[live, 5].max() - t0))
for name, sel in (('heavy (one env per wave)', heavy & live), ('four-env', four & live), ('two-env (solve2_body)', two)):
    if sel.sum() == 0:
        continue
    d = us(a[sel, 5] - a[sel, 4]); e = us(a[sel, 5] - t0); st = us(a[sel, 4] - t0)
    print('%-26s n %4d  start p50 %.1f max %.1f | duration p50 %.1f p90 %.1f max %.1f | end p50 %.1f p90 %.1f max %.1f us' % ((name, sel.sum()) + pct(st, (50, 100)) + pct(d) + pct(e)))
if (heavy & live).any():
    h = a[heavy & live]
    build, sweep, tail = h[:, 1] - h[:, 0], h[:, 2] - h[:, 1], h[:, 3] - h[:, 2]
    nu, nc, nt = h[:, 6] & 255, (h[:, 6] >> 8) & 255, (h[:, 6] >> 16) & 255
    rows = nu + 3 * nc + nt
    print('heavy waves: build cycles p50 %d p90 %d max %d | sweeps p50 %d p90 %d max %d | tail p50 %d max %d' % (pct(build) + pct(sweep) + pct(tail, (50, 100))))
    idx = np.nonzero(heavy & live)[0]
    sub = a[idx + 1, :4] - a[idx, 0:1]      # the idle second wave's slots: header there, tables scattered, X rows loaded, columns built (cycles since the start)
    print('heavy waves: cycles since start p50: header there %d, tables scattered %d, X rows loaded %d, columns built %d, sweeps start %d' % (tuple(np.median(sub, 0)) + (np.median(build),)))
    print('heavy waves: row steps per sweep p50 %d max %d; contacts p50 %d max %d; cycles per row step p50 %.1f p10 %.1f p90 %.1f' % (
        np.median(rows), rows.max(), np.median(nc), nc.max(), *np.percentile(sweep / (50.0 * rows), [50, 10, 90])))
    A = np.stack([np.ones(len(h)), nu.astype(float), nc.astype(float)], 1)
    coef = np.linalg.lstsq(A, sweep / 50.0, rcond=None)[0]
    print('heavy sweep cycles ~ %.0f + %.0f * unit rows + %.0f * contacts' % tuple(coef))
    A = np.stack([np.ones(len(h)), rows.astype(float)], 1)
    coef = np.linalg.lstsq(A, build.astype(float), rcond=None)[0]
    print('heavy build cycles ~ %.0f + %.0f * rows' % tuple(coef))
f4 = four & live & (((fl >> 8) & 255) < 64)
if f4.any() and False:
    T4, nS4, nl4 = fl[f4] & 1, (fl[f4] >> 8) & 255, (fl[f4] >> 16) & 255
    d4 = us(a[f4, 5] - a[f4, 4])
    for T in (0, 1):
        sel = T4 == T
        print('four-env stream %d: slots in use p50 %d p90 %d max %d; limit rows p50 %d max %d; duration by slots:' % (T, *np.percentile(nS4[sel], [50, 90, 100]), np.median(nl4[sel]), nl4[sel].max()),
              ' '.join('%d:%.0f(%d)' % (k, np.median(d4[sel & (nS4 == k)]), (sel & (nS4 == k)).sum()) for k in np.unique(nS4[sel])))
order = np.argsort(-np.where(live, a[:, 5], 0))[:10]
print('last finishers: wave kind start_us end_us')
for b in order:
    print(b, 'four' if four[b] else ('heavy' if heavy[b] else 'two'), '%.1f %.1f' % (us(a[b, 4] - t0), us(a[b, 5] - t0)), 'flags', hex(fl[b] & 0xffffff))
