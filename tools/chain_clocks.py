#!/usr/bin/env python3
"""Where k_chain's time goes: wall clock per block in its two phases (profiling build -DRP_CHAIN_CLOCKS, RP_PLAYROOM_LIB=tools/chain_clocks.so).
    python tools/chain_clocks.py [N]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from roboticsplayroompybullet_amd import VecPlayEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
env.set_fused(2)
env.reset()
acts = bench.make_actions(n, 60, env.device, 1234)
for k in range(60):
    env.step(acts[k])
torch.cuda.synchronize()
nb = min((n + 3) // 4, 1024)
buf = (C.c_int64 * (4 * nb))()
env.lib.rp_debug_chain_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
assert env.lib.rp_debug_chain_clocks(env.h, buf, nb) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(nb, 4) / 100.0      # us
units = -(-((n + 3) // 4) // nb) * 12
print('N = %d, %d blocks, %d units (4 preparations + 1 solve) per block and step' % (n, nb, units))
print('kernel span %.1f us; block lifetime p50 %.1f max %.1f us' % (a[:, 3].max() - a[:, 2].min(), np.median(a[:, 3] - a[:, 2]), (a[:, 3] - a[:, 2]).max()))
print('per unit: four preparations p10 %.1f p50 %.1f p90 %.1f max %.1f us; solve p10 %.1f p50 %.1f p90 %.1f max %.1f us' % (
    tuple(np.percentile(a[:, 0] / units, [10, 50, 90, 100])) + tuple(np.percentile(a[:, 1] / units, [10, 50, 90, 100]))))
