#!/usr/bin/env python3
"""Trace per-step divergence HIP vs oracle (fp64 and fp32) for the 200-step rollout of tests/test_gpu_parity.py."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tests')); sys.path.insert(0, os.path.join(REPO, 'tools'))
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
from gpu_debug import record_from_oracle, oracle_state_from_record
from test_gpu_parity import actions, IDS
np.set_printoptions(precision=5, suppress=True, linewidth=220)
kind = sys.argv[1] if len(sys.argv) > 1 else 'U'
n, steps = 4, 200
env = VecPlayEnv(IDS[kind], n, seed=9); env.reset()
o64 = [OracleEnv(kind, seed=9, env_index=e) for e in range(n)]
for o in o64: o.reset()
recs = np.stack([record_from_oracle(o) for o in o64])
env.set_state(torch.tensor(recs))
o32 = [OracleEnv(kind, seed=9, env_index=e, f32=True) for e in range(n)]
for e, o in enumerate(o32):
    o.reset(); o.set_state(oracle_state_from_record(o, recs[e]))
    # motors/goal: replay not available through set_state; oracle32 reset gives its own motors (defaults) -> same as o64 after reset
acts = actions(kind, steps, n, 5)
na = o64[0].n_arm
for t in range(steps):
    obs, r, d, info = env.step(torch.tensor(acts[t], dtype=torch.float32))
    g = env.get_state().cpu().numpy()
    line = []
    for e in range(n):
        a = acts[t, e].astype(np.float32).astype(np.float64)
        o64[e].step(a); o32[e].step(a)
        q64 = o64[e].get_state(); q32 = o32[e].get_state()
        qg = oracle_state_from_record(o64[e], g[e])
        line.append((np.abs(qg[:na] - q64[:na]).max(), np.abs(qg[:na] - q32[:na]).max(), np.abs(q32[:na] - q64[:na]).max(),
                     np.abs(qg[2*na:] - q64[2*na:]).max() if len(qg) > 2*na else 0))
    if t % 10 == 0 or max(l[0] for l in line) > 5e-4:
        print(t, ' '.join('[g-64 %.1e g-32 %.1e 32-64 %.1e rest %.1e]' % l for l in line))
