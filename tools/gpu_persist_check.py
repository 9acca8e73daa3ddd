#!/usr/bin/env python3
"""GPU box: the device with RP_CFG_PERSISTENT_MANIFOLDS against the oracle with RPO_RULE_PERSIST (fp64 and fp32), 16 envs x `steps` steps from the same
post-reset states; split pipeline vs fused kernel in that mode, bit for bit.   python tools/gpu_persist_check.py [U] [steps]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tools')); sys.path.insert(0, os.path.join(REPO, 'tests'))
from oracle import OracleEnv
from test_gpu_parity import actions, IDS, arm_q
from tolerances import N_MAIN
from gpu_debug import record_from_oracle
from roboticsplayroompybullet_amd import VecPlayEnv
kind = sys.argv[1] if len(sys.argv) > 1 else 'U'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = 16
RULE = int(os.environ['RP_ORACLE_RULE']) if 'RP_ORACLE_RULE' in os.environ else (1015 if 'RP_NO_GJK' in os.environ else None)      # default: the oracle's default rule = the shipped model; RP_NO_GJK=1: the library's RP_NO_GJK switch (= RP_CFG_OBB_EDGES) against rule 1015
env = VecPlayEnv(IDS[kind], n, seed=9, persistent_manifolds=True); env.reset()
fus = VecPlayEnv(IDS[kind], n, seed=9, persistent_manifolds=True); fus.set_fused(1); fus.reset()
KW = {} if RULE is None else dict(rule=RULE)
o64 = [OracleEnv(kind, seed=9, env_index=e, **KW) for e in range(n)]
o32 = [OracleEnv(kind, seed=9, env_index=e, f32=True, **KW) for e in range(n)]
for a, b in zip(o64, o32):
    a.reset(); b.reset(); s = a.get_state(); a.set_state(s); b.set_state(s)       # (set_state empties the caches: all four start without contact history)
rec = torch.tensor(np.stack([record_from_oracle(o) for o in o64]))
env.set_state(rec); fus.set_state(rec)
acts = actions(kind, steps, n, 5)
nm = N_MAIN[kind]; na = o64[0].n_arm
d_dev, d_32 = np.zeros(n), np.zeros(n)
for t in range(steps):
    at = torch.tensor(acts[t], dtype=torch.float32)
    env.step(at); fus.step(at)
    q = arm_q(env, kind)
    for e in range(n):
        a = acts[t, e].astype(np.float32).astype(np.float64)
        o64[e].step(a); o32[e].step(a)
        qo = o64[e].get_state()[:na]
        d_dev[e] = max(d_dev[e], np.abs(q[e] - qo)[:nm].max())
        d_32[e] = max(d_32[e], np.abs(o32[e].get_state()[:na] - qo)[:nm].max())
    if t in (0, 4, 19, 49, steps - 1):
        print('step %3d: device vs fp64 oracle arm max %.2e median %.2e | fp32 oracle max %.2e median %.2e | split == fused: %s'
              % (t, d_dev.max(), np.median(d_dev), d_32.max(), np.median(d_32), bool(torch.equal(env.get_state(), fus.get_state()))))
print('per env device', np.array2string(d_dev, precision=1))
print('per env fp32  ', np.array2string(d_32, precision=1))
