#!/usr/bin/env python3
"""Round 4, experiment (i) of the verdict: the twelve substeps of a step in ONE launch (k_chain, rp_set_fused(h, 2)) against the split pipeline, same workload
(bench.py's distribution B; DIST=A: the literal U(action_space) rollout), N = 1024 / 4096 / 16384.  First checks that both pipelines give the same bits.
    python tools/chain_rate.py [N ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from roboticsplayroompybullet_amd import VecPlayEnv  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]
warm, steps = 200, 100


def actions(env, n, count):
    if os.environ.get('DIST') == 'A':
        return (2 * torch.rand((count, n, 7), generator=torch.Generator(device=env.device).manual_seed(4321), device=env.device) - 1) * env.action_high
    return bench.make_actions(n, count, env.device, 1234)


# same bits?
a, b = VecPlayEnv(bench.ENV_ID, 257, seed=5), VecPlayEnv(bench.ENV_ID, 257, seed=5)
b.set_fused(2)
a.reset(); b.reset()
acts = actions(a, 257, 40)
for t in range(40):
    oa = a.step(acts[t])[0]
    ob = b.step(acts[t])[0]
torch.cuda.synchronize()
same = torch.equal(a.get_state(), b.get_state()) and torch.equal(oa['obs_quat'], ob['obs_quat'])
print('k_chain == split pipeline after 40 steps of 257 envs, bit for bit: %s' % same, flush=True)
a.close(); b.close()
for n in sizes:
    res = {}
    for mode in (0, 2):
        env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
        env.set_fused(mode)
        env.reset()
        acts = actions(env, n, warm + steps)
        for k in range(warm):
            env.step(acts[k])
        reps = []
        for r in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(steps):
                env.step(acts[warm + k])
            torch.cuda.synchronize(); reps.append((time.perf_counter() - t0) / steps)
        res[mode] = sorted(reps)[1]
        env.close()
    print('N = %5d: split pipeline %.3f ms/step = %.3f M env-steps/s; k_chain %.3f ms/step = %.3f M env-steps/s' % (n, res[0] * 1e3, n / res[0] / 1e6, res[2] * 1e3, n / res[2] / 1e6), flush=True)
