#!/usr/bin/env python3
"""How many hardware queues can rp_step afford when RCCL is there too?  (GPU box, one GPU.)

A process has four hardware queues by default; rp_step's env groups take one stream each (the caller's + groups - 1 of the library's).  On an 8-GPU
run RCCL's stream is one more client of those queues on every rank.  This stands in for it on one GPU: a side stream that, after every step, waits for
the step (event), copies the pack-sized message (0.5 MB at N = 4096: what a rank contributes to the all-gather) and records an event that the
step after next waits for - the dependency pattern of bench.py's asynchronous gather.  Reports ms per step with and without the side stream for
2, 3 and 4 env groups; bench.py's default for world > 1 follows from it.
    python tools/queue_budget.py [--steps 300]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from roboticsplayroompybullet_amd import VecPlayEnv  # noqa: E402


def run(groups, side, steps, n=4096):
    env = VecPlayEnv(bench.ENV_ID, n, seed=1234)
    env.set_groups(groups)
    env.reset()
    acts = bench.make_actions(n, steps + 30, env.device, 1234)
    s5 = torch.cuda.Stream()
    dst = torch.empty_like(env.pack)
    evs = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    for k in range(30):
        env.step(acts[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        if side and k >= 2:
            torch.cuda.current_stream().wait_event(done[k & 1])       # the gather of step k - 2 read the buffer this step writes: it must be through
        env.step(acts[30 + k])
        if side:
            evs[k & 1].record()
            with torch.cuda.stream(s5):
                s5.wait_event(evs[k & 1])
                dst.copy_(env.pack, non_blocking=True)
                done[k & 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.close()
    return 1e3 * dt / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    a = ap.parse_args()
    out = {}
    for groups in (2, 3, 4):
        for side in (False, True):
            ms = min(run(groups, side, a.steps) for _ in range(2))
            out['groups=%d%s' % (groups, ' + side stream' if side else '')] = {'ms_per_step': ms, 'env_steps_per_s': 4096 / ms * 1e3}
            print('groups %d, side stream %-5s: %.3f ms per step (%.2f M env-steps/s)' % (groups, side, ms, 4096 / ms * 1e-3), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
