#!/usr/bin/env python3
"""Run-to-run determinism of a library under arm-collision-heavy actions (distribution A): two rollouts from the same seed, states compared bit for bit; prints a checksum
to compare libraries with one another (RP_PLAYROOM_LIB)."""
import os, sys, hashlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from roboticsplayroompybullet_amd import VecPlayEnv
n, steps = 512, 80
outs = []
for rep in range(2):
    env = VecPlayEnv(bench.ENV_ID, n, seed=77); env.reset()
    g = torch.Generator(device=env.device).manual_seed(99)
    acts = (2 * torch.rand((steps, n, 7), generator=g, device=env.device) - 1) * env.action_high
    sums = []
    for k in range(steps):
        env.step(acts[k])
        sums.append(env.get_state().clone())
    outs.append(torch.stack(sums)); env.close()
same = (outs[0].view(torch.int32) == outs[1].view(torch.int32)).all(dim=2)      # [steps, n]
first_bad = [int(torch.nonzero(~same[:, e])[0]) for e in range(n) if not bool(same[:, e].all())]
print('library %s: %d of %d envs identical over %d steps across two runs; first differing steps %s; checksum of run 0: %s' % (
    os.environ.get('RP_PLAYROOM_LIB', 'default'), int(same.all(dim=0).sum()), n, steps, sorted(first_bad)[:8], hashlib.md5(outs[0].cpu().numpy().tobytes()).hexdigest()[:12]) + '; of the 128-float records alone: %s' % hashlib.md5(outs[0][:, :, :128].contiguous().cpu().numpy().tobytes()).hexdigest()[:12])
