"""GPU: the three pipelines of rp_step - split (k_prep2 / k_solve2), one kernel per step (k_step), one launch per step (k_chain) - stepped side by side with a
sync after every call: which one faults, at which step and in which envs and record words the others leave the first, and the row counts of those envs.
(tests/test_gpu_parity.py::test_split_pipeline_equals_fused_kernel_bitwise is the assertion; this is the tool for when it fails - round 6's stream-path experiment,
tools/experiments/r06_stream_path.patch, was debugged with it.)
    python tools/pipelines_bitwise.py [kinds=R,Q,P,U,V] [pipelines=1,0,2] [envs=33] [steps=6] [debug flags of the split pipeline] [groups]"""
import sys
import numpy as np
import torch
from roboticsplayroompybullet_amd import VecPlayEnv

IDS = {'U': 'UR5PlayAbsRPY1Obj-v0', 'P': 'pandaPick-v0', 'R': 'UR5Reach-v0', 'Q': 'pandaReach-v0', 'V': 'pandaPlayAbsRPY1Obj-v0'}
LO = np.array([-0.18, 0.0, 0.02, -0.5, -0.5, -0.5, -1.0]); HI = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])
kinds = sys.argv[1].split(',') if len(sys.argv) > 1 else ['R', 'Q', 'P', 'U', 'V']
fuseds = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 0, 2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 33
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
groups = int(sys.argv[6]) if len(sys.argv) > 6 else 0
for kind in kinds:
    states = {}
    for fused in fuseds:
        print(kind, 'fused', fused, flush=True)
        e = VecPlayEnv(IDS[kind], n, seed=5)
        e.set_fused(fused)
        if fused == 0 and flags:
            e.set_debug_flags(flags)
        if fused == 0 and groups:
            e.set_groups(groups)
        e.reset(); torch.cuda.synchronize()
        print('  reset ok', flush=True)
        rng = np.random.default_rng(8)
        hist = []
        rows = []
        for t in range(steps):
            a7 = LO + (HI - LO) * rng.random((n, 7))
            if kind not in ('U', 'V'):
                a7[..., 0:3] = np.array([-0.18, -0.18, 0.0]) + np.array([0.36, 0.36, 0.2]) * rng.random((n, 3))
            a = torch.tensor(a7, dtype=torch.float32)
            e.step(a); torch.cuda.synchronize()
            hist.append(e.get_state().cpu().numpy().copy())
            rows.append(e.debug_row_counts().numpy().copy())
        print('  steps ok', flush=True)
        states[fused] = hist
        states[('rows', fused)] = rows
    base = fuseds[0]
    for f in fuseds[1:]:
        for t in range(steps):
            d = states[f][t] != states[base][t]
            if d.any():
                envs = np.nonzero(d.any(axis=1))[0]
                cols = np.nonzero(d.any(axis=0))[0]
                print('  fused %d vs %d: first difference at step %d: %d envs (%s ...), record words %s; max gap %.3e' % (f, base, t, len(envs), envs[:8].tolist(), cols[:24].tolist(),
                      np.nanmax(np.abs(states[f][t].astype(np.float64) - states[base][t]))))
                for tt in range(max(0, t - 1), t + 1):
                    rc = states[('rows', f)][tt]
                    print('   step %d: envs with contacts %s; differing envs rows %s' % (tt, [(int(i), rc[i].tolist()) for i in np.nonzero(rc[:, 1])[0][:12]], [(int(i), rc[i].tolist()) for i in envs[:8]]))
                break
        else:
            print('  fused %d == fused %d bitwise over %d steps' % (f, base, steps))
