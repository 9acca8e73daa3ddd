#!/usr/bin/env python3
"""Lock-step comparison (GPU box) like gpu_bisect2.py, but it does not stop at the first difference: before every substep the device gets the fp32
oracle's state, runs ONE fused substep (rp_debug_substep), and the resulting velocities are compared.  Prints the distribution of the per-substep
velocity differences and every substep whose difference exceeds 1e-4 with both row counts (a limit row present on one side only shows there).
Both sides take every substep WITHOUT contact history (the oracle's cache is emptied too): this compares one substep's arithmetic, not the cache.
    python tools/gpu_bisect3.py [id] [steps]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tools'))
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
from gpu_debug import record_from_oracle
np.set_printoptions(precision=7, suppress=True, linewidth=220)
LO = np.array([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0]); HI = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])
kind = sys.argv[1] if len(sys.argv) > 1 else 'U'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
F32 = (sys.argv[3] != 'f64') if len(sys.argv) > 3 else True
VERBOSE = len(sys.argv) > 4
IDS = {'U': 'UR5PlayAbsRPY1Obj-v0', 'R': 'UR5Reach-v0', 'P': 'pandaPick-v0'}
n = 6
rng = np.random.default_rng(3)
a = LO + (HI - LO) * rng.random((steps, n, 7))
env = VecPlayEnv(IDS[kind], 2, seed=8)
hi = np.array([6] * 6 + [1.0])
allmax = []
armmax = []
NM = 7 if kind == 'P' else 6
GRASP = os.environ.get('RP_BISECT_GRASP') == '1'      # drive the gripper onto the block / the table instead of random actions (contacts of the gripper links: torsional rows)
for e in range(n):
    o = OracleEnv(kind, seed=8, env_index=e, f32=F32)
    o.reset()
    na = o.n_arm
    for t in range(steps):
        act = np.clip(a[t, e], -hi, hi)
        if GRASP:
            blk = o.calc_state()['achieved_goal'][:3]
            act = np.array([blk[0], blk[1], (0.0 if kind == 'U' else blk[2] - 0.02) + (0.0 if t % 60 < 40 else 0.1), 0.0, 0.0, 0.3 * np.sin(0.3 * t), 1.0 if (t // 20) % 2 else -1.0])
        o.perform_action(act)
        for sub in range(12):
            rec = record_from_oracle(o)
            env.set_state(torch.tensor(np.tile(rec, (2, 1))))      # records only: the device takes this substep without contact history ...
            dbg = env.debug_substep(0).numpy()
            o.set_state(o.get_state())                               # ... and so does the oracle (set_state empties its contact cache; the motors stay)
            o.substep()
            s1 = o.get_state()
            vg = (dbg[480:480 + 27] + dbg[544:544 + 27])[:na]
            vo = s1[na:2 * na]
            d = np.abs(vg - vo)
            allmax.append(d.max()); armmax.append(d[:NM].max())
            if d[:NM].max() > 1e-4 and VERBOSE:
                print('env', e, 't', t, 'sub', sub, 'max dvel %.2e at dof %d' % (d.max(), d.argmax()), 'device small rows', int(dbg[1]), 'oracle rows', o.num_rows(), 'device ncon', int(dbg[0]))
                print('   v gpu', vg); print('   v cpu', vo)
allmax = np.array(allmax); armmax = np.array(armmax)
print('arm joints only: max %.2e median %.2e p99 %.2e; > 1e-4: %d' % (armmax.max(), np.median(armmax), np.percentile(armmax, 99), int((armmax > 1e-4).sum())))
print('substeps', len(allmax), 'max %.2e median %.2e p99 %.2e; > 1e-4: %d' % (allmax.max(), np.median(allmax), np.percentile(allmax, 99), int((allmax > 1e-4).sum())))
