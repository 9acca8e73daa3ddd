#!/usr/bin/env python3
"""GPU box: the scenario of tests/test_gpu_parity.py::test_play_family_action_types for one id, device vs the fp32 oracle step by step,
then - for the first step whose joints differ by more than 2e-4 - substep by substep in lock step (device restarted from the oracle's state)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tools')); sys.path.insert(0, os.path.join(REPO, 'tests'))
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
from gpu_debug import record_from_oracle
from test_gpu_parity import family_actions
np.set_printoptions(precision=6, suppress=True, linewidth=220)
gid = sys.argv[1] if len(sys.argv) > 1 else 'UR5Play1Obj-v0'
env_i = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n, steps = 5, 12
dev = VecPlayEnv(gid, n, seed=13)
dev.reset()
o = OracleEnv(gid, seed=13, env_index=env_i, f32=True)
o.reset()
acts = family_actions(gid, steps, n, 3)
na = o.n_arm
for t in range(steps):
    rec0 = record_from_oracle(o)
    s_before = o.get_state().copy()
    dev.step(torch.tensor(acts[t], dtype=torch.float32))
    o.step(acts[t, env_i])
    qd = dev.get_state()[env_i, :na].cpu().numpy()
    qo = o.get_state()[:na]
    d = np.abs(qd - qo)
    print('step', t, 'max |dq| %.2e at dof %d' % (d.max(), d.argmax()), 'rows', o.num_rows())
    if d[:6].max() > 2e-4:
        print('  q dev', qd); print('  q cpu', qo)
        # lock step through this env step: a second oracle replays it substep by substep, the device follows from the oracle's state
        o2 = OracleEnv(gid, seed=13, env_index=env_i, f32=True)
        o2.reset(); o2.set_state(s_before)
        import ctypes as C
        g = np.ascontiguousarray(o.calc_state()['desired_goal'], dtype=np.float64); o.clear_quat_memory()
        hi = np.array([1.0] * 8)
        o2.perform_action(acts[t, env_i])
        d2 = VecPlayEnv(gid, 2, seed=13)
        for sub in range(12):
            rec = record_from_oracle(o2)
            d2.set_state(torch.tensor(np.tile(rec, (2, 1))))
            dbg = d2.debug_substep(0).numpy()
            oc = o2.contacts()
            o2.substep()
            s1 = o2.get_state()
            vg = (dbg[480:480 + 27] + dbg[544:544 + 27])[:na]
            vo = s1[na:2 * na]
            print('   sub', sub, 'max dvel %.2e at dof %d' % (np.abs(vg - vo).max(), np.abs(vg - vo).argmax()), 'device rows small', int(dbg[1]), 'ncon', int(dbg[0]), 'oracle rows', o2.num_rows(), 'ncon', len(oc))
            if np.abs(vg - vo).max() > 1e-3:
                print('     v dev', vg); print('     v cpu', vo)
        break
