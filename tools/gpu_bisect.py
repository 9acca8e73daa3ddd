#!/usr/bin/env python3
"""Replay the rollout on the oracle to step T, then compare HIP debug_substep vs oracle one substep at a time."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tests')); sys.path.insert(0, os.path.join(REPO, 'tools'))
from oracle import OracleEnv
from roboticsplayroompybullet_amd import VecPlayEnv
from gpu_debug import record_from_oracle, oracle_state_from_record
from test_gpu_parity import actions, IDS
np.set_printoptions(precision=8, suppress=True, linewidth=220)
kind, T, E = 'U', int(sys.argv[1]), int(sys.argv[2])
n, steps = 4, 200
o = OracleEnv(kind, seed=9, env_index=E); o.reset()
acts = actions(kind, steps, n, 5)
for t in range(T):
    o.step(acts[t, E].astype(np.float32).astype(np.float64))
env = VecPlayEnv(IDS[kind], 2, seed=9)
na = o.n_arm
for t in range(T, T + 3):
    a = acts[t, E].astype(np.float32).astype(np.float64)
    o.perform_action(np.clip(a, [-6]*6+[-1], [6]*6+[1]))
    for sub in range(12):
        rec = record_from_oracle(o)
        env.set_state(torch.tensor(np.tile(rec, (2, 1))))
        dbg = env.debug_substep(0).numpy()
        ncon = int(dbg[0])
        gc = dbg[16:16 + 9 * ncon].reshape(ncon, 9)
        oc = o.contacts()
        s0 = o.get_state()
        o.substep()
        s1 = o.get_state()
        nv = 27
        vg = dbg[480:480 + nv] + dbg[544:544 + nv]
        vo = np.concatenate([s1[na:2*na], s1[2*na+7:2*na+13], s1[2*na+13+7:2*na+26], s1[2*na+26+3:2*na+26+6]])
        dvel = np.abs(vg - vo)
        same = (ncon == len(oc)) and np.allclose(gc[:, :2], oc[:, :2]) and np.allclose(gc[:, 2:], oc[:, 2:], atol=1e-4)
        print('t', t, 'sub', sub, 'ncon', ncon, len(oc), 'contacts_same', same, 'max dvel', dvel.max(), 'argmax', dvel.argmax(), 'rows', o.num_rows(), 'nsmall', int(dbg[1]))
        if not same or dvel.max() > 3e-4:
            print(' gpu contacts\n', gc)
            print(' cpu contacts\n', oc)
            print(' diff contacts\n', gc - oc if gc.shape == oc.shape else None)
            print(' v gpu', vg); print(' v cpu', vo); print(' dv diff', vg - vo)
            sys.exit(0)
