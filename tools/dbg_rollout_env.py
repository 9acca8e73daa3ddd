#!/usr/bin/env python3
"""GPU box: replay tests/test_gpu_parity.py::test_rollout_200_steps_vs_fp64_oracle[kind] and print, for the given envs, the step-by-step distance of
the device and of every fp32 follower from the fp64 oracle (arm joints), plus the oracle's contacts at the step where the device first leaves it.
    python tools/dbg_rollout_env.py U 2,32"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tools')); sys.path.insert(0, os.path.join(REPO, 'tests'))
from tolerances import Followers, N_MAIN
from test_gpu_parity import actions, IDS, arm_q
from gpu_debug import record_from_oracle
from roboticsplayroompybullet_amd import VecPlayEnv
np.set_printoptions(precision=5, suppress=True, linewidth=200)
kind = sys.argv[1]; envs = [int(x) for x in sys.argv[2].split(',')]
n, steps = (64 if kind == 'U' else 8), 200
env = VecPlayEnv(IDS[kind], n, seed=9); env.reset()
fol = [Followers(kind, 9, e) for e in range(n)]
for f in fol:
    f.o64.reset(); f.start_from(f.o64)
env.set_state(torch.tensor(np.stack([record_from_oracle(f.o64) for f in fol])))
acts = actions(kind, steps, n, 5)
nm = N_MAIN[kind]; na = fol[0].o64.n_arm
first = {e: None for e in envs}
for t in range(steps):
    pre = {e: fol[e].o64.get_state().copy() for e in envs}
    obs, r, done, info = env.step(torch.tensor(acts[t], dtype=torch.float32))
    q = arm_q(env, kind)
    for e in envs:
        f = fol[e]
        f.step(acts[t, e].astype(np.float32).astype(np.float64))
        qo = f.o64.get_state()[:na]
        gd = np.abs(q[e] - qo)[:nm].max()
        gf = [np.abs(o.get_state()[:na] - qo)[:nm].max() for o in [f.o32] + f.more]
        if gd > 3e-5 and first[e] is None:
            first[e] = t
            cols = f.o64.collider_list()
            print('env %d: device leaves the fp64 oracle at step %d (gap %.2e, followers %s), status %d' % (e, t, gd, ['%.1e' % x for x in gf], int(info['status'][e])))
            scratch = type(f.o64)(kind, seed=9, env_index=e); scratch.reset(); scratch.set_state(pre[e])
            for c in scratch.contacts():
                a, b = int(c[0]), int(c[1])
                print('   contact c%d(body %d link %d) c%d(body %d link %d) d %+.5f n %s' % (a, cols[a]['body'], cols[a]['link'], b, cols[b]['body'], cols[b]['link'], c[8], c[5:8]))
            # the same step from the oracle's pre-step state on the split pipeline and on the fused kernel
            for fused in (0, 1):
                e3 = VecPlayEnv(IDS[kind], 2, seed=9); e3.set_fused(fused)
                o3 = type(f.o64)(kind, seed=9, env_index=e, f32=True); o3.reset(); o3.set_state(pre[e])
                e3.set_state(torch.tensor(np.tile(record_from_oracle(o3), (2, 1))))
                e3.step(torch.tensor(np.tile(acts[t, e], (2, 1)), dtype=torch.float32))
                o3.step(acts[t, e].astype(np.float32).astype(np.float64))
                print('   one step from the oracle state, %s: arm gap %.2e, all %.2e' % ('fused' if fused else 'split', np.abs(arm_q(e3, kind)[0] - o3.get_state()[:na])[:nm].max(), np.abs(arm_q(e3, kind)[0] - o3.get_state()[:na]).max()))
            # lock-step through the 12 substeps of this step: the device's fused substep against the fp32 oracle from the oracle's state
            o = type(f.o64)(kind, seed=9, env_index=e, f32=True); o.reset(); o.set_state(pre[e])
            o.perform_action(acts[t, e].astype(np.float32).astype(np.float64))
            e2 = VecPlayEnv(IDS[kind], 2, seed=9)
            for sub in range(12):
                rec = record_from_oracle(o)
                e2.set_state(torch.tensor(np.tile(rec, (2, 1))))      # (records only: no contact history on the device ...)
                dbg = e2.debug_substep(0).numpy()
                o.set_state(o.get_state())                               # (... nor in the oracle)
                con = o.contacts()
                o.set_state(o.get_state())
                o.substep()
                s1 = o.get_state()
                vg = (dbg[480:480 + 27] + dbg[544:544 + 27])[:na]
                d = np.abs(vg - s1[na:2 * na])
                print('   sub %2d: fused substep vs fp32 oracle |dqd| arm %.2e gripper %.2e; device ncon %d oracle ncon %d rows %d%s' % (sub, d[:nm].max(), d[nm:].max(), int(dbg[0]), len(con), o.num_rows(),
                      ''.join(' [c%d-c%d d%+.4f]' % (int(c[0]), int(c[1]), c[8]) for c in con if cols[int(c[0])]['link'] >= 0 or cols[int(c[1])]['link'] >= 0)))
        if t % 20 == 19 or (first[e] is not None and t - first[e] < 4):
            print('env %d t %3d device %.2e followers %s' % (e, t, gd, ['%.1e' % x for x in gf]))
