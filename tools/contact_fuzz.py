#!/usr/bin/env python3
"""GPU: the narrowphase of device and oracle on RANDOM scenes, history-free on both sides (records only): the block dropped at a random collider's surface - static furniture,
the robot's base, arm links, the drawer, the door, the other movable bodies - in a random orientation, the arm in a random posture, drawer / door / button / dial anywhere in
their ranges.  Contact lists must agree pair by pair, points and distances to 5e-5.  Round 6: a block thrown at the robot's base found the oracle colliding a box where the
device collided a hull (now tests/test_gpu_gjk_contacts.py); this tool looks for the next such pair class.
    python tools/contact_fuzz.py [kinds=U,P,V] [poses=1500]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle')); sys.path.insert(0, os.path.join(REPO, 'tools'))
from gpu_debug import record_from_oracle  # noqa: E402
from oracle import OracleEnv  # noqa: E402
from roboticsplayroompybullet_amd import VecPlayEnv  # noqa: E402

IDS = {'U': 'UR5PlayAbsRPY1Obj-v0', 'P': 'pandaPick-v0', 'V': 'pandaPlayAbsRPY1Obj-v0'}


def main():
    kinds = sys.argv[1].split(',') if len(sys.argv) > 1 else ['U', 'P', 'V']
    poses = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    for kind in kinds:
        env = VecPlayEnv(IDS[kind], 2, seed=7)
        o = OracleEnv(kind, seed=7, env_index=0, f32=True)
        o.reset()
        na = o.n_arm
        s0 = o.get_state()
        nf = (len(s0) - 2 * na) // 13
        nj = (len(s0) - 2 * na - 13 * nf) // 2
        arm = o.arm_table()
        rng = np.random.default_rng(77)
        classes, bad, shown, touched, ndeep, deepstat = {}, {}, 0, 0, 0, {}
        worst = 0.0
        for t in range(poses):
            s = s0.copy()
            if t % 3:                                        # the arm somewhere near its rest posture (anywhere in its ranges it lies in the furniture most of the time)
                for i in range(na):
                    lo, hi = arm[i][1], arm[i][2]
                    v = s0[i] + rng.uniform(-0.35, 0.35)
                    s[i] = min(max(v, lo), hi) if lo < hi else v
            for k in range(nj):
                s[2 * na + 13 * nf + k] = rng.uniform(-1.5, 0.3) if k != 1 else rng.uniform(0.0, 0.03)
            o.set_state(s)
            cols = o.collider_list()
            for k in range(nf):                              # every free body at a random collider's surface (the drawer keeps its rails: y only)
                base = 2 * na + 13 * k
                if kind in ('U', 'V') and k == 1:
                    s[base + 1] = s0[base + 1] + rng.uniform(-0.06, 0.075)
                    continue
                c = cols[rng.integers(len(cols))]
                q = rng.normal(size=4) if t % 4 else np.array([0.0, 0.0, 0.0, 1.0])
                q /= np.linalg.norm(q)
                x, y, z, w = q
                Rb = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                               [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                # beside one of the collider's six faces (or over an edge: the tangential offset may leave the face), the gap between -4 and +4 mm along that face's normal
                ax, sg = rng.integers(3), rng.choice([-1.0, 1.0])
                nrm = c['R'][:, ax] * sg
                hb = np.array([0.025, 0.025, 0.025])
                ext = float(np.abs(Rb.T @ nrm) @ hb)
                loc = (2 * rng.random(3) - 1) * (np.minimum(c['he'], 0.3) + 0.02)
                loc[ax] = sg * (c['he'][ax] + ext + rng.uniform(-0.004, 0.004))
                s[base:base + 3] = c['p'] + c['R'] @ loc
                s[base + 3:base + 7] = q
                s[base + 7:base + 13] = 0.0
            s[na:2 * na] = 0.0
            o.set_state(s)
            rec = record_from_oracle(o)
            env.set_state(torch.tensor(np.tile(rec, (2, 1))))
            dbg = env.debug_substep(0).numpy()
            o.set_state(s)
            oc = o.contacts()
            ncon = int(dbg[0])
            gc = dbg[16:16 + 9 * ncon].reshape(ncon, 9)
            cols = o.collider_list()
            def cls(a, b):
                f = lambda c: 'static' if cols[c]['body'] == 0 else ('arm' if cols[c]['body'] <= na else 'movable')
                return f(int(a)) + (' (robot)' if cols[int(a)]['body'] == 0 and o.hull_vertices(int(a)) is not None else '') + ' / ' + f(int(b)) + (' (robot)' if cols[int(b)]['body'] == 0 and o.hull_vertices(int(b)) is not None else '')
            deep = (len(oc) and float(oc[:, 8].min()) < -0.006) or (ncon and float(gc[:, 8].min()) < -0.006)      # a body dropped INTO another: overlapping cores, the polytope's own rounding - not what rollouts pass through
            if deep:
                ndeep += 1
                if os.environ.get('FUZZ_DEEP'):              # (study) the deep scenes too, by depth: 6 - 10 mm, 10 - 20, beyond
                    dmin = -min(float(oc[:, 8].min()) if len(oc) else 0.0, float(gc[:, 8].min()) if ncon else 0.0)
                    bk = '6-10 mm' if dmin < 0.010 else ('10-20 mm' if dmin < 0.020 else '> 20 mm')
                    same = ncon == len(oc) and np.array_equal(gc[:, :2], oc[:, :2])
                    err = max(float(np.abs(gc[:, 2:5] - oc[:, 2:5]).max()), float(np.abs(gc[:, 8] - oc[:, 8]).max()), 0.1 * float(np.abs(gc[:, 5:8] - oc[:, 5:8]).max())) if same and ncon else 1.0
                    d = deepstat.setdefault(bk, [0, 0, 0])
                    d[0] += 1; d[1] += int(not same); d[2] += int(same and err > 5e-5)
                    if same and err > 5e-5 and dmin < 0.02 and shown < 6:
                        shown += 1
                        i = int(np.argmax(np.abs(gc[:, 8] - oc[:, 8]) + np.abs(gc[:, 5:8] - oc[:, 5:8]).max(axis=1)))
                        print('%s pose %d (deepest %.1f mm): same pairs, contact %d differs: pair (%d, %d) %s; device %s oracle %s' % (kind, t, 1e3 * dmin, i, gc[i, 0], gc[i, 1], cls(gc[i, 0], gc[i, 1]), np.round(gc[i, 2:], 4).tolist(), np.round(oc[i, 2:], 4).tolist()))
                continue
            touched += int(len(oc) > 0)
            for r in oc:
                classes[cls(r[0], r[1])] = classes.get(cls(r[0], r[1]), 0) + 1
            ok = ncon == len(oc) and np.array_equal(gc[:, :2], oc[:, :2])
            if ok and ncon:
                err = max(float(np.abs(gc[:, 2:5] - oc[:, 2:5]).max()), float(np.abs(gc[:, 8] - oc[:, 8]).max()), 0.1 * float(np.abs(gc[:, 5:8] - oc[:, 5:8]).max()))
                ok = err <= 5e-5
                if ok:
                    worst = max(worst, err)
            if not ok:
                # which pairs differ
                dp = {(int(r[0]), int(r[1])) for r in gc}; op = {(int(r[0]), int(r[1])) for r in oc}
                keys = (dp ^ op) or (dp | op)
                for a, b in keys:
                    bad[cls(a, b)] = bad.get(cls(a, b), 0) + 1
                if shown < 4:
                    shown += 1
                    print('%s pose %d: device %d contacts, oracle %d; pairs only on the device %s, only in the oracle %s' % (kind, t, ncon, len(oc), sorted(dp - op), sorted(op - dp)))
                    if dp == op:
                        print(np.round(gc, 5)); print(np.round(oc, 5))
        print('%s: %d random scenes (%d more with a penetration beyond 6 mm: skipped), %d with contacts; contacts by pair class %s; scenes with another list on the device by pair class: %s; worst point / distance error of the rest %.1e'
              % (kind, poses - ndeep, ndeep, touched, dict(sorted(classes.items())), dict(sorted(bad.items())) or 'none', worst), flush=True)
        if deepstat:
            print('   deep scenes by their deepest contact: [scenes, with another pair list, same pairs but a point / normal / distance beyond 5e-5] %s' % dict(sorted(deepstat.items())), flush=True)


if __name__ == '__main__':
    main()
