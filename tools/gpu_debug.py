#!/usr/bin/env python3
"""Bring-up aid (GPU box): compare HIP intermediates and short rollouts with the CPU oracle, print the diffs."""
import os
import struct
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'oracle'))
from oracle import OracleEnv  # noqa: E402
from roboticsplayroompybullet_amd import VecPlayEnv  # noqa: E402

np.set_printoptions(precision=5, suppress=True, linewidth=200)
IDS = {'U': 'UR5PlayAbsRPY1Obj-v0', 'R': 'UR5Reach-v0', 'P': 'pandaPick-v0'}


def f2i(x):
    return struct.unpack('<f', struct.pack('<i', int(x)))[0]


def record_from_oracle(o):
    """oracle state -> 128-float HIP state record"""
    s = o.get_state()
    n, nf = o.n_arm, (2 if o.kind in (0, 4) else (1 if o.kind == 2 else 0))
    nj = 3 if o.kind in (0, 4) else 0
    r = np.zeros(128, dtype=np.float32)
    r[0:n] = s[0:n]
    r[12:12 + n] = s[n:2 * n]
    p = 2 * n
    for k in range(2):
        if k < nf:
            r[24 + 13 * k:24 + 13 * k + 13] = s[p:p + 13]
            p += 13
        else:
            r[24 + 13 * k + 6] = 1.0
    r[50:50 + nj] = s[p:p + nj]
    r[53:53 + nj] = s[p + nj:p + 2 * nj]
    mode, tgt, mx = o.get_motor()
    r[56:56 + n] = mode
    r[68:68 + n] = tgt
    r[80:92] = 1.0
    r[80:80 + n] = mx
    obs = o.calc_state()
    o.clear_quat_memory()
    g = obs['desired_goal']
    r[92:92 + len(g)] = g
    r[117] = f2i(len(g))
    return r


def oracle_state_from_record(o, r):
    n, nf = o.n_arm, (2 if o.kind in (0, 4) else (1 if o.kind == 2 else 0))
    nj = 3 if o.kind in (0, 4) else 0
    s = list(r[0:n]) + list(r[12:12 + n])
    for k in range(nf):
        s += list(r[24 + 13 * k:24 + 13 * k + 13])
    s += list(r[50:50 + nj]) + list(r[53:53 + nj])
    return np.array(s, dtype=np.float64)


def main(kind='U'):
    print('=== kind', kind)
    o = OracleEnv(kind, seed=7, env_index=0)
    o.reset()
    for _ in range(3):
        o.step([0.05, 0.1, 0.02, 0, 0, 0, 0.5])
    rec = record_from_oracle(o)
    env = VecPlayEnv(IDS[kind], 4, seed=7)
    env.set_state(torch.tensor(np.tile(rec, (4, 1))))
    torch.cuda.synchronize()
    dbg = env.debug_substep(0).numpy()
    ncon = int(dbg[0])
    print('ncon gpu', ncon, 'nsmall', int(dbg[1]))
    oc = o.contacts()
    print('ncon oracle', len(oc))
    gc = dbg[16:16 + 9 * ncon].reshape(ncon, 9)
    for i in range(max(ncon, len(oc))):
        print(' gpu', gc[i] if i < ncon else None)
        print(' cpu', oc[i] if i < len(oc) else None)
    n = o.n_arm
    Mg = dbg[320:320 + 144].reshape(12, 12)[:n, :n]
    Mo = o.mass_matrix_inv()
    print('Minv max abs diff', np.abs(Mg - Mo).max(), 'rel', np.abs(Mg - Mo).max() / np.abs(Mo).max())
    print('Minv diag gpu', np.diag(Mg))
    print('Minv diag cpu', np.diag(Mo))
    qdd = o.forward_dynamics()
    s = o.get_state()
    vstar_o = s[n:2 * n] + qdd / 300.0
    print('vstar arm gpu', dbg[480:480 + n])
    print('vstar arm cpu', vstar_o)
    print('vstar rest gpu', dbg[480 + n:480 + 28])
    print('dv gpu', dbg[544:544 + 28])
    # rollout comparison
    o2 = OracleEnv(kind, seed=7, env_index=0)
    o2.set_state(oracle_state_from_record(o2, rec))
    env.set_state(torch.tensor(np.tile(rec, (4, 1))))
    rng = np.random.default_rng(0)
    # seed motors/goal in the oracle by replaying the same first action on both
    maxd = 0
    for t in range(30):
        a = np.array([rng.uniform(-0.15, 0.15), rng.uniform(0.0, 0.3), rng.uniform(0.0, 0.25), rng.uniform(-0.3, 0.3),
                      rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-1, 1)])
        if kind == 'R':
            a[:3] = [rng.uniform(-0.18, 0.18), rng.uniform(-0.18, 0.18), rng.uniform(0.0, 0.2)]
        og, rg, dg, ig = env.step(torch.tensor(np.tile(a, (4, 1)), dtype=torch.float32))
        oo, ro, do, io = o2.step(a)
        torch.cuda.synchronize()
        g = env.get_state()[0].cpu().numpy()
        so = o2.get_state()
        sg = oracle_state_from_record(o2, g)
        d = np.abs(sg - so)
        maxd = max(maxd, d[:n].max())
        if t % 5 == 0 or t == 29:
            print('t', t, 'max|dq|', d[:n].max(), 'max|dqd|', d[n:2 * n].max(), 'rest', d[2 * n:].max() if len(d) > 2 * n else 0,
                  'status', int(ig['status'][0]), 'tp diff', np.abs(ig['target_poses'][0].cpu().numpy() - io['target_poses']).max())
    print('q gpu', sg[:n])
    print('q cpu', so[:n])
    print('obs gpu', og['obs_quat'][0].cpu().numpy())
    print('obs cpu', oo['obs_quat'])
    print('MAX joint divergence over 30 steps', maxd)
    # reset parity (same counter RNG)
    env2 = VecPlayEnv(IDS[kind], 3, seed=11)
    ob = env2.reset()
    torch.cuda.synchronize()
    for e in range(3):
        oe = OracleEnv(kind, seed=11, env_index=e, f32=True)
        oo = oe.reset()
        print('reset env', e, 'obs diff', np.abs(ob['obs_quat'][e].cpu().numpy() - oo['obs_quat']).max(),
              'goal diff', np.abs(ob['desired_goal'][e].cpu().numpy() - oo['desired_goal']).max())
    print('reset obs gpu', ob['obs_quat'][0].cpu().numpy())
    print('reset obs cpu', oo['obs_quat'])


if __name__ == '__main__':
    for k in (sys.argv[1:] or ['R', 'U']):
        main(k)
