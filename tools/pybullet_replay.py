#!/usr/bin/env python3
"""Replay identical initial states and action sequences through a real PyBullet and through this library, and print
the joint-state divergence (the check BASELINE.json's north_star asks for and that cannot run in the build image:
PyBullet is not installed there).  NOT RUN so far — DESIGN.md §H lists the hypotheses it would confirm or refute.

    python tools/pybullet_replay.py --env UR5Reach-v0 --steps 200          # needs: pip install pybullet, the reference repo

It drives PyBullet with the reference's own semantics restated here (no reference source is imported): URDF + scene
from --reference-root, 300 Hz, 12 substeps per step, POSITION_CONTROL motors with the reference's forces, the shadow-arm
IK (4 chained calculateInverseKinematics calls) and the per-step joint clamps.
"""
import argparse
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--env', default='UR5Reach-v0', choices=['UR5Reach-v0'])
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--reference-root', default='/root/reference')
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    try:
        import pybullet as p
        from pybullet_utils import bullet_client
    except ImportError:
        sys.exit('pybullet is not importable here: this script has to run on a machine where it is installed')
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from roboticsplayroompybullet_amd import VecPlayEnv

    urdf = os.path.join(args.reference_root, 'roboticsPlayroomPybullet', 'envs', 'ur_e_description', 'ur5e2.urdf')
    base_pos, base_orn = [0.5, -0.1, 0.0], p.getQuaternionFromEuler([0, 0, np.pi / 2])
    c = bullet_client.BulletClient(connection_mode=p.DIRECT)
    c.setTimeStep(1.0 / 300)
    c.setGravity(0, 0, -9.8)
    c.setPhysicsEngineParameter(solverResidualThreshold=0)
    plane = c.createCollisionShape(p.GEOM_BOX, halfExtents=[2, 2, 0.0001])
    c.createMultiBody(0, plane, -1, [0, 0, -0.07])
    arm = c.loadURDF(urdf, base_pos, base_orn, useFixedBase=True, flags=p.URDF_ENABLE_CACHED_GRAPHICS_SHAPES)
    for j in range(c.getNumJoints(arm)):
        c.changeDynamics(arm, j, linearDamping=0, angularDamping=0)
    shadow = bullet_client.BulletClient(connection_mode=p.DIRECT)
    sarm = shadow.loadURDF(urdf, base_pos, base_orn, useFixedBase=True)

    env = VecPlayEnv(args.env, 1, seed=args.seed)
    env.reset()
    s = env.get_state()[0].cpu().numpy()
    dofs = [0, 1, 2, 3, 4, 5, 10, 12, 13, 15, 18, 20]
    for d, j in enumerate(dofs):
        c.resetJointState(arm, j, float(s[d]), float(s[12 + d]))
    rng = np.random.default_rng(args.seed)
    ul = np.array([-0.7, 2 * np.pi, -0.5, 2 * np.pi, 2 * np.pi, 2 * np.pi])
    inc = np.array([0.1, 0.1, 0.2, 0.2, 0.2, 0.2])
    worst = 0.0
    for t in range(args.steps):
        a = np.concatenate([rng.uniform(-0.18, 0.18, 2), rng.uniform(0.0, 0.2, 1), rng.uniform(-0.5, 0.5, 3), rng.uniform(-1, 1, 1)])
        # reference semantics on PyBullet
        cur = np.array([c.getJointState(arm, j)[0] for j in range(6)])
        for i in range(6):
            shadow.resetJointState(sarm, i, cur[i])
        orn = p.getQuaternionFromEuler(a[3:6])
        for _ in range(3):
            ang = shadow.calculateInverseKinematics(sarm, 7, a[:3], orn)[:6]
            for i in range(6):
                shadow.resetJointState(sarm, i, ang[i])
        ang = np.array(shadow.calculateInverseKinematics(sarm, 7, a[:3], orn)[:6])
        tgt = np.clip(np.clip(ang, -2 * np.pi, ul), cur - inc, cur + inc)
        c.setJointMotorControlArray(arm, list(range(6)), p.POSITION_CONTROL, targetPositions=tgt, forces=[240.0] * 6)
        amt = a[6] - 0.2
        c.setJointMotorControl2(arm, 18, p.POSITION_CONTROL, amt * 0.055, force=100)
        c.setJointMotorControl2(arm, 20, p.POSITION_CONTROL, c.getJointState(arm, 18)[0], force=1000)
        for j in (12, 15):
            c.setJointMotorControl2(arm, j, p.POSITION_CONTROL, amt * 0.5, force=100)
        for j in (10, 13):
            c.setJointMotorControl2(arm, j, p.POSITION_CONTROL, amt * 0.8, force=100)
        for _ in range(12):
            c.stepSimulation()
        env.step(torch.tensor(a[None], dtype=torch.float32))
        q_ref = np.array([c.getJointState(arm, j)[0] for j in dofs])
        q_hip = env.get_state()[0, :12].cpu().numpy()
        rel = np.abs(q_hip - q_ref) / np.maximum(1.0, np.abs(q_ref))
        worst = max(worst, rel.max())
        if t % 20 == 0:
            print('step %3d  max relative joint divergence so far %.3e  (this step per joint: %s)' % (t, worst, ' '.join('%.1e' % v for v in rel)))
    print('RESULT max relative joint-state divergence over %d steps: %.3e (north_star bound 1e-3)' % (args.steps, worst))


if __name__ == '__main__':
    main()
