#!/usr/bin/env python3
"""Pin the physics against a real PyBullet - the check BASELINE.json's north_star asks for and that cannot run in the build image
(PyBullet is not installed there, nor on the GPU boxes).

Two halves, joined by a JSON fixture:

  1. ON ANY MACHINE WITH `pip install pybullet gym==0.21` AND THE REFERENCE REPO:
         python tools/pybullet_replay.py --all --reference-root /path/to/RoboticsPlayroomPybullet        (five ids x {random, grasp})
     or one at a time:
         python tools/pybullet_replay.py --dump tests/golden/pybullet_UR5PlayAbsRPY1Obj-v0.json \\
                --env UR5PlayAbsRPY1Obj-v0 --reference-root /path/to/RoboticsPlayroomPybullet [--steps 200 --seed 0 --scenario random|grasp]
     runs the REFERENCE'S OWN env class on PyBullet (nothing is restated), and records
       * known answers that pin single hypotheses of DESIGN.md section H: getDynamicsInfo of every arm link (mass, local inertia
         diagonal, inertial frame: H2 / H3), getJointInfo (limits, damping, parents: H1), calculateMassMatrix at the rest pose,
         calculateInverseKinematics at a few (joints, target) pairs incl. the iteration cap and residual defaults (H9),
         getPhysicsEngineParameters, the contact points of the settled scene (positions, normals, distances: margins, H7)
       * a trajectory: the full initial state after reset(), the seeded action sequence, and after every step the 12 (UR5) / 9 (Panda)
         joint positions and velocities, the block pose, drawer y, door / button / dial, obs_quat, target_poses
  2. IN THIS REPO (CPU, and on the GPU box when marked): tests/test_pybullet_golden.py consumes every tests/golden/pybullet_*.json it
     finds - it starts the oracle (and the HIP library) from the recorded initial state, replays the recorded actions and holds the
     joint trajectory to north_star's 1e-3 and the known answers to their own tolerances.  With no such file present the tests say so
     and skip: PARITY STAYS UNPINNED until somebody runs half 1.

`--from-oracle` writes the same fixture from this repo's CPU oracle instead of PyBullet (format check of the pipeline; such a file is
marked `"source": "oracle"` and is never a pin).
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMAT = 1
CLASSES = {'UR5PlayAbsRPY1Obj-v0': 'UR5PlayAbsRPY1Obj', 'UR5Reach-v0': 'UR5Reach', 'pandaPick-v0': 'pandaPick', 'pandaReach-v0': 'pandaReach',
           'pandaPlayAbsRPY1Obj-v0': 'pandaPlayAbsRPY1Obj', 'pandaPush-v0': 'pandaPush'}
# Bullet joint indices of the movable arm joints, in this repo's dof order
ALL_IDS = ['UR5PlayAbsRPY1Obj-v0', 'UR5Reach-v0', 'pandaPick-v0', 'pandaReach-v0', 'pandaPlayAbsRPY1Obj-v0']     # BASELINE.json's ids and their Panda / reach counterparts
ARM_JOINTS = {'UR5': [0, 1, 2, 3, 4, 5, 10, 12, 13, 15, 18, 20], 'Panda': [0, 1, 2, 3, 4, 5, 6, 9, 10]}
LO = np.array([-0.18, 0.0, 0.05, -0.5, -0.5, -0.5, -1.0])
HI = np.array([0.18, 0.3, 0.3, 0.5, 0.5, 0.5, 1.0])


def make_actions(env_id, steps, seed, scenario):
    rng = np.random.default_rng(seed)
    a = LO + (HI - LO) * rng.random((steps, 7))
    if not env_id.startswith(('UR5Play', 'pandaPlay')):
        a[:, 0:3] = np.array([-0.18, -0.18, 0.0]) + np.array([0.36, 0.36, 0.2]) * rng.random((steps, 3))
    return a


def grasp_action(env_id, block_pos, t):
    a = np.zeros(7)
    a[0:3] = block_pos
    if env_id.startswith('UR5'):
        a[2] = 0.02 if t < 60 else 0.15
    else:
        a[2] = block_pos[2] + (0.0 if t < 60 else 0.15)
    a[6] = -1.0 if t < 30 else 1.0
    return a


# ------------------------------------------------------------------ half 1: the reference on PyBullet
def dump_from_pybullet(args):
    try:
        import pybullet as p  # noqa: F401
    except ImportError:
        sys.exit('pybullet is not importable here: run this half on a machine where `pip install pybullet gym==0.21` works')
    sys.path.insert(0, args.reference_root)
    sys.path.insert(0, os.path.join(args.reference_root, 'roboticsPlayroomPybullet', 'envs'))
    from roboticsPlayroomPybullet.envs import envList                       # the reference itself
    np.random.seed(args.seed)
    env = getattr(envList, CLASSES[args.env])()
    obs = env.reset()
    inst, c = env.instance, env.instance.bullet_client
    arm = inst.arm
    arm_type = inst.arm_type
    joints = ARM_JOINTS[arm_type]
    out = {'format': FORMAT, 'source': getattr(args, 'source_label', 'pybullet'), 'env': args.env, 'seed': args.seed, 'scenario': args.scenario, 'arm_type': arm_type,
           'pybullet_api_version': c.getAPIVersion(), 'physics_engine_parameters': {k: v for k, v in c.getPhysicsEngineParameters().items()}}
    # known answers
    n_joints = c.getNumJoints(arm)
    out['joint_info'] = [{'index': j, 'name': c.getJointInfo(arm, j)[1].decode(), 'type': c.getJointInfo(arm, j)[2], 'damping': c.getJointInfo(arm, j)[6],
                          'lower': c.getJointInfo(arm, j)[8], 'upper': c.getJointInfo(arm, j)[9], 'parent': c.getJointInfo(arm, j)[16],
                          'axis': list(c.getJointInfo(arm, j)[13])} for j in range(n_joints)]
    out['dynamics_info'] = [{'link': j, 'mass': c.getDynamicsInfo(arm, j)[0], 'lateral_friction': c.getDynamicsInfo(arm, j)[1],
                             'local_inertia_diagonal': list(c.getDynamicsInfo(arm, j)[2]), 'local_inertial_pos': list(c.getDynamicsInfo(arm, j)[3]),
                             'local_inertial_orn': list(c.getDynamicsInfo(arm, j)[4]), 'contact_damping': c.getDynamicsInfo(arm, j)[8],
                             'contact_stiffness': c.getDynamicsInfo(arm, j)[9], 'collision_margin': c.getDynamicsInfo(arm, j)[11] if len(c.getDynamicsInfo(arm, j)) > 11 else None}
                            for j in range(-1, n_joints)]
    q_now = [c.getJointState(arm, j)[0] for j in joints]
    out['mass_matrix'] = {'q': q_now, 'M': np.array(c.calculateMassMatrix(arm, q_now)).tolist()}
    probes = []
    ee = inst.endEffectorIndex
    for k in range(6):
        tgt = [0.05 * (k - 2), 0.15 + 0.02 * k, 0.12 + 0.03 * k]
        orn = c.getQuaternionFromEuler([0.1 * k, -0.05 * k, 0.2])
        probes.append({'q': q_now, 'target_pos': tgt, 'target_orn': list(orn), 'result_default': list(c.calculateInverseKinematics(arm, ee, tgt, orn)),
                       'result_1_iteration': list(c.calculateInverseKinematics(arm, ee, tgt, orn, maxNumIterations=1)),
                       'result_200_iterations': list(c.calculateInverseKinematics(arm, ee, tgt, orn, maxNumIterations=200, residualThreshold=1e-9))})
    out['ik_probes'] = probes
    out['link_states_at_reset'] = [{'link': j, 'com_pos': list(c.getLinkState(arm, j)[0]), 'com_orn': list(c.getLinkState(arm, j)[1]),
                                    'frame_pos': list(c.getLinkState(arm, j)[4]), 'frame_orn': list(c.getLinkState(arm, j)[5])} for j in range(n_joints)]
    out['contact_points_at_reset'] = [{'bodyA': cp[1], 'bodyB': cp[2], 'linkA': cp[3], 'linkB': cp[4], 'posA': list(cp[5]), 'posB': list(cp[6]),
                                       'normalOnB': list(cp[7]), 'distance': cp[8], 'normal_force': cp[9]} for cp in c.getContactPoints()]

    def snapshot():
        s = {'q': [c.getJointState(arm, j)[0] for j in joints], 'qd': [c.getJointState(arm, j)[1] for j in joints]}
        if inst.objects:
            pos, orn = c.getBasePositionAndOrientation(inst.objects[0])
            lin, ang = c.getBaseVelocity(inst.objects[0])
            s.update(block_pos=list(pos), block_orn=list(orn), block_lin=list(lin), block_ang=list(ang))
        if inst.play:
            dp, do = c.getBasePositionAndOrientation(inst.drawer['drawer'])
            dl, da = c.getBaseVelocity(inst.drawer['drawer'])
            s.update(drawer_pos=list(dp), drawer_orn=list(do), drawer_lin=list(dl), drawer_ang=list(da),
                     scene_joints=[list(c.getJointState(b, 0)[:2]) for b in inst.joints])
        return s
    out['initial_state'] = snapshot()
    out['initial_obs'] = {k: np.asarray(v).tolist() for k, v in obs.items() if v is not None and k != 'img'}
    acts = make_actions(args.env, args.steps, args.seed, args.scenario)
    traj = []
    for t in range(args.steps):
        a = acts[t] if args.scenario == 'random' else grasp_action(args.env, obs['achieved_goal'][:3], t)
        obs, r, _, info = env.step(a)
        s = snapshot()
        s.update(action=a.tolist(), obs_quat=np.asarray(obs['obs_quat']).tolist(), reward=float(r), target_poses=np.asarray(info['target_poses']).tolist())
        traj.append(s)
    out['trajectory'] = traj
    json.dump(out, open(args.dump, 'w'))
    print('wrote %s: %d steps of %s on PyBullet API %s' % (args.dump, args.steps, args.env, out['pybullet_api_version']))


# ------------------------------------------------------------------ the same fixture from the CPU oracle (format check only)
def dump_from_oracle(args):
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    from oracle import OracleEnv
    o = OracleEnv(args.env, seed=args.seed, env_index=0)
    obs = o.reset()
    out = {'format': FORMAT, 'source': 'oracle', 'env': args.env, 'seed': args.seed, 'scenario': args.scenario,
           'arm_type': 'UR5' if args.env.startswith('UR5') else 'Panda'}
    out['initial_state'] = oracle_snapshot(o)
    out['initial_obs'] = {k: np.asarray(v).tolist() for k, v in obs.items() if v is not None and k != 'img'}
    acts = make_actions(args.env, args.steps, args.seed, args.scenario)
    traj = []
    for t in range(args.steps):
        a = acts[t] if args.scenario == 'random' else grasp_action(args.env, obs['achieved_goal'][:3], t)
        obs, r, _, info = o.step(a)
        s = oracle_snapshot(o)
        s.update(action=a.tolist(), obs_quat=obs['obs_quat'].tolist(), reward=float(r), target_poses=np.asarray(info['target_poses']).tolist())
        traj.append(s)
    out['trajectory'] = traj
    json.dump(out, open(args.dump, 'w'))
    print('wrote %s from the CPU oracle (format check, not a pin)' % args.dump)


def oracle_snapshot(o):
    st = o.get_state()
    na = o.n_arm
    s = {'q': st[:na].tolist(), 'qd': st[na:2 * na].tolist()}
    p = 2 * na
    nfree = (o.nv - na) // 6 if o.kind in (0, 2, 4, 5) else 0
    if o.kind in (0, 4):
        nfree = 2
    elif o.kind == 2:
        nfree = 1
    elif o.kind == 5:
        nfree = 3
    if nfree:
        s.update(block_pos=st[p:p + 3].tolist(), block_orn=st[p + 3:p + 7].tolist(), block_lin=st[p + 7:p + 10].tolist(), block_ang=st[p + 10:p + 13].tolist())
    if o.kind in (0, 4):
        d = p + 13
        s.update(drawer_pos=st[d:d + 3].tolist(), drawer_orn=st[d + 3:d + 7].tolist(), drawer_lin=st[d + 7:d + 10].tolist(), drawer_ang=st[d + 10:d + 13].tolist())
        j = p + 26
        # oracle order door, button, dial -> the reference's self.joints order [door, button, dial] (scenes.py complex_scene return)
        s['scene_joints'] = [[st[j + k], st[j + 3 + k]] for k in range(3)]
    return s


def state_vector_from_snapshot(kind, n_arm, snap):
    """recorded snapshot -> the oracle's rpo_set_state vector (tests use it to start from PyBullet's state)"""
    v = list(snap['q']) + list(snap['qd'])
    if 'block_pos' in snap:
        v += list(snap['block_pos']) + list(snap['block_orn']) + list(snap['block_lin']) + list(snap['block_ang'])
    if 'drawer_pos' in snap:
        v += list(snap['drawer_pos']) + list(snap['drawer_orn']) + list(snap['drawer_lin']) + list(snap['drawer_ang'])
        v += [sj[0] for sj in snap['scene_joints']] + [sj[1] for sj in snap['scene_joints']]
    return np.array(v, dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--env', default='UR5PlayAbsRPY1Obj-v0', choices=sorted(CLASSES))
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--scenario', default='random', choices=['random', 'grasp'])
    ap.add_argument('--reference-root', default='/root/reference')
    ap.add_argument('--dump', help='fixture to write, e.g. tests/golden/pybullet_<env id>.json')
    ap.add_argument('--all', action='store_true', help="every BASELINE id x {random, grasp} into tests/golden/pybullet_<env id>_<scenario>.json (the grasp scenario only for ids with a block)")
    ap.add_argument('--from-oracle', action='store_true', help='write the fixture from the CPU oracle (format check only)')
    args = ap.parse_args()
    dump = dump_from_oracle if args.from_oracle else dump_from_pybullet
    if args.all:
        for env_id in ALL_IDS:
            for scenario in ('random', 'grasp'):
                if scenario == 'grasp' and 'Reach' in env_id:
                    continue
                args.env, args.scenario = env_id, scenario
                args.dump = os.path.join(REPO, 'tests', 'golden', 'pybullet_%s_%s.json' % (env_id, scenario))
                dump(args)
        return
    if not args.dump:
        ap.error('--dump FILE or --all')
    dump(args)


if __name__ == '__main__':
    main()
