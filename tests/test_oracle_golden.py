"""CPU oracle vs the reference's own harness arithmetic (tests/golden/*.json, made by executing the reference's
Python against a recording fake client — tests/golden/make_goldens.py).  No GPU."""
import numpy as np
import pytest

from oracle import OracleEnv, quat_from_euler, euler_from_quat, dial_to_0_1_range

DT = 1.0 / 300.0
F32_KEYS = ('obs_quat', 'achieved_goal', 'desired_goal', 'controllable_achieved_goal', 'full_positional_state')
EE = {'U': 7, 'R': 7, 'P': 11, 'Q': 11, 'V': 11, 'W': 11}
GRIP_JOINT = {'U': '18', 'R': '18', 'P': '9', 'Q': '9', 'V': '9', 'W': '9'}


def readings_from_world(kind, w):
    ee = w['link'][str(EE[kind])]
    kw = dict(ee_pos=ee['pos'], ee_orn=ee['orn'], ee_lin=ee['lin'], ee_ang=ee['ang'], grip_q=w['joint'][GRIP_JOINT[kind]],
              joints=[w['joint'][str(j)] for j in range(8)])
    ray = w['ray']
    if kind in ('P', 'Q', 'V', 'W'):
        kw['proprio'] = -1
    else:   # environments.py:736: nothing in hand if the ray misses or hits a pad
        kw['proprio'] = 0 if (ray['fraction'] == 1.0 or ray['link'] in (18, 20)) else 1
    if 'block0' in w['base']:
        b = w['base']['block0']
        kw.update(block_pos=b['pos'], block_orn=b['orn'], block_vel=b['lin'])
    if 'block1' in w['base']:
        b = w['base']['block1']
        kw.update(block2_pos=b['pos'], block2_orn=b['orn'], block2_vel=b['lin'])
    if kind in ('U', 'V', 'W'):
        kw.update(drawer_y=w['base']['drawer']['pos'][1], door_q=w['joint']['door'], button_q=w['joint']['button'],
                  dial_q=w['joint']['dial'])
    return kw


def check_obs(got, want, tol=0.0):
    for k, spec in want.items():
        if spec is None:
            assert got[k] is None
            continue
        v = spec['v']
        if k in F32_KEYS:
            assert spec['dtype'] == 'float32'
            np.testing.assert_array_equal(np.asarray(got[k], dtype=np.float32), np.asarray(v, dtype=np.float32), err_msg=k)
        elif k == 'gripper_proprioception':
            assert int(got[k]) == int(v)
        else:
            g = np.asarray(got[k], dtype=np.float64)
            np.testing.assert_allclose(g, np.asarray(v, dtype=np.float64), rtol=0, atol=tol, err_msg=k)
            assert g.shape == np.asarray(v).shape


def test_joint_index_table_matches_reference_notebook(golden):
    t = golden('ur5_joint_index_table.json')
    lines = [l for l in t['notebook_stdout'].strip().splitlines() if l.strip()]
    names = [l.split("b'")[1].rstrip("'") for l in lines]
    assert names == t['dfs_names']
    assert len(names) == 22
    assert [i for i, ty in enumerate(t['dfs_types']) if ty != 4] == [0, 1, 2, 3, 4, 5, 10, 12, 13, 15, 18, 20]
    assert [i for i, ty in enumerate(t['panda_dfs_types']) if ty != 4] == [0, 1, 2, 3, 4, 5, 6, 9, 10]


@pytest.mark.parametrize('kind', ['U', 'R', 'P'])
def test_calc_state_assembly(golden, kind):
    """obs layout, dtypes, quaternion sign memory, dial mapping, reward (environments.py:799-894)."""
    for seq in golden('calc_state.json')[kind]:
        env = OracleEnv(kind)
        env.set_goal(seq['goal'])
        for step in seq['steps']:
            got = env.assemble_obs(**readings_from_world(kind, step['world']))
            # 'observation' passes through atan2/asin: allow 1 ulp-class differences between libm and python math
            check_obs(got, step['obs'], tol=1e-12)
            r = env.compute_reward(np.float32(got['achieved_goal']), np.float32(got['desired_goal']))
            assert r == pytest.approx(step['reward'], abs=1e-7)


@pytest.mark.parametrize('kind', ['U', 'R', 'P'])
def test_action_to_motor_targets(golden, kind):
    """clip -> rpy->quat -> IK call args -> joint clamps -> motor commands (environments.py:206-208, 955-1073)."""
    high = np.array([6, 6, 6, 6, 6, 6, 1.0])
    n_arm_j = 7 if kind == 'P' else 6
    for case in golden('step.json')[kind]:
        env = OracleEnv(kind)
        a = np.clip(np.array(case['action']), -high, high)
        w = case['world']
        s = env.get_state()
        bullet_dofs = [0, 1, 2, 3, 4, 5, 10, 12, 13, 15, 18, 20] if kind != 'P' else [0, 1, 2, 3, 4, 5, 6, 9, 10]
        for d, j in enumerate(bullet_dofs):
            s[d] = w['joint'][str(j)]
        env.set_state(s)
        ik_log = case['shadow_log'] if kind != 'P' else case['main_log']
        ik_calls = [e for e in ik_log if e['fn'] == 'calculateInverseKinematics']
        assert len(ik_calls) == (4 if kind != 'P' else 1)
        for c in ik_calls:
            np.testing.assert_allclose(c['args'][2], a[:3], atol=0)
            np.testing.assert_allclose(c['args'][3], quat_from_euler(a[3:6]), atol=1e-15)
        if kind != 'P':      # calc_angles seeds the shadow arm with the measured joints (inverseKinematics.py:46)
            first = [e for e in case['shadow_log'] if e['fn'] == 'resetJointState'][:6]
            np.testing.assert_array_equal([e['args'][2] for e in first], s[:6])
        else:
            assert ik_calls[0]['kwargs'] == {'maxNumIterations': 200}
        tp = env.goto_joint_poses(np.array(case['ik_returns'][-1])[:n_arm_j], gripper=a[6])
        np.testing.assert_allclose(tp, case['target_poses'], rtol=0, atol=1e-15)
        arr = [e for e in case['main_log'] if e['fn'] == 'setJointMotorControlArray'][0]
        np.testing.assert_allclose(tp, arr['kwargs']['targetPositions'], rtol=0, atol=1e-15)
        mode, tgt, maximp = env.get_motor()
        for d in range(n_arm_j):
            assert mode[d] == 1 and maximp[d] == pytest.approx(240.0 * DT, rel=1e-15)
        singles = [e for e in case['main_log'] if e['fn'] == 'setJointMotorControl2']
        assert len(singles) == (2 if kind == 'P' else 6)
        for e in singles:
            d = bullet_dofs.index(e['args'][1])
            assert mode[d] == 1
            assert tgt[d] == pytest.approx(e['args'][3], abs=1e-15)
            assert maximp[d] == pytest.approx(e['kwargs']['force'] * DT, rel=1e-15)
        assert sum(1 for e in case['main_log'] if e['fn'] == 'stepSimulation') == 12
        assert case['done'] is False
        assert case['is_success'] == (0 if case['reward'] < 0 else 1)


FAMILY_IDS = ['UR5Play1Obj-v0', 'UR5PlayRel1Obj-v0', 'UR5PlayRelJoints1Obj-v0', 'UR5PlayAbsJoints1Obj-v0', 'UR5PlayRelRPY1Obj-v0']


@pytest.mark.parametrize('gid', FAMILY_IDS)
def test_action_types_of_the_ur5_play_family(golden, gid):
    """The other action types (environments.py:88-113, 915-981): action-space bounds and clip, IK target from the action
    and the measured EE pose (relative types add componentwise, relative_rpy through getEulerFromQuaternion), joint-space
    types without IK, then the same clamps and motor commands as absolute_rpy."""
    from oracle import FAMILY
    g = golden('step_family.json')[gid]
    info = g['info']
    assert info['action_type'] == FAMILY[gid]
    env0 = OracleEnv(gid)
    high = np.array(info['action_high'])
    assert env0.n_action == len(high)
    np.testing.assert_array_equal(np.array(info['action_low']), -high)
    for case in g['cases']:
        env = OracleEnv(gid)
        a = np.clip(np.array(case['action']), -high, high)
        w = case['world']
        s = env.get_state()
        bullet_dofs = [0, 1, 2, 3, 4, 5, 10, 12, 13, 15, 18, 20]
        for d, j in enumerate(bullet_dofs):
            s[d] = w['joint'][str(j)]
        env.set_state(s)
        ik_calls = [e for e in case['shadow_log'] if e['fn'] == 'calculateInverseKinematics']
        if info['action_type'] in ('absolute_joints', 'relative_joints'):
            assert len(ik_calls) == 0
            jp = a[:6] + (s[:6] if info['action_type'] == 'relative_joints' else 0.0)
            tp = env.goto_joint_poses(jp, gripper=a[6])
        else:
            assert len(ik_calls) == 4
            ee = w['link']['7']
            pos, quat = env.action_target(a, ee['pos'], ee['orn'])
            for c in ik_calls:
                np.testing.assert_allclose(c['args'][2], pos, rtol=0, atol=1e-15)
                np.testing.assert_allclose(c['args'][3], quat, rtol=0, atol=1e-14)
            tp = env.goto_joint_poses(np.array(case['ik_returns'][-1])[:6], gripper=a[-1])
        np.testing.assert_allclose(tp, case['target_poses'], rtol=0, atol=1e-15)
        arr = [e for e in case['main_log'] if e['fn'] == 'setJointMotorControlArray'][0]
        np.testing.assert_allclose(tp, arr['kwargs']['targetPositions'], rtol=0, atol=1e-15)
        mode, tgt, maximp = env.get_motor()
        singles = [e for e in case['main_log'] if e['fn'] == 'setJointMotorControl2']
        assert len(singles) == 6
        for e in singles:
            d = bullet_dofs.index(e['args'][1])
            assert mode[d] == 1
            assert tgt[d] == pytest.approx(e['args'][3], abs=1e-15)
            assert maximp[d] == pytest.approx(e['kwargs']['force'] * DT, rel=1e-15)
        assert sum(1 for e in case['main_log'] if e['fn'] == 'stepSimulation') == 12


PANDA_IDS = ['pandaReach-v0', 'pandaReach2D-v0', 'pandaPlay1Obj-v0', 'pandaPlayRel1Obj-v0', 'pandaPlayRelJoints1Obj-v0',
             'pandaPlayAbsJoints1Obj-v0', 'pandaPlayAbsRPY1Obj-v0', 'pandaPlayRelRPY1Obj-v0']


@pytest.mark.parametrize('gid', PANDA_IDS)
def test_panda_reach_and_play_ids(golden, gid):
    """The Panda ids beyond pick / push (envList.py:8-10, 24-88): the Panda in default_scene (reach) and in complex_scene (one-
    object play, six action types).  Attributes, ranges, action space; per step() case the IK call on the live arm, joint
    clamps, motor commands and the observation assembled from the read-back."""
    g = golden('panda_ids.json')[gid]
    info = g['info']
    env0 = OracleEnv(gid)
    kind = 'V' if info['play'] else 'Q'
    assert info['arm_type'] == 'Panda' and info['num_dofs'] == 7 and info['ee_index'] == 11
    initP = golden('scenes.json')['instance_init_P']            # base pose, EE link and rest pose go with the arm type
    np.testing.assert_array_equal(info['base_pos'], initP['base_pos'])
    np.testing.assert_array_equal(info['base_orn'], initP['base_orn'])
    np.testing.assert_array_equal(info['rest'], initP['rest'])
    assert env0.action_type == info['action_type']
    high = np.array(info['action_high'])
    assert env0.n_action == len(high)
    np.testing.assert_array_equal(env0.action_high(), high)
    np.testing.assert_array_equal(np.array(info['action_low']), -high)
    fl = env0.flags()
    assert (fl['play'], fl['use_orientation'], fl['return_velocity'], fl['num_objects']) == \
        (int(info['play']), int(info['use_orientation']), int(info['return_velocity']), info['num_objects'])
    rg = env0.ranges()
    np.testing.assert_allclose(rg['goal_lo'], info['goal_lower_bound'], atol=0)
    np.testing.assert_allclose(rg['goal_hi'], info['goal_upper_bound'], atol=0)
    np.testing.assert_allclose(rg['env_hi'], info['env_upper_bound'], atol=0)
    if info['num_objects']:
        np.testing.assert_allclose(rg['obj_lo'], info['obj_lower_bound'], atol=0)
        np.testing.assert_allclose(rg['obj_hi'], info['obj_upper_bound'], atol=0)
    bullet_dofs = [0, 1, 2, 3, 4, 5, 6, 9, 10]
    for case in g['cases']:
        env = OracleEnv(gid)
        a = np.clip(np.array(case['action']), -high, high)
        w = case['world']
        s = env.get_state()
        for d, j in enumerate(bullet_dofs):
            s[d] = w['joint'][str(j)]
        env.set_state(s)
        ik_calls = [e for e in case['main_log'] if e['fn'] == 'calculateInverseKinematics']
        if info['action_type'] in ('absolute_joints', 'relative_joints'):
            assert len(ik_calls) == 0
            jp = a[:7] + (s[:7] if info['action_type'] == 'relative_joints' else 0.0)
            tp = env.goto_joint_poses(jp, gripper=a[7])
        else:
            assert len(ik_calls) == 1 and ik_calls[0]['kwargs'] == {'maxNumIterations': 200}
            ee = w['link']['11']
            pos, quat = env.action_target(a, ee['pos'], ee['orn'])
            np.testing.assert_allclose(ik_calls[0]['args'][2], pos, rtol=0, atol=1e-15)
            np.testing.assert_allclose(ik_calls[0]['args'][3], quat, rtol=0, atol=1e-14)
            tp = env.goto_joint_poses(np.array(case['ik_returns'][-1])[:7], gripper=a[-1])
        np.testing.assert_allclose(tp, case['target_poses'], rtol=0, atol=1e-15)
        arr = [e for e in case['main_log'] if e['fn'] == 'setJointMotorControlArray'][0]
        np.testing.assert_allclose(tp, arr['kwargs']['targetPositions'], rtol=0, atol=1e-15)
        mode, tgt, maximp = env.get_motor()
        singles = [e for e in case['main_log'] if e['fn'] == 'setJointMotorControl2']
        assert len(singles) == 2
        for e in singles:
            d = bullet_dofs.index(e['args'][1])
            assert mode[d] == 1
            assert tgt[d] == pytest.approx(e['args'][3], abs=1e-15)
            assert maximp[d] == pytest.approx(e['kwargs']['force'] * DT, rel=1e-15)
        assert sum(1 for e in case['main_log'] if e['fn'] == 'stepSimulation') == 12
        env.set_goal(case['goal'])
        got = env.assemble_obs(**readings_from_world(kind, w))
        check_obs(got, case['obs'], tol=1e-12)
        r = env.compute_reward(np.float32(got['achieved_goal']), np.float32(got['desired_goal']))
        assert r == pytest.approx(case['reward'], abs=1e-7)


@pytest.mark.parametrize('gid', ['pandaPlay-v0', 'pandaPlayJoints-v0'])
def test_two_object_play_ids(golden, gid):
    """pandaPlay-v0 / pandaPlayJoints-v0 (envList.py:28-41): two blocks.  Attributes and action space; step() cases (IK call,
    clamps, motor commands, 26 / 18-wide observation); the quaternion sign memory with its (19, 23) pair; reset(): five
    uniforms + choice + random per attempt, both spawn heights, the arm target, the goal of the last attempt."""
    g = golden('two_object_ids.json')[gid]
    info = g['info']
    env0 = OracleEnv(gid)
    assert info['arm_type'] == 'Panda' and info['num_objects'] == 2 and info['num_goals'] == 2 and info['play']
    assert env0.action_type == info['action_type']
    high = np.array(info['action_high'])
    np.testing.assert_array_equal(env0.action_high(), high)
    fl = env0.flags()
    assert (fl['play'], fl['use_orientation'], fl['return_velocity'], fl['num_objects']) == (1, 1, 0, 2)
    rg = env0.ranges()
    np.testing.assert_allclose(rg['goal_lo'], info['goal_lower_bound'], atol=0)
    np.testing.assert_allclose(rg['goal_hi'], info['goal_upper_bound'], atol=0)
    np.testing.assert_allclose(rg['obj_lo'], info['obj_lower_bound'], atol=0)
    np.testing.assert_allclose(rg['obj_hi'], info['obj_upper_bound'], atol=0)
    np.testing.assert_allclose(rg['env_hi'], info['env_upper_bound'], atol=0)
    bullet_dofs = [0, 1, 2, 3, 4, 5, 6, 9, 10]
    for case in g['cases']:
        env = OracleEnv(gid)
        a = np.clip(np.array(case['action']), -high, high)
        w = case['world']
        s = env.get_state()
        for d, j in enumerate(bullet_dofs):
            s[d] = w['joint'][str(j)]
        env.set_state(s)
        ik_calls = [e for e in case['main_log'] if e['fn'] == 'calculateInverseKinematics']
        if info['action_type'] == 'relative_joints':
            assert len(ik_calls) == 0
            tp = env.goto_joint_poses(a[:7] + s[:7], gripper=a[7])
        else:
            assert len(ik_calls) == 1 and ik_calls[0]['kwargs'] == {'maxNumIterations': 200}
            ee = w['link']['11']
            pos, quat = env.action_target(a, ee['pos'], ee['orn'])
            np.testing.assert_allclose(ik_calls[0]['args'][2], pos, rtol=0, atol=1e-15)
            np.testing.assert_allclose(ik_calls[0]['args'][3], quat, rtol=0, atol=1e-14)
            tp = env.goto_joint_poses(np.array(case['ik_returns'][-1])[:7], gripper=a[-1])
        np.testing.assert_allclose(tp, case['target_poses'], rtol=0, atol=1e-15)
        assert sum(1 for e in case['main_log'] if e['fn'] == 'stepSimulation') == 12
        env.set_goal(case['goal'])
        got = env.assemble_obs(**readings_from_world('W', w))
        check_obs(got, case['obs'], tol=1e-12)
        assert len(got['obs_quat']) == 26 and len(got['achieved_goal']) == 18 and len(got['observation']) == 25
        r = env.compute_reward(np.float32(got['achieved_goal']), np.float32(got['desired_goal']))
        assert r == pytest.approx(case['reward'], abs=1e-7)
    for seq in g['calc_state']:
        env = OracleEnv(gid)
        env.set_goal(seq['goal'])
        for step in seq['steps']:
            got = env.assemble_obs(**readings_from_world('W', step['world']))
            check_obs(got, step['obs'], tol=1e-12)
            r = env.compute_reward(np.float32(got['achieved_goal']), np.float32(got['desired_goal']))
            assert r == pytest.approx(step['reward'], abs=1e-7)
    for case in g['resets']:
        draws = case['draws']
        per = ['uniform', 'uniform', 'uniform', 'uniform', 'uniform', 'choice', 'random']
        assert len(draws) % 7 == 0 and [d['fn'] for d in draws][:7] == per
        u = []
        for d in draws[:7]:
            u += d['u'] if isinstance(d['u'], list) else [d['u']]
        assert len(u) == 17
        env = OracleEnv(gid)
        blocks, target = env.reset_samples(u)
        spawns = [e for e in case['log'] if e['fn'] == 'resetBasePositionAndOrientation' and e['args'][2] == [0.0, 0.0, 0.7071, 0.7071]]
        np.testing.assert_allclose(blocks[0:3], spawns[0]['args'][1], rtol=0, atol=1e-15)       # +0.03
        np.testing.assert_allclose(blocks[3:6], spawns[1]['args'][1], rtol=0, atol=1e-15)       # +0.06
        ik = [e for e in case['log'] if e['fn'] == 'calculateInverseKinematics'][0]
        np.testing.assert_allclose(target, ik['args'][2], rtol=0, atol=1e-15)
        assert [e['n'] for e in case['log'] if e['fn'] == 'stepSimulation_x'][0] == 100
        env.reset(u=np.tile(u, 16))
        assert env.last_used % 17 == 0 and env.last_used >= 17
        ag = np.array(case['obs']['achieved_goal']['v'], dtype=np.float32)
        idx = int(draws[-2]['u'] * 18)
        want = ag.copy()
        want[idx] = np.float32(want[idx] + np.float32(draws[-1]['u']))
        np.testing.assert_array_equal(np.array(case['goal'], dtype=np.float32), want)


def test_rewards_and_dial(golden):
    g = golden('rewards.json')
    env = OracleEnv('U')
    for row in g['success_func']:
        assert env.compute_reward(row['ag'], row['g']) == row['r']
    assert {r['r'] for r in g['success_func']} == {0, -1}
    for kind in ('R', 'P'):
        env = OracleEnv(kind)
        for row in g['sparse'][kind]['single']:
            assert env.compute_reward(row['ag'], row['dg']) == pytest.approx(row['r'], abs=1e-15)
        b = g['sparse'][kind]['batch']
        got = [env.compute_reward(a, d) for a, d in zip(b['ag'], b['dg'])]
        np.testing.assert_allclose(got, b['r'], atol=1e-15)
    for row in g['dial']:
        assert dial_to_0_1_range(row['x']) == pytest.approx(row['y'], abs=1e-15)
    assert dial_to_0_1_range(1.0) == pytest.approx(dial_to_0_1_range(3.0))     # (x mod 2)/2.2 quirk


@pytest.mark.parametrize('kind', ['U', 'R', 'P'])
def test_reset_sampling(golden, kind):
    """which uniforms reset() draws, in which order, and where they land (environments.py:492-603)."""
    for case in golden('reset.json')[kind]:
        draws = case['draws']
        per_loop = {'U': ['uniform', 'uniform', 'uniform', 'choice', 'random'], 'R': ['uniform', 'uniform'],
                    'P': ['uniform', 'uniform', 'uniform']}[kind]
        assert [d['fn'] for d in draws][:len(per_loop)] == per_loop
        u = []
        for d in draws[:len(per_loop)]:
            u += d['u'] if isinstance(d['u'], list) else [d['u']]
        env = OracleEnv(kind)
        block, target = env.reset_samples(u)
        log = case['log']
        if kind != 'R':
            spawn = [e for e in log if e['fn'] == 'resetBasePositionAndOrientation' and e['args'][2] == [0.0, 0.0, 0.7071, 0.7071]][0]
            np.testing.assert_allclose(block[:3], spawn['args'][1], rtol=0, atol=1e-15)
        ik = [e for e in log if e['fn'] == 'calculateInverseKinematics'][0]
        np.testing.assert_allclose(target, ik['args'][2], rtol=0, atol=1e-15)
        assert ik['args'][3] == [0.0, 0.0, 0.0, 1.0]
        assert [e['n'] for e in log if e['fn'] == 'stepSimulation_x'][0] == 100
        # the whole reset on the oracle consumes draws in whole attempts
        env.reset(u=np.tile(u, 16))
        assert env.last_used % len(u) == 0 and env.last_used >= len(u)
        if kind == 'U':
            # play goal = float32 achieved_goal with one index bumped (environments.py:511-516)
            ag = np.array(case['obs']['achieved_goal']['v'], dtype=np.float32)
            assert len(draws) % 5 == 0                       # whole attempts; the goal comes from the last one
            idx = int(draws[-2]['u'] * 11)
            want = ag.copy()
            want[idx] = np.float32(want[idx] + np.float32(draws[-1]['u']))
            np.testing.assert_array_equal(np.array(case['goal'], dtype=np.float32), want)
            env2 = OracleEnv('U')
            env2.reset_goal_pos(None, u=[0.1, 0.2, 0.3, draws[-2]['u'], draws[-1]['u']])
            o = env2.calc_state()
            want2 = np.float32(o['achieved_goal']).copy()
            want2[idx] = np.float32(want2[idx] + np.float32(draws[-1]['u']))
            np.testing.assert_array_equal(np.float32(o['desired_goal']), want2)


@pytest.mark.parametrize('kind', ['U', 'R', 'P'])
def test_reset_to_an_observation(golden, kind):
    """reset(o) (environments.py:519-603 with obs given): which entries of o place the object (index 11 / 7, quirk) and the
    arm's IK target, drawer / scene joints back to their defaults, rest pose before the single default IK, first six joints
    written back, no stepSimulation, and the goal draws per attempt."""
    for case in golden('reset_to.json')[kind]:
        o = np.array(case['o'])
        log = case['log']
        assert not any(e['fn'] == 'stepSimulation' for e in log)
        per_loop = ['uniform', 'choice', 'random'] if kind == 'U' else ['uniform']
        draws = case['draws']
        assert [d['fn'] for d in draws][:len(per_loop)] == per_loop and len(draws) % len(per_loop) == 0
        ik = [e for e in log if e['fn'] == 'calculateInverseKinematics'][0]
        np.testing.assert_array_equal(ik['args'][2], o[0:3])
        np.testing.assert_array_equal(ik['args'][3], o[3:7] if kind == 'U' else [0.0, 0.0, 0.0, 1.0])
        if kind != 'R':
            idx = 11 if kind == 'U' else 7
            spawn = [e for e in log if e['fn'] == 'resetBasePositionAndOrientation'][-1 if kind == 'U' else 0]
            np.testing.assert_array_equal(spawn['args'][1], o[idx:idx + 3])
            np.testing.assert_array_equal(spawn['args'][2], o[idx + 3:idx + 7] if kind == 'U' else [0, 0, 0, 1])
        # the oracle: same placement (the IK itself is the oracle's own), same number of draws per attempt
        u = []
        for d in draws:
            u += d['u'] if isinstance(d['u'], list) else [d['u']]
        env = OracleEnv(kind)
        obs = env.reset_to(o, u=np.tile(u, 8))
        per_attempt = 5 if kind == 'U' else 3
        assert env.last_used % per_attempt == 0 and env.last_used >= per_attempt
        s = env.get_state()
        n_arm = 9 if kind == 'P' else 12
        if kind != 'R':
            idx = 11 if kind == 'U' else 7
            np.testing.assert_allclose(s[2 * n_arm:2 * n_arm + 3], o[idx:idx + 3], atol=1e-15)
            np.testing.assert_allclose(s[2 * n_arm + 3:2 * n_arm + 7], o[idx + 3:idx + 7] if kind == 'U' else [0, 0, 0, 1], atol=1e-15)
            np.testing.assert_array_equal(s[2 * n_arm + 7:2 * n_arm + 13], 0.0)
        assert np.all(s[n_arm:2 * n_arm] == 0.0)           # arm at rest velocity
        if kind == 'U':                                     # drawer and scene joints at their defaults
            np.testing.assert_allclose(s[2 * n_arm + 13:2 * n_arm + 16], [-0.1, 0.0, -0.04], atol=1e-12)
            np.testing.assert_array_equal(s[2 * n_arm + 26:2 * n_arm + 32], 0.0)
            ag = np.float32(obs['achieved_goal'])
            np.testing.assert_allclose(ag[0:3], np.float32(o[11:14]), atol=1e-6)


def test_euler_quaternion_identities():
    """getQuaternionFromEuler / getEulerFromQuaternion pinned analytically, not by the stub."""
    from urdf_tree import rpy_to_mat
    rng = np.random.default_rng(3)
    for _ in range(50):
        rpy = rng.uniform([-3.1, -1.5, -3.1], [3.1, 1.5, 3.1])
        q = quat_from_euler(rpy)
        assert abs(np.linalg.norm(q) - 1) < 1e-14
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(R, rpy_to_mat(rpy), atol=1e-14)       # fixed-axis XYZ = Rz Ry Rx
        np.testing.assert_allclose(euler_from_quat(q), rpy, atol=1e-12)
    np.testing.assert_allclose(quat_from_euler([0, 0, np.pi / 2]), [0, 0, np.sqrt(0.5), np.sqrt(0.5)], atol=1e-15)
    np.testing.assert_allclose(euler_from_quat([0, np.sqrt(0.5), 0, np.sqrt(0.5)]), [0, np.pi / 2, 0], atol=1e-12)   # gimbal branch
