#!/usr/bin/env python3
"""Generate tests/golden/*.json by EXECUTING the reference's Python against a fake client.

Runs only in the build container (needs /root/reference); the JSON it writes is data
(inputs + the reference's outputs) and is what travels to the GPU box.  PyBullet itself is
absent, so nothing here pins physics — only the reference's own harness arithmetic
(SURVEY.md §8c): scene construction calls, action -> motor-target mapping, observation
assembly, quaternion sign memory, rewards, dial mapping, declared spaces, reset sampling.

    python tests/golden/make_goldens.py        # rewrites the fixtures deterministically
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import fake_bullet as fb  # noqa: E402
import urdf_tree  # noqa: E402

ENVS = os.path.join(REF, 'roboticsPlayroomPybullet', 'envs')
ur5_tree = urdf_tree.parse_urdf(os.path.join(ENVS, 'ur_e_description', 'ur5e2.urdf'))
panda_tree = urdf_tree.parse_urdf(os.path.join(ENVS, 'franka_panda', 'panda.urdf'))
fb.UR5_JOINT_TYPES = urdf_tree.joint_types(ur5_tree)
fb.PANDA_JOINT_TYPES = urdf_tree.joint_types(panda_tree)

CLIENTS = []
fb.install_stubs(CLIENTS)
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    import roboticsPlayroomPybullet  # noqa: E402,F401  (fills fb.REGISTRY)
    from roboticsPlayroomPybullet.envs import UR5PlayAbsRPY1Obj, UR5Reach, pandaPick  # noqa: E402
    from roboticsPlayroomPybullet.envs import pandaPush  # noqa: E402
    from roboticsPlayroomPybullet.envs import (pandaReach, pandaReach2D, pandaPlay1Obj, pandaPlayRel1Obj, pandaPlayRelJoints1Obj,  # noqa: E402
                                               pandaPlayAbsJoints1Obj, pandaPlayAbsRPY1Obj, pandaPlayRelRPY1Obj)
    from roboticsPlayroomPybullet.envs import pandaPlay, pandaPlayRelJoints  # noqa: E402
    from roboticsPlayroomPybullet.envs import (UR5Play1Obj, UR5PlayRel1Obj, UR5PlayRelJoints1Obj, UR5PlayAbsJoints1Obj,  # noqa: E402
                                               UR5PlayRelRPY1Obj)
    import scenes  # noqa: E402  (the reference puts envs/ on sys.path itself)
    import playRewardFunc  # noqa: E402

KINDS = {'U': UR5PlayAbsRPY1Obj, 'R': UR5Reach, 'P': pandaPick}
IDS = {'U': 'UR5PlayAbsRPY1Obj-v0', 'R': 'UR5Reach-v0', 'P': 'pandaPick-v0'}
# the rest of the UR5 one-object play family: the same scene and arm as U, other action types
FAMILY = {'UR5Play1Obj-v0': UR5Play1Obj, 'UR5PlayRel1Obj-v0': UR5PlayRel1Obj, 'UR5PlayRelJoints1Obj-v0': UR5PlayRelJoints1Obj,
          'UR5PlayAbsJoints1Obj-v0': UR5PlayAbsJoints1Obj, 'UR5PlayRelRPY1Obj-v0': UR5PlayRelRPY1Obj}


def dump(name, obj):
    with open(os.path.join(HERE, name), 'w') as f:
        json.dump(fb._plain(obj), f, indent=None, separators=(',', ':'))
        f.write('\n')
    print('wrote', name, os.path.getsize(os.path.join(HERE, name)), 'bytes')


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def new_env(kind, cls=None):
    """Construct the reference env and activate its (fake) physics client."""
    del CLIENTS[:]
    env = quiet(cls or KINDS[kind])
    quiet(env.activate_physics_client)
    env.physics_client_active = True
    main = CLIENTS[0]
    shadow = CLIENTS[1] if len(CLIENTS) > 1 else None
    return env, main, shadow


def unit_quat(rng):
    q = rng.normal(size=4)
    return (q / np.linalg.norm(q)).tolist()


def fill_world(kind, env, c, rng):
    """Random but plausible world read-back; returns the JSON-able description."""
    inst = env.instance
    arm = inst.arm
    ee = inst.endEffectorIndex
    w = {'joint': {}, 'link': {}, 'base': {}}
    desc = {'joint': {}, 'link': {}, 'base': {}}

    def link(i):
        s = {'pos': rng.uniform(-0.4, 0.6, 3).tolist(), 'orn': unit_quat(rng),
             'lin': rng.uniform(-1, 1, 3).tolist(), 'ang': rng.uniform(-2, 2, 3).tolist()}
        w['link'][(arm, i)] = s
        desc['link'][str(i)] = s

    n_j = c.getNumJoints(arm)
    panda = inst.arm_type == 'Panda'
    for j in range(n_j):
        q = float(rng.uniform(-2.5, 2.5))
        if not panda and j in (18, 20):
            q = float(rng.uniform(0.0, 0.0448))
        if panda and j in (9, 10):
            q = float(rng.uniform(0.0, 0.04))
        w['joint'][(arm, j)] = q
        desc['joint'][str(j)] = q
    link(ee)
    if not panda:
        for i in (ee - 1, 18, 20):
            link(i)
    for k, o in enumerate(inst.objects):
        s = {'pos': rng.uniform(-0.2, 0.3, 3).tolist(), 'orn': unit_quat(rng),
             'lin': rng.uniform(-1, 1, 3).tolist(), 'ang': rng.uniform(-1, 1, 3).tolist()}
        w['base'][o] = s
        desc['base']['block%d' % k] = s
    if inst.play:
        d = inst.drawer['drawer']
        s = {'pos': [-0.1, float(rng.uniform(-0.06, 0.075)), -0.04], 'orn': inst.drawer['defaults']['ori'],
             'lin': [0, 0, 0], 'ang': [0, 0, 0]}
        w['base'][d] = s
        desc['base']['drawer'] = s
        names = ['door', 'button', 'dial']
        ranges = [(-0.16, 0.16), (0.0, 0.035), (-7.0, 7.0)]
        for nm, b, r in zip(names, inst.joints, ranges):
            q = float(rng.uniform(*r))
            w['joint'][(b, 0)] = q
            desc['joint'][nm] = q
    # ray result: miss / pad hit / object hit
    mode = int(rng.integers(0, 3))
    if mode == 0:
        ray = [(-1, -1, 1.0, (0, 0, 0), (0, 0, 0))]
    elif mode == 1:
        ray = [(arm, int(rng.choice([18, 20])), float(rng.uniform(0.1, 0.9)), (0, 0, 0), (0, 0, 1))]
    else:
        ray = [(inst.objects[0] if inst.objects else 0, -1, float(rng.uniform(0.1, 0.9)), (0, 0, 0), (0, 0, 1))]
    w['ray'] = ray
    desc['ray'] = {'uid_is_arm': ray[0][0] == arm, 'link': ray[0][1], 'fraction': ray[0][2]}
    c.world = w
    return desc


def obs_to_json(o):
    out = {}
    for k, v in o.items():
        if v is None:
            out[k] = None
        elif isinstance(v, np.ndarray):
            out[k] = {'dtype': str(v.dtype), 'v': v.astype(np.float64).tolist()}
        elif isinstance(v, list):
            out[k] = {'dtype': 'list', 'v': [float(x) for x in v]}
        else:
            out[k] = {'dtype': type(v).__name__, 'v': v}
    return out


# ---------------------------------------------------------------------------------------------
def gen_registry_and_spaces():
    reg = list(fb.REGISTRY)
    spaces = {}
    for kind in KINDS:
        env = quiet(KINDS[kind])
        sp = {'action_low': env.action_space.low, 'action_high': env.action_space.high,
              'max_episode_steps': env._max_episode_steps, 'num_objects': env.num_objects,
              'num_goals': env.num_goals, 'play': env.play, 'use_orientation': env.use_orientation,
              'return_velocity': env.return_velocity, 'action_type': env.action_type,
              'arm_type': env.arm_type, 'sparse_rew_thresh': env.sparse_rew_thresh,
              'env_lower_bound': env.env_lower_bound, 'env_upper_bound': env.env_upper_bound,
              'goal_lower_bound': env.goal_lower_bound, 'goal_upper_bound': env.goal_upper_bound,
              'obj_lower_bound': env.obj_lower_bound, 'obj_upper_bound': env.obj_upper_bound,
              'observation_space': {k: {'low': v.low, 'high': v.high} for k, v in env.observation_space.spaces.items()}}
        spaces[kind] = sp
    names = urdf_tree.joint_names(ur5_tree)
    dump('registry.json', {'registry': reg, 'ids_in_scope': IDS})
    dump('spaces.json', spaces)
    # NB cell 2 of the reference notebook (testing_bullet_ik.ipynb) prints index -> joint name for ur5e2.urdf.
    nb = json.load(open(os.path.join(ENVS, 'ur_e_description', 'testing_bullet_ik.ipynb')))
    table = None
    for cell in nb['cells']:
        for out in cell.get('outputs', []):
            text = ''.join(out.get('text', []))
            if 'shoulder_pan_joint' in text and 'grasptarget_hand' in text:
                table = text
    dump('ur5_joint_index_table.json', {'notebook_stdout': table, 'dfs_names': names,
                                        'dfs_types': fb.UR5_JOINT_TYPES,
                                        'panda_dfs_names': urdf_tree.joint_names(panda_tree),
                                        'panda_dfs_types': fb.PANDA_JOINT_TYPES})


def gen_scenes():
    out = {}
    for name, fn, extra in (('complex_scene', scenes.complex_scene, (1,)), ('push_scene', scenes.push_scene, ()),
                            ('default_scene', scenes.default_scene, ())):
        c = fb.FakeClient()
        ret = fn(c, [0, 0, 0], c.URDF_ENABLE_CACHED_GRAPHICS_SHAPES, np.array([-1, -1, -0.2]), np.array([1, 1, 1]), *extra)
        out[name] = {'log': c.log, 'ret': ret}
    # instance.__init__ calls that follow the scene (arm load, gear constraint, damping)
    for kind in KINDS:
        env, main, shadow = new_env(kind)
        out['instance_init_' + kind] = {
            'log': [e for e in main.log if e['fn'] in ('loadURDF', 'createConstraint', 'changeConstraint',
                                                       'setPhysicsEngineParameter', 'setTimeStep', 'setGravity')],
            'n_bodies': main.n_bodies, 'arm': env.instance.arm, 'objects': env.instance.objects,
            'ee_index': env.instance.endEffectorIndex, 'rest': env.instance.restJointPositions,
            'base_pos': env.instance.init_arm_base_pos, 'base_orn': env.instance.init_arm_base_orn,
            'shadow_log': shadow.log if shadow else None,
            'drawer_defaults': env.instance.drawer['defaults'] if kind == 'U' else None,
        }
    dump('scenes.json', out)


def gen_calc_state(seed=11, n_seq=3, seq_len=5):
    rng = np.random.default_rng(seed)
    out = {}
    for kind in KINDS:
        seqs = []
        for s in range(n_seq):
            env, c, _ = new_env(kind)
            ng = 11 if kind == 'U' else 3
            env.instance.goal = rng.uniform(-0.3, 0.3, ng)
            steps = []
            prev = None
            for t in range(seq_len):
                desc = fill_world(kind, env, c, rng)
                if kind == 'U' and prev is not None and t % 2 == 1:
                    # exercise quaternion_safe_the_obs: same state as last step with negated quaternions
                    ee, blk = env.instance.endEffectorIndex, env.instance.objects[0]
                    flip_ee, flip_blk = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                    pe = prev['link'][str(ee)]['orn']
                    pb = prev['base']['block0']['orn']
                    c.world['link'][(env.instance.arm, ee)]['orn'] = [(-v if flip_ee else v) for v in pe]
                    c.world['base'][blk]['orn'] = [(-v if flip_blk else v) for v in pb]
                    desc['link'][str(ee)]['orn'] = c.world['link'][(env.instance.arm, ee)]['orn']
                    desc['base']['block0']['orn'] = c.world['base'][blk]['orn']
                obs = quiet(env.instance.calc_state)
                r = env.compute_reward(obs['achieved_goal'], obs['desired_goal'])
                steps.append({'world': desc, 'obs': obs_to_json(obs), 'reward': float(r)})
                prev = desc
            seqs.append({'goal': env.instance.goal, 'steps': steps})
        out[kind] = seqs
    dump('calc_state.json', out)


def gen_step(seed=23, n_cases=6):
    rng = np.random.default_rng(seed)
    out = {}
    for kind in KINDS:
        cases = []
        for k in range(n_cases):
            env, c, shadow = new_env(kind)
            ng = 11 if kind == 'U' else 3
            env.instance.goal = rng.uniform(-0.3, 0.3, ng)
            desc = fill_world(kind, env, c, rng)
            action = rng.uniform(-1.0, 1.0, 7) * np.array([0.5, 0.5, 0.5, 3.3, 3.3, 3.3, 1.4])
            if k % 3 == 2:
                action = rng.uniform(-8, 8, 7)      # exercise the action-space clip
            arm = env.instance.arm
            ndof_ret = 12 if kind != 'P' else 9
            cur = np.array([c.world['joint'][(arm, j)] for j in range(ndof_ret if kind == 'P' else 6)])
            n_ik = 4 if kind != 'P' else 1
            iks = []
            for i in range(n_ik):
                sol = rng.uniform(-3.5, 3.5, ndof_ret)
                if k % 2 == 0:
                    sol[:len(cur)] = cur[:len(sol[:len(cur)])] + rng.uniform(-0.3, 0.3, len(cur))
                iks.append(sol.tolist())
            ik_client = shadow if kind != 'P' else c
            ik_client.ik_queue = [list(v) for v in iks]
            c.clear_log()
            if shadow:
                shadow.clear_log()
            obs, r, done, info = quiet(env.step, action)
            cases.append({'goal': env.instance.goal, 'world': desc, 'action': action, 'ik_returns': iks,
                          'main_log': [e for e in c.log if e['fn'] != 'rayTest'],
                          'shadow_log': shadow.log if shadow else None,
                          'obs': obs_to_json(obs), 'reward': float(r), 'done': bool(done),
                          'is_success': info['is_success'], 'target_poses': info['target_poses']})
        out[kind] = cases
    dump('step.json', out)


def gen_rewards(seed=31):
    rng = np.random.default_rng(seed)
    out = {'success_func': [], 'sparse': {}, 'dial': [], 'euler': []}
    lim = np.array([0.05, 0.05, 0.05, 0.0, 0.0, 0.0, 0.0, 0.025, 0.04, 0.01, 0.3])
    for k in range(40):
        g = np.concatenate([rng.uniform(-0.2, 0.3, 3), unit_quat(rng), rng.uniform(-0.05, 0.05, 1),
                            rng.uniform(-0.1, 0.1, 1), rng.uniform(0, 0.03, 1), rng.uniform(0, 0.9, 1)])
        ag = g.copy()
        mode = k % 8
        scale = 0.9 if k % 2 == 0 else 1.15          # just inside / just outside each threshold
        if mode < 6:
            idx = [0, 2, 7, 8, 9, 10][mode]
            ag[idx] += lim[idx] * scale * (1 if k % 4 < 2 else -1)
        elif mode == 6:
            e = np.array(fb.euler_from_quat(g[3:7]))
            e[k % 3] += (np.pi / 4) * scale
            ag[3:7] = fb.quat_from_euler(e)
        else:
            ag[3:7] = -ag[3:7] if k % 16 == 7 else unit_quat(rng)
        out['success_func'].append({'ag': ag, 'g': g, 'r': int(playRewardFunc.success_func(ag, g))})
    for kind in ('R', 'P'):
        env = quiet(KINDS[kind])
        rows = []
        for k in range(12):
            dg = rng.uniform(-0.2, 0.2, 3)
            ag = dg + rng.normal(size=3) * (0.02 if k % 2 == 0 else 0.06)
            rows.append({'ag': ag, 'dg': dg, 'r': float(env.compute_reward(ag, dg))})
        AG = rng.uniform(-0.2, 0.2, (5, 3))
        DG = AG + rng.normal(size=(5, 3)) * 0.04
        out['sparse'][kind] = {'single': rows, 'batch': {'ag': AG, 'dg': DG, 'r': env.compute_reward(AG, DG)}}
    for x in list(np.linspace(-7, 7, 29)) + [1.0, 3.0, 0.0, 2.0, -0.5]:
        out['dial'].append({'x': float(x), 'y': float(scenes.dial_to_0_1_range(x))})
    dump('rewards.json', out)


def gen_reset(seed=47):
    """reset(): which random draws are made, in what order, and where they land."""
    out = {}
    for kind in KINDS:
        cases = []
        for k in range(3):
            env, c, shadow = new_env(kind)
            rng = np.random.default_rng(seed + k)
            fill_world(kind, env, c, rng)
            ndof_ret = 12 if kind != 'P' else 9
            c.ik_queue = [rng.uniform(-2, 2, ndof_ret).tolist() for _ in range(8)]
            draws = []
            rs = np.random.RandomState(1000 + k)
            orig = (np.random.uniform, np.random.choice, np.random.random)

            def uniform(lo, hi):
                lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
                u = rs.random_sample(lo.shape)
                draws.append({'fn': 'uniform', 'u': u.tolist()})
                return lo + (hi - lo) * u

            def choice(n):
                u = rs.random_sample()
                draws.append({'fn': 'choice', 'n': int(n), 'u': float(u)})
                return int(u * n)

            def random():
                u = rs.random_sample()
                draws.append({'fn': 'random', 'u': float(u)})
                return u

            np.random.uniform, np.random.choice, np.random.random = uniform, choice, random
            try:
                c.clear_log()
                n_ik_before = len(c.ik_queue)
                obs = quiet(env.reset)
            finally:
                np.random.uniform, np.random.choice, np.random.random = orig
            log = []
            n_step = 0
            for e in c.log:
                if e['fn'] == 'stepSimulation':
                    n_step += 1
                    continue
                if e['fn'] in ('changeDynamics', 'rayTest'):
                    continue
                if n_step:
                    log.append({'fn': 'stepSimulation_x', 'n': n_step})
                    n_step = 0
                log.append(e)
            cases.append({'draws': draws, 'log': log, 'n_resets': n_ik_before - len(c.ik_queue),
                          'goal': env.instance.goal, 'obs': obs_to_json(obs)})
        out[kind] = cases
    dump('reset.json', out)


def gen_spaces_more():
    """declared spaces and attributes of further ids that reuse a model in scope: pandaPush-v0 = pandaPick's arm and scene
    with other ranges (envList.py:12-16)"""
    out = {}
    for gid, cls in (('pandaPush-v0', pandaPush),):
        env = quiet(cls)
        out[gid] = {'action_low': env.action_space.low, 'action_high': env.action_space.high,
                    'max_episode_steps': env._max_episode_steps, 'num_objects': env.num_objects, 'num_goals': env.num_goals,
                    'play': env.play, 'use_orientation': env.use_orientation, 'return_velocity': env.return_velocity,
                    'action_type': env.action_type, 'arm_type': env.arm_type, 'sparse_rew_thresh': env.sparse_rew_thresh,
                    'env_lower_bound': env.env_lower_bound, 'env_upper_bound': env.env_upper_bound,
                    'goal_lower_bound': env.goal_lower_bound, 'goal_upper_bound': env.goal_upper_bound,
                    'obj_lower_bound': env.obj_lower_bound, 'obj_upper_bound': env.obj_upper_bound,
                    'observation_space': {k: {'low': v.low, 'high': v.high} for k, v in env.observation_space.spaces.items()}}
    dump('spaces_more.json', out)


def gen_reset_to(seed=83):
    """reset(o): objects and arm placed from an observation vector (environments.py:519-603, obs given) - which indices of o
    are read (the object block starts at index 11 / 7, quirk), the IK target, no settling, and the draws of the goal."""
    out = {}
    for kind in KINDS:
        cases = []
        for k in range(3):
            env, c, shadow = new_env(kind)
            rng = np.random.default_rng(seed + k)
            fill_world(kind, env, c, rng)
            ndof_ret = 12 if kind != 'P' else 9
            c.ik_queue = [rng.uniform(-2, 2, ndof_ret).tolist() for _ in range(8)]
            o = rng.uniform(-0.4, 0.4, 24)
            draws = []
            rs = np.random.RandomState(2000 + k)
            orig = (np.random.uniform, np.random.choice, np.random.random)

            def uniform(lo, hi):
                lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
                u = rs.random_sample(lo.shape)
                draws.append({'fn': 'uniform', 'u': u.tolist()})
                return lo + (hi - lo) * u

            def choice(n):
                u = rs.random_sample()
                draws.append({'fn': 'choice', 'n': int(n), 'u': float(u)})
                return int(u * n)

            def random():
                u = rs.random_sample()
                draws.append({'fn': 'random', 'u': float(u)})
                return u

            np.random.uniform, np.random.choice, np.random.random = uniform, choice, random
            try:
                c.clear_log()
                n_ik_before = len(c.ik_queue)
                obs = quiet(env.reset, o)
            finally:
                np.random.uniform, np.random.choice, np.random.random = orig
            log = [e for e in c.log if e['fn'] not in ('changeDynamics', 'rayTest')]
            cases.append({'o': o, 'draws': draws, 'log': log, 'n_resets': n_ik_before - len(c.ik_queue),
                          'n_objects': len(env.instance.objects), 'goal': env.instance.goal, 'obs': obs_to_json(obs)})
        out[kind] = cases
    dump('reset_to.json', out)


def gen_step_family(seed=61, n_cases=6):
    """step() of the other UR5 one-object play ids: action-space bounds, the IK call arguments each action type produces
    from the action and the measured EE pose / joints, joint clamps and motor commands."""
    rng = np.random.default_rng(seed)
    out = {}
    for gid, cls in FAMILY.items():
        env, c, shadow = new_env('U', cls)
        info = {'action_type': env.action_type, 'action_low': env.action_space.low, 'action_high': env.action_space.high,
                'play': env.play, 'use_orientation': env.use_orientation, 'return_velocity': env.return_velocity,
                'num_objects': env.num_objects, 'max_episode_steps': env._max_episode_steps,
                'observation_space': {k: {'low': v.low, 'high': v.high} for k, v in env.observation_space.spaces.items()}}
        na = len(env.action_space.high)
        cases = []
        for k in range(n_cases):
            env, c, shadow = new_env('U', cls)
            env.instance.goal = rng.uniform(-0.3, 0.3, 11)
            desc = fill_world('U', env, c, rng)
            action = rng.uniform(-1.0, 1.0, na) * 0.9
            if k % 3 == 2:
                action = rng.uniform(-8, 8, na)      # exercise the action-space clip
            arm = env.instance.arm
            cur = np.array([c.world['joint'][(arm, j)] for j in range(6)])
            iks = []
            for i in range(4):
                sol = rng.uniform(-3.5, 3.5, 12)
                if k % 2 == 0:
                    sol[:6] = cur + rng.uniform(-0.3, 0.3, 6)
                iks.append(sol.tolist())
            shadow.ik_queue = [list(v) for v in iks]
            c.clear_log()
            shadow.clear_log()
            obs, r, done, inf = quiet(env.step, action)
            cases.append({'goal': env.instance.goal, 'world': desc, 'action': action, 'ik_returns': iks,
                          'main_log': [e for e in c.log if e['fn'] != 'rayTest'], 'shadow_log': shadow.log,
                          'obs': obs_to_json(obs), 'reward': float(r), 'done': bool(done),
                          'is_success': inf['is_success'], 'target_poses': inf['target_poses']})
        out[gid] = {'info': info, 'cases': cases}
    dump('step_family.json', out)


# the Panda ids beyond pandaPick / pandaPush: Panda + default_scene and Panda + complex_scene
PANDA_IDS = {'pandaReach-v0': pandaReach, 'pandaReach2D-v0': pandaReach2D, 'pandaPlay1Obj-v0': pandaPlay1Obj,
             'pandaPlayRel1Obj-v0': pandaPlayRel1Obj, 'pandaPlayRelJoints1Obj-v0': pandaPlayRelJoints1Obj,
             'pandaPlayAbsJoints1Obj-v0': pandaPlayAbsJoints1Obj, 'pandaPlayAbsRPY1Obj-v0': pandaPlayAbsRPY1Obj,
             'pandaPlayRelRPY1Obj-v0': pandaPlayRelRPY1Obj}


def gen_panda_ids(seed=97, n_cases=5):
    """Per id: declared spaces / attributes / ranges, the scene and arm it builds (call log), and step() cases: the IK call
    the action type produces on the live arm (maxNumIterations=200), joint clamps, motor commands, and the observation
    assembled from the fake world read-back."""
    rng = np.random.default_rng(seed)
    out = {}
    for gid, cls in PANDA_IDS.items():
        env, c, shadow = new_env(None, cls)
        assert shadow is None
        inst = env.instance
        info = {'action_type': env.action_type, 'action_low': env.action_space.low, 'action_high': env.action_space.high,
                'play': env.play, 'use_orientation': env.use_orientation, 'return_velocity': env.return_velocity,
                'num_objects': env.num_objects, 'num_goals': env.num_goals, 'max_episode_steps': env._max_episode_steps,
                'arm_type': env.arm_type, 'sparse_rew_thresh': env.sparse_rew_thresh,
                'env_lower_bound': env.env_lower_bound, 'env_upper_bound': env.env_upper_bound,
                'goal_lower_bound': env.goal_lower_bound, 'goal_upper_bound': env.goal_upper_bound,
                'obj_lower_bound': env.obj_lower_bound, 'obj_upper_bound': env.obj_upper_bound,
                'observation_space': {k: {'low': v.low, 'high': v.high} for k, v in env.observation_space.spaces.items()},
                'base_pos': inst.init_arm_base_pos, 'base_orn': inst.init_arm_base_orn, 'ee_index': inst.endEffectorIndex,
                'rest': inst.restJointPositions, 'num_dofs': inst.numDofs,
                'scene_fns': [e['fn'] for e in c.log if e['fn'] in ('loadURDF', 'createMultiBody', 'createConstraint')]}
        na = len(env.action_space.high)
        ng = 11 if env.play else 3
        cases = []
        for k in range(n_cases):
            env, c, shadow = new_env(None, cls)
            env.instance.goal = rng.uniform(-0.3, 0.3, ng)
            desc = fill_world(None, env, c, rng)
            action = rng.uniform(-1.0, 1.0, na) * 0.9
            if k % 3 == 2:
                action = rng.uniform(-8, 8, na)      # exercise the action-space clip
            arm = env.instance.arm
            cur = np.array([c.world['joint'][(arm, j)] for j in range(7)])
            sol = rng.uniform(-3.5, 3.5, 9)
            if k % 2 == 0:
                sol[:7] = cur + rng.uniform(-0.3, 0.3, 7)
            c.ik_queue = [sol.tolist()]
            c.clear_log()
            obs, r, done, inf = quiet(env.step, action)
            cases.append({'goal': env.instance.goal, 'world': desc, 'action': action, 'ik_returns': [sol.tolist()],
                          'main_log': [e for e in c.log if e['fn'] != 'rayTest'],
                          'obs': obs_to_json(obs), 'reward': float(r), 'done': bool(done),
                          'is_success': inf['is_success'], 'target_poses': inf['target_poses']})
        out[gid] = {'info': info, 'cases': cases}
    dump('panda_ids.json', out)


# the two-object play ids: Panda + complex_scene with two blocks
TWO_OBJ_IDS = {'pandaPlay-v0': pandaPlay, 'pandaPlayJoints-v0': pandaPlayRelJoints}


def gen_two_object_ids(seed=131, n_cases=5, n_seq=3, seq_len=5):
    """pandaPlay-v0 / pandaPlayJoints-v0: the scene with two blocks (call log), declared spaces / attributes, step() cases
    (IK call, clamps, motor commands, observation of both objects), calc_state sequences that exercise the quaternion sign
    memory (incl. its (19, 23) index pair for the second object), and reset(): the draws and where they land."""
    rng = np.random.default_rng(seed)
    c0 = fb.FakeClient()
    ret = scenes.complex_scene(c0, [0, 0, 0], c0.URDF_ENABLE_CACHED_GRAPHICS_SHAPES, np.array([-1, -1, -0.2]), np.array([1, 1, 1]), 2)
    out = {'complex_scene_2obj': {'log': c0.log, 'ret': ret}}
    for gid, cls in TWO_OBJ_IDS.items():
        env, c, shadow = new_env(None, cls)
        assert shadow is None
        inst = env.instance
        info = {'action_type': env.action_type, 'action_low': env.action_space.low, 'action_high': env.action_space.high,
                'play': env.play, 'use_orientation': env.use_orientation, 'return_velocity': env.return_velocity,
                'num_objects': env.num_objects, 'num_goals': env.num_goals, 'max_episode_steps': env._max_episode_steps,
                'arm_type': env.arm_type, 'sparse_rew_thresh': env.sparse_rew_thresh,
                'env_lower_bound': env.env_lower_bound, 'env_upper_bound': env.env_upper_bound,
                'goal_lower_bound': env.goal_lower_bound, 'goal_upper_bound': env.goal_upper_bound,
                'obj_lower_bound': env.obj_lower_bound, 'obj_upper_bound': env.obj_upper_bound,
                'observation_space': {k: {'low': v.low, 'high': v.high} for k, v in env.observation_space.spaces.items()},
                'base_pos': inst.init_arm_base_pos, 'base_orn': inst.init_arm_base_orn, 'ee_index': inst.endEffectorIndex,
                'rest': inst.restJointPositions, 'num_dofs': inst.numDofs, 'objects': inst.objects, 'arm': inst.arm,
                'drawer_defaults': inst.drawer['defaults']}
        na = len(env.action_space.high)
        cases = []
        for k in range(n_cases):
            env, c, shadow = new_env(None, cls)
            env.instance.goal = rng.uniform(-0.3, 0.3, 18)
            desc = fill_world(None, env, c, rng)
            action = rng.uniform(-1.0, 1.0, na) * 0.9
            if k % 3 == 2:
                action = rng.uniform(-8, 8, na)
            arm = env.instance.arm
            cur = np.array([c.world['joint'][(arm, j)] for j in range(7)])
            sol = rng.uniform(-3.5, 3.5, 9)
            if k % 2 == 0:
                sol[:7] = cur + rng.uniform(-0.3, 0.3, 7)
            c.ik_queue = [sol.tolist()]
            c.clear_log()
            obs, r, done, inf = quiet(env.step, action)
            cases.append({'goal': env.instance.goal, 'world': desc, 'action': action, 'ik_returns': [sol.tolist()],
                          'main_log': [e for e in c.log if e['fn'] != 'rayTest'],
                          'obs': obs_to_json(obs), 'reward': float(r), 'done': bool(done),
                          'is_success': inf['is_success'], 'target_poses': inf['target_poses']})
        seqs = []
        for s_ in range(n_seq):
            env, c, _ = new_env(None, cls)
            env.instance.goal = rng.uniform(-0.3, 0.3, 18)
            steps, prev = [], None
            for t in range(seq_len):
                desc = fill_world(None, env, c, rng)
                if prev is not None and t % 2 == 1:     # same state as last step with some quaternions negated
                    ee = env.instance.endEffectorIndex
                    arm = env.instance.arm
                    if rng.integers(0, 2):
                        c.world['link'][(arm, ee)]['orn'] = [-v for v in prev['link'][str(ee)]['orn']]
                    else:
                        c.world['link'][(arm, ee)]['orn'] = list(prev['link'][str(ee)]['orn'])
                    desc['link'][str(ee)]['orn'] = c.world['link'][(arm, ee)]['orn']
                    for b, uid in enumerate(env.instance.objects):
                        pb = prev['base']['block%d' % b]
                        sgn = -1.0 if rng.integers(0, 2) else 1.0
                        c.world['base'][uid]['orn'] = [sgn * v for v in pb['orn']]
                        c.world['base'][uid]['pos'] = list(pb['pos'])
                        desc['base']['block%d' % b]['orn'] = c.world['base'][uid]['orn']
                        desc['base']['block%d' % b]['pos'] = c.world['base'][uid]['pos']
                    if rng.integers(0, 2):               # the (19, 23) pair also covers the drawer entry: flip it along
                        d = env.instance.drawer['drawer']
                        c.world['base'][d]['pos'] = [prev['base']['drawer']['pos'][0], -prev['base']['drawer']['pos'][1], prev['base']['drawer']['pos'][2]]
                        desc['base']['drawer']['pos'] = c.world['base'][d]['pos']
                obs = quiet(env.instance.calc_state)
                r = env.compute_reward(obs['achieved_goal'], obs['desired_goal'])
                steps.append({'world': desc, 'obs': obs_to_json(obs), 'reward': float(r)})
                prev = desc
            seqs.append({'goal': env.instance.goal, 'steps': steps})
        resets = []
        for k in range(3):
            env, c, shadow = new_env(None, cls)
            rr = np.random.default_rng(seed + 10 + k)
            fill_world(None, env, c, rr)
            c.ik_queue = [rr.uniform(-2, 2, 9).tolist() for _ in range(64)]
            draws = []
            rs = np.random.RandomState(2000 + k)
            orig = (np.random.uniform, np.random.choice, np.random.random)

            def uniform(lo, hi):
                lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
                u = rs.random_sample(lo.shape)
                draws.append({'fn': 'uniform', 'u': u.tolist()})
                return lo + (hi - lo) * u

            def choice(n):
                u = rs.random_sample()
                draws.append({'fn': 'choice', 'n': int(n), 'u': float(u)})
                return int(u * n)

            def random():
                u = rs.random_sample()
                draws.append({'fn': 'random', 'u': float(u)})
                return u

            np.random.uniform, np.random.choice, np.random.random = uniform, choice, random
            try:
                c.clear_log()
                n_ik_before = len(c.ik_queue)
                obs = quiet(env.reset)
            finally:
                np.random.uniform, np.random.choice, np.random.random = orig
            log, n_step = [], 0
            for e in c.log:
                if e['fn'] == 'stepSimulation':
                    n_step += 1
                    continue
                if e['fn'] in ('changeDynamics', 'rayTest'):
                    continue
                if n_step:
                    log.append({'fn': 'stepSimulation_x', 'n': n_step})
                    n_step = 0
                log.append(e)
            resets.append({'draws': draws, 'log': log, 'n_resets': n_ik_before - len(c.ik_queue),
                           'goal': env.instance.goal, 'obs': obs_to_json(obs)})
        out[gid] = {'info': info, 'cases': cases, 'calc_state': seqs, 'resets': resets}
    dump('two_object_ids.json', out)


if __name__ == '__main__':
    gen_two_object_ids()
    gen_panda_ids()
    gen_spaces_more()
    gen_reset_to()
    gen_step_family()
    gen_registry_and_spaces()
    gen_scenes()
    gen_calc_state()
    gen_step()
    gen_rewards()
    gen_reset()
