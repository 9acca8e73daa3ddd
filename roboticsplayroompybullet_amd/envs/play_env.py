"""Single-env adapters with the reference's gym surface (environments.py:58-314, envList.py:18-22, 89-99).

`playEnv` keeps the reference's method names, argument meaning, observation-dict keys, shapes and dtypes; underneath
it is a VecPlayEnv of one environment on the GPU.  Like the reference, nothing touches the simulator until the first
reset() ("activate_physics_client", environments.py:175-177).  The constructor kwargs that reach the simulation (ranges,
reward threshold, sparse / dense reward, action type) are handed to the library through rp_config; kwargs that would change
the observation layout of the registered id (num_objects, use_orientation, return_velocity, play, arm_type) raise when they
differ from the id's.  Out of scope and raising NotImplementedError: GUI / VR clients.
"""
import numpy as np

from .. import spaces

F32_KEYS = ('obs_quat', 'achieved_goal', 'desired_goal', 'controllable_achieved_goal', 'full_positional_state')


class _InstanceShim:
    """The attributes in-tree callers reach through env.instance (interactive.py:8,19,46,53; environments.py:184,210)."""

    def __init__(self, env):
        self._env = env
        self.arm_type = env.arm_type
        self.default_arm_orn_RPY = [0, 0, 0]                                   # environments.py:357,365
        self.restJointPositions = ([-0.6, 0.437, 0.217, -2.09, 1.1, 1.4, 1.3, 0.0, 0.0, 0.0] if env.arm_type == 'Panda' else
                                   [-1.50189075, -1.6291067, -1.87020409, -1.21324173, 1.57003561, 0.06970189])
        self.endEffectorIndex = 11 if env.arm_type == 'Panda' else 7
        self.record_images = False

    def calc_state(self):
        return self._env._to_reference_obs(self._env._vec.calc_state())

    def calc_actor_state(self):
        o = self._env._vec.calc_state()
        g = lambda k: o[k][0].cpu().numpy().astype(np.float64)   # noqa: E731
        q = g('obs_quat')
        ee_orn = q[3:7] if self._env.use_orientation else None
        return {'pos': g('controllable_achieved_goal')[:3], 'orn': ee_orn, 'pos_vel': g('velocity')[:3], 'orn_vel': g('velocity')[3:],
                'gripper': [g('controllable_achieved_goal')[3]], 'joints': list(g('joints')),
                'proprioception': int(o['gripper_proprioception'][0])}


class playEnv:
    metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 60}
    ENV_ID = None

    def __init__(self, num_objects=0, env_range_low=(-0.18, -0.18, -0.05), env_range_high=(0.18, 0.18, 0.15),
                 goal_range_low=(-0.18, -0.18, -0.05), goal_range_high=(0.18, 0.18, 0.05), obj_lower_bound=(-0.18, -0.18, -0.05),
                 obj_upper_bound=(-0.18, -0.18, -0.05), sparse=True, use_orientation=False, sparse_rew_thresh=0.05,
                 fixed_gripper=False, return_velocity=True, max_episode_steps=250, play=False, action_type='absolute_rpy',
                 show_goal=True, arm_type='Panda', device=0, seed=None, contact_margin=None, persistent_manifolds=True, hull_gjk=True, speculative_limits=False, hull_epa=None):
        # seed=None: like the reference, which draws from the global np.random (environments.py:496, 530, 579), every new env gets
        # its own episode stream and np.random.seed(k) makes it repeatable
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        self.sparse, self._contact_margin = bool(sparse), contact_margin
        self._model_opts = dict(persistent_manifolds=bool(persistent_manifolds), hull_gjk=bool(hull_gjk), speculative_limits=bool(speculative_limits), hull_epa=hull_epa)      # the library's contact-model switches (rp_config.flags)
        self.timeStep = 1.0 / 300
        self.render_scene = False
        self.physics_client_active = 0
        self.num_objects, self.use_orientation, self.return_velocity = num_objects, use_orientation, return_velocity
        self.fixed_gripper, self.sparse_reward_threshold, self.sparse_rew_thresh = fixed_gripper, sparse_rew_thresh, sparse_rew_thresh
        self.num_goals = max(num_objects, 1)
        self.play, self.action_type, self.show_goal, self.arm_type = play, action_type, show_goal, arm_type
        self._max_episode_steps = max_episode_steps
        self._device, self._seed = device, seed
        ndof = 6 if arm_type == 'UR5' else 7                                    # environments.py:88-113
        high = {'absolute_rpy': [6] * 6 + [1], 'relative_rpy': [1] * 7, 'absolute_quat': [1] * 8, 'relative_quat': [1] * 8,
                'absolute_joints': [6] * ndof + [1], 'relative_joints': [1] * (ndof + 1)}.get(action_type)
        if high is None:
            raise NotImplementedError(action_type)
        if action_type == 'absolute_quat' and not use_orientation:
            high = [1, 1, 1, 1]              # the reference declares a 4-wide space here (environments.py:90-95) and then cannot step it: absolute_quat_step asserts 8 (937)
        high = np.array(high)
        self.action_space = spaces.Box(-high, high)
        # declared spaces, reproduced as-is including the arm_lower_obs_lim typo (environments.py:120-166, quirk F11)
        eu, el = np.array(env_range_high, dtype=float), np.array(env_range_low, dtype=float)
        self.env_upper_bound, self.env_lower_bound = eu, el
        self.goal_upper_bound, self.goal_lower_bound = np.array(goal_range_high, dtype=float), np.array(goal_range_low, dtype=float)
        self.obj_lower_bound, self.obj_upper_bound = list(obj_lower_bound), list(obj_upper_bound)
        if use_orientation:
            self.arm_upper_lim = np.concatenate([eu, np.array([1, 1, 1, 1, 0.04])])
            self.arm_lower_lim = np.concatenate([el, -np.array([1, 1, 1, 1, 0.0])])
            arm_upper_obs_lim = np.concatenate([eu, np.array([1, 1, 1, 1, 1, 1, 1, 0.04])])
            arm_lower_obs_lim = np.concatenate([eu, -np.array([1, 1, 1, 1, 1, 1, 1, 0.0])])
            obj_upper_lim = np.concatenate([self.obj_upper_bound, np.ones(7)])
            obj_lower_lim = np.concatenate([self.obj_lower_bound, -np.ones(7)])
            obj_upper_positional_lim = np.concatenate([eu, np.ones(4)])
            obj_lower_positional_lim = np.concatenate([el, -np.ones(4)])
        else:
            self.arm_upper_lim = np.concatenate([eu, np.array([0.04])])
            self.arm_lower_lim = np.concatenate([el, -np.array([0.0])])
            arm_upper_obs_lim = np.concatenate([eu, np.array([1, 1, 1, 0.04])])
            arm_lower_obs_lim = np.concatenate([eu, -np.array([1, 1, 1, 0.0])])
            obj_upper_lim = np.concatenate([self.obj_upper_bound, np.ones(3)])
            obj_lower_lim = np.concatenate([self.obj_lower_bound, -np.ones(3)])
            obj_upper_positional_lim, obj_lower_positional_lim = eu, el
        cat = np.concatenate
        self.observation_space = spaces.Dict(dict(
            desired_goal=spaces.Box(cat([el] * self.num_goals), cat([eu] * self.num_goals)),
            achieved_goal=spaces.Box(cat([el] * self.num_goals), cat([eu] * self.num_goals)),
            observation=spaces.Box(cat([arm_lower_obs_lim] + [obj_lower_lim] * num_objects), cat([arm_upper_obs_lim] + [obj_upper_lim] * num_objects)),
            controllable_achieved_goal=spaces.Box(self.arm_lower_lim, self.arm_upper_lim),
            full_positional_state=spaces.Box(cat([self.arm_lower_lim] + [obj_lower_positional_lim] * num_objects),
                                             cat([self.arm_upper_lim] + [obj_upper_positional_lim] * num_objects))))
        self._vec = None
        self.instance = None

    # -- reference surface ------------------------------------------------------------------------------------------
    def activate_physics_client(self, vr=None):
        if vr is not None or self.render_scene:
            raise NotImplementedError('GUI / VR clients are out of scope (SURVEY.md §2.1)')
        from ..vec_env import VecPlayEnv
        self._vec = VecPlayEnv(self.ENV_ID, 1, device=self._device, seed=self._seed, action_type=self.action_type,
                               goal_range_low=self.goal_lower_bound, goal_range_high=self.goal_upper_bound,
                               obj_lower_bound=self.obj_lower_bound, obj_upper_bound=self.obj_upper_bound,
                               env_range_high=self.env_upper_bound, sparse_rew_thresh=self.sparse_rew_thresh, sparse=self.sparse,
                               contact_margin=self._contact_margin, **self._model_opts)
        # kwargs that define the layout belong to the registered id: refuse silently different envs
        d = self._vec.dims
        per_obj = 3 + (4 if self.use_orientation else 0) + (3 if self.return_velocity else 0)
        want_obs = 3 + (3 if self.return_velocity else 0) + (4 if self.use_orientation else 0) + 1 + self.num_objects * per_obj + (4 if self.play and self.num_objects else 0)
        want_tp = 6 if self.arm_type == 'UR5' else 7
        if d['obs_quat'] != want_obs or d['target_poses'] != want_tp:
            self._vec.close()
            self._vec = None
            raise NotImplementedError('%s is registered with another num_objects / use_orientation / return_velocity / play / arm_type; '
                                      'these kwargs cannot be overridden (the id fixes the baked scene and the observation layout)' % self.ENV_ID)
        self.instance = _InstanceShim(self)
        self._vec.record_images = self.instance.record_images = getattr(self, '_record_images', False)

    def reset(self, o=None, vr=None):
        if not self.physics_client_active:
            self.activate_physics_client(vr)
            self.physics_client_active = True
        if o is not None:                                      # environments.py:542-556, 575-590: place objects and arm from o
            import torch
            return self._to_reference_obs(self._vec.reset(o=torch.as_tensor(np.asarray(o, dtype=np.float32))[None]))
        return self._to_reference_obs(self._vec.reset())      # the "reset until not already solved" loop runs on device

    def reset_goal_pos(self, goal):
        import torch
        g = None if goal is None else torch.as_tensor(np.asarray(goal, dtype=np.float32))[None]
        self._vec.reset_goal_pos(g)

    def render(self, mode):
        if mode == 'human':
            self.render_scene = True
            return np.array([])
        if mode in ('rgb_array', 'playback'):      # environments.py:200-203: from now on calc_state fills obs['img']
            self._record_images = True
            if self._vec is not None:
                self._vec.record_images = True
                self.instance.record_images = True

    def step(self, action):
        import torch
        if self.action_type in ('absolute_quat', 'relative_quat'):
            assert len(action) == 8          # environments.py:937, 948 (an id without use_orientation declares 4 and fails here, like the reference)
        a = np.clip(action, self.action_space.low, self.action_space.high)        # environments.py:207 (again on device)
        obs, r, done, info = self._vec.step(torch.as_tensor(np.asarray(a, dtype=np.float32))[None])
        o = self._to_reference_obs(obs)
        reward = float(r[0])
        if not self.play:
            reward = float(np.float64(reward))
        return o, reward, False, {'is_success': int(info['is_success'][0]), 'target_poses': info['target_poses'][0].cpu().numpy().astype(np.float64)}

    def compute_reward(self, achieved_goal, desired_goal, info=None):
        import torch
        ag = torch.as_tensor(np.asarray(achieved_goal, dtype=np.float32))
        dg = torch.as_tensor(np.asarray(desired_goal, dtype=np.float32))
        r = self._vec.compute_reward(ag, dg).cpu().numpy().astype(np.float64)
        return float(r) if r.ndim == 0 else r

    def compute_reward_sparse(self, achieved_goal, desired_goal, info=None):
        """environments.py:278-304; stays the sparse formula when the env was built with sparse=False (the reference rebinds only compute_reward, 169-170)"""
        import torch
        ag = torch.as_tensor(np.asarray(achieved_goal, dtype=np.float32))
        dg = torch.as_tensor(np.asarray(desired_goal, dtype=np.float32))
        r = self._vec.compute_reward_sparse(ag, dg).cpu().numpy().astype(np.float64)
        return float(r) if r.ndim == 0 else r

    def calc_target_distance(self, achieved_goal, desired_goal):
        return float(np.linalg.norm(np.asarray(achieved_goal) - np.asarray(desired_goal)))

    def visualise_sub_goal(self, sub_goal, sub_goal_state='full_positional_state'):
        """environments.py:606-690: draw a sub-goal into the images as half-transparent ghosts: the objects (and for the play ids drawer / door / button / dial) of
        'achieved_goal' / 'full_positional_state', and the ghost ARM of 'full_positional_state' / 'controllable_achieved_goal' - for the Panda; like the reference
        (environments.py:629-630) the UR5 raises NotImplementedError.  (The reference's Panda branch refers to an attribute that is never set, `self.ghost_panda`,
        and cannot run as written; what it evidently means is implemented: reset_arm(ghost_arm, sub_goal, from_init=False) - rest pose, one default IK call towards
        the sub-goal's EE pose, joints [0:6] - environments.py:575-590.)

        Deviations from the reference's lines, all on the ghost arm (INTEGRATION.md section 4 lists them too):
          * UR5 ids with 'full_positional_state' / 'controllable_achieved_goal' raise NotImplementedError as upstream does (rounds 1 - 4 of this library drew the object and
            fixture ghosts and no arm for them: callers that relied on that get the upstream behaviour now; 'achieved_goal' still draws the objects);
          * the ghost's orientation is sub_goal[3:7] (the EE quaternion of a use_orientation state); the reference's reset_arm reads o[6:10] when return_velocity is set -
            no registered Panda id has both flags, so the two agree on every id that can be made;
          * the ghost's IK starts from the rest pose on EVERY call; the reference passes from_init=False, i.e. the second and later calls start from the previous ghost's
            joints.  Stateless on purpose: the ghost of a sub-goal does not depend on which ghosts were drawn before it."""
        import torch
        if self._vec is None:
            raise RuntimeError('visualise_sub_goal before the first reset(): the physics client is not active yet')
        g = np.asarray(sub_goal, dtype=np.float32)
        if sub_goal_state not in ('achieved_goal', 'full_positional_state', 'controllable_achieved_goal'):
            raise ValueError(sub_goal_state)
        arm = None
        if sub_goal_state in ('full_positional_state', 'controllable_achieved_goal'):
            if self.arm_type != 'Panda':
                raise NotImplementedError      # environments.py:629-630
            orn = g[3:7] if (sub_goal_state == 'full_positional_state' and self.use_orientation) else np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32)      # default_arm_orn = quaternion of RPY (0, 0, 0): environments.py:357-366
            arm = np.concatenate([g[0:3], orn, [0.0]]).astype(np.float32)
        if sub_goal_state == 'full_positional_state':
            g = g[(8 if self.use_orientation else 4):]
        self._vec.ghost_arm = torch.as_tensor(arm)[None] if arm is not None else None
        if sub_goal_state == 'controllable_achieved_goal' or self.num_objects == 0:
            self._vec.sub_goal = None
            return
        assert g.shape == (self._vec.dims['achieved_goal'],), g.shape
        self._vec.sub_goal = torch.as_tensor(g)[None]

    def delete_sub_goal(self):
        if self._vec is not None:
            self._vec.sub_goal = None
            self._vec.ghost_arm = None

    def close(self):
        if self._vec is not None:
            self._vec.close()

    # -- helpers ----------------------------------------------------------------------------------------------------
    def _to_reference_obs(self, o):
        """dtypes of environments.py:849-861: float32 for five keys, float64 observation/velocity, list joints, int flag."""
        out = {}
        for k in F32_KEYS:
            out[k] = o[k][0].cpu().numpy().astype(np.float32)
        out['joints'] = [float(v) for v in o['joints'][0].cpu().numpy()]
        out['velocity'] = o['velocity'][0].cpu().numpy().astype(np.float64)
        out['img'] = None if o.get('img') is None else o['img'][0].cpu().numpy()      # [200, 200, 3] uint8 (environments.py:841-845)
        out['observation'] = o['observation'][0].cpu().numpy().astype(np.float64)
        out['gripper_proprioception'] = int(o['gripper_proprioception'][0])
        return out


class pandaPick(playEnv):                     # envList.py:18-22
    ENV_ID = 'pandaPick-v0'

    def __init__(self, num_objects=1, env_range_low=(-0.18, -0.18, -0.055), env_range_high=(0.18, 0.18, 0.2), goal_range_low=(-0.18, -0.18, 0.0),
                 goal_range_high=(0.18, 0.18, 0.1), use_orientation=False, **kw):
        super().__init__(num_objects=num_objects, env_range_low=env_range_low, env_range_high=env_range_high, goal_range_low=goal_range_low,
                         goal_range_high=goal_range_high, use_orientation=use_orientation, obj_lower_bound=goal_range_low,
                         obj_upper_bound=goal_range_high, **kw)


class pandaPush(playEnv):                     # envList.py:12-16
    ENV_ID = 'pandaPush-v0'

    def __init__(self, num_objects=1, env_range_low=(-0.18, -0.18, -0.055), env_range_high=(0.18, 0.18, -0.04), goal_range_low=(-0.1, -0.1, -0.06),
                 goal_range_high=(0.1, 0.1, -0.05), use_orientation=False, **kw):
        super().__init__(num_objects=num_objects, env_range_low=env_range_low, env_range_high=env_range_high, goal_range_low=goal_range_low,
                         goal_range_high=goal_range_high, use_orientation=use_orientation, obj_lower_bound=goal_range_low,
                         obj_upper_bound=goal_range_high, **kw)


class UR5Reach(playEnv):                      # envList.py:89-91
    ENV_ID = 'UR5Reach-v0'

    def __init__(self, num_objects=0, **kw):
        super().__init__(num_objects=num_objects, use_orientation=False, arm_type='UR5', **kw)


class UR5PlayAbsRPY1Obj(playEnv):             # envList.py:93-99
    ENV_ID = 'UR5PlayAbsRPY1Obj-v0'

    def __init__(self, num_objects=1, env_range_low=(-1.0, -1.0, -0.2), env_range_high=(1.0, 1.0, 1.0), goal_range_low=(-0.18, 0, 0.05),
                 goal_range_high=(0.18, 0.3, 0.1), use_orientation=True, **kw):
        super().__init__(num_objects=num_objects, env_range_low=env_range_low, env_range_high=env_range_high, goal_range_low=goal_range_low,
                         goal_range_high=goal_range_high, use_orientation=use_orientation, obj_lower_bound=[-0.18, 0, 0.05],
                         obj_upper_bound=[0.18, 0.3, 0.1], return_velocity=False, max_episode_steps=None, play=True,
                         action_type='absolute_rpy', show_goal=False, arm_type='UR5', **kw)


class pandaReach(playEnv):                    # envList.py:8-10
    ENV_ID = 'pandaReach-v0'

    def __init__(self, num_objects=0, **kw):
        super().__init__(num_objects=num_objects, use_orientation=False, **kw)


class pandaReach2D(playEnv):                  # envList.py:24-26
    ENV_ID = 'pandaReach2D-v0'

    def __init__(self, num_objects=0, env_range_low=(-0.18, -0.18, -0.07), env_range_high=(0.18, 0.18, 0.0), goal_range_low=(-0.18, -0.18, -0.06),
                 goal_range_high=(0.18, 0.18, -0.05), use_orientation=False, **kw):
        super().__init__(num_objects=num_objects, env_range_low=env_range_low, env_range_high=env_range_high, goal_range_low=goal_range_low,
                         goal_range_high=goal_range_high, use_orientation=use_orientation, **kw)


def _play_1obj(name, env_id, action_type, anchor, arm_type='UR5'):
    """a member of a one-object play family: UR5PlayAbsRPY1Obj's configuration with another action type and / or arm"""
    def __init__(self, num_objects=1, env_range_low=(-1.0, -1.0, -0.2), env_range_high=(1.0, 1.0, 1.0), goal_range_low=(-0.18, 0, 0.05),
                 goal_range_high=(0.18, 0.3, 0.1), use_orientation=True, **kw):
        playEnv.__init__(self, num_objects=num_objects, env_range_low=env_range_low, env_range_high=env_range_high,
                         goal_range_low=goal_range_low, goal_range_high=goal_range_high, use_orientation=use_orientation,
                         obj_lower_bound=[-0.18, 0, 0.05], obj_upper_bound=[0.18, 0.3, 0.1], return_velocity=False,
                         max_episode_steps=None, play=True, action_type=action_type, show_goal=False, arm_type=arm_type, **kw)
    return type(name, (playEnv,), {'ENV_ID': env_id, '__init__': __init__, '__doc__': anchor})


UR5PlayRelRPY1Obj = _play_1obj('UR5PlayRelRPY1Obj', 'UR5PlayRelRPY1Obj-v0', 'relative_rpy', 'envList.py:101-107')
UR5PlayRelJoints1Obj = _play_1obj('UR5PlayRelJoints1Obj', 'UR5PlayRelJoints1Obj-v0', 'relative_joints', 'envList.py:109-115')
UR5PlayAbsJoints1Obj = _play_1obj('UR5PlayAbsJoints1Obj', 'UR5PlayAbsJoints1Obj-v0', 'absolute_joints', 'envList.py:117-123')
UR5Play1Obj = _play_1obj('UR5Play1Obj', 'UR5Play1Obj-v0', 'absolute_quat', 'envList.py:126-132')
UR5PlayRel1Obj = _play_1obj('UR5PlayRel1Obj', 'UR5PlayRel1Obj-v0', 'relative_quat', 'envList.py:134-140')

# the Panda one-object play family (envList.py:43-88): same configuration, arm_type left at its default 'Panda'
pandaPlayRelJoints1Obj = _play_1obj('pandaPlayRelJoints1Obj', 'pandaPlayRelJoints1Obj-v0', 'relative_joints', 'envList.py:43-48', 'Panda')
pandaPlayAbsJoints1Obj = _play_1obj('pandaPlayAbsJoints1Obj', 'pandaPlayAbsJoints1Obj-v0', 'absolute_joints', 'envList.py:50-55', 'Panda')
pandaPlay1Obj = _play_1obj('pandaPlay1Obj', 'pandaPlay1Obj-v0', 'absolute_quat', 'envList.py:58-63', 'Panda')
pandaPlayRel1Obj = _play_1obj('pandaPlayRel1Obj', 'pandaPlayRel1Obj-v0', 'relative_quat', 'envList.py:65-70', 'Panda')
pandaPlayAbsRPY1Obj = _play_1obj('pandaPlayAbsRPY1Obj', 'pandaPlayAbsRPY1Obj-v0', 'absolute_rpy', 'envList.py:72-78', 'Panda')
pandaPlayRelRPY1Obj = _play_1obj('pandaPlayRelRPY1Obj', 'pandaPlayRelRPY1Obj-v0', 'relative_rpy', 'envList.py:80-86', 'Panda')


class pandaPlay(playEnv):                     # envList.py:28-33: two blocks, absolute_quat
    ENV_ID = 'pandaPlay-v0'

    def __init__(self, num_objects=2, env_range_low=(-1.0, -1.0, -0.4), env_range_high=(1.0, 1.0, 1.0), goal_range_low=(-0.18, 0, 0.05),
                 goal_range_high=(0.18, 0.3, 0.1), use_orientation=True, **kw):
        super().__init__(num_objects=num_objects, env_range_low=env_range_low, env_range_high=env_range_high, goal_range_low=goal_range_low,
                         goal_range_high=goal_range_high, use_orientation=use_orientation, obj_lower_bound=[-0.18, 0, 0.05],
                         obj_upper_bound=[0.18, 0.3, 0.1], return_velocity=False, max_episode_steps=None, play=True,
                         action_type='absolute_quat', show_goal=False, **kw)


class pandaPlayRelJoints(playEnv):            # envList.py:36-41 (registered as pandaPlayJoints-v0): two blocks, relative_joints
    ENV_ID = 'pandaPlayJoints-v0'

    def __init__(self, num_objects=2, env_range_low=(-1.0, -1.0, -0.2), env_range_high=(1.0, 1.0, 1.0), goal_range_low=(-0.18, 0, 0.05),
                 goal_range_high=(0.18, 0.3, 0.1), use_orientation=True, **kw):
        super().__init__(num_objects=num_objects, env_range_low=env_range_low, env_range_high=env_range_high, goal_range_low=goal_range_low,
                         goal_range_high=goal_range_high, use_orientation=use_orientation, obj_lower_bound=[-0.18, 0, 0.05],
                         obj_upper_bound=[0.18, 0.3, 0.1], return_velocity=False, max_episode_steps=None, play=True,
                         action_type='relative_joints', show_goal=False, **kw)
