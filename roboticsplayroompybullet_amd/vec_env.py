"""VecPlayEnv — N reference environments stepped at once on one MI355X through the C ABI.

Method names, argument meaning and dict keys mirror the reference's playEnv (environments.py:58-314); every value
is a [N, ...] torch tensor on the env's device.  State lives inside the HIP library; the tensors returned by
step()/reset() are owned by this object and overwritten by the next call (clone to keep).
"""
import ctypes as C

import torch

from . import _lib

OBS_KEYS = ('obs_quat', 'achieved_goal', 'desired_goal', 'controllable_achieved_goal', 'full_positional_state', 'joints',
            'velocity', 'observation')

# record layout of rp_get_state / rp_set_state (csrc/rp_device_model.h ST_*), floats per env = 128
STATE_LAYOUT = {'q': (0, 12), 'qd': (12, 24), 'free0': (24, 37), 'free1': (37, 50), 'jq': (50, 53), 'jqd': (53, 56),
                'motor_mode': (56, 68), 'motor_target': (68, 80), 'motor_maximp': (80, 92), 'goal': (92, 103),
                'last_ee_quat': (103, 107), 'last_block_quat': (107, 111), 'last_ag_quat': (111, 115), 'have_last': (115, 116)}

# the same record in the RP_WIDE build (two-object ids: 9 arm dofs, three free bodies, 18-wide goal)
WIDE_STATE_LAYOUT = {'q': (0, 9), 'qd': (9, 18), 'free0': (18, 31), 'free1': (31, 44), 'free2': (44, 57), 'jq': (57, 60), 'jqd': (60, 63),
                     'motor_mode': (63, 72), 'motor_target': (72, 81), 'motor_maximp': (81, 90), 'goal': (90, 108),
                     'last_ee_quat': (108, 112), 'last_block_quat': (112, 116), 'last_obs_19_23': (116, 120), 'last_ag_10_14': (120, 124),
                     'have_last': (124, 125)}


class VecPlayEnv:
    def __init__(self, env_id, num_envs, device=0, seed=0, env_offset=0, action_type=None, goal_range_low=None, goal_range_high=None,
                 obj_lower_bound=None, obj_upper_bound=None, env_range_high=None, sparse_rew_thresh=None, sparse=True,
                 contact_margin=None, persistent_manifolds=True, hull_gjk=True, speculative_limits=False, hull_epa=None):
        """The keyword arguments after env_offset are the constructor kwargs of the reference's env classes that reach the
        simulation (envList.py -> environments.py:64-67); None keeps what the id registers.  contact_margin: rp_config."""
        if env_id not in _lib.ENV_KINDS:
            raise NotImplementedError('env id %r is outside the hot-path scope (SURVEY.md §8)' % (env_id,))
        if not torch.cuda.is_available():
            raise RuntimeError('VecPlayEnv needs a ROCm GPU: the hot path is HIP-only, there is no CPU fallback')
        self.wide = env_id in _lib.WIDE_IDS
        self.lib = _lib.load(wide=self.wide)
        self.state_layout = WIDE_STATE_LAYOUT if self.wide else STATE_LAYOUT
        self.env_id = env_id
        self.num_envs = int(num_envs)
        idx = device if isinstance(device, int) else (torch.device(device).index or 0)
        self.device = torch.device('cuda', idx)
        cfg = _lib.RpConfig(_lib.ENV_KINDS[env_id], self.num_envs, self.device.index, int(env_offset), int(seed))
        flags = 0
        if (goal_range_low is None) != (goal_range_high is None) or (obj_lower_bound is None) != (obj_upper_bound is None):
            raise ValueError('range kwargs come in low / high pairs')
        if goal_range_low is not None:
            flags |= _lib.CFG_GOAL_RANGE
            cfg.goal_range_low[:] = [float(v) for v in goal_range_low]
            cfg.goal_range_high[:] = [float(v) for v in goal_range_high]
        if obj_lower_bound is not None:
            flags |= _lib.CFG_OBJ_RANGE
            cfg.obj_lower_bound[:] = [float(v) for v in obj_lower_bound]
            cfg.obj_upper_bound[:] = [float(v) for v in obj_upper_bound]
        if env_range_high is not None:
            flags |= _lib.CFG_ENV_RANGE
            cfg.env_range_high[:] = [float(v) for v in env_range_high]
        if sparse_rew_thresh is not None:
            flags |= _lib.CFG_REW_THRESH
            cfg.sparse_rew_thresh = float(sparse_rew_thresh)
        if not sparse:
            flags |= _lib.CFG_DENSE_REWARD
        if action_type is not None:
            flags |= _lib.CFG_ACTION_TYPE
            cfg.action_type = _lib.ACTION_TYPE_CODES[action_type]
        if contact_margin is not None:
            flags |= _lib.CFG_CONTACT_MARGIN
            cfg.contact_margin = float(contact_margin)
        if not persistent_manifolds:
            flags |= _lib.CFG_STATELESS_CONTACTS      # rp_config_flags: no contact cache, points rebuilt every substep (round 3's first model)
        if speculative_limits:
            flags |= _lib.CFG_SPECULATIVE_LIMITS      # round 2's joint-limit rows (oracle rule without bit 2): no gripper chatter, further from Bullet's limit rule
        if not hull_gjk:
            flags |= _lib.CFG_OBB_EDGES               # round 3's contacts where a link's deepest hull vertex lies beside the box face: its OBB instead of GJK on the hull (oracle rule 1015)
        if hull_epa is not None:                      # None: the arm's default (Panda ids: on, UR5 ids: off - include/rp_playroom.h RP_CFG_HULL_EPA)
            flags |= _lib.CFG_HULL_EPA if hull_epa else _lib.CFG_NO_HULL_EPA
        cfg.flags = flags
        self.h = C.c_void_p()
        self._done = None
        _lib.check(self.lib, None, self.lib.rp_create(C.byref(cfg), C.byref(self.h)), 'rp_create')
        d = _lib.RpDims()
        self.lib.rp_get_dims(self.h, C.byref(d))
        self.dims = {n: getattr(d, n) for n, _ in _lib.RpDims._fields_}
        N, dev = self.num_envs, self.device

        def f(w):
            return torch.zeros((N, w), dtype=torch.float32, device=dev)

        self.buf = {k: f(self.dims[k]) for k in OBS_KEYS}
        self.buf['gripper_proprioception'] = torch.zeros(N, dtype=torch.int32, device=dev)
        self.buf['reward'] = torch.zeros(N, dtype=torch.float32, device=dev)
        self.buf['is_success'] = torch.zeros(N, dtype=torch.int32, device=dev)
        self.buf['target_poses'] = f(self.dims['target_poses'])
        self.buf['status'] = torch.zeros(N, dtype=torch.int32, device=dev)
        # obs_quat | achieved_goal | reward | is_success in one row per env: the per-step multi-GPU gather's message (sharding.py)
        # (two buffers, alternating per step: an asynchronous gather of step k's pack may still be reading it while step k + 1 runs)
        self._packs = [f(self.dims['obs_quat'] + self.dims['achieved_goal'] + 2) for _ in range(2)]
        self._pack_i = 0
        self.buf['pack'] = self._packs[0]
        self.out = _lib.RpOut(**{k: self.buf[k].data_ptr() for k, _ in _lib.RpOut._fields_})
        at = action_type or _lib.ACTION_TYPES.get(env_id, 'absolute_rpy')                           # environments.py:88-113
        hi = {'absolute_rpy': [6] * 6 + [1], 'absolute_joints': [6] * (self.dims['action'] - 1) + [1]}.get(at, [1] * self.dims['action'])
        self.action_type = at
        self.action_high = torch.tensor(hi, dtype=torch.float32, device=dev)
        self._max_episode_steps = None if env_id.startswith('UR5Play') else 250
        self.record_images = False          # instance.record_images (environments.py:201, 203)
        self.image_envs = None              # (lo, hi): the envs whose img is rendered while record_images is set (default: all - 120 KB per env and call)
        self._img = None                    # reused image buffer: obs['img'] is overwritten by the next step / reset / calc_state (clone() to keep one)
        self.sub_goal = None                # [N, dims.achieved_goal] ghosts drawn into img (visualise_sub_goal)
        self.ghost_arm = None               # [N, 8] ghost arm poses drawn into img (visualise_sub_goal's arm part, Panda ids)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _obs(self):
        o = {k: self.buf[k] for k in OBS_KEYS}
        # environments.py:841-845: img only while record_images is set (render('rgb_array')); all envs, [N, 200, 200, 3] uint8
        o['img'] = None
        if self.record_images:
            lo, hi = self.image_envs or (0, self.num_envs)
            if self._img is None or self._img.shape[0] != hi - lo:
                self._img = torch.empty((hi - lo, 200, 200, 3), dtype=torch.uint8, device=self.device)
            sg = self.sub_goal[lo:hi] if (self.sub_goal is not None and self.sub_goal.shape[0] == self.num_envs) else self.sub_goal
            ga = self.ghost_arm[lo:hi] if (self.ghost_arm is not None and self.ghost_arm.shape[0] == self.num_envs) else self.ghost_arm
            o['img'] = self.render('rgb_array', envs=(lo, hi), sub_goal=sg, out=self._img, ghost_arm=ga)
        o['gripper_proprioception'] = self.buf['gripper_proprioception']
        return o

    def close(self):
        if getattr(self, 'h', None):
            self.lib.rp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _flip_pack(self, keep_rows=False):
        """the next writer of the pack gets the other buffer: an asynchronous gather of the previous step's pack (sharding.gather_observations,
        on RCCL's stream) may still be reading the current one, and nothing orders a write on our stream after that read.  keep_rows: a masked
        reset rewrites only some rows, so the others are carried over first (a device copy on our stream that only READS the old buffer)."""
        new = self._packs[self._pack_i ^ 1]
        if keep_rows:
            new.copy_(self._packs[self._pack_i])
        self._pack_i ^= 1
        self.buf['pack'] = new
        self.out.pack = new.data_ptr()

    def reset(self, mask=None, o=None):
        """playEnv.reset(o=None) for all envs (or those where mask != 0).  With o [N, >= 18 / 10 / 3]: playEnv.reset(o) - objects
        and arm are placed from the observation vectors instead of being sampled (environments.py:542-556, 575-590)."""
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mp = C.c_void_p(mask.data_ptr())
        self._flip_pack(keep_rows=mask is not None)
        if o is None:
            _lib.check(self.lib, self.h, self.lib.rp_reset(self.h, mp, C.byref(self.out), self._stream()), 'rp_reset')
        else:
            o = o.to(device=self.device, dtype=torch.float32).contiguous()
            assert o.dim() == 2 and o.shape[0] == self.num_envs, o.shape
            _lib.check(self.lib, self.h, self.lib.rp_reset_to(self.h, C.c_void_p(o.data_ptr()), o.shape[1], mp, C.byref(self.out), self._stream()),
                       'rp_reset_to')
        return self._obs()

    def step(self, action):
        a = action.to(device=self.device, dtype=torch.float32).contiguous()
        assert a.shape == (self.num_envs, self.dims['action']), a.shape
        self._flip_pack()
        _lib.check(self.lib, self.h, self.lib.rp_step(self.h, C.c_void_p(a.data_ptr()), C.byref(self.out), self._stream()), 'rp_step')
        info = {'is_success': self.buf['is_success'], 'target_poses': self.buf['target_poses'], 'status': self.buf['status']}
        if self._done is None:      # environments.py:212: always False - one tensor for the handle's life (a fill kernel per step sat at the end of the step's chain: 8 us)
            self._done = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        return self._obs(), self.buf['reward'], self._done, info

    def calc_state(self):
        self._flip_pack()
        _lib.check(self.lib, self.h, self.lib.rp_calc_state(self.h, C.byref(self.out), self._stream()), 'rp_calc_state')
        return self._obs()

    def reset_goal_pos(self, goal=None, mask=None):
        gp = mp = None
        if goal is not None:
            goal = goal.to(device=self.device, dtype=torch.float32).contiguous()
            assert goal.shape == (self.num_envs, self.dims['desired_goal'])
            gp = C.c_void_p(goal.data_ptr())
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mp = C.c_void_p(mask.data_ptr())
        _lib.check(self.lib, self.h, self.lib.rp_reset_goal(self.h, gp, mp, self._stream()), 'rp_reset_goal')

    def compute_reward_sparse(self, achieved_goal, desired_goal, info=None):
        """the sparse formula (environments.py:278-304) whatever `sparse` the env was built with"""
        return self.compute_reward(achieved_goal, desired_goal, info, _fn='rp_compute_reward_sparse')

    def compute_reward(self, achieved_goal, desired_goal, info=None, _fn='rp_compute_reward'):
        ag = achieved_goal.to(device=self.device, dtype=torch.float32).contiguous()
        dg = desired_goal.to(device=self.device, dtype=torch.float32).contiguous()
        w = self.dims['achieved_goal']
        ag2, dg2 = ag.reshape(-1, w), dg.reshape(-1, w)
        r = torch.empty(ag2.shape[0], dtype=torch.float32, device=self.device)
        _lib.check(self.lib, self.h, getattr(self.lib, _fn)(self.h, C.c_void_p(ag2.data_ptr()), C.c_void_p(dg2.data_ptr()),
                                                             C.c_void_p(r.data_ptr()), ag2.shape[0], self._stream()), _fn)
        return r.reshape(ag.shape[:-1])

    @property
    def pack(self):
        """[N, obs_quat + achieved_goal + 2] written by the latest step / reset / calc_state: obs_quat | achieved_goal | reward |
        is_success (float)"""
        return self.buf['pack']

    def camera(self, target=(0.0, 0.25, 0.0), distance=1.3, yaw=-30.0, pitch=-30.0, roll=0.0, fov=50.0, aspect=1.0, gripper=False):
        """an rp_camera: p.computeViewMatrixFromYawPitchRoll + computeProjectionMatrixFOV (the defaults are the reference's fixed camera,
        environments.py:21-30); gripper=True: the gripper camera of environments.py:33-49"""
        cam = _lib.RpCamera()
        _lib.check(self.lib, self.h, self.lib.rp_camera_from_yaw_pitch_roll((C.c_float * 3)(*[float(v) for v in target]), float(distance), float(yaw),
                                                                             float(pitch), float(roll), C.byref(cam)), 'rp_camera_from_yaw_pitch_roll')
        cam.fov_deg, cam.aspect, cam.mode = float(fov), float(aspect), 1 if gripper else 0
        return cam

    def render(self, mode='rgb_array', width=200, height=200, envs=None, camera=None, sub_goal=None, out=None, ghost_arm=None):
        """obs['img'] of the reference (environments.py:841-845: getCameraImage(200, 200, ...)[2][:, :, :3]) for envs [lo, hi) (default: all):
        uint8 [n, height, width, 3] on the device.  sub_goal [n, dims.achieved_goal]: draw the sub-goal's ghosts
        (visualise_sub_goal, environments.py:606-690).  ghost_arm [n, 8] = EE position, orientation quaternion (xyzw), gripper: the ghost ARM of
        visualise_sub_goal's 'controllable_achieved_goal' / 'full_positional_state' (environments.py:623-637, 671-674; Panda ids only - the reference raises for the
        UR5).  mode 'human' (a GUI window) does not exist here and returns None."""
        if mode == 'human':
            return None
        lo, hi = (0, self.num_envs) if envs is None else (int(envs[0]), int(envs[1]))
        n = hi - lo
        img = out if out is not None else torch.empty((n, int(height), int(width), 3), dtype=torch.uint8, device=self.device)
        assert img.shape == (n, int(height), int(width), 3) and img.dtype == torch.uint8 and img.is_contiguous()
        sg = None
        if sub_goal is not None:
            sub_goal = sub_goal.to(device=self.device, dtype=torch.float32).contiguous()
            assert sub_goal.shape == (n, self.dims['achieved_goal']), sub_goal.shape
            sg = C.c_void_p(sub_goal.data_ptr())
        if ghost_arm is not None:
            ghost_arm = ghost_arm.to(device=self.device, dtype=torch.float32).contiguous()
            assert ghost_arm.shape == (n, 8), ghost_arm.shape
            _lib.check(self.lib, self.h, self.lib.rp_render_ex(self.h, C.byref(camera) if camera is not None else None, int(width), int(height), lo, n,
                                                               C.c_void_p(img.data_ptr()), sg, C.c_void_p(ghost_arm.data_ptr()), self._stream()), 'rp_render_ex')
            return img
        _lib.check(self.lib, self.h, self.lib.rp_render(self.h, C.byref(camera) if camera is not None else None, int(width), int(height), lo, n,
                                                        C.c_void_p(img.data_ptr()), sg, self._stream()), 'rp_render')
        return img

    def ray_test(self, ray_from, ray_to):
        """bullet_client.rayTest for k rays per env: ray_from / ray_to [N, k, 3] (world).  Returns dict(hit_fraction [N, k] (1 = miss),
        collider [N, k] (-1 = miss), link [N, k] (Bullet link index of an arm collider, else -1), hit_position, hit_normal [N, k, 3])"""
        f = ray_from.to(device=self.device, dtype=torch.float32).contiguous()
        t = ray_to.to(device=self.device, dtype=torch.float32).contiguous()
        assert f.shape == t.shape and f.dim() == 3 and f.shape[0] == self.num_envs and f.shape[2] == 3, f.shape
        k = f.shape[1]
        out = {'hit_fraction': torch.empty((self.num_envs, k), dtype=torch.float32, device=self.device),
               'collider': torch.empty((self.num_envs, k), dtype=torch.int32, device=self.device),
               'link': torch.empty((self.num_envs, k), dtype=torch.int32, device=self.device),
               'hit_position': torch.empty((self.num_envs, k, 3), dtype=torch.float32, device=self.device),
               'hit_normal': torch.empty((self.num_envs, k, 3), dtype=torch.float32, device=self.device)}
        _lib.check(self.lib, self.h, self.lib.rp_ray_test(self.h, C.c_void_p(f.data_ptr()), C.c_void_p(t.data_ptr()), k, C.c_void_p(out['hit_fraction'].data_ptr()),
                                                          C.c_void_p(out['collider'].data_ptr()), C.c_void_p(out['link'].data_ptr()),
                                                          C.c_void_p(out['hit_position'].data_ptr()), C.c_void_p(out['hit_normal'].data_ptr()), self._stream()),
                   'rp_ray_test')
        return out

    def get_state(self):
        n = self.lib.rp_state_bytes(self.h) // 4
        s = torch.empty((self.num_envs, n), dtype=torch.float32, device=self.device)
        _lib.check(self.lib, self.h, self.lib.rp_get_state(self.h, C.c_void_p(s.data_ptr()), self._stream()), 'rp_get_state')
        return s

    def set_state(self, s):
        """s: [N or 1, rp_state_bytes / 4] as get_state returns it - or only the state records ([.., 128], e.g. built from an oracle's state): the envs then
        start without contact history (an all-zero contact cache)"""
        s = s.to(device=self.device, dtype=torch.float32)
        if s.dim() == 1:
            s = s[None]
        n = self.lib.rp_state_bytes(self.h) // 4
        if s.shape[1] < n:
            s = torch.cat([s, torch.zeros((s.shape[0], n - s.shape[1]), dtype=torch.float32, device=self.device)], 1)
        s = s.contiguous()
        _lib.check(self.lib, self.h, self.lib.rp_set_state(self.h, C.c_void_p(s.data_ptr()), s.shape[0], self._stream()), 'rp_set_state')

    def replay(self, o0, actions, keys=('obs_quat', 'achieved_goal')):
        """Play recorded trajectories back (the reference's README use: "playing out the teleop data", "reset the environment to
        specific locations"): every env is placed from its first recorded observation o0[e] with reset(o) - objects and arm from
        the observation vector, nothing settles - then the recorded actions [T, N, action] are stepped open loop.  Returns
        {key: [T + 1, N, dim]} (index 0 = after the reset) plus 'reward' and 'is_success' [T, N]."""
        obs = self.reset(o=o0)
        out = {k: [obs[k].clone()] for k in keys}
        rew, suc = [], []
        for t in range(actions.shape[0]):
            obs, r, _, info = self.step(actions[t])
            for k in keys:
                out[k].append(obs[k].clone())
            rew.append(r.clone())
            suc.append(info['is_success'].clone())
        res = {k: torch.stack(v) for k, v in out.items()}
        res['reward'] = torch.stack(rew) if rew else torch.zeros((0, self.num_envs), device=self.device)
        res['is_success'] = torch.stack(suc) if suc else torch.zeros((0, self.num_envs), dtype=torch.int32, device=self.device)
        return res

    def set_fused(self, mode=1):
        """step pipeline: 0 = default (k_action, k_prep2, k_solve2, k_calc_state), 1 = one fused kernel per step (the
        library's in-GPU cross-check path).  Both are bit-identical."""
        _lib.check(self.lib, self.h, self.lib.rp_set_fused(self.h, int(mode)), 'rp_set_fused')

    def set_groups(self, groups):
        """number of env groups (each with its own stream and kernel chain) rp_step uses; results do not depend on it"""
        _lib.check(self.lib, self.h, self.lib.rp_set_groups(self.h, int(groups)), 'rp_set_groups')

    def set_debug_flags(self, flags):
        """test hook: bit 0 makes the solver give every contact its own folded slot (its fallback layout) instead of solving
        arm-only and non-arm contacts side by side; results are bit-identical either way"""
        _lib.check(self.lib, self.h, self.lib.rp_set_debug_flags(self.h, int(flags)), 'rp_set_debug_flags')

    def debug_row_counts(self):
        """test hook: per env of the latest substep, [unit rows, contacts, 1 if an arm contact, spanning contacts] (host tensor)"""
        buf = (C.c_int32 * (2 * self.num_envs))()
        _lib.check(self.lib, self.h, self.lib.rp_debug_row_counts(self.h, buf), 'rp_debug_row_counts')
        a = torch.tensor(list(buf), dtype=torch.int64).reshape(self.num_envs, 2)
        return torch.stack([a[:, 0], a[:, 1] % 1000, (a[:, 1] // 1000) % 100, a[:, 1] // 100000], 1)

    def enable_timers(self, steps=64):
        """keep per-launch hipEvent timings for the next `steps` rp_step calls (0 disables)"""
        _lib.check(self.lib, self.h, self.lib.rp_enable_timers(self.h, int(steps)), 'rp_enable_timers')

    def timers(self):
        t = _lib.RpTimers()
        _lib.check(self.lib, self.h, self.lib.rp_get_timers(self.h, C.byref(t)), 'rp_get_timers')
        return {n: getattr(t, n) for n, _ in _lib.RpTimers._fields_}

    def debug_substep(self, env=0):
        buf = (C.c_float * 4096)()
        _lib.check(self.lib, self.h, self.lib.rp_debug_substep(self.h, env, buf), 'rp_debug_substep')
        return torch.tensor(list(buf))
