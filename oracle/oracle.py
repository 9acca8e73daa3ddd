"""ctypes wrapper over the CPU oracle (oracle/librp_oracle*.so).  TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product package
(roboticsplayroompybullet_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
KINDS = {'U': 0, 'R': 1, 'P': 2, 'Q': 3, 'V': 4, 'W': 5, 'pandaPlay-v0': 5, 'UR5PlayAbsRPY1Obj-v0': 0, 'UR5Reach-v0': 1, 'pandaPick-v0': 2, 'pandaReach-v0': 3,
         'pandaPlayAbsRPY1Obj-v0': 4}


class RpoObs(C.Structure):
    _fields_ = [('obs_quat', C.c_double * 26), ('achieved_goal', C.c_double * 18), ('desired_goal', C.c_double * 18),
                ('controllable_achieved_goal', C.c_double * 4), ('full_positional_state', C.c_double * 26),
                ('joints', C.c_double * 8), ('velocity', C.c_double * 6), ('observation', C.c_double * 25),
                ('gripper_proprioception', C.c_int), ('n_obs', C.c_int), ('n_ag', C.c_int), ('n_fps', C.c_int),
                ('n_observation', C.c_int)]

    def to_dict(self, n_goal):
        return {
            'obs_quat': np.array(self.obs_quat[:self.n_obs]),
            'achieved_goal': np.array(self.achieved_goal[:self.n_ag]),
            'desired_goal': np.array(self.desired_goal[:n_goal]),
            'controllable_achieved_goal': np.array(self.controllable_achieved_goal[:4]),
            'full_positional_state': np.array(self.full_positional_state[:self.n_fps]),
            'joints': list(self.joints[:8]),
            'velocity': np.array(self.velocity[:6]),
            'img': None,
            'observation': np.array(self.observation[:self.n_observation]),
            'gripper_proprioception': int(self.gripper_proprioception),
        }


class RpoReadings(C.Structure):
    _fields_ = [('ee_pos', C.c_double * 3), ('ee_orn', C.c_double * 4), ('ee_lin', C.c_double * 3), ('ee_ang', C.c_double * 3),
                ('grip_q', C.c_double), ('joints', C.c_double * 8), ('proprio', C.c_int),
                ('block_pos', C.c_double * 3), ('block_orn', C.c_double * 4), ('block_vel', C.c_double * 3),
                ('drawer_y', C.c_double), ('door_q', C.c_double), ('button_q', C.c_double), ('dial_q', C.c_double),
                ('block2_pos', C.c_double * 3), ('block2_orn', C.c_double * 4), ('block2_vel', C.c_double * 3)]


_LIBS = {}


def build():
    subprocess.run(['make', '-C', HERE, '-s'], check=True)


REF_FLAGS = {'hull': 1, 'persist': 2, 'order': 4, 'lever': 8, 'soft': 16, 'anchor': 32, 'spin': 64, 'fricskip': 128, 'warm': 256, 'limit': 512}
REF_DEFAULT = 255 + 512          # everything but warm starting (rp_bullet_ref.c RPB_DEFAULT)


def load(f32=False, bullet_ref=False, abx=False):
    name = 'librp_oracle_abx.so' if abx else ('librp_oracle_bullet.so' if bullet_ref else ('librp_oracle_f32.so' if f32 else 'librp_oracle.so'))
    if name in _LIBS:
        return _LIBS[name]
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        if abx:
            subprocess.run(['make', '-C', HERE, '-s', name], check=True)      # (the experiment build is not part of `make all`)
        else:
            build()
    lib = C.CDLL(path)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    vp = C.c_void_p
    lib.rpo_create.restype = vp
    lib.rpo_create.argtypes = [C.c_int, C.c_ulonglong, C.c_int]
    lib.rpo_destroy.argtypes = [vp]
    lib.rpo_nv.argtypes = [vp]
    lib.rpo_n_arm.argtypes = [vp]
    lib.rpo_reset.argtypes = [vp, dp, C.c_int, C.POINTER(RpoObs)]
    lib.rpo_reset_to.argtypes = [vp, dp, C.c_int, dp, C.c_int, C.POINTER(RpoObs)]
    lib.rpo_reset_to.restype = C.c_int
    lib.rpo_reset.restype = C.c_int
    lib.rpo_reset_goal.argtypes = [vp, dp, dp, C.c_int]
    lib.rpo_reset_samples.argtypes = [vp, dp, dp, dp]
    lib.rpo_step.argtypes = [vp, dp, C.POINTER(RpoObs), dp, ip, dp]
    lib.rpo_calc_state.argtypes = [vp, C.POINTER(RpoObs)]
    lib.rpo_compute_reward.argtypes = [vp, dp, dp]
    lib.rpo_compute_reward.restype = C.c_double
    lib.rpo_read_world.argtypes = [vp, C.POINTER(RpoReadings)]
    lib.rpo_assemble_obs.argtypes = [vp, C.POINTER(RpoReadings), C.POINTER(RpoObs)]
    lib.rpo_quat_from_euler.argtypes = [dp, dp]
    lib.rpo_euler_from_quat.argtypes = [dp, dp]
    lib.rpo_dial_to_0_1_range.argtypes = [C.c_double]
    lib.rpo_dial_to_0_1_range.restype = C.c_double
    lib.rpo_perform_action.argtypes = [vp, dp, dp]
    lib.rpo_action_target.argtypes = [C.c_int, dp, dp, dp, dp, dp]
    lib.rpo_set_ranges.argtypes = [vp, dp, dp, dp, dp, dp]
    lib.rpo_set_action_type.argtypes = [vp, C.c_int]
    lib.rpo_set_margin.argtypes = [vp, C.c_double]
    lib.rpo_set_rule.argtypes = [vp, C.c_int]
    lib.rpo_last_num_tors.argtypes = [vp]
    lib.rpo_cache_size.argtypes = [vp, ip]
    lib.rpo_shift_free_body.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double]
    fp = C.POINTER(C.c_float)
    lib.rpo_get_cache_row.argtypes = [vp, fp]
    lib.rpo_set_cache_row.argtypes = [vp, fp]
    lib.rpo_gjk_stats.argtypes = [C.POINTER(C.c_long), C.c_int]
    lib.rpo_get_rule.argtypes = [vp]
    lib.rpo_set_reward_cfg.argtypes = [vp, C.c_double, C.c_int]
    lib.rpo_action_dim.argtypes = [vp]
    lib.rpo_get_config.argtypes = [vp, dp]
    lib.rpo_action_dim.restype = C.c_int
    lib.rpo_goto_joint_poses.argtypes = [vp, dp, C.c_int, C.c_double, dp]
    lib.rpo_ik.argtypes = [vp, dp, dp, dp, C.c_int, dp]
    lib.rpo_calc_angles.argtypes = [vp, dp, dp, dp, dp]
    lib.rpo_substep.argtypes = [vp]
    lib.rpo_run_simulation.argtypes = [vp]
    lib.rpo_state_size.argtypes = [vp]
    lib.rpo_get_state.argtypes = [vp, dp]
    lib.rpo_set_state.argtypes = [vp, dp]
    lib.rpo_get_motor.argtypes = [vp, ip, dp, dp]
    lib.rpo_set_goal.argtypes = [vp, dp]
    lib.rpo_clear_quat_memory.argtypes = [vp]
    lib.rpo_site_pose.argtypes = [vp, C.c_int, dp, dp, dp, dp]
    lib.rpo_mass_matrix_inv.argtypes = [vp, dp]
    lib.rpo_forward_dynamics.argtypes = [vp, dp]
    lib.rpo_contacts.argtypes = [vp, dp, C.c_int]
    lib.rpo_last_num_rows.argtypes = [vp]
    lib.rpo_contact_substeps.argtypes = [vp]
    lib.rpo_residual_substeps.argtypes = [vp]
    lib.rpo_ik_iterations.argtypes = [C.c_int]
    lib.rpo_ik_iterations.restype = C.c_long
    lib.rpo_rest_pose.argtypes = [vp, dp]
    lib.rpo_set_proprioception_boxes.argtypes = [C.c_int]
    lib.rpo_arm_table.argtypes = [vp, dp]
    lib.rpo_collider_dynamics.argtypes = [vp, dp]
    lib.rpo_set_arm_q.argtypes = [vp, dp]
    lib.rpo_box_box.argtypes = [dp, dp, dp, dp, dp, dp, C.c_double, dp]
    lib.rpo_collider_poses.argtypes = [vp, dp]
    lib.rpo_collider_table.argtypes = [vp, dp]
    lib.rpo_hull_vertices_world.argtypes = [vp, C.c_int, dp, C.c_int]
    lib.rpo_pair_table.argtypes = [vp, C.POINTER(C.c_int)]
    lib.rpo_pair_table.restype = C.c_int
    lib.rpo_bench_rollout.argtypes = [C.c_int, C.c_ulonglong, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.c_double]
    lib.rpo_bench_rollout.restype = C.c_double
    lib.rpo_rng_uniform.argtypes = [C.c_ulonglong, C.c_uint, C.c_uint]
    lib.rpo_rng_uniform.restype = C.c_double
    if bullet_ref:
        lib.rpo_set_ref_flags.argtypes = [vp, C.c_uint]
        lib.rpo_get_ref_flags.argtypes = [vp]
        lib.rpo_get_ref_flags.restype = C.c_uint
        lib.rpo_ref_num_contacts.argtypes = [vp]
        lib.rpo_ref_num_manifolds.argtypes = [vp]
        lib.rpo_ref_contacts.argtypes = [vp, dp, C.c_int]
    _LIBS[name] = lib
    return lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


ACTION_TYPES = {'absolute_rpy': 0, 'relative_rpy': 1, 'absolute_quat': 2, 'relative_quat': 3, 'absolute_joints': 4, 'relative_joints': 5}
FAMILY = {'UR5PlayAbsRPY1Obj-v0': 'absolute_rpy', 'UR5PlayRelRPY1Obj-v0': 'relative_rpy', 'UR5Play1Obj-v0': 'absolute_quat',
          'UR5PlayRel1Obj-v0': 'relative_quat', 'UR5PlayAbsJoints1Obj-v0': 'absolute_joints', 'UR5PlayRelJoints1Obj-v0': 'relative_joints'}


# ids that reuse an arm + scene in scope with other ranges: (kind, goal_lo, goal_hi, obj_lo, obj_hi, env_hi)  (envList.py:12-16)
RANGES = {'pandaPush-v0': ('P', [-0.1, -0.1, -0.06], [0.1, 0.1, -0.05], [-0.1, -0.1, -0.06], [0.1, 0.1, -0.05], [0.18, 0.18, -0.04]),
          # pandaReach2D-v0 (envList.py:24-26): no objects, so the object range is unused
          'pandaReach2D-v0': ('Q', [-0.18, -0.18, -0.06], [0.18, 0.18, -0.05], [-0.18, -0.18, -0.05], [-0.18, -0.18, -0.05], [0.18, 0.18, 0.0])}
# the Panda one-object play family (envList.py:43-88): Panda + complex_scene (model V), one id per action type
PANDA_FAMILY = {'pandaPlayAbsRPY1Obj-v0': 'absolute_rpy', 'pandaPlayRelRPY1Obj-v0': 'relative_rpy', 'pandaPlay1Obj-v0': 'absolute_quat',
                'pandaPlayRel1Obj-v0': 'relative_quat', 'pandaPlayAbsJoints1Obj-v0': 'absolute_joints',
                'pandaPlayRelJoints1Obj-v0': 'relative_joints'}


# the two-object play ids (envList.py:28-41): Panda + complex_scene with two blocks (model W)
TWO_OBJECT = {'pandaPlay-v0': 'absolute_quat', 'pandaPlayJoints-v0': 'relative_joints'}


class OracleEnv:
    """One reference env (instance + playEnv) on the CPU oracle."""

    def __init__(self, kind, seed=0, env_index=0, f32=False, action_type=None, margin=None, ranges=None, sparse_rew_thresh=None,
                 dense_reward=False, bullet_ref=False, ref_flags=None, rule=None, abx=False):
        """ranges = (goal_lo, goal_hi, obj_lo, obj_hi, env_hi): the env class's range kwargs (envList.py); margin: contact margin
        in metres for every pair (default, like the library: per pair the smaller of the two objects' Bullet breaking thresholds, rp_model.col_thr)"""
        self.bullet_ref = bullet_ref
        self.lib = load(f32, bullet_ref, abx)      # bullet_ref: the frozen Bullet-like step (rp_bullet_ref.c) under the same harness; abx: the experiment build (Makefile)
        user_ranges = ranges
        ranges = None
        if kind in RANGES:
            kind, *ranges = RANGES[kind]
        if kind in FAMILY:                  # a registered id of the UR5 one-object play family: scene U, other action type
            kind, action_type = 'U', FAMILY[kind]
        if kind in PANDA_FAMILY:
            kind, action_type = 'V', PANDA_FAMILY[kind]
        if kind in TWO_OBJECT:
            kind, action_type = 'W', TWO_OBJECT[kind]
        self.kind = KINDS[kind]
        self.h = self.lib.rpo_create(self.kind, seed, env_index)
        self.action_type = action_type or 'absolute_rpy'
        self.lib.rpo_set_action_type(self.h, ACTION_TYPES[self.action_type])
        self.n_action = self.lib.rpo_action_dim(self.h)
        if user_ranges is not None:
            ranges = user_ranges
        if ranges:
            self.lib.rpo_set_ranges(self.h, *[_d(r)[1] for r in ranges])
        if margin is not None:
            self.lib.rpo_set_margin(self.h, float(margin))
        if rule is not None:
            self.lib.rpo_set_rule(self.h, int(rule))
        if bullet_ref and ref_flags is not None:
            self.lib.rpo_set_ref_flags(self.h, int(ref_flags))
        if sparse_rew_thresh is not None or dense_reward:
            self.lib.rpo_set_reward_cfg(self.h, 0.05 if sparse_rew_thresh is None else float(sparse_rew_thresh), int(bool(dense_reward)))
        self.n_arm = self.lib.rpo_n_arm(self.h)
        self.nv = self.lib.rpo_nv(self.h)
        self.n_goal = {0: 11, 4: 11, 5: 18}.get(self.kind, 3)
        self.n_target = 7 if self.kind in (2, 3, 4, 5) else 6

    def __del__(self):
        if getattr(self, 'h', None):
            self.lib.rpo_destroy(self.h)
            self.h = None

    def reset(self, u=None):
        o = RpoObs()
        if u is None:
            used = self.lib.rpo_reset(self.h, None, 0, C.byref(o))
        else:
            ua, up = _d(u)
            used = self.lib.rpo_reset(self.h, up, len(ua), C.byref(o))
        self.last_used = used
        return o.to_dict(self.n_goal)

    def reset_to(self, o, u=None):
        """playEnv.reset(o): place objects and arm from an observation vector"""
        ob = RpoObs()
        oa, op = _d(o)
        if u is None:
            used = self.lib.rpo_reset_to(self.h, op, len(oa), None, 0, C.byref(ob))
        else:
            ua, up = _d(u)
            used = self.lib.rpo_reset_to(self.h, op, len(oa), up, len(ua), C.byref(ob))
        self.last_used = used
        return ob.to_dict(self.n_goal)

    def reset_samples(self, u):
        b, t = np.zeros(6), np.zeros(3)
        dp = C.POINTER(C.c_double)
        self.lib.rpo_reset_samples(self.h, _d(u)[1], b.ctypes.data_as(dp), t.ctypes.data_as(dp))
        return b, t

    def reset_goal_pos(self, goal=None, u=None):
        gp = _d(goal)[1] if goal is not None else None
        if u is None:
            self.lib.rpo_reset_goal(self.h, gp, None, 0)
        else:
            ua, up = _d(u)
            self.lib.rpo_reset_goal(self.h, gp, up, len(ua))

    def step(self, action):
        a, ap = _d(action)
        o = RpoObs()
        r = C.c_double()
        s = C.c_int()
        tp = np.zeros(7)
        self.lib.rpo_step(self.h, ap, C.byref(o), C.byref(r), C.byref(s), tp.ctypes.data_as(C.POINTER(C.c_double)))
        return o.to_dict(self.n_goal), r.value, False, {'is_success': s.value, 'target_poses': tp[:self.n_target].copy()}

    def action_target(self, action, ee_pos, ee_orn):
        """IK target of a pose-type action from the clipped action and a measured EE pose (test hook)"""
        a = np.zeros(8); a[:len(action)] = action
        pos, quat = np.zeros(3), np.zeros(4)
        dp = C.POINTER(C.c_double)
        at = ACTION_TYPES[self.action_type]
        self.lib.rpo_action_target(at, _d(a)[1], _d(ee_pos)[1], _d(ee_orn)[1], pos.ctypes.data_as(dp), quat.ctypes.data_as(dp))
        return pos, quat

    def _config(self):
        c = np.zeros(27)
        self.lib.rpo_get_config(self.h, c.ctypes.data_as(C.POINTER(C.c_double)))
        return c

    def flags(self):
        c = self._config()
        return {'play': int(c[0]), 'use_orientation': int(c[1]), 'return_velocity': int(c[2]), 'num_objects': int(c[3])}

    def ranges(self):
        c = self._config()
        return {'goal_lo': c[4:7], 'goal_hi': c[7:10], 'obj_lo': c[10:13], 'obj_hi': c[13:16], 'env_hi': c[16:19]}

    def action_high(self):
        return self._config()[19:19 + self.n_action]

    def calc_state(self):
        o = RpoObs()
        self.lib.rpo_calc_state(self.h, C.byref(o))
        return o.to_dict(self.n_goal)

    def assemble_obs(self, **kw):
        rd = RpoReadings()
        for k, v in kw.items():
            if isinstance(v, (list, tuple, np.ndarray)):
                for i, x in enumerate(v):
                    getattr(rd, k)[i] = float(x)
            else:
                setattr(rd, k, v)
        o = RpoObs()
        self.lib.rpo_assemble_obs(self.h, C.byref(rd), C.byref(o))
        return o.to_dict(self.n_goal)

    def colliders(self):
        """(R [n, 3, 3], p [n, 3], table [n, 9] = type, he3, rgb3, toggle, link) of the current state"""
        buf = np.zeros(64 * 12)
        n = self.lib.rpo_collider_poses(self.h, buf.ctypes.data_as(C.POINTER(C.c_double)))
        tab = np.zeros(64 * 9)
        self.lib.rpo_collider_table(self.h, tab.ctypes.data_as(C.POINTER(C.c_double)))
        b = buf[:12 * n].reshape(n, 12)
        return b[:, :9].reshape(n, 3, 3), b[:, 9:], tab[:9 * n].reshape(n, 9)

    def hull_vertices(self, c):
        """world-frame vertices [n, 3] of the convex hull collider c collides as (arm links), None for boxes / spheres"""
        buf = np.zeros(3 * 1024)
        n = self.lib.rpo_hull_vertices_world(self.h, int(c), buf.ctypes.data_as(C.POINTER(C.c_double)), 1024)
        return buf[:3 * n].reshape(n, 3).copy() if n > 0 else None

    def compute_reward(self, ag, dg):
        return self.lib.rpo_compute_reward(self.h, _d(ag)[1], _d(dg)[1])

    def perform_action(self, action):
        tp = np.zeros(7)
        self.lib.rpo_perform_action(self.h, _d(action)[1], tp.ctypes.data_as(C.POINTER(C.c_double)))
        return tp[:self.n_target].copy()

    def goto_joint_poses(self, poses, gripper=None):
        tp = np.zeros(7)
        self.lib.rpo_goto_joint_poses(self.h, _d(poses)[1], int(gripper is not None), float(gripper or 0.0),
                                      tp.ctypes.data_as(C.POINTER(C.c_double)))
        return tp[:self.n_target].copy()

    def rest_pose(self):
        out = np.zeros(self.n_arm)
        self.lib.rpo_rest_pose(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def ik(self, pos, quat, q_seed, max_iter=20):
        out = np.zeros(self.n_arm)
        self.lib.rpo_ik(self.h, _d(pos)[1], _d(quat)[1], _d(q_seed)[1], max_iter, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def calc_angles(self, pos, quat, current):
        out = np.zeros(6)
        self.lib.rpo_calc_angles(self.h, _d(pos)[1], _d(quat)[1], _d(current)[1], out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def substep(self, n=1):
        for _ in range(n):
            self.lib.rpo_substep(self.h)

    def run_simulation(self):
        self.lib.rpo_run_simulation(self.h)

    def get_state(self):
        s = np.zeros(self.lib.rpo_state_size(self.h))
        self.lib.rpo_get_state(self.h, s.ctypes.data_as(C.POINTER(C.c_double)))
        return s

    def set_state(self, s):
        self.lib.rpo_set_state(self.h, _d(s)[1])

    def get_cache_row(self):
        """the contact cache in the HIP library's row layout (float32 [704], integers as bit patterns: rp_kernels.cuh PMC_*) - what rp_get_state rows carry
        behind the 128-float record"""
        row = np.zeros(self.lib.rpo_cache_row_words(), dtype=np.float32)
        n = self.lib.rpo_get_cache_row(self.h, row.ctypes.data_as(C.POINTER(C.c_float)))
        assert n == len(row), n
        return row

    def set_cache_row(self, row):
        """takes a cache row (e.g. the device's own, rp_get_state[:, 128:]) as this env's contact history; call after set_state, which clears it"""
        row = np.ascontiguousarray(row, dtype=np.float32)
        assert len(row) == self.lib.rpo_cache_row_words(), len(row)
        if self.lib.rpo_set_cache_row(self.h, row.ctypes.data_as(C.POINTER(C.c_float))) != 0:
            raise ValueError('malformed cache row')

    def gjk_stats(self, reset=False):
        """GJK counters of the calling thread: calls, rounds, two-point seeds, contacts, apart, overlap, tetrahedra, rounds of the apart exits"""
        st = (C.c_long * 8)()
        self.lib.rpo_gjk_stats(st, int(bool(reset)))
        return list(st)

    def get_motor(self):
        mode = np.zeros(self.n_arm, dtype=np.int32)
        tgt = np.zeros(self.n_arm)
        mx = np.zeros(self.n_arm)
        self.lib.rpo_get_motor(self.h, mode.ctypes.data_as(C.POINTER(C.c_int)), tgt.ctypes.data_as(C.POINTER(C.c_double)),
                               mx.ctypes.data_as(C.POINTER(C.c_double)))
        return mode, tgt, mx

    def set_goal(self, goal):
        self.lib.rpo_set_goal(self.h, _d(goal)[1])

    def clear_quat_memory(self):
        self.lib.rpo_clear_quat_memory(self.h)

    def site_pose(self, site=0):
        p, q, l, a = np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(3)
        dp = C.POINTER(C.c_double)
        self.lib.rpo_site_pose(self.h, site, p.ctypes.data_as(dp), q.ctypes.data_as(dp), l.ctypes.data_as(dp), a.ctypes.data_as(dp))
        return p, q, l, a

    def mass_matrix_inv(self):
        M = np.zeros((self.n_arm, self.n_arm))
        self.lib.rpo_mass_matrix_inv(self.h, M.ctypes.data_as(C.POINTER(C.c_double)))
        return M

    def forward_dynamics(self):
        a = np.zeros(self.n_arm)
        self.lib.rpo_forward_dynamics(self.h, a.ctypes.data_as(C.POINTER(C.c_double)))
        return a

    def contacts(self, max_n=96):
        """[n, 9]: collider a, collider b, point, normal (from b toward a), distance.  Mode B: the points of its persistent manifolds (this call advances
        them as a substep's collision phase would)"""
        out = np.zeros((max_n, 9))
        fn = self.lib.rpo_ref_contacts if self.bullet_ref else self.lib.rpo_contacts
        n = fn(self.h, out.ctypes.data_as(C.POINTER(C.c_double)), max_n)
        return out[:min(n, max_n)]

    def arm_table(self):
        """[n_arm, 6]: jtype, lower, upper, body mass, Bullet joint index, parent dof"""
        t = np.zeros((self.n_arm, 6))
        self.lib.rpo_arm_table(self.h, t.ctypes.data_as(C.POINTER(C.c_double)))
        return t

    def pair_list(self):
        """the baked candidate pairs (collider a, collider b) the broadphase sweeps"""
        buf = (C.c_int * 4096)()
        n = self.lib.rpo_pair_table(self.h, buf)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]

    def collider_list(self):
        """every collider at the current state: dict(type, he [3], R [3, 3], p [3], body, friction, mass, stiffness, damping, threshold, link)"""
        tab, dyn, pose = np.zeros((64, 9)), np.zeros((64, 6)), np.zeros((64, 12))
        dp = C.POINTER(C.c_double)
        n = self.lib.rpo_collider_table(self.h, tab.ctypes.data_as(dp))
        self.lib.rpo_collider_dynamics(self.h, dyn.ctypes.data_as(dp))
        self.lib.rpo_collider_poses(self.h, pose.ctypes.data_as(dp))
        return [dict(type=int(tab[c, 0]), he=tab[c, 1:4].copy(), R=pose[c, :9].reshape(3, 3).copy(), p=pose[c, 9:12].copy(), body=int(dyn[c, 0]), friction=dyn[c, 1],
                     mass=dyn[c, 2], stiffness=dyn[c, 3], damping=dyn[c, 4], threshold=dyn[c, 5], link=int(tab[c, 8])) for c in range(n)]

    def set_arm_q(self, q):
        self.lib.rpo_set_arm_q(self.h, _d(q)[1])

    def num_rows(self):
        return self.lib.rpo_last_num_rows(self.h)


def box_box(ca, Ra, ha, cb, Rb, hb, margin=0.02, f32=False):
    out = np.zeros((4, 7))
    n = load(f32).rpo_box_box(_d(ca)[1], _d(np.asarray(Ra).reshape(-1))[1], _d(ha)[1], _d(cb)[1],
                              _d(np.asarray(Rb).reshape(-1))[1], _d(hb)[1], margin, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:n]


def rng_uniform(seed, env, counter):
    return load().rpo_rng_uniform(seed, env, counter)


def quat_from_euler(rpy, f32=False):
    q = np.zeros(4)
    load(f32).rpo_quat_from_euler(_d(rpy)[1], q.ctypes.data_as(C.POINTER(C.c_double)))
    return q


def euler_from_quat(q, f32=False):
    e = np.zeros(3)
    load(f32).rpo_euler_from_quat(_d(q)[1], e.ctypes.data_as(C.POINTER(C.c_double)))
    return e


def dial_to_0_1_range(x):
    return load().rpo_dial_to_0_1_range(float(x))


def bench_rollout(kind, seed, actions, n_threads, margin=None, f32=False):
    """cpu_baseline leg: actions [n_envs, n_steps, action_dim] stepped on n_threads threads (envs statically partitioned);
    returns env-steps per second of the stepping phase"""
    lib = load(f32)
    a, ap = _d(actions)
    n_envs, n_steps, na = a.shape
    t = lib.rpo_bench_rollout(KINDS[kind], seed, n_envs, n_steps, na, ap, int(n_threads), -1.0 if margin is None else float(margin))
    if t <= 0:
        raise RuntimeError('rpo_bench_rollout failed')
    return n_envs * n_steps / t
