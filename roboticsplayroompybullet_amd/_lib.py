"""ctypes binding of the C ABI in include/rp_playroom.h (librp_playroom_hip.so, built in-tree for gfx950).

There is no CPU fallback: if the HIP library is missing or fails to load, every entry point raises.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_PATH = os.environ.get('RP_PLAYROOM_LIB', os.path.join(CSRC, 'librp_playroom_hip.so'))   # env override: profiling builds
# the RP_WIDE build of the same sources: the two-object play ids (three free bodies in the record, the drawer in the arm's DPP row)
WIDE_LIB_PATH = os.path.join(CSRC, 'librp_playroom_hip_wide.so')
WIDE_IDS = ('pandaPlay-v0', 'pandaPlayJoints-v0')

ENV_KINDS = {'UR5PlayAbsRPY1Obj-v0': 0, 'UR5Reach-v0': 1, 'pandaPick-v0': 2,
             # the rest of the UR5 one-object play family (same scene, other action types)
             'UR5Play1Obj-v0': 3, 'UR5PlayRel1Obj-v0': 4, 'UR5PlayRelJoints1Obj-v0': 5, 'UR5PlayAbsJoints1Obj-v0': 6,
             'UR5PlayRelRPY1Obj-v0': 7,
             'pandaPush-v0': 8,                # pandaPick's arm and scene, other ranges
             'pandaReach-v0': 9, 'pandaReach2D-v0': 10,                     # Panda + default_scene
             # the Panda one-object play family: Panda + complex_scene
             'pandaPlay1Obj-v0': 11, 'pandaPlayRel1Obj-v0': 12, 'pandaPlayRelJoints1Obj-v0': 13, 'pandaPlayAbsJoints1Obj-v0': 14,
             'pandaPlayAbsRPY1Obj-v0': 15, 'pandaPlayRelRPY1Obj-v0': 16,
             'pandaPlay-v0': 17, 'pandaPlayJoints-v0': 18}      # two blocks: served by the wide build (WIDE_LIB_PATH)


ACTION_TYPES = {'UR5PlayAbsRPY1Obj-v0': 'absolute_rpy', 'UR5Reach-v0': 'absolute_rpy', 'pandaPick-v0': 'absolute_rpy',
                'UR5Play1Obj-v0': 'absolute_quat', 'UR5PlayRel1Obj-v0': 'relative_quat', 'UR5PlayRelJoints1Obj-v0': 'relative_joints',
                'UR5PlayAbsJoints1Obj-v0': 'absolute_joints', 'UR5PlayRelRPY1Obj-v0': 'relative_rpy', 'pandaPush-v0': 'absolute_rpy',
                'pandaReach-v0': 'absolute_rpy', 'pandaReach2D-v0': 'absolute_rpy',
                'pandaPlay1Obj-v0': 'absolute_quat', 'pandaPlayRel1Obj-v0': 'relative_quat', 'pandaPlayRelJoints1Obj-v0': 'relative_joints',
                'pandaPlayAbsJoints1Obj-v0': 'absolute_joints', 'pandaPlayAbsRPY1Obj-v0': 'absolute_rpy',
                'pandaPlayRelRPY1Obj-v0': 'relative_rpy', 'pandaPlay-v0': 'absolute_quat', 'pandaPlayJoints-v0': 'relative_joints'}


class RpConfig(C.Structure):
    _fields_ = [('env_kind', C.c_int32), ('num_envs', C.c_int32), ('device', C.c_int32), ('env_offset', C.c_int32),
                ('seed', C.c_uint64), ('flags', C.c_uint32), ('action_type', C.c_int32),
                ('goal_range_low', C.c_float * 3), ('goal_range_high', C.c_float * 3),
                ('obj_lower_bound', C.c_float * 3), ('obj_upper_bound', C.c_float * 3), ('env_range_high', C.c_float * 3),
                ('sparse_rew_thresh', C.c_float), ('contact_margin', C.c_float)]


# rp_config_flags / rp_action_type (include/rp_playroom.h)
CFG_GOAL_RANGE, CFG_OBJ_RANGE, CFG_ENV_RANGE, CFG_REW_THRESH, CFG_DENSE_REWARD, CFG_ACTION_TYPE, CFG_CONTACT_MARGIN, CFG_STATELESS_CONTACTS, CFG_HULL_GJK, CFG_OBB_EDGES, CFG_SPECULATIVE_LIMITS, CFG_HULL_EPA, CFG_NO_HULL_EPA = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096
ACTION_TYPE_CODES = {'absolute_rpy': 0, 'relative_rpy': 1, 'absolute_quat': 2, 'relative_quat': 3, 'absolute_joints': 4, 'relative_joints': 5}


class RpDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('obs_quat', 'achieved_goal', 'desired_goal', 'controllable_achieved_goal',
                                         'full_positional_state', 'joints', 'velocity', 'observation', 'target_poses', 'action')]


class RpOut(C.Structure):
    _fields_ = [('obs_quat', C.c_void_p), ('achieved_goal', C.c_void_p), ('desired_goal', C.c_void_p),
                ('controllable_achieved_goal', C.c_void_p), ('full_positional_state', C.c_void_p), ('joints', C.c_void_p),
                ('velocity', C.c_void_p), ('observation', C.c_void_p), ('gripper_proprioception', C.c_void_p),
                ('reward', C.c_void_p), ('is_success', C.c_void_p), ('target_poses', C.c_void_p), ('status', C.c_void_p),
                ('pack', C.c_void_p)]


class RpCamera(C.Structure):
    _fields_ = [('eye', C.c_float * 3), ('target', C.c_float * 3), ('up', C.c_float * 3), ('fov_deg', C.c_float), ('aspect', C.c_float),
                ('mode', C.c_int32)]


class RpTimers(C.Structure):
    _fields_ = [('last_step_ms', C.c_float), ('last_reset_ms', C.c_float), ('steps', C.c_uint64), ('steps_timed', C.c_uint32),
                ('avg_step_ms', C.c_float), ('avg_action_ms', C.c_float), ('avg_prep_ms', C.c_float), ('avg_solve_ms', C.c_float),
                ('avg_obs_ms', C.c_float)]


EXPORTS = ['rp_create', 'rp_destroy', 'rp_get_dims', 'rp_reset', 'rp_reset_to', 'rp_reset_goal', 'rp_step', 'rp_calc_state',
           'rp_compute_reward', 'rp_compute_reward_sparse', 'rp_state_bytes', 'rp_get_state', 'rp_set_state', 'rp_get_timers', 'rp_enable_timers',
           'rp_last_error', 'rp_version', 'rp_default_camera', 'rp_camera_from_yaw_pitch_roll', 'rp_render', 'rp_render_ex', 'rp_ray_test']
# include/rp_playroom_debug.h: test / tuning hooks
DEBUG_EXPORTS = ['rp_set_fused', 'rp_set_groups', 'rp_set_debug_flags', 'rp_debug_substep', 'rp_debug_row_counts', 'rp_debug_reset_rounds', 'rp_debug_ghost_joints']

_lib = None
_libs = {}


def build(force=False):
    """Compile the HIP library in-tree (hipcc --offload-arch=gfx950); cross-compiles without a GPU."""
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(WIDE_LIB_PATH):
        subprocess.run(['make', '-C', CSRC, '-s'], check=True)
    return LIB_PATH


def load(wide=False):
    """the HIP library (wide=True: its RP_WIDE build, which serves WIDE_IDS)"""
    global _lib
    path = WIDE_LIB_PATH if wide else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError('%s is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` '
                           'or `make -C roboticsplayroompybullet_amd/csrc`. There is no CPU fallback.' % (os.path.basename(path), path))
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.rp_create.argtypes = [C.POINTER(RpConfig), C.POINTER(vp)]
    lib.rp_destroy.argtypes = [vp]
    lib.rp_get_dims.argtypes = [vp, C.POINTER(RpDims)]
    lib.rp_reset.argtypes = [vp, vp, C.POINTER(RpOut), vp]
    lib.rp_reset_to.argtypes = [vp, vp, C.c_int32, vp, C.POINTER(RpOut), vp]
    lib.rp_reset_goal.argtypes = [vp, vp, vp, vp]
    lib.rp_step.argtypes = [vp, vp, C.POINTER(RpOut), vp]
    lib.rp_calc_state.argtypes = [vp, C.POINTER(RpOut), vp]
    lib.rp_compute_reward.argtypes = [vp, vp, vp, vp, C.c_int32, vp]
    lib.rp_compute_reward_sparse.argtypes = [vp, vp, vp, vp, C.c_int32, vp]
    lib.rp_state_bytes.argtypes = [vp]
    lib.rp_state_bytes.restype = C.c_size_t
    lib.rp_get_state.argtypes = [vp, vp, vp]
    lib.rp_set_state.argtypes = [vp, vp, C.c_int32, vp]
    lib.rp_get_timers.argtypes = [vp, C.POINTER(RpTimers)]
    lib.rp_enable_timers.argtypes = [vp, C.c_int32]
    lib.rp_last_error.argtypes = [vp]
    lib.rp_last_error.restype = C.c_char_p
    lib.rp_version.restype = C.c_char_p
    lib.rp_default_camera.argtypes = [C.POINTER(RpCamera)]
    lib.rp_camera_from_yaw_pitch_roll.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(RpCamera)]
    lib.rp_render.argtypes = [vp, C.POINTER(RpCamera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
    lib.rp_render_ex.argtypes = [vp, C.POINTER(RpCamera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]
    lib.rp_ray_test.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    lib.rp_debug_substep.argtypes = [vp, C.c_int32, C.POINTER(C.c_float)]
    lib.rp_set_fused.argtypes = [vp, C.c_int32]
    lib.rp_set_groups.argtypes = [vp, C.c_int32]
    lib.rp_set_debug_flags.argtypes = [vp, C.c_int32]
    lib.rp_debug_ghost_joints.argtypes = [vp, C.c_void_p, C.c_int32]
    lib.rp_debug_row_counts.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.rp_debug_reset_rounds.argtypes = [vp]
    _libs[path] = lib
    if not wide:
        _lib = lib
    return lib


def check(lib, handle, code, what):
    if code != 0:
        msg = lib.rp_last_error(handle)
        raise RuntimeError('%s failed (%d): %s' % (what, code, msg.decode() if msg else ''))
