/* rp_playroom.h — C ABI of the MI355X-native batched playroom simulator (librp_playroom_hip.so).
 *
 * This is the drop-in boundary of SURVEY.md §8b.  The reference has no FFI of its own: its hot path calls the
 * PyBullet client (third-party, CPU) from Python.  Each entry point below REPLACES a group of those calls for N
 * environments at once; the reference interface it stands in for is cited per function
 * (ENV = roboticsPlayroomPybullet/envs/environments.py, IKS = inverseKinematics.py, RWD = playRewardFunc.py).
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions; every function returns 0 on success or a negative rp_status.
 *   - all array arguments are DEVICE pointers to caller-owned, contiguous, row-major [N, dim] buffers
 *     (float32 unless noted), valid until `stream` reaches the call; the library owns only its internal state.
 *   - a handle is bound to one device, is not thread-safe, and enqueues on the caller's stream
 *     (pass torch.cuda.current_stream().cuda_stream); `stream` is a hipStream_t passed as void*.
 *   - no host<->device copies and no synchronisation inside rp_step / rp_reset_to / rp_compute_reward; rp_reset reads one
 *     4-byte counter back per round of settle substeps (see rp_reset).  (Debug path only: with rp_set_fused(h, 1) AND timers
 *     enabled, rp_step waits for its own kernel to read the timer - rp_playroom_debug.h.)
 *   - every entry point makes the handle's device current for the duration of the call and restores the caller's device,
 *     so handles on different devices (or several on one device, each driven on its own stream) can live in one process.
 *   - per-env numerical blow-ups are not errors: they set status[env] != 0 in rp_out.
 */
#ifndef RP_PLAYROOM_H
#define RP_PLAYROOM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rp_sim* rp_handle;

enum rp_status { RP_OK = 0, RP_ERR_ARG = -1, RP_ERR_HIP = -2, RP_ERR_UNSUPPORTED = -3, RP_ERR_STATE_SIZE = -4,
                 RP_ERR_INCOMPLETE = -5 /* rp_reset ran out of rounds with envs still pending (their status[] is set to 4) */ };

/* registered ids in scope (roboticsPlayroomPybullet/__init__.py:92,66,24) and, from SURVEY.md 8f rank 1, the rest of
 * the UR5 one-object play family (__init__.py:72,77,82,87,97; envList.py:101-140): same scene and arm as
 * UR5PlayAbsRPY1Obj-v0, other action types (environments.py:915-981) */
enum rp_env_kind {
  RP_ENV_UR5_PLAY_ABS_RPY_1OBJ = 0, RP_ENV_UR5_REACH = 1, RP_ENV_PANDA_PICK = 2,
  RP_ENV_UR5_PLAY_1OBJ = 3,            /* absolute_quat   [x y z qx qy qz qw grip]  */
  RP_ENV_UR5_PLAY_REL_1OBJ = 4,        /* relative_quat   (added to the measured EE pose, componentwise) */
  RP_ENV_UR5_PLAY_REL_JOINTS_1OBJ = 5, /* relative_joints [dq0..dq5 grip], no IK */
  RP_ENV_UR5_PLAY_ABS_JOINTS_1OBJ = 6, /* absolute_joints [q0..q5 grip], no IK */
  RP_ENV_UR5_PLAY_REL_RPY_1OBJ = 7,    /* relative_rpy    [dx dy dz droll dpitch dyaw grip] */
  RP_ENV_PANDA_PUSH = 8,               /* pandaPush-v0 (__init__.py:19, envList.py:12-16): pandaPick's arm and scene, other ranges */
  /* the Panda in the other two scenes */
  RP_ENV_PANDA_REACH = 9,              /* pandaReach-v0 (__init__.py:4, envList.py:8-10): Panda + default_scene, no object */
  RP_ENV_PANDA_REACH_2D = 10,          /* pandaReach2D-v0 (__init__.py:9, envList.py:24-26): same, goals just above the plane */
  /* the Panda one-object play family (__init__.py:33-64, envList.py:43-88): Panda + complex_scene, one id per action type;
   * joint-space actions carry 7 joints: [q0..q6 grip] */
  RP_ENV_PANDA_PLAY_1OBJ = 11,            /* absolute_quat   */
  RP_ENV_PANDA_PLAY_REL_1OBJ = 12,        /* relative_quat   */
  RP_ENV_PANDA_PLAY_REL_JOINTS_1OBJ = 13, /* relative_joints */
  RP_ENV_PANDA_PLAY_ABS_JOINTS_1OBJ = 14, /* absolute_joints */
  RP_ENV_PANDA_PLAY_ABS_RPY_1OBJ = 15,    /* absolute_rpy    */
  RP_ENV_PANDA_PLAY_REL_RPY_1OBJ = 16,    /* relative_rpy    */
  /* the two-object play ids (__init__.py:29, 41; envList.py:28-41): Panda + complex_scene with two blocks.  Served by the
   * RP_WIDE build of this library (librp_playroom_hip_wide.so: three free bodies in the state record, observations 26 / 18 wide,
   * the drawer in the arm's half of the solver's lane layout); librp_playroom_hip.so answers RP_ERR_UNSUPPORTED for them, and
   * the wide build for every other id. */
  RP_ENV_PANDA_PLAY = 17,                 /* pandaPlay-v0: absolute_quat */
  RP_ENV_PANDA_PLAY_JOINTS = 18,          /* pandaPlayJoints-v0: relative_joints */
  RP_ENV_COUNT = 19
};

/* perform_action's dispatch (ENV:915-934); values of rp_config.action_type */
enum rp_action_type { RP_ACTION_ABSOLUTE_RPY = 0, RP_ACTION_RELATIVE_RPY = 1, RP_ACTION_ABSOLUTE_QUAT = 2, RP_ACTION_RELATIVE_QUAT = 3,
                      RP_ACTION_ABSOLUTE_JOINTS = 4, RP_ACTION_RELATIVE_JOINTS = 5 };

/* rp_config.flags: which of the optional fields are valid.  0 = the id exactly as envList.py registers it. */
enum rp_config_flags {
  RP_CFG_GOAL_RANGE = 1,    /* goal_range_low / goal_range_high    (ENV:495-498 goal draw, ENV:577-581 arm reset target) */
  RP_CFG_OBJ_RANGE = 2,     /* obj_lower_bound / obj_upper_bound   (ENV:527-533 object spawn) */
  RP_CFG_ENV_RANGE = 4,     /* env_range_high                      (ENV:537-540 out-of-bounds re-sample after settling) */
  RP_CFG_REW_THRESH = 8,    /* sparse_rew_thresh                   (ENV:297; ids that are not play ids) */
  RP_CFG_DENSE_REWARD = 16, /* sparse=False: compute_reward = -||ag - dg|| over the whole goal vector (ENV:169-170, 273-275) */
  RP_CFG_ACTION_TYPE = 32,  /* action_type                         (ENV:88-113, 915-981) */
  RP_CFG_CONTACT_MARGIN = 64, /* contact_margin, see below */
  RP_CFG_STATELESS_CONTACTS = 128 /* (no field) rebuild the contact points every substep instead of keeping them across substeps.  Default (flag clear): a
                                   * per-env contact cache (2.1 KB beside the state record) with btPersistentManifold's life cycle - one manifold per object pair
                                   * in creation order, <= 4 points in the two bodies' frames, refreshed every substep, dropped beyond the pair's breaking
                                   * threshold; a box pair makes new points only while the boxes overlap.  The cache is part of the state (rp_get_state /
                                   * rp_set_state rows carry it behind the record).  With the flag: round 3's first model - points
                                   * exist out to the pair's margin, nothing is remembered (13 % faster, further from Bullet: DESIGN.md section 2) */
  ,
  RP_CFG_HULL_GJK = 256,          /* (no field; the DEFAULT since 0.3, accepted for callers written against 0.2) an arm link whose deepest hull vertex lies BESIDE the
                                   * box face it approaches (box edges and corners) gets its contact from GJK's distance phase on hull and box (oracle
                                   * RPO_RULE_GJK): the reference loads mesh colliders (environments.py:397, 409-411) and Bullet runs GJK on their hulls */
  RP_CFG_SPECULATIVE_LIMITS = 1024, /* (no field) round 2's joint-limit rows: a row exists from 0.1 rad (m) BEFORE the limit on and lets the joint close the gap within the
                                   * substep (a contact-like speculative row, erp of the contacts).  Default (flag clear): Bullet's rule as recalled - a row only
                                   * while the limit is violated, erp 0.2 (btMultiBodyJointLimitConstraint::createConstraintRows) - under which a gripper joint whose
                                   * position motor is commanded past its limit (every "open" action, environments.py:1037-1073) chatters at the limit: 1.1 mm for a
                                   * Robotiq pad = 0.026 in obs_quat's gripper entry.  The flag trades that sawtooth for 1e-2 of free-motion divergence from the
                                   * reference step: an A / B switch for learners that see the gripper observation */
  RP_CFG_OBB_EDGES = 512,         /* (no field) round 3's contacts for that case: the link's OBB against the box (SAT + face clipping) instead of GJK on the hull -
                                   * a few per cent faster, further from Bullet (the headline id's block position: 8 cm instead of 2 mm median divergence from the
                                   * reference step over 200 steps) */
  RP_CFG_HULL_EPA = 2048,         /* (no field) where GJK finds the CORES of a link's hull and a box overlapping, depth, normal and witness come from the expanding polytope
                                   * on the two cores (Bullet's btGjkEpaSolver2; oracle RPO_RULE_EPA) instead of the OBB path.  The DEFAULT of the Panda ids (pandaPick: the
                                   * worst arm divergence from the reference step over 200 steps 1.2e-3 -> 5.6e-5 rad), not of the UR5 ids (nothing moves in their table rows;
                                   * 4 % of the headline, 19 % under the literal random-action rollout): this flag turns it on for those too ... */
  RP_CFG_NO_HULL_EPA = 4096       /* ... and this one off for the Panda ids.  Both set: RP_ERR_ARG.  Ignored under RP_CFG_OBB_EDGES (no GJK, no polytope). */
};

typedef struct rp_config {
  int32_t env_kind;      /* rp_env_kind */
  int32_t num_envs;      /* N, 1 .. 4194304 */
  int32_t device;        /* HIP device ordinal */
  int32_t env_offset;    /* global index of env 0 of this handle (multi-GPU shards: rank * N); keys the per-env RNG */
  uint64_t seed;         /* counter RNG: u = f(seed, env_offset + env, draw#) — shard-invariant */
  /* the kwargs an env class of envList.py hands to playEnv.__init__ (ENV:64-67) that reach the simulation; each is read only
   * when its rp_config_flags bit is set.  Kwargs that change the layout of the observation (num_objects, use_orientation,
   * return_velocity, play, arm_type) belong to the id and cannot be overridden. */
  uint32_t flags;
  int32_t action_type;   /* rp_action_type */
  float goal_range_low[3], goal_range_high[3];
  float obj_lower_bound[3], obj_upper_bound[3];
  float env_range_high[3];
  float sparse_rew_thresh;
  /* one distance, in metres, out to which the narrowphase creates contact points for every collider pair (0 .. 0.05).
   * Default (flag clear): per pair, Bullet's contact breaking threshold - gContactBreakingThreshold (0.02) x the smaller of
   * the two collision objects' angular-motion discs (btCollisionDispatcher's CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD
   * default): 1.2 mm for the block, 0.4 - 6 mm for the arm links, 9 mm for the table.  Bullet keeps the points of its
   * persistent manifolds out to that distance, and so does this library's contact cache (default).  Setting this field is a
   * study of the STATELESS contacts (it implies RP_CFG_STATELESS_CONTACTS): points are rebuilt every substep and the margin
   * is the distance within which a point exists.  See DESIGN.md H7. */
  float contact_margin;
} rp_config;

/* per-kind output widths */
typedef struct rp_dims {
  int32_t obs_quat, achieved_goal, desired_goal, controllable_achieved_goal, full_positional_state, joints, velocity,
      observation, target_poses, action;
} rp_dims;

/* outputs of one step / reset: the reference's obs dict (ENV:849-861) + reward/info (ENV:211-214). NULL = skip. */
typedef struct rp_out {
  float* obs_quat;                   /* [N, dims.obs_quat] */
  float* achieved_goal;              /* [N, dims.achieved_goal] */
  float* desired_goal;               /* [N, dims.desired_goal] */
  float* controllable_achieved_goal; /* [N, 4] */
  float* full_positional_state;      /* [N, dims.full_positional_state] */
  float* joints;                     /* [N, 8] */
  float* velocity;                   /* [N, 6] */
  float* observation;                /* [N, dims.observation] */
  int32_t* gripper_proprioception;   /* [N] */
  float* reward;                     /* [N] */
  int32_t* is_success;               /* [N] */
  float* target_poses;               /* [N, dims.target_poses]; written by rp_step only */
  int32_t* status;                   /* [N] bit flags, 0 = ok: 1 non-finite state, 2 an object fell through the scene's ground plate
                                      * (tunnelled: it is below the lowest static collider), 4 rp_reset left this env unfinished,
                                      * 8 (informational, not a fault) the IK of the latest rp_step ran out of its iterations for this env
                                      * (inverseKinematics.py:44-50: 4 x 20; Panda: 200) - its joint targets then hang on the measured joints,
                                      * 16 (informational) one of that IK's stopping tests was decided within 0.5 % of the residual threshold: another
                                      * evaluation order of the same arithmetic may have stopped an iteration apart (~5e-5 rad in the joint targets) */
  float* pack;                       /* [N, dims.obs_quat + dims.achieved_goal + 2]: obs_quat | achieved_goal | reward | is_success
                                      * in one row, the message of the per-step multi-GPU observation gather,
                                      * written by the same kernel so the gather needs no packing pass */
} rp_out;

typedef struct rp_timers {
  float last_step_ms;    /* device time of the most recent rp_step (hipEvent pair on the call's stream) */
  float last_reset_ms;
  uint64_t steps;        /* rp_step calls so far */
  /* averages over the steps recorded since rp_enable_timers (per-launch hipEvent pairs on the call's stream;
   * nothing synchronises inside rp_step, rp_get_timers does) */
  uint32_t steps_timed;
  float avg_step_ms;     /* k_action .. k_calc_state, one rp_step */
  float avg_action_ms;   /* k_action, one launch */
  float avg_prep_ms;     /* k_prep, one launch (12 per step) */
  float avg_solve_ms;    /* k_solve, one launch (12 per step) */
  float avg_obs_ms;      /* k_calc_state, one launch */
} rp_timers;

/* gym.make(id) + playEnv.__init__ + activate_physics_client (ENV:64-170, 218-249): builds N identical worlds. */
int rp_create(const rp_config* cfg, rp_handle* out);
int rp_destroy(rp_handle h);
int rp_get_dims(rp_handle h, rp_dims* dims);

/* playEnv.reset(o=None) (ENV:173-187) for every env whose mask byte is non-zero (mask NULL = all):
 * resample block / arm / goal, 100 settle substeps, repeat while the goal is already satisfied.  The settle substeps run
 * through the step pipeline's kernels over the envs that still need them, in rounds; the host waits on `stream` once per round
 * (1-3 rounds for a few envs, more for the unluckiest env of a large batch), so the call returns when the reset is done.
 * Uses the handle's row workspace: do not overlap with rp_step on another stream. */
int rp_reset(rp_handle h, const uint8_t* mask, const rp_out* out, void* stream);

/* playEnv.reset(o) (ENV:173-187 with ENV:542-556, 575-590): objects and arm placed from an observation vector o [N, n_o]
 * instead of being sampled - object pose from o[11:18] (use_orientation ids) or o[7:10], arm IK target o[0:3] with
 * orientation o[3:7]; no settling; the goal is still drawn.  SURVEY.md 8f rank 2 (state restore). */
int rp_reset_to(rp_handle h, const float* o, int32_t n_o, const uint8_t* mask, const rp_out* out, void* stream);

/* playEnv.reset_goal_pos(goal) (ENV:190-191, 492-516). goal [N, dims.desired_goal] or NULL (random goal).
 * Play envs then overwrite the goal with a random perturbation of the achieved goal, as the reference does. */
int rp_reset_goal(rp_handle h, const float* goal, const uint8_t* mask, void* stream);

/* playEnv.step(action) (ENV:206-214): clip -> absolute_rpy IK (IKS:44-50 / ENV:995-997) -> motor targets
 * (ENV:1010-1073) -> 12 x stepSimulation at 300 Hz (ENV:485-490) -> calc_state (ENV:799-864) -> reward.
 * action [N, dims.action]: x y z roll pitch yaw gripper for the absolute_rpy ids; see rp_env_kind for the others. */
int rp_step(rp_handle h, const float* action, const rp_out* out, void* stream);

/* instance.calc_state() without stepping (ENV:799-864); updates the quaternion sign memory like the reference. */
int rp_calc_state(rp_handle h, const rp_out* out, void* stream);

/* playEnv.compute_reward(achieved_goal, desired_goal) (ENV:278-304, RWD:66-77) for M rows. */
int rp_compute_reward(rp_handle h, const float* achieved_goal, const float* desired_goal, float* reward, int32_t m, void* stream);
/* playEnv.compute_reward_sparse (environments.py:278-304): the sparse formula whatever `sparse` the env was built with (the reference rebinds only
 * compute_reward when sparse=False, environments.py:169-170; compute_reward_sparse stays callable) */
int rp_compute_reward_sparse(rp_handle h, const float* achieved_goal, const float* desired_goal, float* reward, int32_t m, void* stream);

/* full simulator state (positions, velocities, motor targets, goal, quaternion memory, RNG counters - the first 512 bytes of a row: the state record -
 * and, unless RP_CFG_STATELESS_CONTACTS, the env's contact cache behind it): the explicit save/restore the reference lacks.
 * src_env_count == 1 broadcasts one env to all N.  A row whose cache part is all zeros is a state without contact history. */
size_t rp_state_bytes(rp_handle h);           /* bytes per env (a row of the buffers below) */
int rp_get_state(rp_handle h, void* dst, void* stream);
int rp_set_state(rp_handle h, const void* src, int32_t src_env_count, void* stream);

/* ---- images and ray queries ----------------------------------------------------------------------
 * The reference renders obs['img'] with PyBullet's OpenGL rasteriser and asks Bullet for one ray per step; this library casts rays
 * against the boxes and spheres of its contact model, coloured like the reference's visual shapes (flat shading, no textures). */
typedef struct rp_camera {
  float eye[3], target[3], up[3];   /* computeViewMatrix(eye, target, up) */
  float fov_deg, aspect;            /* computeProjectionMatrixFOV(fov, aspect, near, far); near / far only clip: rays run 0 .. 10 m */
  int32_t mode;                     /* 0 = this camera; 1 = gripper camera (ENV:33-49): eye at the EE link, looking along its -z, up = its x */
} rp_camera;
/* the reference's fixed camera (ENV:21-30): computeViewMatrixFromYawPitchRoll(target [0, 0.25, 0], distance 1.3, yaw -30, pitch -30, roll 0,
 * upAxisIndex 2), computeProjectionMatrixFOV(fov 50, aspect 1, near 0.01, far 10) */
int rp_default_camera(rp_camera* cam);
/* p.computeViewMatrixFromYawPitchRoll's camera placement (degrees, upAxisIndex 2) */
int rp_camera_from_yaw_pitch_roll(const float target[3], float distance, float yaw_deg, float pitch_deg, float roll_deg, rp_camera* cam);
/* instance.calc_state()'s img (ENV:841-845: getCameraImage(width, height, view, projection)[2][:, :, :3]) for envs [first_env, first_env +
 * num_envs): rgb [num_envs, height, width, 3] uint8, row 0 = top.  cam NULL = rp_default_camera.  sub_goal (may be NULL):
 * [num_envs, dims.achieved_goal] - visualise_sub_goal(sub_goal, 'achieved_goal') (ENV:606-690): the objects and, for the play ids,
 * drawer / door / button / dial drawn a second time, half transparent, at the poses the vectors name. */
int rp_render(rp_handle h, const rp_camera* cam, int32_t width, int32_t height, int32_t first_env, int32_t num_envs, uint8_t* rgb,
              const float* sub_goal, void* stream);
/* ... with the ghost ARM of visualise_sub_goal(sub_goal, 'controllable_achieved_goal' / 'full_positional_state') (ENV:623-637, 671-674): ghost_arm (may be NULL)
 * [num_envs, 8] = EE position 3, orientation quaternion 4 (xyzw), gripper 1 (unused, as in the reference's reset_arm) - a second arm, half transparent, at the joints one
 * default IK call from the rest pose finds for that pose (ENV:575-590).  Panda models only: RP_ERR_UNSUPPORTED for a UR5 (the reference raises NotImplementedError). */
int rp_render_ex(rp_handle h, const rp_camera* cam, int32_t width, int32_t height, int32_t first_env, int32_t num_envs, uint8_t* rgb,
                 const float* sub_goal, const float* ghost_arm, void* stream);
/* bullet_client.rayTest(from, to) (ENV:738-741) for k rays per env: from / to [N, k, 3] world coordinates.  Outputs (each may be NULL):
 * hit_fraction [N, k] (1 on a miss), collider [N, k] (index into the model's collider table, -1 on a miss), link [N, k] (Bullet link
 * index of an arm collider, -1 otherwise), hit_position [N, k, 3], hit_normal [N, k, 3].  A ray that starts inside a shape does not hit it. */
int rp_ray_test(rp_handle h, const float* from, const float* to, int32_t k, float* hit_fraction, int32_t* collider, int32_t* link,
                float* hit_position, float* hit_normal, void* stream);

int rp_get_timers(rp_handle h, rp_timers* t);
/* on = number of rp_step calls to keep per-launch timings for (a ring); 0 disables */
int rp_enable_timers(rp_handle h, int32_t on);
const char* rp_last_error(rp_handle h);
const char* rp_version(void);

#ifdef __cplusplus
}
#endif
#endif
