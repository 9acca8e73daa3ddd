/* oracle/rp_oracle.h — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * PARITY UNPINNED for physics: the reference's physics lives in PyBullet (unpinned third-party dependency,
 * absent from /root/reference and from this image).  The harness arithmetic (action
 * mapping, observation assembly, rewards, reset sampling) IS pinned against tests/golden JSON fixtures, which were
 * produced by executing the reference's own Python (tests/golden/make_goldens.py).
 *
 * One rpo_env = one reference `instance` + `playEnv` (environments.py:58-1073) without Python.
 * All I/O is double regardless of the internal `real` (build with -DRP_FLOAT for an fp32 oracle). */
#ifndef RP_ORACLE_H
#define RP_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rpo_env rpo_env;

/* observation bundle of one env, reference calc_state() (environments.py:799-864); sizes are maxima */
typedef struct rpo_obs {
  double obs_quat[26], achieved_goal[18], desired_goal[18], controllable_achieved_goal[4];
  double full_positional_state[26], joints[8], velocity[6], observation[25];
  int gripper_proprioception;
  int n_obs, n_ag, n_fps, n_observation;
} rpo_obs;

/* raw world read-back feeding calc_state (what the reference pulls from PyBullet getters) */
typedef struct rpo_readings {
  double ee_pos[3], ee_orn[4], ee_lin[3], ee_ang[3], grip_q, joints[8];
  int proprio;
  double block_pos[3], block_orn[4], block_vel[3], drawer_y, door_q, button_q, dial_q;
  double block2_pos[3], block2_orn[4], block2_vel[3];      /* second object of the two-object play ids */
} rpo_readings;

rpo_env* rpo_create(int kind /*0 U, 1 R, 2 P, 3 Q, 4 V, 5 W (rp_model.h)*/, unsigned long long seed, int env_index);
/* perform_action's dispatch (environments.py:915-934); default RPO_ACT_ABS_RPY.  Action length: rpo_action_dim. */
enum { RPO_ACT_ABS_RPY = 0, RPO_ACT_REL_RPY = 1, RPO_ACT_ABS_QUAT = 2, RPO_ACT_REL_QUAT = 3, RPO_ACT_ABS_JOINTS = 4, RPO_ACT_REL_JOINTS = 5 };
void rpo_set_action_type(rpo_env* e, int action_type);
/* what the fast model shares with the frozen reference step (rp_oracle.c RPO_RULE_*; default: all of them = 247, rule 0 = round 2's model).  1: Bullet's
 * non-contact row order, walked in alternating direction; 2: joint-limit rows only while violated, erp 0.2; 4: arm links touch static boxes with their hulls'
 * vertices; 16: an arm link's box touches a box only on overlap; 32: box-box points in btBoxBoxDetector's order; 64: a contact acts at its point on A / on B;
 * 128: torsional friction rows of links with spinning_friction */
void rpo_set_rule(rpo_env* e, int rule);
int rpo_get_rule(const rpo_env* e);
int rpo_pair_table(const rpo_env* e, int* out);      /* the baked candidate pairs: (a, b) collider indices; returns their number */
void rpo_set_margin(rpo_env* e, double margin);                     /* one contact margin for all pairs, metres (default: per pair, rp_model.col_thr) */
void rpo_set_reward_cfg(rpo_env* e, double sparse_rew_thresh, int dense);   /* environments.py:66, 169-170 */
int rpo_action_dim(const rpo_env* e);
void rpo_get_config(const rpo_env* e, double* out27);   /* flags, ranges, action-space high (test hook) */
void rpo_set_ranges(rpo_env* e, const double* goal_lo, const double* goal_hi, const double* obj_lo, const double* obj_hi, const double* env_hi);
/* test hook: IK target (position, quaternion) of a pose-type action given the measured EE link pose */
void rpo_action_target(int action_type, const double* action8, const double* ee_pos, const double* ee_orn, double* pos, double* quat);
void rpo_destroy(rpo_env*);
int rpo_nv(const rpo_env*);
int rpo_n_arm(const rpo_env*);

/* playEnv.reset(o=None) (environments.py:173-187).  If `u` is non-NULL the uniforms are taken from it in the
 * order np.random would be consumed (returns how many were used), else from the counter RNG. */
int rpo_reset(rpo_env*, const double* u, int n_u, rpo_obs* out);
/* playEnv.reset(o): objects and arm from an observation vector (environments.py:173-187, 519-603); o needs 18 entries (U), 10 (P), 3 (R) */
int rpo_reset_to(rpo_env* e, const double* o, int n_o, const double* u, int n_u, rpo_obs* out);
void rpo_reset_samples(const rpo_env*, const double* u, double* block_pos, double* arm_target);
/* instance.reset_goal_pos(goal) (environments.py:492-516). goal may be NULL. */
void rpo_reset_goal(rpo_env*, const double* goal, const double* u, int n_u);
/* playEnv.step (environments.py:206-214) */
void rpo_step(rpo_env*, const double* action, rpo_obs* out, double* reward, int* is_success, double* target_poses);
void rpo_calc_state(rpo_env*, rpo_obs* out);
double rpo_compute_reward(const rpo_env*, const double* ag, const double* dg);
void rpo_read_world(rpo_env*, rpo_readings*);
void rpo_assemble_obs(rpo_env*, const rpo_readings*, rpo_obs* out);   /* stateful in play mode (quaternion memory) */
void rpo_quat_from_euler(const double* rpy, double* q);
void rpo_euler_from_quat(const double* q, double* rpy);
double rpo_dial_to_0_1_range(double x);

/* pieces, exposed for unit tests and goldens */
void rpo_perform_action(rpo_env*, const double* action_clipped, double* target_poses);   /* environments.py:915-1073 */
void rpo_goto_joint_poses(rpo_env*, const double* joint_poses, int has_gripper, double gripper, double* target_poses);
void rpo_ik(const rpo_env*, const double* pos, const double* quat, const double* q_seed, int max_iter, double* q_out);
void rpo_calc_angles(const rpo_env*, const double* pos, const double* quat, const double* current, double* q_out); /* inverseKinematics.py:44-50 */
void rpo_substep(rpo_env*);                                /* one stepSimulation() */
void rpo_run_simulation(rpo_env*);                         /* environments.py:485-490 */

/* raw state access: q[n_arm] qd[n_arm] | per free body pos3 quat4 vel3 omega3 | joint1 q, qd */
int rpo_state_size(const rpo_env*);
void rpo_get_state(const rpo_env*, double* s);
void rpo_set_state(rpo_env*, const double* s);
void rpo_get_motor(const rpo_env*, int* mode, double* target, double* maximp);
void rpo_set_goal(rpo_env*, const double* goal);
void rpo_clear_quat_memory(rpo_env*);

/* kinematics/dynamics probes */
void rpo_site_pose(const rpo_env*, int site, double* pos, double* quat, double* linvel, double* angvel);
void rpo_mass_matrix_inv(rpo_env*, double* Minv /* nv*nv, arm block via unit impulse responses */);
void rpo_forward_dynamics(rpo_env*, double* qdd /* n_arm */);
int rpo_contacts(rpo_env*, double* out /* per contact: colA colB px py pz nx ny nz dist */, int max);
int rpo_last_num_tors(const rpo_env* e);        /* torsional friction rows of the latest substep */
int rpo_cache_size(const rpo_env* e, int* points);     /* RPO_RULE_PERSIST: cached manifolds (empty ones included), their points */
/* the contact cache in the HIP library's row layout (704 words, rp_kernels.cuh PMC_*): export, and import of a row (e.g. the device's own) - tests only */
int rpo_cache_row_words(void);
int rpo_get_cache_row(const rpo_env* e, float* row);
int rpo_set_cache_row(rpo_env* e, const float* row);
void rpo_gjk_stats(long* out8, int reset);
void rpo_epa_stats(long* out4, int reset);       /* RPO_RULE_EPA counters of the calling thread: calls, rounds, converged, gave up */       /* GJK counters of the calling thread: calls, rounds, two-point seeds, results 1 / 0 / -1, tetrahedra, rounds of the "apart" exits */
void rpo_shift_free_body(rpo_env* e, int k, double dx, double dy, double dz);      /* test hook: moves a free body, keeps the contact cache */
int rpo_last_num_rows(const rpo_env*);
int rpo_arm_table(const rpo_env* e, double* out);                      /* [n_arm][6]: jtype, lower, upper, body mass, Bullet joint index, parent dof */
int rpo_collider_dynamics(const rpo_env* e, double* out);              /* [n_col][6]: body, friction, body mass, contact stiffness, damping, breaking threshold */
void rpo_set_arm_q(rpo_env* e, const double* q);                      /* arm joints to q, at rest */
int rpo_rest_pose(const rpo_env* e, double* out);                     /* the arm's rest joints (environments.py:361, 371), n_arm values; returns how many the reference resets (6 UR5, 8 Panda) */
void rpo_set_proprioception_boxes(int on);                            /* test hook, process-wide: gripper_proprioception's ray against the links' boxes (rounds 1 - 5) instead of their hulls */
int rpo_residual_substeps(const rpo_env* e);                          /* substeps since creation solved in the residual form (rule bit 262144) */
long rpo_ik_iterations(int reset);                                    /* loop passes of the IK in the calling thread since the last reset of the counter (a pass that ends at the residual test included) */
int rpo_contact_substeps(const rpo_env* e);                           /* substeps since creation whose solve had a contact row */
int rpo_box_box(const double* ca, const double* Ra, const double* ha, const double* cb, const double* Rb, const double* hb,
                double margin, double* out /* per point: p3 n3 dist */);
double rpo_rng_uniform(unsigned long long seed, unsigned env_index, unsigned counter);

int rpo_hull_vertices_world(rpo_env* e, int c, double* out, int max);      /* world-frame hull vertices of an arm collider (0: none) */
int rpo_collider_poses(rpo_env* e, double* out12_per_collider);      /* world R (row-major) and p of every collider, returns the count */
int rpo_collider_table(const rpo_env* e, double* out9_per_collider); /* type, he3, rgb3, toggle, link */

/* bench.py's cpu_baseline: n_envs envs over n_threads threads (static partition), n_steps steps each of actions [n_envs][n_steps][action_dim];
 * returns wall seconds of the stepping phase (resets excluded).  margin < 0 keeps the default. */
double rpo_bench_rollout(int kind, unsigned long long seed, int n_envs, int n_steps, int action_dim, const double* actions, int n_threads, double margin);

#ifdef __cplusplus
}
#endif

#endif
