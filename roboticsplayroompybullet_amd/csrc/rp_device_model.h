/* rp_device_model.h — fp32 device-side copy of the baked model tables (rp_model.h) plus derived tree masks.
 * One DevModel per handle lives in device memory; kernels read it through wave-uniform (scalar) loads. */
#ifndef RP_DEVICE_MODEL_H
#define RP_DEVICE_MODEL_H
#include <stdint.h>
#include <math.h>
#include <string.h>

#include "rp_model.h"

#define RP_REC_FLOATS 128 /* per-env state record, 512 B: one coalesced wave load */

/* state record layout (floats) */
#ifndef RP_WIDE
#define ST_NARM 12       /* stride of the per-dof arrays */
#define ST_NFREE 2       /* free-body slots in the record */
#define ST_Q 0
#define ST_QD 12
#define ST_FREE 24       /* per free body 13: pos3 quat4 vel3 om3 */
#define ST_JQ 50
#define ST_JQD 53
#define ST_MMODE 56
#define ST_MTARGET 68
#define ST_MMAXIMP 80
#define ST_GOAL 92
#define ST_LAST_EE_Q 103 /* quaternion sign memory (environments.py:868-894) */
#define ST_LAST_BLK_Q 107
#define ST_LAST_AG_Q 111
#define ST_HAVE_LAST 115
#define ST_RNG 116       /* uint32 draw counter (bit pattern) */
#define ST_NGOAL 117
#define ST_STATUS 118
#else
/* RP_WIDE build (librp_playroom_hip_wide.so): the two-object play ids - Panda only (9 arm dofs), three free bodies (block,
 * block, drawer), 18-wide goal.  Same 128-float record, other offsets; lane layout: rp_kernels.cuh lane_pos.  The sign memory keeps obs[3:7], obs[11:15] (= ag[3:7]:
 * the same quaternion with the same history, so one copy serves both), obs[19:23] and ag[10:14]. */
#define ST_NARM 9
#define ST_NFREE 3
#define ST_Q 0
#define ST_QD 9
#define ST_FREE 18
#define ST_JQ 57
#define ST_JQD 60
#define ST_MMODE 63
#define ST_MTARGET 72
#define ST_MMAXIMP 81
#define ST_GOAL 90
#define ST_LAST_EE_Q 108
#define ST_LAST_BLK_Q 112
#define ST_LAST_AG_Q 112
#define ST_LAST_OBS19 116
#define ST_LAST_AG10 120
#define ST_HAVE_LAST 124
#define ST_RNG 125
#define ST_NGOAL 126
#define ST_STATUS 127
#endif

typedef struct DevModel {
  int kind, n_arm, n_free, n_j1, n_col, n_pair, nv, nbody, n_site;
  int arm_type, scene, drawer_free, free_row0;
  int play, use_orientation, return_velocity, num_objects, n_goal_init;
  int n_obs, n_ag, n_fps, n_observation, n_target;
  int action_type, n_action;    /* RP_ACT_* (environments.py:915-934) and the action length: 7, 8 (quaternion types) or n_target + 1 (joint types) */
  int arm_parent[RP_MAX_ARM], arm_jtype[RP_MAX_ARM], arm_limited[RP_MAX_ARM];
  uint32_t arm_anc[RP_MAX_ARM]; /* bit k set: dof k is an ancestor-or-self of body i */
  uint32_t arm_sub[RP_MAX_ARM]; /* bit j set: body j is in the subtree of body i (incl. i) */
  float arm_jpos[RP_MAX_ARM][3], arm_jrot[RP_MAX_ARM][9], arm_axis[RP_MAX_ARM][3];
  float arm_mass[RP_MAX_ARM], arm_com[RP_MAX_ARM][3], arm_inertia[RP_MAX_ARM][9];
  float arm_lower[RP_MAX_ARM], arm_upper[RP_MAX_ARM];
  float base_pos[3], base_rot[9], rest[RP_MAX_ARM];
  int site_body[RP_MAX_SITE];
  float site_pos[RP_MAX_SITE][3], site_rot[RP_MAX_SITE][9];
  int ee_chain; /* number of serial dofs from the base to the EE site body */
  float free_mass[RP_MAX_FREE], free_inertia[RP_MAX_FREE][3], free_pos0[RP_MAX_FREE][3], free_quat0[RP_MAX_FREE][4];
  int free_rot_locked[RP_MAX_FREE];
  int j1_type[RP_MAX_J1], j1_has_pos_motor[RP_MAX_J1];
  float j1_pos[RP_MAX_J1][3], j1_rot[RP_MAX_J1][9], j1_axis[RP_MAX_J1][3], j1_minv[RP_MAX_J1];
  float j1_motor_target[RP_MAX_J1], j1_motor_maximp[RP_MAX_J1];
  int col_body[RP_MAX_COL], col_type[RP_MAX_COL], col_link[RP_MAX_COL], col_obj[RP_MAX_COL];
  float col_he[RP_MAX_COL][3], col_pos[RP_MAX_COL][3], col_rot[RP_MAX_COL][9], col_friction[RP_MAX_COL];
  unsigned char pair[RP_MAX_PAIR][2];
  /* gripper dofs (environments.py:1037-1073) */
  int d_grip_obs;                 /* dof whose position is the gripper observation (UR5 joint 18, Panda joint 9) */
  int d18, d20, d12, d15, d10, d13, d9p, d10p;
  int joints_dof[8];              /* dof of Bullet joints 0..7 or -1 (fixed) */
  float goal_lo[3], goal_hi[3], obj_lo[3], obj_hi[3], env_hi[3];
  float rew_thresh;               /* sparse_rew_thresh (environments.py:297) */
  int dense_reward;               /* sparse=False: compute_reward = -distance (environments.py:169-170, 273-275) */
  float col_rgb[RP_MAX_COL][3];   /* rendering: colour of the collider's visual shape; col_toggle 1 globe (button), 2 grill (dial) */
  int col_toggle[RP_MAX_COL];
  float col_stiff[RP_MAX_COL], col_damp[RP_MAX_COL];   /* URDF <contact> stiffness / damping of the collider's link, 0 = none */
  float col_spin[RP_MAX_COL];     /* URDF <contact> spinning_friction of the collider's link (the gripper links: 0.1), 0 = none: torsional friction rows */
  float col_margin[RP_MAX_COL];   /* distance out to which a collider's contact points exist; a pair's margin is the smaller of the two.
                                   * Default: Bullet's relative contact breaking threshold of the collider's object (rp_model.col_thr);
                                   * rp_config.contact_margin replaces it by one value for all */
  float boxbox_margin;            /* >= 0: the margin of box-against-box pairs (default 0: points only while the boxes overlap, as btBoxBoxDetector makes them); < 0: the pair's
                                   * margin like every other pair (rp_config.contact_margin given: one value for all) */
  float floor_z;                  /* bottom of the lowest static collider: an object below it has left the scene (status bit 2) */
  /* joint clamps of goto_joint_poses (environments.py:1015-1021) */
  float ll[7], ul[7], inc[7];
  int spec_limits;                /* RP_CFG_SPECULATIVE_LIMITS: round 2's joint-limit rows (oracle rule without RPO_RULE_LIMIT) */
  int gjk;                        /* GJK's distance phase where a hull's deepest vertex lies beside the box face (collide(); oracle RPO_RULE_GJK) */
  int epa;                        /* ... and the expanding polytope where it finds the cores overlapping (hull_epa16; oracle RPO_RULE_EPA) */
  int persist;                    /* unless RP_CFG_STATELESS_CONTACTS: collide() keeps its manifolds in pmcache (rp_kernels.cuh PMC_*) */
  float* pmcache;                 /* [N][PMC_FLOATS], device memory owned by the handle */
  /* convex-hull vertices of the arm links' collision meshes (generated/rp_hullverts_gen.h): device pointer to the arm's table (x, y, z, 0 in the owning
   * body's frame), per collider the first vertex and the count (0 = no hull).  rp_create uploads the table and sets the pointer. */
  const float* hullv;
  int hull_off[RP_MAX_COL], hull_cnt[RP_MAX_COL];
  /* support-vertex candidate tables of those hulls (generated/rp_hullcells_gen.h; rp_kernels.cuh hcell_of): hcv = every cell's candidates as (x, y, z, vertex number) -
   * device memory, expanded from the baked vertex numbers by rp_create -, hco = per hull RP_HCELL_N + 1 offsets into it, hcell_first[collider] = the hull's place in hco (-1: no hull) */
  const float* hcv;
  const int* hco;
  int hcell_first[RP_MAX_COL];
  /* face planes of the hulls for the ray caster (generated/rp_hullplanes_gen.h; rp_render.cuh rc_ray_hull): (nx, ny, nz, w), n . x + w <= 0 inside, in the COLLIDER's frame
   * (rp_create takes them there from the body frame they are baked in); per collider the first plane and the count (0: the collider is drawn as its box / sphere) */
  const float* hpl;
  int hpl_off[RP_MAX_COL], hpl_cnt[RP_MAX_COL];
} DevModel;

static inline int rp_dm_dof_of_joint(const rp_model* m, int j) {
  for (int i = 0; i < m->n_arm; i++)
    if (m->arm_bullet_index[i] == j) return i;
  return -1;
}

static inline void rp_mat_to_quat(const double* M, float* q) { /* btMatrix3x3::getRotation */
  double tr = M[0] + M[4] + M[8], t[4];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    t[3] = s * 0.5; s = 0.5 / s;
    t[0] = (M[7] - M[5]) * s; t[1] = (M[2] - M[6]) * s; t[2] = (M[3] - M[1]) * s;
  } else {
    int i = M[0] < M[4] ? (M[4] < M[8] ? 2 : 1) : (M[0] < M[8] ? 2 : 0);
    int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = sqrt(M[4 * i] - M[4 * j] - M[4 * k] + 1.0);
    t[i] = s * 0.5; s = 0.5 / s;
    t[3] = (M[3 * k + j] - M[3 * j + k]) * s;
    t[j] = (M[3 * j + i] + M[3 * i + j]) * s;
    t[k] = (M[3 * k + i] + M[3 * i + k]) * s;
  }
  for (int a = 0; a < 4; a++) q[a] = (float)t[a];
}

static inline void rp_build_dev_model(const rp_model* m, DevModel* d) {
  memset(d, 0, sizeof(*d));
  d->kind = m->kind; d->n_arm = m->n_arm; d->n_free = m->n_free; d->n_j1 = m->n_joint1; d->n_col = m->n_col;
  d->n_pair = m->n_pair; d->n_site = m->n_site;
  d->nv = m->n_arm + 6 * m->n_free + m->n_joint1;
  d->nbody = 1 + m->n_arm + m->n_free + m->n_joint1;
  for (int i = 0; i < m->n_arm; i++) {
    d->arm_parent[i] = m->arm_parent[i]; d->arm_jtype[i] = m->arm_jtype[i];
    d->arm_limited[i] = m->arm_lower[i] < m->arm_upper[i];
    d->arm_mass[i] = (float)m->arm_mass[i]; d->arm_lower[i] = (float)m->arm_lower[i]; d->arm_upper[i] = (float)m->arm_upper[i];
    d->rest[i] = (float)m->rest[i];
    for (int k = 0; k < 3; k++) { d->arm_jpos[i][k] = (float)m->arm_jpos[i][k]; d->arm_axis[i][k] = (float)m->arm_axis[i][k]; d->arm_com[i][k] = (float)m->arm_com[i][k]; }
    for (int k = 0; k < 9; k++) { d->arm_jrot[i][k] = (float)m->arm_jrot[i][k]; d->arm_inertia[i][k] = (float)m->arm_inertia[i][k]; }
    uint32_t anc = 0;
    for (int k = i; k >= 0; k = m->arm_parent[k]) anc |= 1u << k;
    d->arm_anc[i] = anc;
  }
  for (int i = 0; i < m->n_arm; i++) {
    uint32_t sub = 0;
    for (int j = 0; j < m->n_arm; j++) if (d->arm_anc[j] & (1u << i)) sub |= 1u << j;
    d->arm_sub[i] = sub;
  }
  for (int k = 0; k < 3; k++) d->base_pos[k] = (float)m->base_pos[k];
  for (int k = 0; k < 9; k++) d->base_rot[k] = (float)m->base_rot[k];
  for (int s = 0; s < m->n_site; s++) {
    d->site_body[s] = m->site_body[s];
    for (int k = 0; k < 3; k++) d->site_pos[s][k] = (float)m->site_pos[s][k];
    for (int k = 0; k < 9; k++) d->site_rot[s][k] = (float)m->site_rot[s][k];
  }
  d->ee_chain = m->site_body[RP_SITE_EE];     /* the EE body's ancestors are dofs 0..body-1 (serial chain) */
  for (int f = 0; f < m->n_free; f++) {
    d->free_mass[f] = (float)m->free_mass[f]; d->free_rot_locked[f] = m->free_rot_locked[f];
    for (int k = 0; k < 3; k++) { d->free_inertia[f][k] = (float)m->free_inertia[f][k]; d->free_pos0[f][k] = (float)m->free_pos0[f][k]; }
    rp_mat_to_quat(m->free_rot0[f], d->free_quat0[f]);
  }
  for (int j = 0; j < m->n_joint1; j++) {
    d->j1_type[j] = m->j1_type[j]; d->j1_has_pos_motor[j] = m->j1_has_pos_motor[j];
    d->j1_minv[j] = (float)(m->j1_type[j] == 1 ? 1.0 / m->j1_mass[j] : 1.0 / m->j1_inertia_axis[j]);
    d->j1_motor_target[j] = (float)m->j1_motor_target[j];
    d->j1_motor_maximp[j] = (float)(m->j1_has_pos_motor[j] ? m->j1_motor_force[j] / 300.0 : 1.0);
    for (int k = 0; k < 3; k++) { d->j1_pos[j][k] = (float)m->j1_pos[j][k]; d->j1_axis[j][k] = (float)m->j1_axis[j][k]; }
    for (int k = 0; k < 9; k++) d->j1_rot[j][k] = (float)m->j1_rot[j][k];
  }
  for (int c = 0; c < m->n_col; c++) {
    d->col_body[c] = m->col_body[c]; d->col_type[c] = m->col_type[c]; d->col_link[c] = m->col_link[c]; d->col_obj[c] = m->col_obj[c];
    d->col_friction[c] = (float)m->col_friction[c];
    for (int k = 0; k < 3; k++) { d->col_he[c][k] = (float)m->col_he[c][k]; d->col_pos[c][k] = (float)m->col_pos[c][k]; }
    for (int k = 0; k < 9; k++) d->col_rot[c][k] = (float)m->col_rot[c][k];
  }
  memcpy(d->pair, m->pair, sizeof(d->pair));
  int isP = m->arm_type == RP_ARM_PANDA;
  d->arm_type = m->arm_type; d->scene = m->scene; d->drawer_free = m->drawer_free; d->free_row0 = m->free_row0;
  d->d_grip_obs = rp_dm_dof_of_joint(m, isP ? 9 : 18);
  d->d18 = rp_dm_dof_of_joint(m, 18); d->d20 = rp_dm_dof_of_joint(m, 20); d->d12 = rp_dm_dof_of_joint(m, 12);
  d->d15 = rp_dm_dof_of_joint(m, 15); d->d10 = rp_dm_dof_of_joint(m, 10); d->d13 = rp_dm_dof_of_joint(m, 13);
  d->d9p = rp_dm_dof_of_joint(m, 9); d->d10p = rp_dm_dof_of_joint(m, 10);
  for (int j = 0; j < 8; j++) d->joints_dof[j] = rp_dm_dof_of_joint(m, j);
  /* envList.py:18-22, 89-99 */
  const float PI = 3.14159265358979323846f;
  if (m->scene == RP_SCENE_COMPLEX) {
    d->play = 1; d->use_orientation = 1; d->return_velocity = 0; d->num_objects = m->n_free - 1;      /* blocks, then the drawer */
    float gl[3] = {-0.18f, 0.f, 0.05f}, gh[3] = {0.18f, 0.3f, 0.1f};
    for (int k = 0; k < 3; k++) { d->goal_lo[k] = d->obj_lo[k] = gl[k]; d->goal_hi[k] = d->obj_hi[k] = gh[k]; d->env_hi[k] = 1.f; }
    d->n_ag = 7 * d->num_objects + 4; d->n_goal_init = d->n_ag;
    d->n_obs = 8 + d->n_ag; d->n_fps = 8 + d->n_ag; d->n_observation = 7 + d->n_ag; d->n_target = 6;
  } else if (m->scene == RP_SCENE_DEFAULT) {
    d->play = 0; d->use_orientation = 0; d->return_velocity = 1; d->num_objects = 0; d->n_goal_init = 3;
    float gl[3] = {-0.18f, -0.18f, -0.05f}, gh[3] = {0.18f, 0.18f, 0.05f}, eh[3] = {0.18f, 0.18f, 0.15f};
    for (int k = 0; k < 3; k++) { d->goal_lo[k] = gl[k]; d->goal_hi[k] = gh[k]; d->env_hi[k] = eh[k]; }
    d->n_obs = 7; d->n_ag = 3; d->n_fps = 4; d->n_observation = 6; d->n_target = 6;
  } else {
    d->play = 0; d->use_orientation = 0; d->return_velocity = 1; d->num_objects = 1; d->n_goal_init = 3;
    float gl[3] = {-0.18f, -0.18f, 0.0f}, gh[3] = {0.18f, 0.18f, 0.1f}, eh[3] = {0.18f, 0.18f, 0.2f};
    for (int k = 0; k < 3; k++) { d->goal_lo[k] = d->obj_lo[k] = gl[k]; d->goal_hi[k] = d->obj_hi[k] = gh[k]; d->env_hi[k] = eh[k]; }
    d->n_obs = 13; d->n_ag = 3; d->n_fps = 7; d->n_observation = 12; d->n_target = 7;
  }
  d->n_target = isP ? 7 : 6;            /* numDofs (environments.py:361, 371) */
  d->rew_thresh = 0.05f; d->dense_reward = 0; d->boxbox_margin = 0.f; d->persist = 0; d->pmcache = nullptr; d->gjk = 0; d->spec_limits = 0;
  for (int c = 0; c < m->n_col; c++) { d->col_margin[c] = (float)m->col_thr[c]; d->col_stiff[c] = (float)m->col_stiffness[c]; d->col_damp[c] = (float)m->col_damping[c]; d->col_spin[c] = (float)m->col_spin[c];
    d->col_toggle[c] = m->col_toggle[c]; for (int k = 0; k < 3; k++) d->col_rgb[c][k] = (float)m->col_rgb[c][k]; }
  d->floor_z = 1e30f;
  for (int c = 0; c < m->n_col; c++) {
    if (m->col_body[c] != 0) continue;
    double ext = m->col_type[c] == 0 ? fabs(m->col_rot[c][6]) * m->col_he[c][0] + fabs(m->col_rot[c][7]) * m->col_he[c][1] + fabs(m->col_rot[c][8]) * m->col_he[c][2]
                                     : m->col_he[c][0];
    float lo = (float)(m->col_pos[c][2] - ext);
    if (lo < d->floor_z) d->floor_z = lo;
  }
  if (isP) {   /* environments.py:1015-1017 */
    const float ll[7] = {-0.6f, -2.2f, -3.0f, -3.04878596f, -PI, -PI, -PI};
    const float ul[7] = {3.f, 1.8f, 0.5f, -0.5002492f, 3.f, 3.45266257f, 2.40072908f};
    const float inc[7] = {0.1f, 0.1f, 0.2f, 0.2f, 0.2f, 0.2f, 0.2f};
    for (int k = 0; k < 7; k++) { d->ll[k] = ll[k]; d->ul[k] = ul[k]; d->inc[k] = inc[k]; }
  } else {     /* environments.py:1019-1021 */
    const float ul[6] = {-0.7f, 2 * PI, -0.5f, 2 * PI, 2 * PI, 2 * PI};
    const float inc[6] = {0.1f, 0.1f, 0.2f, 0.2f, 0.2f, 0.2f};
    for (int k = 0; k < 6; k++) { d->ll[k] = -2 * PI; d->ul[k] = ul[k]; d->inc[k] = inc[k]; }
  }
}

#endif
