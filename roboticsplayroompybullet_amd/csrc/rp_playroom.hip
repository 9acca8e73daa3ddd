/* rp_playroom.hip — C ABI (include/rp_playroom.h) over the gfx950 kernels in rp_kernels.cuh. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/rp_playroom.h"
#include "../../include/rp_playroom_debug.h"
#include "generated/rp_models_gen.h"
#include "generated/rp_hullverts_gen.h"
#include "generated/rp_hullcells_gen.h"
#include "generated/rp_hullplanes_gen.h"
#include "rp_device_model.h"
#include "rp_kernels.cuh"
#include "rp_render.cuh"


struct rp_sim {
  rp_config cfg;
  DevModel host_model;
  DevModel* dev_model;
  float* state;            /* [N][RP_REC_FLOATS] */
  float* ws;               /* [N][W3_FLOATS] constraint-row workspace of the split step pipeline */
  float* dbg;
  float* hullv;            /* convex-hull vertices of the arm's collision meshes (DevModel.hullv points here) */
  float* hcv; int* hco;    /* their support-vertex candidate tables (DevModel.hcv / hco) */
  float* hpl;              /* their face planes, collider frame (DevModel.hpl: the ray caster) */
  float* pmcache;          /* the contact caches, [N][PMC_FLOATS] (DevModel.pmcache points here); nullptr under RP_CFG_STATELESS_CONTACTS */
  int* sort_cnt;           /* [2][RP_MAX_GROUPS][SORT_BINS] load-class histograms for pairing envs in k_solve2 (double-buffered) */
  int* sort_slot;          /* [N] per env: (bin << 16) | rank inside the bin, from the latest k_solve2 */
  int* pair_env;           /* [N] per group range: env ids sorted by load class, heaviest first (k_solve2 pairs neighbours) */
  int* hv_list;            /* [N] per group range: the heavy envs of the current substep (k_prep2 appends, k_solve2's worker blocks take them: one env per wave) */
  int hv_waves;            /* worker blocks per group (0: an eighth of the group's envs, 16 .. 512; RP_HV_WAVES overrides) */
  int* hv_cnt;             /* [2][RP_MAX_GROUPS] their number, double-buffered by substep parity (zeroed by k_member and by the k_prep2 before) */
  GroupBounds gb;          /* place ranges of the groups the tables were built for */
  int gsplit[RP_MAX_GROUPS]; /* RP_GROUP_SPLIT: relative sizes of the groups, heaviest first (0 = equal) */
  int* member[2];          /* [N] place -> env, all envs ranked by load class (k_member); group g owns places [N g / G, N (g + 1) / G) */
  int member_cur;
  int debug_flags;         /* rp_set_debug_flags: bit 0 = k_solve2 never solves contacts side by side (one folded slot per contact) */
  int sort_G, sort_par;    /* group count the tables were built for (0 = none yet); buffer that the next k_solve2 reads */
  hipEvent_t ev0, ev1;
  hipEvent_t* pool;        /* per-launch timing ring: EV_PER_STEP events per recorded step */
  int pool_steps, pool_next, pool_count;
  int timers_on;
  int groups;              /* env groups of the default pipeline, each on its own stream (tail overlap) */
  hipStream_t gstream[RP_MAX_GROUPS];
  hipEvent_t gfork, gjoin[RP_MAX_GROUPS];
  int fused;               /* 0: split pipeline (default), 1: single fused k_step kernel (in-library reference path), 2: k_chain - one launch for the twelve substeps (round 4's experiment) */
  int chain_blocks;        /* k_chain's grid (RP_CHAIN_BLOCKS at rp_create, default 1024) */
  /* rp_reset through the split pipeline (allocated at the first reset): dense scratch records of the envs that are settling,
   * their env ids, per-env progress {pending, attempt, depth}, pairing tables of the scratch range */
  float* rs_state; int* rs_idx; int4* rs_meta; int* rs_count; int* rs_sort_cnt; int* rs_sort_slot; int* rs_pair; int* rs_count_host;
  int reset_rounds;        /* rounds the latest rp_reset took (rp_debug) */
  float* rc_tab; int* rc_cnt; float* rc_ee; int rc_cap; int rc_last_num;      /* rp_render / rp_ray_test: collider poses of rc_cap envs (allocated on first use) */
  rp_timers timers;
  char err[256];
};

#define EV_PER_STEP (2 + 2 * (2 * K_NSUB + 2))   /* step pair + a pair per launch (action, 12 prep, 12 solve, obs) */
static thread_local char g_err[256] = "";   /* rp_create / NULL-handle errors: per calling thread (rp_last_error(NULL) reads the caller's own) */

#define HIPCHK(h, call)                                                                             \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      snprintf((h) ? (h)->err : g_err, 256, "%s failed: %s", #call, hipGetErrorString(e_));         \
      return RP_ERR_HIP;                                                                            \
    }                                                                                               \
  } while (0)

/* makes the handle's device current for one entry point and restores the caller's on the way out (handles on several devices,
 * or a torch caller whose current device is another one) */
struct DevGuard {
  int prev = -1, cur = -1;
  explicit DevGuard(int dev) : cur(dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != cur) (void)hipSetDevice(cur);
  }
  ~DevGuard() { if (prev >= 0 && prev != cur) (void)hipSetDevice(prev); }
};

static OutPtrs to_ptrs(const rp_out* o) {
  OutPtrs p;
  memset(&p, 0, sizeof(p));
  if (!o) return p;
  p.obs_quat = o->obs_quat; p.achieved_goal = o->achieved_goal; p.desired_goal = o->desired_goal;
  p.cag = o->controllable_achieved_goal; p.fps = o->full_positional_state; p.joints = o->joints; p.velocity = o->velocity;
  p.observation = o->observation; p.proprio = o->gripper_proprioception; p.reward = o->reward; p.is_success = o->is_success;
  p.target_poses = o->target_poses; p.status = o->status; p.pack = o->pack;
  return p;
}

extern "C" {

#ifndef RP_BUILD_ID
#define RP_BUILD_ID "unversioned"
#endif
#ifdef RP_WIDE
const char* rp_version(void) { return "rp_playroom 0.4 (gfx950, wide build: two-object play ids) build " RP_BUILD_ID; }
#else
const char* rp_version(void) { return "rp_playroom 0.4 (gfx950) build " RP_BUILD_ID; }
#endif

static void destroy_handle(rp_sim* h) {        /* frees whatever a (possibly partial) handle owns; hipFree(nullptr) etc. are no-ops */
  if (!h) return;
  hipFree(h->hullv); hipFree(h->hcv); hipFree(h->hco); hipFree(h->hpl); hipFree(h->pmcache); hipFree(h->dev_model); hipFree(h->state); hipFree(h->ws); hipFree(h->dbg); hipFree(h->sort_cnt); hipFree(h->sort_slot); hipFree(h->pair_env); hipFree(h->hv_list); hipFree(h->hv_cnt); hipFree(h->member[0]); hipFree(h->member[1]);
  hipFree(h->rc_tab); hipFree(h->rc_cnt); hipFree(h->rc_ee);
  hipFree(h->rs_state); hipFree(h->rs_idx); hipFree(h->rs_meta); hipFree(h->rs_count); hipFree(h->rs_sort_cnt); hipFree(h->rs_sort_slot); hipFree(h->rs_pair);
  if (h->rs_count_host) hipHostFree(h->rs_count_host);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->gfork) hipEventDestroy(h->gfork);
  for (int i = 0; i < RP_MAX_GROUPS; i++) { if (h->gstream[i]) hipStreamDestroy(h->gstream[i]); if (h->gjoin[i]) hipEventDestroy(h->gjoin[i]); }
  if (h->pool) { for (int i = 0; i < h->pool_steps * EV_PER_STEP; i++) if (h->pool[i]) hipEventDestroy(h->pool[i]); free(h->pool); }
  free(h);
}

int rp_create(const rp_config* cfg, rp_handle* out) {
  if (!cfg || !out || cfg->num_envs <= 0) { snprintf(g_err, 256, "rp_create: bad argument"); return RP_ERR_ARG; }
  if (cfg->num_envs > RP_MAX_ENVS) { snprintf(g_err, 256, "rp_create: num_envs %d exceeds %d (rank field of the env pairing tables)", cfg->num_envs, RP_MAX_ENVS); return RP_ERR_ARG; }
  if (cfg->env_kind < 0 || cfg->env_kind >= RP_ENV_COUNT) { snprintf(g_err, 256, "rp_create: unsupported env kind %d", cfg->env_kind); return RP_ERR_UNSUPPORTED; }
  if ((cfg->flags & RP_CFG_ACTION_TYPE) && (cfg->action_type < 0 || cfg->action_type > RP_ACTION_RELATIVE_JOINTS)) { snprintf(g_err, 256, "rp_create: action_type %d", cfg->action_type); return RP_ERR_ARG; }
  if ((cfg->flags & RP_CFG_HULL_EPA) && (cfg->flags & RP_CFG_NO_HULL_EPA)) { snprintf(g_err, 256, "rp_create: RP_CFG_HULL_EPA and RP_CFG_NO_HULL_EPA"); return RP_ERR_ARG; }
  if ((cfg->flags & RP_CFG_CONTACT_MARGIN) && !(cfg->contact_margin >= 0.f && cfg->contact_margin <= 0.05f)) { snprintf(g_err, 256, "rp_create: contact_margin %g outside [0, 0.05]", (double)cfg->contact_margin); return RP_ERR_ARG; }
  /* registered id -> baked model (arm + scene) and action type.  The ids of a play family share scene, arm and configuration
   * (envList.py:43-140); only perform_action differs. */
  static const struct { char model; int action_type; } SPEC[RP_ENV_COUNT] = {
    {'U', RP_ACT_ABS_RPY}, {'R', RP_ACT_ABS_RPY}, {'P', RP_ACT_ABS_RPY},
    {'U', RP_ACT_ABS_QUAT}, {'U', RP_ACT_REL_QUAT}, {'U', RP_ACT_REL_JOINTS}, {'U', RP_ACT_ABS_JOINTS}, {'U', RP_ACT_REL_RPY},
    {'P', RP_ACT_ABS_RPY},
    {'Q', RP_ACT_ABS_RPY}, {'Q', RP_ACT_ABS_RPY},
    {'V', RP_ACT_ABS_QUAT}, {'V', RP_ACT_REL_QUAT}, {'V', RP_ACT_REL_JOINTS}, {'V', RP_ACT_ABS_JOINTS}, {'V', RP_ACT_ABS_RPY}, {'V', RP_ACT_REL_RPY},
    {'W', RP_ACT_ABS_QUAT}, {'W', RP_ACT_REL_JOINTS}};
  static_assert(RP_ACT_ABS_RPY == RP_ACTION_ABSOLUTE_RPY && RP_ACT_REL_RPY == RP_ACTION_RELATIVE_RPY && RP_ACT_ABS_QUAT == RP_ACTION_ABSOLUTE_QUAT &&
                RP_ACT_REL_QUAT == RP_ACTION_RELATIVE_QUAT && RP_ACT_ABS_JOINTS == RP_ACTION_ABSOLUTE_JOINTS && RP_ACT_REL_JOINTS == RP_ACTION_RELATIVE_JOINTS,
                "public rp_action_type values are the kernels' RP_ACT_*");
  const int action_type = (cfg->flags & RP_CFG_ACTION_TYPE) ? cfg->action_type : SPEC[cfg->env_kind].action_type;
#ifdef RP_WIDE
  if (SPEC[cfg->env_kind].model != 'W') { snprintf(g_err, 256, "rp_create: env kind %d is served by librp_playroom_hip.so, not by the wide build", cfg->env_kind); return RP_ERR_UNSUPPORTED; }
#else
  if (SPEC[cfg->env_kind].model == 'W') { snprintf(g_err, 256, "rp_create: the two-object ids (env kind %d) are served by librp_playroom_hip_wide.so", cfg->env_kind); return RP_ERR_UNSUPPORTED; }
#endif
  {
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess) { (void)hipGetLastError(); snprintf(g_err, 256, "rp_create: hipGetDeviceCount: %s", hipGetErrorString(de)); return RP_ERR_HIP; }
    if (cfg->device < 0 || cfg->device >= ndev) { snprintf(g_err, 256, "rp_create: device %d of %d", cfg->device, ndev); return RP_ERR_ARG; }
  }
  rp_sim* h = (rp_sim*)calloc(1, sizeof(rp_sim));
  rp_model* m = (rp_model*)malloc(sizeof(rp_model));
  if (!h || !m) { free(h); free(m); snprintf(g_err, 256, "rp_create: out of host memory"); return RP_ERR_ARG; }
  h->cfg = *cfg;
  switch (SPEC[cfg->env_kind].model) {
    case 'W': rp_fill_model_W(m); break;
    case 'R': rp_fill_model_R(m); break;
    case 'P': rp_fill_model_P(m); break;
    case 'Q': rp_fill_model_Q(m); break;
    case 'V': rp_fill_model_V(m); break;
    default: rp_fill_model_U(m); break;
  }
  rp_build_dev_model(m, &h->host_model);
  free(m);
  DevModel* d = &h->host_model;
  if (cfg->env_kind == RP_ENV_PANDA_PUSH) {      /* pandaPick's arm and scene with pandaPush's ranges (envList.py:12-16) */
    const float gl[3] = {-0.1f, -0.1f, -0.06f}, gh[3] = {0.1f, 0.1f, -0.05f}, eh[3] = {0.18f, 0.18f, -0.04f};
    for (int k = 0; k < 3; k++) { d->goal_lo[k] = d->obj_lo[k] = gl[k]; d->goal_hi[k] = d->obj_hi[k] = gh[k]; d->env_hi[k] = eh[k]; }
  }
  if (cfg->env_kind == RP_ENV_PANDA_REACH_2D) {  /* envList.py:24-26 */
    const float gl[3] = {-0.18f, -0.18f, -0.06f}, gh[3] = {0.18f, 0.18f, -0.05f}, eh[3] = {0.18f, 0.18f, 0.0f};
    for (int k = 0; k < 3; k++) { d->goal_lo[k] = gl[k]; d->goal_hi[k] = gh[k]; d->env_hi[k] = eh[k]; }
  }
  /* the constructor kwargs of the env class (rp_config.flags; environments.py:64-67) */
  for (int k = 0; k < 3; k++) {
    if (cfg->flags & RP_CFG_GOAL_RANGE) { d->goal_lo[k] = cfg->goal_range_low[k]; d->goal_hi[k] = cfg->goal_range_high[k]; }
    if (cfg->flags & RP_CFG_OBJ_RANGE) { d->obj_lo[k] = cfg->obj_lower_bound[k]; d->obj_hi[k] = cfg->obj_upper_bound[k]; }
    if (cfg->flags & RP_CFG_ENV_RANGE) d->env_hi[k] = cfg->env_range_high[k];
  }
  if (cfg->flags & RP_CFG_REW_THRESH) d->rew_thresh = cfg->sparse_rew_thresh;
  if (cfg->flags & RP_CFG_DENSE_REWARD) d->dense_reward = 1;
  if (cfg->flags & RP_CFG_CONTACT_MARGIN) { for (int c = 0; c < RP_MAX_COL; c++) d->col_margin[c] = cfg->contact_margin; d->boxbox_margin = -1.f; }
  d->action_type = action_type;
  d->n_action = (action_type == RP_ACT_ABS_QUAT || action_type == RP_ACT_REL_QUAT) ? 8
              : ((action_type == RP_ACT_ABS_JOINTS || action_type == RP_ACT_REL_JOINTS) ? d->n_target + 1 : 7);
  DevGuard guard(cfg->device);
  const int N = cfg->num_envs;
  int rc = RP_ERR_HIP;
  hipError_t e = hipSuccess;
#define CREATE_CHK(call) do { e = (call); if (e != hipSuccess) { snprintf(g_err, 256, "rp_create: %s: %s", #call, hipGetErrorString(e)); goto fail; } } while (0)
  (void)hipGetLastError();          /* a stale error of an earlier, unrelated call must not fail this one */
  CREATE_CHK(hipSetDevice(cfg->device));
  CREATE_CHK(hipMalloc((void**)&h->dev_model, sizeof(DevModel)));
  CREATE_CHK(hipMalloc((void**)&h->state, (size_t)N * RP_REC_FLOATS * sizeof(float)));
  CREATE_CHK(hipMalloc((void**)&h->ws, (size_t)N * W3_FLOATS * sizeof(float)));
  CREATE_CHK(hipMalloc((void**)&h->dbg, 4096 * sizeof(float)));
  CREATE_CHK(hipMalloc((void**)&h->sort_cnt, (size_t)2 * RP_MAX_GROUPS * SORT_BINS * sizeof(int)));
  CREATE_CHK(hipMalloc((void**)&h->sort_slot, (size_t)N * sizeof(int)));
  CREATE_CHK(hipMalloc((void**)&h->pair_env, (size_t)N * sizeof(int)));
  { const char* e = getenv("RP_HV_WAVES"); h->hv_waves = e ? atoi(e) : 0; }
  CREATE_CHK(hipMalloc((void**)&h->hv_list, (size_t)N * sizeof(int)));
  CREATE_CHK(hipMalloc((void**)&h->hv_cnt, 2 * RP_MAX_GROUPS * sizeof(int)));
  CREATE_CHK(hipMemset(h->hv_cnt, 0, 2 * RP_MAX_GROUPS * sizeof(int)));
  CREATE_CHK(hipMalloc((void**)&h->member[0], (size_t)N * sizeof(int)));
  CREATE_CHK(hipMalloc((void**)&h->member[1], (size_t)N * sizeof(int)));
  {
    const float (*hv)[4]; const int *hoff, *hcnt;
    const int nhv = rp_hull_tables(d->kind, &hv, &hoff, &hcnt);
    if (nhv > 0) {
      CREATE_CHK(hipMalloc((void**)&h->hullv, (size_t)nhv * 4 * sizeof(float)));
      CREATE_CHK(hipMemcpy(h->hullv, hv, (size_t)nhv * 4 * sizeof(float), hipMemcpyHostToDevice));
      const bool off = getenv("RP_NO_HULL") != nullptr;      /* timing / model studies only: arm links as their OBBs everywhere (round 2's contacts) */
      for (int c = 0; c < RP_MAX_COL; c++) { d->hull_off[c] = hoff[c]; d->hull_cnt[c] = off ? 0 : hcnt[c]; }
      /* the candidate tables: the baked lists hold vertex numbers; the kernels read (x, y, z, number) in one 16-byte load */
      const unsigned short* cidx; const int *coff, *cfirst; int ctotal = 0;
      const int ncoff = rp_hcell_tables(d->kind, &cidx, &coff, &cfirst, &ctotal);
      for (int c = 0; c < RP_MAX_COL; c++) d->hcell_first[c] = -1;
      if (ncoff <= 0) {      /*  hulls without candidate tables - a new kind, a bake that went wrong: the narrowphase would read through a null table */
        for (int c = 0; c < RP_MAX_COL; c++)
          if (d->hull_cnt[c] > 0) { snprintf(g_err, 256, "rp_create: collider %d collides as a hull of %d vertices but the model has no support-vertex candidate tables (rp_hullcells_gen.h)", c, d->hull_cnt[c]); e = hipErrorInvalidValue; goto fail; }
      }
      if (ncoff > 0) {
        std::vector<float> cv((size_t)ctotal * 4, 0.f);
        std::vector<char> done((size_t)ncoff, 0);
        for (int c = 0; c < RP_MAX_COL; c++) {
          d->hcell_first[c] = hcnt[c] > 0 ? cfirst[c] : -1;
          if (d->hull_cnt[c] > 0 && cfirst[c] < 0) { snprintf(g_err, 256, "rp_create: collider %d collides as a hull but has no candidate table", c); e = hipErrorInvalidValue; goto fail; }
          if (hcnt[c] <= 0 || cfirst[c] < 0) continue;
          const bool first_user = !done[cfirst[c]];
          done[cfirst[c]] = 1;
          for (int k = coff[cfirst[c]]; k < coff[cfirst[c] + RP_HCELL_N]; k++) {
            const int vi = cidx[k];
            if (vi >= hcnt[c]) {      /* (checked for EVERY collider that shares the table, not only the first) */ snprintf(g_err, 256, "rp_create: the candidate table refers to vertex %d of a %d-vertex hull", vi, hcnt[c]); e = hipErrorInvalidValue; goto fail; }
            if (!first_user) continue;
            const float* v = hv[hoff[c] + vi];
            cv[4 * (size_t)k] = v[0]; cv[4 * (size_t)k + 1] = v[1]; cv[4 * (size_t)k + 2] = v[2]; memcpy(&cv[4 * (size_t)k + 3], &vi, 4);
          }
        }
        CREATE_CHK(hipMalloc((void**)&h->hcv, cv.size() * sizeof(float)));
        CREATE_CHK(hipMemcpy(h->hcv, cv.data(), cv.size() * sizeof(float), hipMemcpyHostToDevice));
        CREATE_CHK(hipMalloc((void**)&h->hco, (size_t)ncoff * sizeof(int)));
        CREATE_CHK(hipMemcpy(h->hco, coff, (size_t)ncoff * sizeof(int), hipMemcpyHostToDevice));
      }
    }
    {      /* the hulls' face planes for the ray caster: baked in the body frame, used in the collider's (x_body = Rc x_col + pc: n_col = Rc^T n, w_col = w + n . pc) */
      const float (*pl)[4]; const int *poff, *pcnt;
      const int npl = rp_hplane_tables(d->kind, &pl, &poff, &pcnt);
      if (npl > 0 && nhv > 0 && getenv("RP_NO_HULL") == nullptr) {
        std::vector<float> hp((size_t)npl * 4, 0.f);
        std::vector<char> done((size_t)npl, 0);
        for (int c = 0; c < RP_MAX_COL; c++) {
          d->hpl_off[c] = poff[c]; d->hpl_cnt[c] = hcnt[c] > 0 ? pcnt[c] : 0;
          if (d->hpl_cnt[c] <= 0) continue;
          if (done[poff[c]]) { d->hpl_cnt[c] = 0; continue; }      /* (a plane list serves ONE collider frame; no model shares one between two) */
          done[poff[c]] = 1;
          const float* Rc = d->col_rot[c]; const float* pc = d->col_pos[c];
          for (int k = poff[c]; k < poff[c] + pcnt[c]; k++) {
            const float* n = pl[k];
            hp[4 * (size_t)k] = Rc[0] * n[0] + Rc[3] * n[1] + Rc[6] * n[2];
            hp[4 * (size_t)k + 1] = Rc[1] * n[0] + Rc[4] * n[1] + Rc[7] * n[2];
            hp[4 * (size_t)k + 2] = Rc[2] * n[0] + Rc[5] * n[1] + Rc[8] * n[2];
            hp[4 * (size_t)k + 3] = n[3] + n[0] * pc[0] + n[1] * pc[1] + n[2] * pc[2];
          }
        }
        CREATE_CHK(hipMalloc((void**)&h->hpl, hp.size() * sizeof(float)));
        CREATE_CHK(hipMemcpy(h->hpl, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
      }
    }
    d->hullv = h->hullv; d->hcv = h->hcv; d->hco = h->hco; d->hpl = h->hpl;
    if (!(cfg->flags & RP_CFG_STATELESS_CONTACTS) && !(cfg->flags & RP_CFG_CONTACT_MARGIN) && getenv("RP_STATELESS") == nullptr) {      /* (a uniform contact margin is a stateless-model study: it keeps the stateless contacts) */
      CREATE_CHK(hipMalloc((void**)&h->pmcache, (size_t)N * PMC_FLOATS * sizeof(float)));
      CREATE_CHK(hipMemset(h->pmcache, 0, (size_t)N * PMC_FLOATS * sizeof(float)));
      d->persist = 1; d->pmcache = h->pmcache;
    }
    d->gjk = ((cfg->flags & RP_CFG_OBB_EDGES) || getenv("RP_NO_GJK") != nullptr) ? 0 : 1;      /* (oracle RPO_RULE_GJK: the default; RP_NO_GJK=1: the tools' switch to round 3's OBB edges) */
    {      /* oracle RPO_RULE_EPA: the Panda kinds' default (rpo_create); RP_CFG_HULL_EPA / RP_CFG_NO_HULL_EPA or the tools' RP_EPA=1 / RP_NO_EPA=1 force it */
      int epa = (SPEC[cfg->env_kind].model == 'U' || SPEC[cfg->env_kind].model == 'R') ? 0 : 1;
      if ((cfg->flags & RP_CFG_HULL_EPA) || getenv("RP_EPA") != nullptr) epa = 1;
      if ((cfg->flags & RP_CFG_NO_HULL_EPA) || getenv("RP_NO_EPA") != nullptr) epa = 0;
      d->epa = d->gjk ? epa : 0;
    }
    d->spec_limits = (cfg->flags & RP_CFG_SPECULATIVE_LIMITS) ? 1 : 0;
    if (getenv("RP_NO_SPIN") != nullptr)                     /* timing / model studies only: no torsional friction rows */
      for (int c = 0; c < RP_MAX_COL; c++) d->col_spin[c] = 0.f;
  }
  CREATE_CHK(hipMemcpy(h->dev_model, &h->host_model, sizeof(DevModel), hipMemcpyHostToDevice));
  CREATE_CHK(hipEventCreate(&h->ev0));
  CREATE_CHK(hipEventCreate(&h->ev1));
  {
    const char* g = getenv("RP_STEP_GROUPS");
    h->groups = g ? atoi(g) : 3;      /* 3 group streams + nothing else stays within the 4 hardware queues ROCm multiplexes onto */
    if (h->groups < 1) h->groups = 1;
    if (h->groups > RP_MAX_GROUPS) h->groups = RP_MAX_GROUPS;
    { const char* cb = getenv("RP_CHAIN_BLOCKS"); h->chain_blocks = cb ? (atoi(cb) > 0 ? atoi(cb) : 1) : 1024; }      /* k_chain's grid: 256 CUs x 4 blocks of 40 KB LDS and 2 x 217 VGPRs - everything resident at once */
    CREATE_CHK(hipEventCreateWithFlags(&h->gfork, hipEventDisableTiming));
    /* RP_GROUP_SPLIT="30,30,40": relative group sizes, heaviest group first (default: 25,35,40 for 3 groups, equal otherwise).  Stream priorities for the heavy
     * group were tried and lose (2.69 vs 2.61 ms per step). */
    const char* sp = getenv("RP_GROUP_SPLIT");
    for (int i = 0; i < RP_MAX_GROUPS; i++) h->gsplit[i] = 0;
    if (sp) { int i = 0; while (*sp && i < RP_MAX_GROUPS) { h->gsplit[i++] = atoi(sp); while (*sp && *sp != ',') sp++; if (*sp == ',') sp++; } }
    for (int i = 0; i < RP_MAX_GROUPS; i++) {
      CREATE_CHK(hipStreamCreateWithFlags(&h->gstream[i], hipStreamNonBlocking));
      CREATE_CHK(hipEventCreateWithFlags(&h->gjoin[i], hipEventDisableTiming));
    }
  }
  hipLaunchKernelGGL(k_init, dim3((N + 255) / 256), dim3(256), 0, 0, h->dev_model, h->state, N);
  CREATE_CHK(hipGetLastError());
  CREATE_CHK(hipDeviceSynchronize());
#undef CREATE_CHK
  *out = h;
  return RP_OK;
fail:
  (void)hipGetLastError();
  destroy_handle(h);
  return rc;
}

int rp_destroy(rp_handle h) {
  if (!h) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  hipDeviceSynchronize();
  destroy_handle(h);
  return RP_OK;
}

int rp_get_dims(rp_handle h, rp_dims* d) {
  if (!h || !d) return RP_ERR_ARG;
  const DevModel* m = &h->host_model;
  d->obs_quat = m->n_obs; d->achieved_goal = m->n_ag; d->desired_goal = m->n_ag; d->controllable_achieved_goal = 4;
  d->full_positional_state = m->n_fps; d->joints = 8; d->velocity = 6; d->observation = m->n_observation;
  d->target_poses = m->n_target; d->action = m->n_action;
  return RP_OK;
}

/* playEnv.reset() for the masked envs through the split pipeline: rounds of { gather the envs that still have a settle phase
 * ahead, sample their objects, 100 x (k_settle_prep, k_settle_solve: k_prep2 / k_solve2 under other names) over the dense scratch range, finish / re-sample / next attempt }.
 * One host sync per round (the pending count sizes the launches); a round is ~100 substeps, most resets take 1-3 rounds, the
 * unluckiest env of a large batch a few more (play ids repeat the reset while the drawn goal is already satisfied). */
static int reset_split(rp_handle h, const uint8_t* mask, const rp_out* out, hipStream_t s) {
  const int N = h->cfg.num_envs;
  if (!h->rs_state) {
    if (hipMalloc((void**)&h->rs_state, (size_t)N * RP_REC_FLOATS * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&h->rs_idx, (size_t)N * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&h->rs_meta, (size_t)N * sizeof(int4)) != hipSuccess ||
        hipMalloc((void**)&h->rs_count, sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&h->rs_sort_cnt, (size_t)2 * SORT_BINS * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&h->rs_sort_slot, (size_t)N * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&h->rs_pair, (size_t)N * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void**)&h->rs_count_host, sizeof(int)) != hipSuccess) {
      snprintf(h->err, 256, "rp_reset: scratch allocation failed"); return RP_ERR_HIP;
    }
  }
  const OutPtrs op = to_ptrs(out);
  const uint64_t seed = h->cfg.seed;
  const uint32_t off = (uint32_t)h->cfg.env_offset;
  hipLaunchKernelGGL(k_reset_mark, dim3((N + 255) / 256), dim3(256), 0, s, mask, h->rs_meta, N);
  int* cnt[2] = {h->rs_sort_cnt, h->rs_sort_cnt + SORT_BINS};
  h->reset_rounds = 0;
  bool exhausted = true;            /* the loop ran out of rounds (otherwise it left through M <= 0, which already proved that nothing is pending) */
  for (int round = 0; round < 64 * 9; round++) {
    hipLaunchKernelGGL(k_reset_list, dim3(1), dim3(64), 0, s, h->rs_meta, h->rs_idx, h->rs_count, N);
    HIPCHK(h, hipMemcpyAsync(h->rs_count_host, h->rs_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    const int M = *h->rs_count_host;
    if (M <= 0) { exhausted = false; break; }
    h->reset_rounds++;
    hipLaunchKernelGGL(k_reset_sample, dim3(M), dim3(64), 0, s, h->dev_model, h->state, h->rs_state, h->rs_idx, M, seed, off);
    hipLaunchKernelGGL(k_sort_init, dim3((max(M, SORT_BINS) + 255) / 256), dim3(256), 0, s, cnt[0], h->rs_sort_slot, 0, M);
    int par = 0;
    for (int i = 0; i < K_NSETTLE; i++) {
      hipLaunchKernelGGL(k_settle_prep, dim3(M), dim3(PREP_THREADS), 0, s, h->dev_model, h->rs_state, h->ws, 0, M, cnt[par], cnt[par ^ 1], h->rs_sort_slot, h->rs_pair, (const int*)nullptr, (const int*)h->rs_idx, h->debug_flags & 1);
      hipLaunchKernelGGL(k_settle_solve, dim3((M + 2 * SOLVE_WAVES - 1) / (2 * SOLVE_WAVES)), dim3(64 * SOLVE_WAVES), 0, s, h->dev_model, h->rs_state, h->ws, 0, M, h->rs_pair, cnt[par ^ 1], h->rs_sort_slot, h->debug_flags);
      par ^= 1;
    }
    hipLaunchKernelGGL(k_reset_finish, dim3(M), dim3(64), 0, s, h->dev_model, h->rs_state, h->state, h->rs_idx, h->rs_meta, op, M, seed, off);
    HIPCHK(h, hipGetLastError());
  }
  /* the round budget is 9 object re-samples x 64 attempts, which k_reset_finish's own caps (depth < 8, attempt < 64) cannot
   * exceed; if envs are pending all the same, say so instead of handing back half-reset records */
  if (!exhausted) return RP_OK;
  hipLaunchKernelGGL(k_reset_list, dim3(1), dim3(64), 0, s, h->rs_meta, h->rs_idx, h->rs_count, N);
  HIPCHK(h, hipMemcpyAsync(h->rs_count_host, h->rs_count, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  if (*h->rs_count_host > 0) {
    hipLaunchKernelGGL(k_reset_flag_pending, dim3((N + 255) / 256), dim3(256), 0, s, h->rs_meta, op.status, N);
    snprintf(h->err, 256, "rp_reset: %d envs still pending after %d rounds (status bit 4 set)", *h->rs_count_host, h->reset_rounds);
    return RP_ERR_INCOMPLETE;
  }
  return RP_OK;
}

static int reset_impl(rp_handle h, const float* o, int32_t n_o, const uint8_t* mask, const rp_out* out, void* stream) {
  DevGuard guard(h->cfg.device);
  hipStream_t s = (hipStream_t)stream;
  int N = h->cfg.num_envs;
  if (h->timers_on) hipEventRecord(h->ev0, s);
  if (!o && h->fused != 1) {
    int rc = reset_split(h, mask, out, s);
    if (rc != RP_OK) return rc;
  } else {          /* reset(o) has no settle phase; the fused path keeps everything in one kernel */
    hipLaunchKernelGGL(k_reset, dim3(N), dim3(64), 0, s, h->dev_model, h->state, mask, to_ptrs(out), N, h->cfg.seed, (uint32_t)h->cfg.env_offset, o, (int)n_o);
  }
  HIPCHK(h, hipGetLastError());
  if (h->timers_on) { hipEventRecord(h->ev1, s); hipEventSynchronize(h->ev1); hipEventElapsedTime(&h->timers.last_reset_ms, h->ev0, h->ev1); }
  return RP_OK;
}

int rp_reset(rp_handle h, const uint8_t* mask, const rp_out* out, void* stream) {
  if (!h) return RP_ERR_ARG;
  return reset_impl(h, nullptr, 0, mask, out, stream);
}

int rp_reset_to(rp_handle h, const float* o, int32_t n_o, const uint8_t* mask, const rp_out* out, void* stream) {
  if (!h || !o) return RP_ERR_ARG;
  const DevModel* m = &h->host_model;
  /* reset_object_pos(obs) reads object b at o[11 + 10 b : 18 + 10 b] (use_orientation) or o[7 + 6 b : 10 + 6 b] (environments.py:544-556) */
  int need = m->num_objects > 0 ? (m->use_orientation ? 18 + 10 * (m->num_objects - 1) : 10 + 6 * (m->num_objects - 1))
                                : (m->use_orientation ? (m->return_velocity ? 10 : 7) : 3);
  if (n_o < need) { snprintf(h->err, 256, "rp_reset_to: o has %d entries per env, this env reads %d", n_o, need); return RP_ERR_ARG; }
  return reset_impl(h, o, n_o, mask, out, stream);
}

int rp_reset_goal(rp_handle h, const float* goal, const uint8_t* mask, void* stream) {
  if (!h) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  int N = h->cfg.num_envs;
  hipLaunchKernelGGL(k_reset_goal, dim3(N), dim3(64), 0, (hipStream_t)stream, h->dev_model, h->state, goal, mask, N, h->cfg.seed,
                     (uint32_t)h->cfg.env_offset);
  HIPCHK(h, hipGetLastError());
  return RP_OK;
}

int rp_step(rp_handle h, const float* action, const rp_out* out, void* stream) {
  if (!h || !action) { if (h) snprintf(h->err, 256, "rp_step: action is NULL"); return RP_ERR_ARG; }
  DevGuard guard(h->cfg.device);
  hipStream_t s = (hipStream_t)stream;
  int N = h->cfg.num_envs;
  if (h->timers_on && h->fused == 1) hipEventRecord(h->ev0, s);
  if (h->fused == 1) {
    hipLaunchKernelGGL(k_step, dim3(N), dim3(64), 0, s, h->dev_model, h->state, action, to_ptrs(out), N);
  } else if (h->fused == 2) {
    /* round 4's experiment (rp_set_fused(h, 2)): the twelve substeps in ONE launch, k_chain - blocks of two waves own four envs each (in the load ranking's order) for
     * the whole step.  Ranking, action kernel, the chain, observations: four launches on the caller's stream */
    OutPtrs op = to_ptrs(out);
    GroupBounds gb;
    gb.b[0] = 0;
    for (int g = 1; g <= RP_MAX_GROUPS; g++) gb.b[g] = N;
    if (h->sort_G == 0) hipLaunchKernelGGL(k_member_identity, dim3((N + 255) / 256), dim3(256), 0, s, h->member[h->member_cur], N, h->hv_cnt);
    else {
      hipLaunchKernelGGL(k_member, dim3(1), dim3(1024), 0, s, h->member[h->member_cur], h->member[h->member_cur ^ 1], h->sort_cnt + (size_t)h->sort_par * RP_MAX_GROUPS * SORT_BINS,
                         h->sort_slot, N, h->sort_G, h->gb, 1, gb, h->hv_cnt);
      h->member_cur ^= 1;
    }
    const int* member = h->member[h->member_cur];
    int* cnt_next = h->sort_cnt + (size_t)(h->sort_par ^ 1) * RP_MAX_GROUPS * SORT_BINS;      /* the last substep's load classes: the next step's ranking */
    HIPCHK(h, hipMemsetAsync(cnt_next, 0, SORT_BINS * sizeof(int), s));
    hipLaunchKernelGGL(k_action, dim3((N + 3) / 4), dim3(64), 0, s, h->dev_model, h->state, action, op.target_poses, 0, N, member);
    const int nq = (N + 3) / 4;
    const int blocks = min(nq, h->chain_blocks);          /* 256 CUs x 4 blocks of 40 KB LDS and 2 x 217 VGPRs: everything resident at once */
    hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64 * SOLVE_WAVES), 0, s, h->dev_model, h->state, h->ws, N, member, h->pair_env, cnt_next, h->sort_slot, K_NSUB, h->debug_flags);
    hipLaunchKernelGGL(k_calc_state, dim3(N), dim3(64), 0, s, h->dev_model, h->state, op, 0, N, member);
    h->sort_par ^= 1; h->sort_G = 1; h->gb = gb;
  } else {
    OutPtrs op = to_ptrs(out);
    hipEvent_t* ev = h->pool ? h->pool + (size_t)h->pool_next * EV_PER_STEP : nullptr;
    /* Env groups: substeps of different envs are independent, so the envs are cut into `groups` contiguous ranges and
     * each range runs its own 25-kernel chain on its own stream; the tail of one group's k_solve2 (its heaviest wave)
     * overlaps with the other groups' kernels.  Per-launch timing (rp_enable_timers) uses one group so that the event
     * pairs bracket exactly one kernel each. */
    int G = ev ? 1 : h->groups;
    if (G > (N + 63) / 64) G = (N + 63) / 64;
    int e = 2;
#ifdef RP_SYNC_DEBUG      /* debugging builds: synchronise after every launch of the chain and say which one died */
#define TIMED(launch) do { fprintf(stderr, "[rp sync] launching %.40s\n", #launch); fflush(stderr); launch; hipError_t e_ = hipDeviceSynchronize(); fprintf(stderr, "[rp sync]   -> %d\n", (int)e_); fflush(stderr); } while (0)
#else
#define TIMED(launch) do { if (ev) hipEventRecord(ev[e++], gs); launch; if (ev) hipEventRecord(ev[e++], gs); } while (0)
#endif
    /* groups by load: rank all envs by the load class of the latest k_solve2 and cut the ranking into the G groups */
    GroupBounds gb;
    {
      long long tot = 0, acc = 0;
      static const int split3[3] = {25, 35, 40};     /* default for 3 groups: the heavy group smaller (2.54 vs 2.59 ms per step, equal thirds) */
      const int* split = h->gsplit;
      if (h->gsplit[0] <= 0 && G == 3) split = split3;
      for (int g = 0; g < G; g++) tot += split[g] > 0 ? split[g] : 0;
      bool custom = tot > 0;
      for (int g = 0; g < G && custom; g++) if (split[g] <= 0) custom = false;
      gb.b[0] = 0;
      for (int g = 0; g < G; g++) {
        acc += custom ? split[g] : 1;
        gb.b[g + 1] = (int)((long long)N * acc / (custom ? tot : G));
      }
      for (int g = G + 1; g <= RP_MAX_GROUPS; g++) gb.b[g] = N;
    }
    const int* member = h->member[h->member_cur];
    if (h->sort_G == 0) {
      hipLaunchKernelGGL(k_member_identity, dim3((N + 255) / 256), dim3(256), 0, s, h->member[h->member_cur], N, h->hv_cnt);
    } else {
      hipLaunchKernelGGL(k_member, dim3(1), dim3(1024), 0, s, member, h->member[h->member_cur ^ 1], h->sort_cnt + (size_t)h->sort_par * RP_MAX_GROUPS * SORT_BINS,
                         h->sort_slot, N, h->sort_G, h->gb, G, gb, h->hv_cnt);
      h->member_cur ^= 1;
      member = h->member[h->member_cur];
    }
    if (G > 1) hipEventRecord(h->gfork, s);
    /* The groups' chains are enqueued ROUND-ROBIN, one launch per group at a time: a chain is 26 dependent launches and its length (not the
     * machine's width) sets the step time at N = 4096, so all chains have to start at once - enqueued group by group, the last group's chain
     * started a whole group's worth of host launch time late (2.38 -> 2.1x ms per step).  Same kernels, same arguments, same order inside
     * every stream. */
    struct GroupCtx { hipStream_t gs; int e0, e1, ng, nab, hb; int* gcnt[2]; int* hvc[2]; } gc[RP_MAX_GROUPS];
    const int par0 = h->sort_par;
    for (int g = 0; g < G; g++) {
      GroupCtx& c = gc[g];
      /* group 0 stays on the caller's stream; a fourth group takes gstream[0], the stream created first: with the default four hardware queues of a process it is
       * the one that has a queue to itself (the kernel trace shows queues 1, 3, 4 for three groups and 2 for gstream[0]); gstream[3] shares one, and two chains on
       * one queue run one after the other (3.6 ms per step).  Four groups are 0.7 % faster than three and leave no queue for anybody else: the default stays 3 */
      c.gs = g == 0 ? s : h->gstream[g == 3 ? 0 : g];
      c.e0 = gb.b[g]; c.e1 = gb.b[g + 1]; c.ng = c.e1 - c.e0; c.nab = (c.ng + 3) / 4;      /* action blocks of 64 threads: one env per 16 lanes (k_action_prep: 128 threads, half as many) */
      /* env pairing of this group: k_solve2 ranks its envs by load class (histogram, double-buffered), the next k_prep2
       * turns the ranks into the table the next k_solve2 reads */
      c.gcnt[0] = h->sort_cnt + (size_t)g * SORT_BINS; c.gcnt[1] = h->sort_cnt + (size_t)(RP_MAX_GROUPS + g) * SORT_BINS;
      /* the heavy envs of a substep (1 - 2 % of the envs: a grasp, a push, a crowded drawer) are listed by its k_prep2 and solved one per wave by worker blocks at the head
       * of k_solve2's grid: an eighth of the group's envs' worth of waves (a longer list is walked in strides) */
      c.hvc[0] = h->hv_cnt + g; c.hvc[1] = h->hv_cnt + RP_MAX_GROUPS + g;
      c.hb = (h->debug_flags & 1) ? c.ng : (h->hv_waves > 0 ? h->hv_waves : max((c.ng + 1) / 2, 16));      /* worker blocks: one heavy env each at a time (debug flag 1: every env is on the list) */
      if (c.gs != s) hipStreamWaitEvent(c.gs, h->gfork, 0);
      if (ev) hipEventRecord(ev[0], c.gs);
      if (h->sort_G == 0) hipLaunchKernelGGL(k_sort_init, dim3((max(c.ng, SORT_BINS) + 255) / 256), dim3(256), 0, c.gs, c.gcnt[par0], h->sort_slot, c.e0, c.ng);
    }
    /* first substep: action kernel and first k_prep2 in one launch (k_action_prep), the motor rows rebuilt by the k_solve2 after
     * it (flag bit 1); with per-launch timers on (one group), the kernels stay apart so that every event pair brackets one of them */
    if (ev) { hipStream_t gs = gc[0].gs; TIMED(hipLaunchKernelGGL(k_action, dim3(gc[0].nab), dim3(64), 0, gs, h->dev_model, h->state, action, op.target_poses, gc[0].e0, gc[0].e1, member)); }
    int par = par0;
    for (int sub = 0; sub < K_NSUB; sub++) {
      const bool merged = !ev && sub == 0;
      for (int g = 0; g < G; g++) {
        const GroupCtx& c = gc[g];
        hipStream_t gs = c.gs;
        if (merged) {
#ifdef RP_SYNC_DEBUG
          fprintf(stderr, "[rp sync] before k_action_prep: %d\n", (int)hipDeviceSynchronize()); fflush(stderr);
#endif
          hipLaunchKernelGGL(k_action_prep, dim3((c.ng + 7) / 8 + c.ng), dim3(PREP_THREADS), 0, gs, h->dev_model, h->state, h->ws, c.e0, c.e1, c.gcnt[par], c.gcnt[par ^ 1], h->sort_slot,
                             h->pair_env, member, action, op.target_poses, (c.ng + 7) / 8, c.hvc[sub & 1], c.hvc[(sub & 1) ^ 1], h->hv_list, h->debug_flags & 1);
#ifdef RP_SYNC_DEBUG
          fprintf(stderr, "[rp sync] after k_action_prep: %d\n", (int)hipDeviceSynchronize()); fflush(stderr);
#endif
        } else
          TIMED(hipLaunchKernelGGL(k_prep2, dim3(c.ng), dim3(PREP_THREADS), 0, gs, h->dev_model, h->state, h->ws, c.e0, c.e1, c.gcnt[par], c.gcnt[par ^ 1], h->sort_slot, h->pair_env, member,
                                   c.hvc[sub & 1], c.hvc[(sub & 1) ^ 1], h->hv_list, h->debug_flags & 1));
      }
      for (int g = 0; g < G; g++) {
        const GroupCtx& c = gc[g];
        hipStream_t gs = c.gs;
        TIMED(hipLaunchKernelGGL(k_solve2, dim3(c.hb + (c.ng + 2 * SOLVE_WAVES - 1) / (2 * SOLVE_WAVES)), dim3(64 * SOLVE_WAVES), 0, gs, h->dev_model, h->state, h->ws, c.e0, c.e1, h->pair_env, c.gcnt[par ^ 1], h->sort_slot,
                                 h->debug_flags | (merged ? RP_DBG_MOTOR : 0) | RP_DBG_WORKERS, (const int*)c.hvc[sub & 1], (const int*)h->hv_list, c.hb));
      }
      par ^= 1;
    }
    h->sort_par = par; h->sort_G = G; h->gb = gb;
    for (int g = 0; g < G; g++) {
      const GroupCtx& c = gc[g];
      hipStream_t gs = c.gs;
      TIMED(hipLaunchKernelGGL(k_calc_state, dim3(c.ng), dim3(64), 0, gs, h->dev_model, h->state, op, c.e0, c.e1, member));
      if (ev) hipEventRecord(ev[1], gs);
      if (gs != s) hipEventRecord(h->gjoin[g], gs);
    }
    for (int g = 1; g < G; g++) hipStreamWaitEvent(s, h->gjoin[g], 0);
    if (ev) {
      h->pool_next = (h->pool_next + 1) % h->pool_steps;
      if (h->pool_count < h->pool_steps) h->pool_count++;
    }
#undef TIMED
  }
  HIPCHK(h, hipGetLastError());
  if (h->timers_on && h->fused == 1) { hipEventRecord(h->ev1, s); hipEventSynchronize(h->ev1); hipEventElapsedTime(&h->timers.last_step_ms, h->ev0, h->ev1); }
  h->timers.steps++;
  return RP_OK;
}

int rp_calc_state(rp_handle h, const rp_out* out, void* stream) {
  if (!h) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  int N = h->cfg.num_envs;
  hipLaunchKernelGGL(k_calc_state, dim3(N), dim3(64), 0, (hipStream_t)stream, h->dev_model, h->state, to_ptrs(out), 0, N, (const int*)nullptr);
  HIPCHK(h, hipGetLastError());
  return RP_OK;
}

static int reward_impl(rp_handle h, const float* ag, const float* dg, float* r, int32_t m, void* stream, int force_sparse) {
  if (!h || !ag || !dg || !r || m < 0) return RP_ERR_ARG;
  if (m == 0) return RP_OK;
  DevGuard guard(h->cfg.device);
  hipLaunchKernelGGL(k_reward, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->dev_model, ag, dg, r, m, force_sparse);
  HIPCHK(h, hipGetLastError());
  return RP_OK;
}
int rp_compute_reward(rp_handle h, const float* ag, const float* dg, float* r, int32_t m, void* stream) { return reward_impl(h, ag, dg, r, m, stream, 0); }
int rp_compute_reward_sparse(rp_handle h, const float* ag, const float* dg, float* r, int32_t m, void* stream) { return reward_impl(h, ag, dg, r, m, stream, 1); }

size_t rp_state_bytes(rp_handle h) { return (RP_REC_FLOATS + (h && h->pmcache ? PMC_FLOATS : 0)) * sizeof(float); }

int rp_get_state(rp_handle h, void* dst, void* stream) {
  if (!h || !dst) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  const int N = h->cfg.num_envs, nc = h->pmcache ? PMC_FLOATS : 0;
  const size_t total = (size_t)N * (RP_REC_FLOATS + nc);
  hipLaunchKernelGGL(k_read_state, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float*)dst, h->state, h->pmcache, N, nc);
  HIPCHK(h, hipGetLastError());
  return RP_OK;
}

int rp_set_state(rp_handle h, const void* src, int32_t src_env_count, void* stream) {
  if (!h || !src) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  int N = h->cfg.num_envs;
  if (src_env_count != 1 && src_env_count != N) { snprintf(h->err, 256, "rp_set_state: src_env_count %d is neither 1 nor %d", src_env_count, N); return RP_ERR_STATE_SIZE; }
  const int nc = h->pmcache ? PMC_FLOATS : 0;
  const size_t total = (size_t)N * (RP_REC_FLOATS + nc);
  hipLaunchKernelGGL(k_copy_state, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->state, h->pmcache, (const float*)src, N, src_env_count, nc);
  HIPCHK(h, hipGetLastError());
  return RP_OK;
}

int rp_set_debug_flags(rp_handle h, int32_t flags) { if (!h) return RP_ERR_ARG; h->debug_flags = flags & 1; return RP_OK; }   /* bit 1 is internal (k_action_prep) */
int rp_set_groups(rp_handle h, int32_t groups) { if (!h || groups < 1 || groups > RP_MAX_GROUPS) return RP_ERR_ARG; h->groups = groups; return RP_OK; }
int rp_set_fused(rp_handle h, int32_t fused) {
  if (!h || fused < 0 || fused > 2) return RP_ERR_ARG;
  if (fused == 2 && h->timers_on) { snprintf(h->err, 256, "rp_set_fused: the k_chain pipeline has no per-launch timers (rp_enable_timers(h, 0) first)"); return RP_ERR_ARG; }
  if (fused != h->fused) h->sort_G = 0;          /* the load tables of one pipeline mean nothing to the other: start from the identity ranking */
  h->fused = fused;
  return RP_OK;
}
static int rc_reserve(rp_handle h, int num) {
  if (num <= h->rc_cap) return RP_OK;
  hipFree(h->rc_tab); hipFree(h->rc_cnt); hipFree(h->rc_ee);
  h->rc_tab = nullptr; h->rc_cnt = nullptr; h->rc_ee = nullptr; h->rc_cap = 0;
  HIPCHK(h, hipMalloc((void**)&h->rc_tab, (size_t)num * RC_MAX * RC_STRIDE * sizeof(float)));
  HIPCHK(h, hipMalloc((void**)&h->rc_cnt, (size_t)num * sizeof(int)));
  HIPCHK(h, hipMalloc((void**)&h->rc_ee, (size_t)num * 20 * sizeof(float)));      /* [num][12] EE poses, then [num][8] the ghost arm's joints of the latest rp_render_ex (rp_debug_ghost_joints) */
  h->rc_cap = num;
  return RP_OK;
}

int rp_camera_from_yaw_pitch_roll(const float target[3], float distance, float yaw_deg, float pitch_deg, float roll_deg, rp_camera* cam) {
  if (!target || !cam) return RP_ERR_ARG;
  /* b3ComputeViewMatrixFromYawPitchRoll, upAxis 2: eye = target + R (0, -distance, 0), up = R (0, 0, 1), R = Rz(yaw) Ry(roll) Rx(pitch) */
  const double d2r = 0.01745329251994329547, y = yaw_deg * d2r, p = pitch_deg * d2r, r = roll_deg * d2r;
  const double cy = cos(y), sy = sin(y), cp = cos(p), sp = sin(p), cr = cos(r), sr = sin(r);
  const double R[9] = {cy * cr, cy * sr * sp - sy * cp, cy * sr * cp + sy * sp, sy * cr, sy * sr * sp + cy * cp, sy * sr * cp - cy * sp, -sr, cr * sp, cr * cp};
  memset(cam, 0, sizeof(*cam));
  for (int k = 0; k < 3; k++) { cam->target[k] = target[k]; cam->eye[k] = target[k] + (float)(R[3 * k + 1] * -distance); cam->up[k] = (float)R[3 * k + 2]; }
  cam->fov_deg = 50.f; cam->aspect = 1.f; cam->mode = 0;
  return RP_OK;
}
int rp_default_camera(rp_camera* cam) {
  const float target[3] = {0.f, 0.25f, 0.f};
  return rp_camera_from_yaw_pitch_roll(target, 1.3f, -30.f, -30.f, 0.f, cam);
}

static RpCamera device_camera(const rp_camera* c) {
  RpCamera d; memset(&d, 0, sizeof(d));
  double f[3], u[3], s[3], n = 0;
  for (int k = 0; k < 3; k++) { f[k] = c->target[k] - c->eye[k]; n += f[k] * f[k]; }
  n = sqrt(n > 0 ? n : 1);
  for (int k = 0; k < 3; k++) f[k] /= n;
  s[0] = f[1] * c->up[2] - f[2] * c->up[1]; s[1] = f[2] * c->up[0] - f[0] * c->up[2]; s[2] = f[0] * c->up[1] - f[1] * c->up[0];
  n = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]); n = n > 0 ? n : 1;
  for (int k = 0; k < 3; k++) s[k] /= n;
  u[0] = s[1] * f[2] - s[2] * f[1]; u[1] = s[2] * f[0] - s[0] * f[2]; u[2] = s[0] * f[1] - s[1] * f[0];
  for (int k = 0; k < 3; k++) { d.eye[k] = c->eye[k]; d.fwd[k] = (float)f[k]; d.right[k] = (float)s[k]; d.up[k] = (float)u[k]; }
  d.tan_half_fov = (float)tan(0.5 * c->fov_deg * 0.01745329251994329547); d.aspect = c->aspect; d.mode = c->mode;
  return d;
}

int rp_render_ex(rp_handle h, const rp_camera* cam, int32_t width, int32_t height, int32_t first_env, int32_t num_envs, uint8_t* rgb,
                 const float* sub_goal, const float* ghost_arm, void* stream) {
  if (!h || !rgb || width <= 0 || height <= 0 || width > 4096 || height > 4096 || first_env < 0 || num_envs <= 0 || first_env + num_envs > h->cfg.num_envs) {
    if (h) snprintf(h->err, 256, "rp_render: bad argument"); return RP_ERR_ARG;
  }
  if (ghost_arm && h->host_model.arm_type != RP_ARM_PANDA) {      /* environments.py:629-630: the reference has a ghost arm for the Panda only and raises for the UR5 */
    snprintf(h->err, 256, "rp_render_ex: the ghost arm exists for the Panda only (environments.py:623-630)"); return RP_ERR_UNSUPPORTED;
  }
  DevGuard guard(h->cfg.device);
  rp_camera def;
  if (!cam) { rp_default_camera(&def); cam = &def; }
  if (cam->mode < 0 || cam->mode > 1 || !(cam->fov_deg > 0.f && cam->fov_deg < 180.f) || !(cam->aspect > 0.f)) { snprintf(h->err, 256, "rp_render: bad camera"); return RP_ERR_ARG; }
  int rc = rc_reserve(h, num_envs);
  if (rc != RP_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  h->rc_last_num = num_envs;
  hipLaunchKernelGGL(k_collider_poses, dim3(num_envs), dim3(64), 0, s, h->dev_model, h->state, first_env, num_envs, h->rc_tab, h->rc_cnt, sub_goal, h->rc_ee, ghost_arm);
  const int tiles = (width * height + 255) / 256;
  hipLaunchKernelGGL(k_render, dim3((unsigned)num_envs * tiles), dim3(256), 0, s, h->dev_model, h->rc_tab, h->rc_cnt, num_envs, device_camera(cam), h->rc_ee, width, height, rgb);
  HIPCHK(h, hipGetLastError());
  return RP_OK;
}

int rp_render(rp_handle h, const rp_camera* cam, int32_t width, int32_t height, int32_t first_env, int32_t num_envs, uint8_t* rgb,
              const float* sub_goal, void* stream) {
  return rp_render_ex(h, cam, width, height, first_env, num_envs, rgb, sub_goal, nullptr, stream);
}

int rp_ray_test(rp_handle h, const float* from, const float* to, int32_t k, float* hit_fraction, int32_t* collider, int32_t* link,
                float* hit_position, float* hit_normal, void* stream) {
  if (!h || !from || !to || k <= 0) { if (h) snprintf(h->err, 256, "rp_ray_test: bad argument"); return RP_ERR_ARG; }
  DevGuard guard(h->cfg.device);
  const int N = h->cfg.num_envs;
  int rc = rc_reserve(h, N);
  if (rc != RP_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_collider_poses, dim3(N), dim3(64), 0, s, h->dev_model, h->state, 0, N, h->rc_tab, h->rc_cnt, (const float*)nullptr, (float*)nullptr, (const float*)nullptr);
  hipLaunchKernelGGL(k_ray_test, dim3(N), dim3(64), 0, s, h->dev_model, h->rc_tab, h->rc_cnt, N, k, from, to, hit_fraction, collider, link, hit_position, hit_normal);
  HIPCHK(h, hipGetLastError());
  return RP_OK;
}

int rp_get_timers(rp_handle h, rp_timers* t) {
  if (!h || !t) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  rp_timers r = h->timers;
  r.steps_timed = 0; r.avg_step_ms = r.avg_action_ms = r.avg_prep_ms = r.avg_solve_ms = r.avg_obs_ms = 0.f;
  if (h->pool && h->pool_count > 0) {
    double step = 0, act = 0, prep = 0, solve = 0, obs = 0;
    for (int k = 0; k < h->pool_count; k++) {
      hipEvent_t* ev = h->pool + (size_t)k * EV_PER_STEP;
      HIPCHK(h, hipEventSynchronize(ev[1]));
      float ms;
      hipEventElapsedTime(&ms, ev[0], ev[1]); step += ms;
      hipEventElapsedTime(&ms, ev[2], ev[3]); act += ms;
      for (int sub = 0; sub < K_NSUB; sub++) {
        hipEventElapsedTime(&ms, ev[4 + 4 * sub], ev[5 + 4 * sub]); prep += ms;
        hipEventElapsedTime(&ms, ev[6 + 4 * sub], ev[7 + 4 * sub]); solve += ms;
      }
      hipEventElapsedTime(&ms, ev[4 + 4 * K_NSUB], ev[5 + 4 * K_NSUB]); obs += ms;
    }
    int c = h->pool_count;
    r.steps_timed = (uint32_t)c;
    r.avg_step_ms = (float)(step / c); r.avg_action_ms = (float)(act / c); r.avg_prep_ms = (float)(prep / (c * K_NSUB));
    r.avg_solve_ms = (float)(solve / (c * K_NSUB)); r.avg_obs_ms = (float)(obs / c);
    r.last_step_ms = r.avg_step_ms;
  }
  *t = r;
  return RP_OK;
}
int rp_enable_timers(rp_handle h, int32_t on) {
  if (!h || on < 0) return RP_ERR_ARG;
  if (on > 0 && h->fused == 2) { snprintf(h->err, 256, "rp_enable_timers: the k_chain pipeline (rp_set_fused(h, 2)) is one launch for twelve substeps - no per-launch timers"); return RP_ERR_ARG; }
  DevGuard guard(h->cfg.device);
  if (h->pool) { for (int i = 0; i < h->pool_steps * EV_PER_STEP; i++) hipEventDestroy(h->pool[i]); free(h->pool); h->pool = nullptr; }
  h->timers_on = on; h->pool_steps = on; h->pool_next = 0; h->pool_count = 0;
  if (on > 0) {
    h->pool = (hipEvent_t*)calloc((size_t)on * EV_PER_STEP, sizeof(hipEvent_t));
    for (int i = 0; i < on * EV_PER_STEP; i++) HIPCHK(h, hipEventCreate(&h->pool[i]));
  }
  return RP_OK;
}
const char* rp_last_error(rp_handle h) { return h ? h->err : g_err; }

int rp_debug_reset_rounds(rp_handle h) { return h ? h->reset_rounds : RP_ERR_ARG; }

/* test hooks (include/rp_playroom_debug.h): one substep on every env, intermediates of env `env` into host buf[4096] */
int rp_debug_substep(rp_handle h, int32_t env, float* host_buf) {
  if (!h || !host_buf) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  int N = h->cfg.num_envs;
  HIPCHK(h, hipMemset(h->dbg, 0, 4096 * sizeof(float)));
  hipLaunchKernelGGL(k_debug_substep, dim3(N), dim3(64), 0, 0, h->dev_model, h->state, h->dbg, N, env);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(host_buf, h->dbg, 4096 * sizeof(float), hipMemcpyDeviceToHost));
  return RP_OK;
}

/* test hook: the joints of the ghost arm(s) of the latest rp_render_ex with a ghost_arm, [num_envs of that call][8], into host_buf */
int rp_debug_ghost_joints(rp_handle h, float* host_buf, int32_t num_envs) {
  if (!h || !host_buf || num_envs < 1 || num_envs > h->rc_cap || !h->rc_ee) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(host_buf, h->rc_ee + (size_t)12 * h->rc_last_num, (size_t)num_envs * 8 * sizeof(float), hipMemcpyDeviceToHost));
  return RP_OK;
}

/* test hook: per-env (nsmall, ncon) of the most recent k_prep2 into host_buf[2*N] */
int rp_debug_row_counts(rp_handle h, int32_t* host_buf) {
  if (!h || !host_buf) return RP_ERR_ARG;
  DevGuard guard(h->cfg.device);
  HIPCHK(h, hipDeviceSynchronize());
  int N = h->cfg.num_envs;
  for (int e = 0; e < N; e++) {   /* header: maskL, maskU, nj, ncon, nA, nB, gear, nC -> (unit rows, ncon + 1000 * (arm contact) + 1e5 * spanning contacts) */
    int32_t hdr[8];
    HIPCHK(h, hipMemcpy(hdr, h->ws + (size_t)e * W3_FLOATS, 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
    host_buf[2 * e] = 12 + __builtin_popcount((unsigned)hdr[0]) + __builtin_popcount((unsigned)hdr[1]) + hdr[2];
    host_buf[2 * e + 1] = hdr[3] + 1000 * ((hdr[4] + hdr[7]) > 0) + 100000 * hdr[7];
  }
  return RP_OK;
}

#ifdef RP_PROLOGUE_CLOCKS
int rp_debug_prologue_clocks(rp_handle h, uint64_t* host_buf, int32_t nwaves) {
  if (!h || !host_buf || nwaves > 4096) return RP_ERR_ARG;
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_pclk), (size_t)nwaves * 8 * sizeof(uint64_t)));
  return RP_OK;
}
#endif
#ifdef RP_CHAIN_CLOCKS
int rp_debug_chain_clocks(rp_handle h, int64_t* host_buf, int32_t nblocks) {
  if (!h || !host_buf || nblocks > 4096) return RP_ERR_ARG;
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_chain_clk), (size_t)nblocks * 4 * sizeof(int64_t)));
  return RP_OK;
}
#endif
#ifdef RP_CLOCKS
#if RP_CLOCKS == 2
#define RP_CLK_STRIDE 32
#else
#define RP_CLK_STRIDE 8
#endif
int rp_debug_clocks(rp_handle h, uint64_t* host_buf, int32_t nblocks) {
  if (!h || !host_buf || nblocks > 4096) return RP_ERR_ARG;
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_clk), (size_t)nblocks * RP_CLK_STRIDE * sizeof(uint64_t)));
  return RP_OK;
}
#endif

}  /* extern "C" */
