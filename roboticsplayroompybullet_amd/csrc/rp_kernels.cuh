/* rp_kernels.cuh — the batched playroom env-step for gfx950 (MI355X), hand-written HIP.
 *
 * rp_step runs right-sized kernels per env group (see "split pipeline" below): k_action_prep (16 lanes per env for the
 * cooperative IK, in one launch with the first substep's preparation), 12 x { k_prep2 (one env per block of two waves, PrepLds:
 * wave 0 bodies (FK) -> colliders (AABBs) -> candidate pairs (broadphase, 64 per sweep) -> active pairs (cooperative narrowphase,
 * 8 lanes per pair) -> manifolds -> contact list, beside wave 1 joint subspaces -> arm links (CRBA / RNEA about a per-substep
 * reference point) -> fp64 Cholesky -> v* -> unit rows; joined for the contact rows (Jacobians, M^-1 J^T) -> per-env workspace),
 * k_solve2 (two envs per wave: 50 PGS sweeps over register-resident rows + integration; lane l owns velocity component l of its
 * env, row dot products are DPP reductions) }, k_calc_state (one wave per env: observation, reward, outputs).
 * The first design - the whole env step as ONE kernel, one wave per env, everything staged in LDS - is kept as k_step / k_reset:
 * the in-library cross-check (bit-identical to the split pipeline, tests/) and the path of rp_reset_to.
 *
 * What the pieces restate (reference file:line via the CPU oracle, oracle/rp_oracle.c, which tests compare against):
 *   perform_action/goto/close_gripper  environments.py:915-1073, inverseKinematics.py:44-50
 *   substep (stepSimulation)           SURVEY.md App. E recollection of Bullet's multibody step (parity unpinned)
 *   calc_state / rewards               environments.py:746-894, 278-304; playRewardFunc.py:16-77
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rp_device_model.h"
#include "rp_math.cuh"

#define NB_MAX (1 + RP_MAX_ARM + RP_MAX_FREE + RP_MAX_J1)
#define NVP 28               /* padded row stride (nv <= 27) */
#define MAXC 21              /* contact points kept per env per substep (shared cap with the oracle): 63 contact rows */
#define MAXACT 64            /* AABB-overlapping pairs examined per substep (shared cap with the oracle) */
#define MAXT 4               /* torsional friction rows kept per env per substep (shared cap with the oracle): one per collider pair in contact whose links carry spinning_friction */
#define MAXROWC (3 * MAXC + MAXT)
static_assert(MAXACT <= 64 && NB_MAX <= 32, "a candidate record packs its pair index in 6 bits and its bodies in 5 bits each");
/* persistent contact manifolds (DevModel.persist; oracle RPO_RULE_PERSIST, collide_persistent): the per-env contact cache.  Header: number of manifolds, 3 pad.
 * Manifold (PMC_MAN floats): object-pair key (objA * 256 + objB) | points | breaking threshold | flags of the pair (bits 16.. of LDS.key, rebuilt every substep)
 * | 4 pad | 4 points x PMC_PT: point in A's body frame, in B's, normal, distance, colliders (a | b << 8) */
#define PM_MAX 11
#define PMC_HDR 4
#define PMC_PT 11
#define PMC_MAN (8 + 4 * PMC_PT)
#define PMC_MANIFOLDS (PMC_HDR + PM_MAX * PMC_MAN)   /* 576: the manifolds (staged in LDS by collide()) */
/* ... and behind them what GJK's distance phase last ended with for a (hull, box) collider pair (oracle hull_box_gjk, rpo_env.gax): PMC_AXN slots, direct-mapped by the
 * baked pair index, 8 words each: tag (pair index + 1, 0 = empty) | n + corner codes << 4 (three bits per simplex point, four bits apart) | three hull vertex numbers |
 * the direction v in the box's frame.  Read and written in place (global memory), one slot per GJK call */
#define PMC_AX PMC_MANIFOLDS
#define PMC_AXN 16
#define PMC_FLOATS (PMC_MANIFOLDS + 8 * PMC_AXN)      /* 704: a row of the contact cache (part of the state rows) */
#define RP_MAX_GROUPS 16      /* env groups (streams) rp_step can cut the envs into */
#define SORT_KEYS 64          /* load classes for pairing envs in k_solve2: 8 * min(folded slots, 7) + min((side-by-side slots - 1) / 2, 7) */
#define SORT_REPS 8           /* histogram replicas (env & 7): one hot word would serialise ~4096 atomics at ~90 per us */
#define SORT_BINS (SORT_KEYS * SORT_REPS)
#define SORT_RANK_BITS 22     /* sort_slot = (bin << 22) | rank inside the bin: 512 bins, ranks < 4 M (rp_create bounds num_envs) */
#define SORT_RANK_MASK ((1 << SORT_RANK_BITS) - 1)
#define RP_MAX_ENVS (1 << SORT_RANK_BITS)
#define MAXSMALL 44          /* arm motors 12 + scene-joint motors 3 + limits 24 + gear 1 (+ pad) */

#ifndef RP_PREP_WAVES
#define RP_PREP_WAVES 4      /* k_prep2: at most 128 VGPRs, four waves per SIMD (its LDS block allows 16 blocks per CU) */
#endif
#ifndef RP_WAVES_PER_EU
#define RP_WAVES_PER_EU 1      /* register budget: 512 / RP_WAVES_PER_EU VGPR+AGPR per lane */
#endif

#define K_DT (1.0f / 300.0f)
#define K_GRAVITY (-9.8f)
#define K_NSUB 12
#define K_NSETTLE 100
#define K_MAXVEL 100.0f       /* btMultiBody::m_maxCoordinateVelocity: clamp of every generalized velocity after the solve */
#ifdef K_NITER_OVERRIDE   /* timing ablations only */
#define K_NITER K_NITER_OVERRIDE
#else
#define K_NITER 50
#endif
#define K_ERP 0.08f
#define K_SLOP 1e-5f
#define K_TIE_EPS 1e-6f
#define K_KP 0.1f
#define K_DEFMOTOR 1.0f
#define K_LIMIT_MAXIMP 100.0f
#define K_ERP_LIMIT 0.2f      /* btContactSolverInfo::m_erp: what a violated joint limit pushes back with */
#define K_LIMIT_ACTIVATION 0.1f   /* RP_CFG_SPECULATIVE_LIMITS (round 2's rule): a limit row exists from this distance before the limit on */
#define K_LIN_DAMP 0.04f
#define K_ANG_DAMP 0.04f
#define K_IK_DAMP 0.1f
#define K_IK_RES 1e-4f
#define K_IK_MAXSTEP (45.0f * RP_PI_F / 180.0f)

#define ROWW 18              /* compact Jacobian row: slot0 = 12 entries at dof offset off0, slot1 = 6 entries at off1 */
#define ROWREG ((ROWW * MAXROWC + 3) & ~3)   /* floats reserved for all compact rows: 16-byte copies must not overrun */

/* Per-env LDS block.  Phase-local scratch (collision, dynamics, Jacobian rows) shares one union: the phases of a
 * substep run strictly one after another, separated by barriers. */
#ifdef RP_WIDE
#define O_FLOATS 160            /* output block of one env (O_* offsets at calc_state) */
#else
#define O_FLOATS 128
#endif
#define CANDMAX 64             /* candidate points examined per env per substep, in pair order (shared cap with the oracle) */
#define MANPTS (MAXC + 3)      /* merged manifold points that can still reach the solver: the manifold that crosses MAXC is merged whole */
#define NPSCR_FLOATS 768       /* narrowphase scratch: 96 floats for each of the 8 lane groups */
#define PREP_CH 8              /* k_prep2 builds the contact rows of PREP_CH contacts at a time (PrepLds) */

/* The phases of a substep are templates over the LDS block; both blocks below carry the same member names.
 * EnvLds: the one-kernel path (k_step, resets, calc_state): everything of a substep at once, the solver sweeps read all rows from LDS. */
/* The hull pairs of a substep, pooled over all active pairs (narrowphase_coop's first phase; hull_item): per active pair its colliders | margin | baked pair index,
 * its outcome (-1 no hull pair / the OBB path, 0 apart, 1 contact) and the staged contact; the pairs of each GJK cache slot as a 64-bit mask (a CLASS: worked off in
 * pair order by one wave); a GJK scratch per wave; and k_prep2's hand-over between its two waves: unclaimed classes | classes done | - | narrowphase over */
#define HULL_POOL_FIELDS int hinfo[MAXACT][4]; int hout[MAXACT]; float hpt[MAXACT][7]; unsigned long long hcls[PMC_AXN];
struct __align__(16) EnvLds {
  float st[RP_REC_FLOATS];
  float xR[NB_MAX * 9], xp[NB_MAX * 3];
  float O[4];
  float S[RP_MAX_ARM * 6];
  float Minv[144], tau[RP_MAX_ARM];
  float finv[RP_MAX_FREE * 9];
  float vstar[32];
  float conp[MAXC * 3], conn[MAXC * 3], cond[MAXC], conmu[MAXC];
  int cona[MAXC], conb[MAXC], conk[MAXC];     /* colliders of the contact; class: 0 no arm dof, 1 arm only, 2 spanning */
  int torc[MAXT]; float tors[MAXT];           /* torsional rows: parent contact, coefficient (tors_list) */
  alignas(16) float srow[MAXSMALL * 8];      /* type, dofA, sign/ratio, rhs | dinv, lo, hi, dofB */
  alignas(16) float rowS[MAXROWC * 4];       /* rhs, cfm * dinv (soft normal rows, else 0), mu, parent */
  alignas(16) float rowT[MAXROWC * 4];       /* lo_c, hi_c, off0, off1 */
  union alignas(16) {
    struct {                                   /* collide() */
      float aabb[RP_MAX_COL * 8];              /* lo.xyz, contact margin of the collider (DevModel.col_margin; a pair's is the smaller) | hi.xyz, - */
      int act[MAXACT], candn[MAXACT], key[MAXACT];   /* active pair -> baked pair index | number of candidate points + 256 * their offset | manifold key */
      float pmu[MAXACT];                       /* ... | friction coefficient of the pair */
      alignas(16) float cand[CANDMAX * 8];     /* candidate points, compact, in pair order (fk_bodies' scratch before that) */
      alignas(16) float man[MANPTS * 8];                 /* merged manifolds, in solver-bound order */
      float npscr[NPSCR_FLOATS];
      HULL_POOL_FIELDS
    };
    struct {                                   /* arm_dynamics() */
      float inert[RP_MAX_ARM * 10], compI[RP_MAX_ARM * 10];
      float vsp[RP_MAX_ARM * 6], csp[RP_MAX_ARM * 6], fsp[RP_MAX_ARM * 6], Fv[RP_MAX_ARM * 6];
      double Md[144];                          /* mass matrix / Cholesky factor in fp64 */
    };
    struct { float J[ROWREG]; float B[ROWREG]; };                    /* contact rows */
  };
  alignas(16) float out[O_FLOATS];
};

/* PrepLds: k_prep2 / k_settle_prep / the prep blocks of k_action_prep: one env per block of TWO waves.  A substep's preparation is a chain
 * of latency-bound phases; collision detection (AABBs, broadphase, narrowphase, manifolds: 26 k cycles) and the arm's dynamics (joint
 * subspaces, CRBA, Cholesky, bias forces, v*, the unit rows: 24 k) need nothing from each other, so wave 0 runs the first and wave 1 the
 * second, and only the contact rows wait for both.  What the kernel needs beyond that is resident waves: its registers (108 - 128 VGPRs) allow sixteen per CU = eight blocks, so
 * the block may take 160 KB / 8 = 20 KB of LDS and no more.  Lifetimes:
 *   whole kernel   st, body transforms, joint subspaces, the contact list, slot tables, M^-1, tau, v*, free-body inverse inertias
 *   wave 0         AABBs (dead after the broadphase: the narrowphase scratch and then the merged manifolds take their place), active-pair
 *                  tables, candidate points; the hull pool (HULL_POOL_FIELDS)
 *   wave 1         the dynamics scratch, then over it the small rows; after the join ONE CHUNK of contact rows (PREP_CH contacts: built,
 *                  copied to the workspace, next chunk); aout (the unit rows in solver form) has left by then */
struct __align__(16) PrepLds {
  float st[RP_REC_FLOATS];
  int roff[64];
  int slot[64];
  unsigned amask[4];
  int hdr[4];                                  /* ncon (wave 0) | gear present (wave 1) */
  float O[4];
  float S[RP_MAX_ARM * 6];
  float xR[NB_MAX * 9], xp[NB_MAX * 3];
  float conp[MAXC * 3], conn[MAXC * 3], cond[MAXC], conmu[MAXC];
  int cona[MAXC], conb[MAXC], conk[MAXC];
  int torc[MAXT]; float tors[MAXT];
  union alignas(16) {                          /* wave 0: collide() */
    float aabb[RP_MAX_COL * 8];
    float npscr[NPSCR_FLOATS];
    float man[MANPTS * 8];
  };
  int act[MAXACT], candn[MAXACT], key[MAXACT];
  float pmu[MAXACT];
  alignas(16) float cand[CANDMAX * 8];
  alignas(16) float Minv[144];
  float tau[RP_MAX_ARM];
  float finv[RP_MAX_FREE * 9];
  float vstar[32];
  union alignas(16) {
    struct {                                   /* wave 1: arm_dynamics() */
      float inert[RP_MAX_ARM * 10], compI[RP_MAX_ARM * 10];
      float vsp[RP_MAX_ARM * 6], csp[RP_MAX_ARM * 6], fsp[RP_MAX_ARM * 6], Fv[RP_MAX_ARM * 6];
      double Md[144];
    };
    struct {
      float srow[MAXSMALL * 8];
      float rowS[3 * PREP_CH * 4], rowT[3 * PREP_CH * 4];
      union {
        struct { float J[3 * PREP_CH * ROWW], B[3 * PREP_CH * ROWW]; };
        float aout[192];
      };
    };
  };
  HULL_POOL_FIELDS
#ifdef RP_LDS_PAD                 /* occupancy experiments only: fewer k_prep2 blocks per CU */
  float pad[RP_LDS_PAD];
#endif
};
static_assert(offsetof(EnvLds, aabb) % 16 == 0 && offsetof(PrepLds, aabb) % 16 == 0 && offsetof(EnvLds, cand) % 16 == 0 && offsetof(EnvLds, man) % 16 == 0 && offsetof(EnvLds, srow) % 16 == 0 && offsetof(EnvLds, rowS) % 16 == 0 &&
              offsetof(EnvLds, rowT) % 16 == 0 && offsetof(EnvLds, J) % 16 == 0 && offsetof(EnvLds, B) % 16 == 0, "16-byte LDS accesses");
static_assert(sizeof(PrepLds) <= 20480 - 512, "k_prep2: eight blocks (sixteen waves: what its 128 VGPRs allow) per CU");
static_assert(offsetof(PrepLds, roff) % 16 == 0 && offsetof(PrepLds, slot) % 16 == 0 && offsetof(PrepLds, Minv) % 16 == 0 && offsetof(PrepLds, aout) % 16 == 0 &&
              offsetof(PrepLds, Md) % 8 == 0 && offsetof(PrepLds, cand) % 16 == 0 && offsetof(PrepLds, man) % 16 == 0, "16-byte copies out of LDS");

/* ObsLds: k_calc_state alone - the record, the body transforms (and fk_bodies' scratch), the joint subspaces, the output block: 2.8 KB instead of EnvLds' 17, so that
 * every env of a 4096-env step is resident at once (EnvLds: six blocks per CU, three rounds - 0.048 ms for a kernel that assembles an observation) */
struct __align__(16) ObsLds {
  float st[RP_REC_FLOATS];
  float xR[NB_MAX * 9], xp[NB_MAX * 3];
  float O[4];
  float S[RP_MAX_ARM * 6];
  alignas(16) float cand[2 * 12 * RP_MAX_ARM + 2 * RP_MAX_ARM + 4];      /* (fk_bodies' scratch) */
  alignas(16) float out[O_FLOATS];
};
#ifdef RP_CLOCKS      /* profiling build only: per-wave phase timestamps of the last k_solve2 (1) / k_prep2 (2) launch */
__device__ unsigned long long g_clk[32 * 4096];
#define CLK_MARK(i) if (lane == 0) { g_clk[8 * wb + (i)] = __builtin_readcyclecounter(); }      /* (k_solve2: wb = the wave's number in the launch) */
#if RP_CLOCKS == 2
#define PCLK(i) if (lane == 0) { g_clk[32 * (blockIdx.x & 4095) + (i)] = (i) >= 6 && (i) < 8 ? wall_clock64() : __builtin_readcyclecounter(); }
#define PCLK_ZERO(i) if (lane == 0) { g_clk[32 * (blockIdx.x & 4095) + (i)] = 0ull; }
#define PCLK_ADD(i, v) if (lane == 0) { g_clk[32 * (blockIdx.x & 4095) + (i)] += (unsigned long long)(v); }
#define HCLK_ADD(i, v) if (l16 == 0) { atomicAdd(&g_clk[32 * (blockIdx.x & 4095) + (i)], (unsigned long long)(v)); }      /* (hull_item16: the four rows of a wave count side by side) */
#ifdef RP_HPROF      /* experiment: slots 29-31 = cycles inside hull_item | in the narrowphase's hull section | waiting for the other wave (first wave only) */
#define PCLK_G(i, v)
#define PCLK_H(i, v) PCLK_ADD(i, v)
#else
#define PCLK_G(i, v) PCLK_ADD(i, v)
#define PCLK_H(i, v)
#endif
#define CLK_MARK2(i)
#else
#define PCLK(i)
#define PCLK_ZERO(i)
#define PCLK_ADD(i, v)
#define HCLK_ADD(i, v)
#define PCLK_G(i, v)
#define PCLK_H(i, v)
#define CLK_MARK2(i) CLK_MARK(i)
#endif
#else
#define CLK_MARK(i)
#define CLK_MARK2(i)
#define PCLK(i)
#define PCLK_ZERO(i)
#define PCLK_ADD(i, v)
#define HCLK_ADD(i, v)
#define PCLK_G(i, v)
#define PCLK_H(i, v)
#endif

/* The phases of a substep each run inside ONE wave (k_prep2 runs two of them side by side in the two waves of its block), so what they need
 * between a lane's LDS write and another lane's read is program order, not a workgroup barrier: LDS operations of a wave execute in order. */
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

/* ------------------------------------------------------------------ small helpers */
__device__ __forceinline__ int dof_free(const DevModel* m, int k) { return m->n_arm + 6 * k; }
__device__ __forceinline__ int dof_j1(const DevModel* m, int k) { return m->n_arm + 6 * m->n_free + k; }
/* lane layout of the velocity vector inside a 32-lane group: arm dof i at lane i (DPP row 0); in DPP row 1 scene
 * joint k at lane 16 + k (k < 3) and component c of free body f at lane 19 + 6 f + c, so that arm-only and non-arm
 * rows reduce in different DPP rows and every unit row (motor, limit) sits at a compile-time lane of its DPP row.
 * A free body of rp_model.free_row0 (the rotation-locked drawer) lives in DPP row 0 instead, its three translation components
 * at lanes n_arm .. n_arm + 2: its contacts against the world (always there: it rests on its rails) then ride in row 0 side
 * by side with the block's contacts in row 1, instead of queueing behind them.  Its angular velocity has no lane: it is zero and
 * stays zero (no inverse inertia), so every product it would enter is an exact zero. */
#ifndef RP_WIDE
#define LANE_J1 16
#define LANE_FREE 19
__device__ __forceinline__ int lane_pos(const DevModel* m, int d) {
  if (d < m->n_arm) return d;
  const int e = d - m->n_arm, f6 = 6 * m->n_free;
  if (e >= f6) return LANE_J1 + (e - f6);
  const int f = e / 6, c = e - 6 * f;
  if ((m->free_row0 >> f) & 1) return c < 3 ? m->n_arm + c : -1;
  return LANE_FREE + e;
}
__device__ __forceinline__ int lane_dof(const DevModel* m, int l) {       /* inverse of lane_pos; -1 = no dof at this lane */
  if (l < 16) {
    if (l < m->n_arm) return l;
    const int c = l - m->n_arm;
    return (m->free_row0 != 0 && c < 3) ? m->n_arm + 6 * (__ffs(m->free_row0) - 1) + c : -1;
  }
  if (l >= 32) return -1;
  if (l < LANE_FREE) return l - LANE_J1 < m->n_j1 ? m->n_arm + 6 * m->n_free + (l - LANE_J1) : -1;
  const int e = l - LANE_FREE, f = e / 6;
  return (e < 6 * m->n_free && !((m->free_row0 >> f) & 1)) ? m->n_arm + e : -1;
}
#else
/* RP_WIDE (two blocks + drawer, Panda): DPP row 0 = arm dofs 0..8 and the drawer's six components at lanes 9..14 (the arm's row
 * has room for them, the other row does not); DPP row 1 = scene joints 16..18, the two blocks at 19..30.  Contacts are classed
 * by the rows they touch (collide()), so a drawer-against-stop contact rides in row 0 next to a block-against-table contact in
 * row 1, and drawer-against-block folds like arm-against-block. */
#define LANE_J1 16
#define LANE_FREE 19
__device__ __forceinline__ int lane_pos(const DevModel* m, int d) {
  if (d < m->n_arm) return d;
  const int e = d - m->n_arm, f6 = 6 * m->n_free;
  if (e >= f6) return LANE_J1 + (e - f6);
  const int f = e / 6, c = e - 6 * f;
  return f == m->drawer_free ? m->n_arm + c : LANE_FREE + 6 * f + c;        /* the blocks are free bodies 0 and 1 */
}
__device__ __forceinline__ int lane_dof(const DevModel* m, int l) {
  if (l < m->n_arm) return l;
  if (l < 16) return l - m->n_arm < 6 ? m->n_arm + 6 * m->drawer_free + (l - m->n_arm) : -1;
  if (l >= 32) return -1;
  if (l < LANE_FREE) return l - LANE_J1 < m->n_j1 ? m->n_arm + 6 * m->n_free + (l - LANE_J1) : -1;
  return l - LANE_FREE < 12 ? m->n_arm + (l - LANE_FREE) : -1;
}
#endif
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float safe_inv(float d) { return d > 1e-9f ? 1.0f / d : 0.0f; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float lane_read(float v, int src_lane_uniform) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane_uniform));
}

__device__ __forceinline__ double readlane_d(double v, int src_lane_uniform) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane_uniform), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane_uniform);
  return __hiloint2double(hi, lo);
}

/* value of lane K (0..15) of each 16-lane DPP row, delivered to the whole row: one v_mov_b32_dpp row_newbcast.
 * The DPP control is an immediate, so labels are template constants (static_for below, not #pragma unroll). */
template <int K>
__device__ __forceinline__ float bcast16(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + (K & 15), 0xF, 0xF, true));
}
template <int T> struct IdxC { static constexpr int v = T; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(IdxC<I>{}); static_for<I + 1, N>(f); }
}

/* value of lane (l16 - D) of the same 16-lane DPP row; lanes with l16 < D keep their own value */
template <int D>
__device__ __forceinline__ float shr16(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x110 + D, 0xF, 0xF, false));
}

/* sum over lanes 0..31 (lanes 32..63 must hold 0); result in every lane.  DPP butterflies inside rows of 16. */
__device__ __forceinline__ float wave_sum32(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));            /* quad_perm [1,0,3,2] */
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));            /* quad_perm [2,3,0,1] */
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));           /* row_half_mirror */
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));           /* row_mirror */
  return lane_read(v, 0) + lane_read(v, 16);
}

/* counter RNG shared with the oracle (oracle/rp_oracle.c rpo_rng_uniform) */
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint32_t env, uint32_t counter) {
  uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)env << 32) | counter));
  return (float)(h >> 40) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------ kinematics */
struct Xf { M3 R; V3 p; };

__device__ __forceinline__ Xf joint_compose(const DevModel* m, const Xf& P, int j, float q) {
  Xf r;
  M3 Rj = mul(P.R, ldm3(m->arm_jrot[j]));
  r.p = P.p + mulv(P.R, ld3(m->arm_jpos[j]));
  V3 ax = ld3(m->arm_axis[j]);
  if (m->arm_jtype[j] == 0) {
    r.R = mul(Rj, axis_angle(ax, q));
  } else {
    r.R = Rj;
    r.p = r.p + mulv(Rj, ax) * q;
  }
  return r;
}

/* World transform of every body from the state record.  Arm links by pointer jumping over the kinematic tree: lane j
 * starts with link j's transform relative to its parent link (joint frame times joint motion; root links composed with
 * the base), then every round composes it with its current ancestor's transform and jumps to that ancestor's ancestor -
 * ceil(log2(depth)) rounds of one 3x4 composition instead of a walk of up to 12 joints per link. */
template <class LDS>
__device__ __forceinline__ void fk_bodies(const DevModel* m, LDS& L, int lane) {
  float* T = L.cand;                     /* two buffers of 12 links x (R row-major, p), then 2 x 12 ancestor indices; dead space until collide() */
  static_assert(2 * 12 * RP_MAX_ARM + 2 * RP_MAX_ARM <= CANDMAX * 8 && 2 * 12 * RP_MAX_ARM + 2 * RP_MAX_ARM <= (int)(sizeof(L.cand) / sizeof(float)), "fk scratch fits the candidate list");
  int* P = (int*)(T + 2 * 12 * RP_MAX_ARM);
  const int n = m->n_arm;
  Xf x; x.R = ident3(); x.p = mk3(0, 0, 0);
  int par = -1;
  if (lane < n) {
    M3 R0 = ldm3(m->arm_jrot[lane]);
    V3 p0 = ld3(m->arm_jpos[lane]), ax = ld3(m->arm_axis[lane]);
    float q = L.st[ST_Q + lane];
    x.R = R0; x.p = p0;
    if (m->arm_jtype[lane] == 0) x.R = mul(R0, axis_angle(ax, q));
    else x.p = p0 + mulv(R0, ax) * q;
    par = m->arm_parent[lane];
    if (par < 0) { M3 Rb = ldm3(m->base_rot); V3 pb = ld3(m->base_pos); x.p = pb + mulv(Rb, x.p); x.R = mul(Rb, x.R); }
  }
  for (int it = 0; __ballot(par >= 0) != 0ull; it++) {      /* wave-uniform trip count */
    float* Tb = T + (it & 1) * 12 * RP_MAX_ARM;
    int* Pb = P + (it & 1) * RP_MAX_ARM;
    if (lane < n) { stm3(&Tb[12 * lane], x.R); st3(&Tb[12 * lane + 9], x.p); Pb[lane] = par; }
    WSYNC();
    if (par >= 0) {
      M3 Ra = ldm3(&Tb[12 * par]); V3 pa = ld3(&Tb[12 * par + 9]);
      x.p = pa + mulv(Ra, x.p); x.R = mul(Ra, x.R);
      par = Pb[par];
    }
  }
  /* link j (lane j) is body j + 1; the other bodies (world, free bodies, scene joints) are stored by the lane of their index */
  if (lane < n) { stm3(&L.xR[9 * (lane + 1)], x.R); st3(&L.xp[3 * (lane + 1)], x.p); }
  if (lane < m->nbody && !(lane >= 1 && lane <= n)) {
    int b = lane;
    Xf y; y.R = ident3(); y.p = mk3(0, 0, 0);
    if (b > n && b <= n + m->n_free) {
      int k = b - 1 - n;
      const float* f = &L.st[ST_FREE + 13 * k];
      Q4 q = {f[3], f[4], f[5], f[6]};
      y.R = quat_to_m3(q); y.p = ld3(f);
    } else if (b > n + m->n_free) {
      int k = b - 1 - n - m->n_free;
      M3 R0 = ldm3(m->j1_rot[k]);
      V3 ax = ld3(m->j1_axis[k]);
      y.p = ld3(m->j1_pos[k]);
      float q = L.st[ST_JQ + k];
      if (m->j1_type[k] == 0) y.R = mul(R0, axis_angle(ax, q));
      else { y.R = R0; y.p = y.p + mulv(R0, ax) * q; }
    }
    stm3(&L.xR[9 * b], y.R); st3(&L.xp[3 * b], y.p);
  }
}

/* joint motion subspaces about the reference point O (the EE body's origin), world axes */
template <class LDS>
__device__ __forceinline__ void joint_subspaces(const DevModel* m, LDS& L, int lane) {
  if (lane == 0) st3(L.O, ld3(&L.xp[3 * m->site_body[RP_SITE_EE]]));
  WSYNC();
  if (lane < m->n_arm) {
    V3 O = ld3(L.O);
    M3 R = ldm3(&L.xR[9 * (1 + lane)]);
    V3 a = mulv(R, ld3(m->arm_axis[lane]));
    V6 S;
    if (m->arm_jtype[lane] == 0) { S.a = a; S.l = cross(ld3(&L.xp[3 * (1 + lane)]) - O, a); }
    else { S.a = mk3(0, 0, 0); S.l = a; }
    st6(&L.S[6 * lane], S);
  }
}

template <class LDS>
__device__ __forceinline__ Xf collider_xf(const DevModel* m, const LDS& L, int c) {
  int b = m->col_body[c];
  M3 Rb = ldm3(&L.xR[9 * b]);
  Xf x;
  x.R = mul(Rb, ldm3(m->col_rot[c]));
  x.p = ld3(&L.xp[3 * b]) + mulv(Rb, ld3(m->col_pos[c]));
  return x;
}

template <class LDS>
__device__ __forceinline__ void collider_aabbs(const DevModel* m, LDS& L, int lane) {
  if (lane < m->n_col) {
    Xf x = collider_xf(m, L, lane);
    V3 he = ld3(m->col_he[lane]);
    float e[3];
    for (int i = 0; i < 3; i++)
      e[i] = m->col_type[lane] == 0 ? fabsf(x.R.m[3 * i]) * he.x + fabsf(x.R.m[3 * i + 1]) * he.y + fabsf(x.R.m[3 * i + 2]) * he.z : he.x;
    *(float4*)&L.aabb[8 * lane] = make_float4(x.p.x - e[0], x.p.y - e[1], x.p.z - e[2], m->col_margin[lane]);
    *(float4*)&L.aabb[8 * lane + 4] = make_float4(x.p.x + e[0], x.p.y + e[1], x.p.z + e[2], 0.f);
  }
}

/* ------------------------------------------------------------------ narrowphase (same decisions as oracle box_box) */
struct CPt { V3 p, n; float dist; };

/* sphere against box (same decisions and arithmetic as the oracle's sphere_box).  Inlined and written without dynamically indexed
 * arrays: as a called function it took the box's rotation by reference, and the caller spilled both pairs' matrices to scratch on
 * every narrowphase pass (6 KB per env and launch of write traffic for a path that almost never runs). */
__device__ __forceinline__ int sphere_box(V3 cs, float r, V3 cb, const M3& Rb, V3 hb, float margin, int sphere_is_b, CPt* out) {
  const V3 l = tmulv(Rb, cs - cb);
  V3 cl = l;
  bool inside = true;
  if (cl.x > hb.x) { cl.x = hb.x; inside = false; }
  if (cl.x < -hb.x) { cl.x = -hb.x; inside = false; }
  if (cl.y > hb.y) { cl.y = hb.y; inside = false; }
  if (cl.y < -hb.y) { cl.y = -hb.y; inside = false; }
  if (cl.z > hb.z) { cl.z = hb.z; inside = false; }
  if (cl.z < -hb.z) { cl.z = -hb.z; inside = false; }
  V3 nl; float dist;
  if (!inside) {
    const V3 df = cl - l;
    const float len = norm(df);
    dist = len - r;
    if (dist > margin) return 0;
    nl = df * (1.f / len);
  } else {
    const float px = hb.x - fabsf(l.x), py = hb.y - fabsf(l.y), pz = hb.z - fabsf(l.z);
    int k0 = 0; float best = 1e30f;          /* first strict minimum, in axis order (the oracle's loop) */
    if (px < best) { best = px; k0 = 0; }
    if (py < best) { best = py; k0 = 1; }
    if (pz < best) { best = pz; k0 = 2; }
    const float lk = k0 == 0 ? l.x : (k0 == 1 ? l.y : l.z), hk = k0 == 0 ? hb.x : (k0 == 1 ? hb.y : hb.z);
    const float nk = lk > 0.f ? -1.f : 1.f, ck = lk > 0.f ? hk : -hk;
    nl = mk3(k0 == 0 ? nk : 0.f, k0 == 1 ? nk : 0.f, k0 == 2 ? nk : 0.f);
    cl = mk3(k0 == 0 ? ck : cl.x, k0 == 1 ? ck : cl.y, k0 == 2 ? ck : cl.z);
    dist = -best - r;
  }
  const V3 nw = mulv(Rb, nl), pw = mulv(Rb, cl) + cb;
  out[0].p = pw - nw * (0.5f * dist);
  out[0].n = sphere_is_b ? nw : -nw;
  out[0].dist = dist;
  return 1;
}

/* ------------------------------------------------------------------ cooperative narrowphase: 8 lanes per active pair.
 * Same decisions and the same arithmetic per value as the oracle's sequential box_box (oracle/rp_oracle.c): SAT over
 * 15 axes with first-wins tie tolerance, edge-edge closest points or reference-face clipping (Sutherland-Hodgman, a
 * polygon never exceeds 8 vertices), at most 4 points kept around the deepest.  Here the 15 axis tests run one per
 * lane, the clip handles one polygon edge per lane (ballot + popcount give every output vertex its place, in the
 * sequential order), and all intermediate data sits in registers or in a 96-float LDS scratch per pair - no private
 * (scratch) memory. */
#define NPG 8
#define NPG_SCRATCH 96         /* sv[16] | poly[2][8][3] | kept[8][4] */
__device__ __forceinline__ V3 pick3(int i, V3 a, V3 b, V3 c) { return i == 0 ? a : (i == 1 ? b : c); }
__device__ __forceinline__ float pick1(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

/* coordinate of hull vertex v along a box axis (u = the axis in the hull's body frame, c = the box centre along it): ONE instruction sequence wherever it is
 * needed, so that the second scan finds the extreme of the first by equality */
__device__ __forceinline__ float hull_coord(V3 u, float4 v, float c) { return __fmaf_rn(u.z, v.z, __fmaf_rn(u.y, v.y, u.x * v.x)) - c; }

/* ---- GJK distance between an arm link's hull and a box (oracle hull_box_gjk, RPO_RULE_GJK): the simplex lives in the pair's narrowphase scratch
 * (point k: w, a, b = 9 floats at S + 9 k), every lane runs the same closest-point arithmetic on it (LDS broadcasts; all lanes store the same values), the
 * support queries are the whole-wave vertex scans of the hull contact above. */
#define GJK_DUP 1e-24
#define GJK_REL 1e-12
#define GJK_ZERO 1e-20
#define GJK_STALL (1.0 - 1e-14)
/* Support-vertex candidate tables (generated/rp_hullcells_gen.h, tools/bake_hull_cells.py): the cube-map cell of a direction holds every vertex of the link's hull that can
 * have the largest coordinate along some direction of the cell - a superset wide enough for fp32 rounding, so that scanning the cell returns the very vertex a scan of the
 * whole hull returns (the largest computed coordinate, the lowest vertex number among equals; tests/test_hull_cells.py: 10^6 directions per arm against the oracle's full
 * scan).  A cell has 3 - 8 candidates at the median where a link has 200 - 1000 vertices.  DevModel.hcv: the lists as (x, y, z, vertex number), hull after hull; DevModel.hco:
 * per hull RP_HCELL_N + 1 offsets; DevModel.hcell_first[collider]: the hull's place in hco.  (The oracle keeps scanning whole hulls: the tables are the library's alone.) */
__device__ __forceinline__ int hcell_of(V3 d) {
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  int m = 0; float dm = d.x, dp = d.y, dq = d.z, am = ax;
  if (ay > am) { m = 1; dm = d.y; dp = d.z; dq = d.x; am = ay; }
  if (az > am) { m = 2; dm = d.z; dp = d.x; dq = d.y; am = az; }
  const float inv = __builtin_amdgcn_rcpf(fmaxf(am, 1e-30f));      /* (1 ulp: the cells are baked 1e-4 wider than they are; a zero vector - an unused probe - lands in some cell) */
  int i = (int)floorf((dp * inv + 1.f) * (0.5f * RP_HCELL_G)), j = (int)floorf((dq * inv + 1.f) * (0.5f * RP_HCELL_G));
  i = min(max(i, 0), RP_HCELL_G - 1); j = min(max(j, 0), RP_HCELL_G - 1);
  return ((2 * m + (dm < 0.f ? 1 : 0)) * RP_HCELL_G + i) * RP_HCELL_G + j;
}
/* wave-wide reductions through DPP (no LDS round trip: six butterflies through ds_bpermute cost ~900 cycles of latency per scan): every lane ends with the result */
__device__ __forceinline__ float wave_max_f(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
  return fmaxf(fmaxf(lane_read(v, 0), lane_read(v, 16)), fmaxf(lane_read(v, 32), lane_read(v, 48)));
}
__device__ __forceinline__ float wave_min_f(float v) { return -wave_max_f(-v); }
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
/* ---- the simplex arithmetic of the GJK below, in DOUBLE (as in every build of the oracle): the sub-case determinants of a sliver simplex - three vertices of a finely
 * tessellated link, a few millimetres from the origin - cancel to 1e-3 relative in fp32, and the witness points of nearly parallel features then move by centimetres
 * (measured on the CPU: fp32 against fp64 oracle, 15 of 360 poses with another contact; with this arithmetic in double: 1). */
struct D3 { double x, y, z; };
__device__ __forceinline__ D3 mkd(double x, double y, double z) { D3 r = {x, y, z}; return r; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return mkd(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 dcross(D3 a, D3 b) { return mkd(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ D3 dsel(bool c, D3 a, D3 b) { return mkd(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
/* 1 / x: v_rcp_f64 and two Newton steps (full double precision; an IEEE division is a twenty-instruction sequence, four of them per triangle) */
__device__ __forceinline__ double drcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
/* closest point of the triangle a b c to the origin (Ericson's sub-cases in the oracle's order, gjk_closest), without branches: which vertices support it
 * (bit k = vertex k stays) and their weights */
struct GjkTri { int keep; double l0, l1, l2; };
__device__ __forceinline__ GjkTri gjk_tri(D3 a, D3 b, D3 c) {
  const D3 ab = b - a, ac = c - a;
  const double d1 = -ddot(ab, a), d2 = -ddot(ac, a), d3 = -ddot(ab, b), d4 = -ddot(ac, b), d5 = -ddot(ab, c), d6 = -ddot(ac, c);
  const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  const double tab = d1 * drcp(d1 - d3), tac = d2 * drcp(d2 - d6), tbc = (d4 - d3) * drcp((d4 - d3) + (d5 - d6)), den = drcp(va + vb + vc);
  GjkTri r;
  r.keep = 7; r.l1 = vb * den; r.l2 = vc * den; r.l0 = 1.0 - r.l1 - r.l2;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) { r.keep = 6; r.l0 = 0.0; r.l1 = 1.0 - tbc; r.l2 = tbc; }
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { r.keep = 5; r.l0 = 1.0 - tac; r.l1 = 0.0; r.l2 = tac; }
  if (d6 >= 0.0 && d5 <= d6) { r.keep = 4; r.l0 = 0.0; r.l1 = 0.0; r.l2 = 1.0; }
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { r.keep = 3; r.l0 = 1.0 - tab; r.l1 = tab; r.l2 = 0.0; }
  if (d3 >= 0.0 && d4 <= d3) { r.keep = 2; r.l0 = 0.0; r.l1 = 1.0; r.l2 = 0.0; }
  if (d1 <= 0.0 && d2 <= 0.0) { r.keep = 1; r.l0 = 1.0; r.l1 = 0.0; r.l2 = 0.0; }      /* (the first test of the sequence wins: applied last) */
  return r;
}
__device__ __forceinline__ GjkTri gjk_seg(D3 a, D3 b) {
  const D3 ab = b - a;
  const double t = -ddot(a, ab), den = ddot(ab, ab);
  GjkTri r; r.l2 = 0.0;
  r.keep = 3; r.l1 = t * drcp(den); r.l0 = 1.0 - r.l1;
  if (t >= den) { r.keep = 2; r.l0 = 0.0; r.l1 = 1.0; }
  if (t <= 0.0 || den <= 0.0) { r.keep = 1; r.l0 = 1.0; r.l1 = 0.0; }
  return r;
}
__device__ __forceinline__ V3 sel3(bool c, V3 a, V3 b) { return mk3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
/* The simplex of the GJK below: Minkowski-difference points w = (hull vertex) - (box-core corner) with the corner kept as the witness, all in the BOX's frame.  The
 * points live in LDS (W: four points of three doubles in the pair's narrowphase scratch; every lane holds the same values and stores the same values), the corners
 * (one sign bit per axis: bit k set = +hbc_k), the count and the weights in registers.  reduce(): keeps the vertices of `keep` (a subsequence of p0 p1 p2, in that
 * order) with their weights */
struct GjkSimplex {
  double* W; int b0, b1, b2, b3; int i0, i1, i2, i3; int n;      /* W: four points of three doubles, then the three weights (W[12 .. 15)); b: the points' box-core corners; i: their hull vertices (what the contact cache keeps of a call) */
  __device__ __forceinline__ D3 pt(int i) const { return mkd(W[3 * i], W[3 * i + 1], W[3 * i + 2]); }
  __device__ __forceinline__ void put(int i, D3 p) { W[3 * i] = p.x; W[3 * i + 1] = p.y; W[3 * i + 2] = p.z; }
  __device__ __forceinline__ void weights(double a, double b, double c) { W[12] = a; W[13] = b; W[14] = c; }
  /* (scalars by value: handed a GjkTri by reference, the compiler keeps it in private memory and turns the selects below into indexed loads) */
  __device__ __forceinline__ void reduce(D3 p0, D3 p1, D3 p2, int q0, int q1, int q2, int j0, int j1, int j2, int keep, double r0, double r1, double r2) {
    const bool k0 = keep & 1, k1 = keep & 2;
    const double t12 = k1 ? r1 : r2;
    put(0, dsel(k0, p0, dsel(k1, p1, p2))); b0 = k0 ? q0 : (k1 ? q1 : q2); i0 = k0 ? j0 : (k1 ? j1 : j2);
    put(1, dsel(k0 && k1, p1, p2)); b1 = (k0 && k1) ? q1 : q2; i1 = (k0 && k1) ? j1 : j2;
    put(2, p2); b2 = q2; i2 = j2;
    weights(k0 ? r0 : t12, (k0 && k1) ? r1 : r2, r2);
    n = __popc(keep);
  }
  __device__ __forceinline__ D3 closest() const {
    D3 p = pt(0);
    D3 q = mkd(p.x * W[12], p.y * W[12], p.z * W[12]);
    if (n > 1) { p = pt(1); q = mkd(q.x + p.x * W[13], q.y + p.y * W[13], q.z + p.z * W[13]); }
    if (n > 2) { p = pt(2); q = mkd(q.x + p.x * W[14], q.y + p.y * W[14], q.z + p.z * W[14]); }
    return q;
  }
  static __device__ __forceinline__ V3 corner(int code, V3 h) { return mk3((code & 1) ? h.x : -h.x, (code & 2) ? h.y : -h.y, (code & 4) ? h.z : -h.z); }
  __device__ __forceinline__ D3 witness(V3 h) const {
    V3 c = corner(b0, h);
    D3 q = mkd(c.x * W[12], c.y * W[12], c.z * W[12]);
    if (n > 1) { c = corner(b1, h); q = mkd(q.x + c.x * W[13], q.y + c.y * W[13], q.z + c.z * W[13]); }
    if (n > 2) { c = corner(b2, h); q = mkd(q.x + c.x * W[14], q.y + c.y * W[14], q.z + c.z * W[14]); }
    return q;
  }
};
/* closest point of the simplex to the origin; the simplex shrinks to the supporting sub-simplex.  n = 4 afterwards: the origin lies inside the tetrahedron */
__device__ __forceinline__ void gjk_closest(GjkSimplex& S, int lane) {
  (void)lane;
  if (S.n == 1) { S.weights(1.0, 0.0, 0.0); WSYNC(); return; }
  if (S.n == 2) { const D3 a = S.pt(0), b = S.pt(1); const GjkTri r = gjk_seg(a, b); WSYNC(); S.reduce(a, b, b, S.b0, S.b1, S.b1, S.i0, S.i1, S.i1, r.keep, r.l0, r.l1, r.l2); WSYNC(); return; }
  if (S.n == 3) { const D3 a = S.pt(0), b = S.pt(1), c = S.pt(2); const GjkTri r = gjk_tri(a, b, c); WSYNC(); S.reduce(a, b, c, S.b0, S.b1, S.b2, S.i0, S.i1, S.i2, r.keep, r.l0, r.l1, r.l2); WSYNC(); return; }
  /* tetrahedron: the closest of the faces the origin lies outside of (the oracle's F / OPP tables).  The four faces side by side in the four lanes of a quad - every lane
   * of the row runs this code anyway, and one after the other the faces were four fifths of a deep GJK round's ~14 k cycles (an arm pressed into the furniture under the
   * literal random-action rollout: a dozen such rounds per substep) -, then the quad's minimum, the lowest face among equals as the oracle's `dd < best` keeps it */
  const int f = lane & 3;
  double dd;
  GjkTri r;
  {
    const int i0 = f == 3 ? 1 : 0, i1 = f == 0 ? 1 : (f == 1 ? 2 : 3), i2 = f == 0 ? 2 : (f == 1 ? 3 : (f == 2 ? 1 : 2)), io = f == 0 ? 3 : (f == 1 ? 1 : (f == 2 ? 2 : 0));
    const D3 a = S.pt(i0), b = S.pt(i1), c = S.pt(i2);
    double so, sd;
    { const D3 d = S.pt(io), nrm = dcross(b - a, c - a); so = -ddot(a, nrm); sd = ddot(d - a, nrm); }
    const bool skip = so * sd > 0.0 || (sd == 0.0 && so == 0.0);
    r = gjk_tri(a, b, c);
    const D3 q = mkd(a.x * r.l0 + b.x * r.l1 + c.x * r.l2, a.y * r.l0 + b.y * r.l1 + c.y * r.l2, a.z * r.l0 + b.z * r.l1 + c.z * r.l2);
    dd = ddot(q, q);
    if (skip || !(dd < 1e30)) dd = 1e30;
  }
  double mn = fmin(dd, __shfl_xor(dd, 1));
  mn = fmin(mn, __shfl_xor(mn, 2));
  int bf = (dd == mn && dd < 1e30) ? f : 4;
  bf = min(bf, __shfl_xor(bf, 1));
  bf = min(bf, __shfl_xor(bf, 2));
  if (bf > 3) return;                                        /* the origin lies inside all four: the cores overlap (S.n stays 4) */
  const int src = (lane & ~3) | bf;                          /* the lane of this quad that solved the winning face */
  const int keep = __shfl(r.keep, src);
  const double l0 = __shfl(r.l0, src), l1 = __shfl(r.l1, src), l2 = __shfl(r.l2, src);
  const int i0 = bf == 3 ? 1 : 0, i1 = bf == 0 ? 1 : (bf == 1 ? 2 : 3), i2 = bf == 0 ? 2 : (bf == 1 ? 3 : (bf == 2 ? 1 : 2));
  const D3 a = S.pt(i0), b = S.pt(i1), c = S.pt(i2);
  auto code = [&](int i) { return i == 0 ? S.b0 : (i == 1 ? S.b1 : (i == 2 ? S.b2 : S.b3)); };
  auto vert = [&](int i) { return i == 0 ? S.i0 : (i == 1 ? S.i1 : (i == 2 ? S.i2 : S.i3)); };
  const int qa = code(i0), qb = code(i1), qc = code(i2), ja = vert(i0), jb = vert(i1), jc = vert(i2);
  WSYNC();
  S.reduce(a, b, c, qa, qb, qc, ja, jb, jc, keep, l0, l1, l2);
  WSYNC();
}

template <int K>
__device__ __forceinline__ int bcast16i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + (K & 15), 0xF, 0xF, true); }
__device__ __forceinline__ float row_max_f(float v) {      /* max over the 16 lanes of a DPP row, in every lane */
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ int row_min_i(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ int row_max_i(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ unsigned row_or_u(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
  return v;
}
#define RP_DPP_D(v, ctrl) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ double row_min_d(double v) {      /* (no NaNs here) */
  v = fmin(v, RP_DPP_D(v, 0xB1));
  v = fmin(v, RP_DPP_D(v, 0x4E));
  v = fmin(v, RP_DPP_D(v, 0x141));
  v = fmin(v, RP_DPP_D(v, 0x140));
  return v;
}
/* ---- penetration of a hull core into a box core: the expanding polytope (oracle hull_box_epa, RPO_RULE_EPA; btGjkEpaSolver2 restated for two polytopes).  GJK's distance
 * phase ended with a tetrahedron around the origin - the cores overlap; until round 5 that pair went to the OBB path.  The Minkowski difference of two polytopes is one, so
 * the loop ends exactly: per round the face nearest the origin (lowest number among equals), the support point of hull - box along its normal (the same cell scan as
 * GJK's), done if that is not beyond the face (1e-9) - else the faces that see it go and the horizon's edges get faces to it (dead slots first, rising).
 * The sixteen lanes of the row share the polytope: its vertices (difference points in double, box-core corner codes) lie in the row's scratch; FACE f LIVES IN LANE f & 15
 * (faces 16 - 19: a second set of registers in lanes 0 - 3) as its vertex word, unit normal and distance, computed ONCE when the face is made - a face never changes, and
 * the oracle's re-evaluations return the same bits (its cross and dot products are unfused, so the turned normal of a swapped word is the exact negative).  Per round: a
 * row minimum over the lanes' distances, the support scan, every lane's visibility test, the horizon's edge list (entry k in lane k & 15, the oracle's order of
 * insertions and swap-removals kept: it fixes which slot a new face gets, and the lowest-number rule reads slots) and the new faces' planes in parallel.  The first
 * version (every lane recomputing every face's plane from scratch twice a round, the edge list in LDS) cost 100 - 300 k cycles a call: the literal random-action
 * rollout, whose arms lie IN the furniture, lost 40 %.  Caps shared with the oracle. */
#define EPA_MAXV 12
#define EPA_MAXF 20
#define EPA_MAXE 24
#define EPA_ITERS 8
#define HROW_FLOATS 192      /* a row's scratch in L.npscr: [0, 64) GJK (hull_item16), [64, 136) the polytope's vertices, [136, 148) their corner codes */
struct EpaFace { D3 n; double d; int word; bool ok; };
__device__ __forceinline__ EpaFace epa_face(const double* Vw, int f) {      /* unit normal away from the origin, distance, the word with its last two vertices swapped if the normal had to turn */
  const int ia = f & 31, ib = (f >> 5) & 31, ic = (f >> 10) & 31;
  const D3 a = mkd(Vw[3 * ia], Vw[3 * ia + 1], Vw[3 * ia + 2]), b = mkd(Vw[3 * ib], Vw[3 * ib + 1], Vw[3 * ib + 2]), c = mkd(Vw[3 * ic], Vw[3 * ic + 1], Vw[3 * ic + 2]);
  D3 n = dcross(b - a, c - a);
  const double l2 = ddot(n, n);
  EpaFace r; r.ok = l2 > 1e-36; r.word = f;
  const double inv = drcp(sqrt(r.ok ? l2 : 1.0));
  n = mkd(n.x * inv, n.y * inv, n.z * inv);
  double d = ddot(n, a);
  if (d < 0.0) { r.word = ia | (ic << 5) | (ib << 10); n = mkd(-n.x, -n.y, -n.z); d = -d; }
  r.n = n; r.d = d;
  return r;
}
template <class LDS>
__device__ __forceinline__ bool hull_epa16(LDS& L, const int lane, const GjkSimplex& S, const float4* __restrict__ cv, const int* __restrict__ co, D3& nrm, double& depth, D3& witb) {      /* (the box axes in the hull's frame, the centre and the core's half extents wait in the row's GJK scratch: fetched around each support query) */
  const int l16 = lane & 15, row = lane >> 4, rbase = lane & 48;
  double* Vw = (double*)&L.npscr[HROW_FLOATS * row + 64];
  int* Vm = (int*)&L.npscr[HROW_FLOATS * row + 136];
  const float* Y = &L.npscr[HROW_FLOATS * row + 2 * 18];
  static_assert(64 + 6 * EPA_MAXV == 136 && 136 + EPA_MAXV <= HROW_FLOATS && EPA_MAXF <= 32 && EPA_MAXE <= 32, "the row's scratch; two faces and two edges a lane");
  for (int i = 0; i < 4; i++) { const D3 p = S.pt(i); Vw[3 * i] = p.x; Vw[3 * i + 1] = p.y; Vw[3 * i + 2] = p.z; }
  Vm[0] = S.b0; Vm[1] = S.b1; Vm[2] = S.b2; Vm[3] = S.b3;
  WSYNC();
  int fw[2] = {0, 0}; D3 fn[2] = {mkd(0, 0, 0), mkd(0, 0, 0)}; double fd[2] = {1e300, 1e300}; int E[2] = {0, 0};
  int nv = 4, nf = 4; unsigned alive = 0xFu;
  {
    const int w4 = l16 == 1 ? (0 | (2 << 5) | (3 << 10)) : l16 == 2 ? (0 | (3 << 5) | (1 << 10)) : l16 == 3 ? (1 | (3 << 5) | (2 << 10)) : (0 | (1 << 5) | (2 << 10));
    const EpaFace r = epa_face(Vw, w4);
    fw[0] = r.word; fn[0] = r.n; fd[0] = r.d;
    if (row_or_u(l16 < 4 && !r.ok ? 1u : 0u) != 0u) return false;
  }
#pragma unroll 1
  for (int it = 0; it < EPA_ITERS; it++) {
    /* the face nearest the origin */
    const bool a0 = (alive >> l16) & 1u, a1 = (alive >> (l16 + 16)) & 1u;
    const double c0 = a0 ? fd[0] : 1e300, c1 = a1 ? fd[1] : 1e300;
    const double bd = row_min_d(fmin(c0, c1));
    const int bf = row_min_i(a0 && c0 == bd ? l16 : (a1 && c1 == bd ? l16 + 16 : 99));
    if (bf >= 99) return false;
    const int bsrc = rbase | (bf & 15);
    const bool bhi = bf >= 16;
    const D3 bn = mkd(__shfl(bhi ? fn[1].x : fn[0].x, bsrc), __shfl(bhi ? fn[1].y : fn[0].y, bsrc), __shfl(bhi ? fn[1].z : fn[0].z, bsrc));
    /* the point of hull - box core farthest along bn: the hull's vertex by the scan of the direction's cell (lowest number among equals), the corner by signs */
    const V3 bnf = mk3((float)bn.x, (float)bn.y, (float)bn.z);
    V3 wa; int wb; V3 hbc;
    {
      const V3 u0 = ld3(Y), u1 = ld3(Y + 3), u2 = ld3(Y + 6);
      const float c0 = Y[9], c1 = Y[10], c2 = Y[11];
      hbc = ld3(Y + 12);
      const V3 dl = u0 * bnf.x + u1 * bnf.y + u2 * bnf.z;
      float sd = -1e30f; int bi = 0x7fffffff; V3 bq = mk3(0, 0, 0);
      const int cell = hcell_of(dl);
      const int o0 = co[cell], o1 = co[cell + 1];
      for (int base = o0 + l16; __any(base - l16 < o1); base += 16) {
        const float4 q = cv[base < o1 ? base : o0];
        const float dq = __fmaf_rn(dl.z, q.z, __fmaf_rn(dl.y, q.y, dl.x * q.x));
        if (base < o1 && dq > sd) { sd = dq; bi = __float_as_int(q.w); bq = mk3(q.x, q.y, q.z); }
      }
      const float top = row_max_f(sd);
      const int key = row_min_i(sd == top ? ((bi << 4) | l16) : 0x7fffffff);
      const int win = rbase | (key & 15);
      const V3 q = mk3(__shfl(bq.x, win), __shfl(bq.y, win), __shfl(bq.z, win));
      wa = mk3(hull_coord(u0, make_float4(q.x, q.y, q.z, 0.f), c0), hull_coord(u1, make_float4(q.x, q.y, q.z, 0.f), c1), hull_coord(u2, make_float4(q.x, q.y, q.z, 0.f), c2));
      wb = (-bnf.x >= 0.f ? 1 : 0) | (-bnf.y >= 0.f ? 2 : 0) | (-bnf.z >= 0.f ? 4 : 0);
    }
    const V3 cb = GjkSimplex::corner(wb, hbc);
    const D3 w = mkd((double)wa.x - (double)cb.x, (double)wa.y - (double)cb.y, (double)wa.z - (double)cb.z);
    const double ext = ddot(w, bn);
    bool done = ext - bd < 1e-9 || nv >= EPA_MAXV || it == EPA_ITERS - 1;
    int ne = 0; unsigned kill = 0u;
    if (!done) {
      /* every lane: does its face see w */
      {
        const int i0 = fw[0] & 31, i1 = fw[1] & 31;
        const bool v0 = a0 && ddot(fn[0], w - mkd(Vw[3 * i0], Vw[3 * i0 + 1], Vw[3 * i0 + 2])) > 1e-12;
        bool v1 = false;
        if ((alive >> 16) != 0u) v1 = a1 && ddot(fn[1], w - mkd(Vw[3 * i1], Vw[3 * i1 + 1], Vw[3 * i1 + 2])) > 1e-12;
        kill = row_or_u((v0 ? 1u << l16 : 0u) | (v1 ? 1u << (l16 + 16) : 0u));
      }
      /* the horizon: the edges of the faces that go, face after face (rising), an edge whose reverse is in the list cancels it (the list's last entry takes the place) */
      bool over = false;
      unsigned rest = kill;
#pragma unroll 1
      while (rest != 0u) {
        const int i = __ffs(rest) - 1; rest &= rest - 1u;
        const int f = __shfl(i >= 16 ? fw[1] : fw[0], rbase | (i & 15));
#pragma unroll 1
        for (int e2 = 0; e2 < 3; e2++) {
          const int x = (f >> (5 * e2)) & 31, y = (f >> (e2 == 2 ? 0 : 5 * e2 + 5)) & 31;
          const int rev = y | (x << 8);
          const int found = row_max_i(l16 + 16 < ne && E[1] == rev ? l16 + 16 : (l16 < ne && E[0] == rev ? l16 : -1));
          if (found >= 0) {
            const int last = ne - 1;
            const int lv = __shfl(last >= 16 ? E[1] : E[0], rbase | (last & 15));
            if (found == l16) E[0] = lv;
            if (found == l16 + 16) E[1] = lv;
            ne--;
          } else if (ne >= EPA_MAXE) over = true;
          else {
            if (ne == l16) E[0] = x | (y << 8);
            if (ne == l16 + 16) E[1] = x | (y << 8);
            ne++;
          }
        }
      }
      if (over || ne == 0) return false;
      if (__popc(alive) - __popc(kill) + ne > EPA_MAXF) done = true;      /* no room for the new faces: the nearest face as it is */
    }
    if (done) {
      /* the origin's projection on the nearest face in barycentric coordinates: the witness on the box core */
      const int f = __shfl(bhi ? fw[1] : fw[0], bsrc);
      const int ia = f & 31, ib = (f >> 5) & 31, ic = (f >> 10) & 31;
      const D3 a = mkd(Vw[3 * ia], Vw[3 * ia + 1], Vw[3 * ia + 2]), b = mkd(Vw[3 * ib], Vw[3 * ib + 1], Vw[3 * ib + 2]), c = mkd(Vw[3 * ic], Vw[3 * ic + 1], Vw[3 * ic + 2]);
      const D3 p = mkd(bn.x * bd, bn.y * bd, bn.z * bd);
      const D3 v0 = b - a, v1 = c - a, v2 = p - a;
      const double d00 = ddot(v0, v0), d01 = ddot(v0, v1), d11 = ddot(v1, v1), d20 = ddot(v2, v0), d21 = ddot(v2, v1);
      const double den = d00 * d11 - d01 * d01;
      const double iden = den != 0.0 ? drcp(den) : 0.0;
      const double bv = (d11 * d20 - d01 * d21) * iden, bw = (d00 * d21 - d01 * d20) * iden, bu = 1.0 - bv - bw;
      const V3 hb2 = ld3(Y + 12);
      const V3 ca = GjkSimplex::corner(Vm[ia], hb2), cbb = GjkSimplex::corner(Vm[ib], hb2), cc = GjkSimplex::corner(Vm[ic], hb2);
      witb = mkd(bu * ca.x + bv * cbb.x + bw * cc.x, bu * ca.y + bv * cbb.y + bw * cc.y, bu * ca.z + bv * cbb.z + bw * cc.z);
      nrm = bn; depth = bd;
      return true;
    }
    alive &= ~kill;
    Vw[3 * nv] = w.x; Vw[3 * nv + 1] = w.y; Vw[3 * nv + 2] = w.z; Vm[nv] = wb;
    WSYNC();
    /* edge k of the list makes a face with w in the k-th free slot (the dead ones rising, then new ones): each lane finds its slots' ranks and so their edges */
    {
      const unsigned dead = ~alive & ((1u << nf) - 1u);
      const int nd = __popc(dead);
      unsigned made = 0u; bool bad = false;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int j = l16 + 16 * s;
        const int rank = ((dead >> j) & 1u) ? __popc(dead & ((1u << j) - 1u)) : (j >= nf ? nd + (j - nf) : 1000);
        const bool mk = rank < ne;
        made |= mk ? 1u << j : 0u;
        if (s == 1 && max(nf, nf + ne - nd) <= 16) continue;      /* (row-uniform: there is no slot beyond the sixteenth) */
        const int e0 = __shfl(E[0], rbase | (rank & 15)), e1 = __shfl(E[1], rbase | (rank & 15));
        const int e = rank >= 16 ? e1 : e0;
        const EpaFace r = epa_face(Vw, mk ? ((e & 255) | ((e >> 8) << 5) | (nv << 10)) : (0 | (1 << 5) | (2 << 10)));
        if (mk) { fw[s] = r.word; fn[s] = r.n; fd[s] = r.d; bad |= !r.ok; }
      }
      alive |= row_or_u(made);
      nf = max(nf, nf + ne - nd);
      nv++;
      if (row_or_u(bad ? 1u : 0u) != 0u) return false;
    }
  }
  return false;
}

/* One hull pair by the SIXTEEN LANES OF A DPP ROW, the four rows of a wave on four pairs at once (round 5; until then a whole wave did one pair after the other: since the
 * support-vertex tables a query looks at a handful of vertices, and under the literal random-action rollout an arm lying on the furniture has ten such pairs per substep).
 * Everything below is the same in the sixteen lanes of a row except the candidates a lane scans; nothing in it is wave-uniform - no readfirstlane, no ballot, the cross-lane
 * steps are row DPP - and the rows diverge freely (a row whose pair needs no GJK idles while another iterates).  gi = the row's active pair of the substep, -1: none; the
 * pair is described in L.hinfo[gi] by narrowphase_coop's first phase (hull collider | box collider << 8 | hull is collider b << 16, margin, baked pair index); the outcome
 * goes to L.hout[gi] (1: hull contact, staged in L.hpt[gi]; 0: apart; -1: the OBB path).  The GJK simplex of a row lives in its 64 floats of the narrowphase scratch
 * (free until the batches start). */
template <class LDS>
__device__ __forceinline__ void hull_item16(const DevModel* m, LDS& L, const int lane, const int gi, float* gax) {
  const int l16 = lane & 15, row = lane >> 4;
  if (gi >= 0) {
        PCLK_ADD(29, -(long long)__builtin_readcyclecounter())      /* (profiling build, row 0's items: cycles in the set-up and the 15-axis test | 30: the face scan | 31: GJK) */
        const int hinf = L.hinfo[gi][0];
        const int ca = hinf & 255, cb = (hinf >> 8) & 255;
        const bool flip = ((hinf >> 16) & 1) != 0;                  /* the pair's normal points from b toward a: from the hull toward the box when the hull is b */
        const float mg = __int_as_float(L.hinfo[gi][1]);
        /* what this pair's last GJK call left in the contact cache (issued now, used after the scan) */
        const int pi_u = L.hinfo[gi][2];
        float* gslot = gax ? gax + 8 * (pi_u & (PMC_AXN - 1)) : nullptr;
        float4 gs1 = make_float4(0.f, 0.f, 0.f, 0.f); int gtag = 0;
        if (gslot && m->gjk) { gtag = __float_as_int(gslot[0]); gs1 = *(const float4*)(gslot + 4); }      /* (the direction for the scan; the simplex is fetched when GJK is reached) */
        const bool warm = gtag == pi_u + 1;
        const V3 hc0 = ld3(m->col_he[cb]);
        const V3 hcm = mk3(fmaxf(hc0.x, RP_HULL_MARGIN), fmaxf(hc0.y, RP_HULL_MARGIN), fmaxf(hc0.z, RP_HULL_MARGIN));      /* the box as the reference step's GJK sees it: core + margin, a box thinner than the margin comes out 0.001 thick (oracle hull_face) */
        /* the box in the hull's body frame: a vertex v has the box coordinate u_k . v - c_k.  First the link's OBB against the box over the fifteen directions of the box-box
         * SAT (one per lane; oracle obb_apart): clear by more than the margin along one of them = apart - the diagonal arrangements, a link passing a table edge, that
         * the six face directions let through.  And one more direction for the scan, up / cp: the direction this pair's last GJK call ended with (if the cache has
         * one): the scan measures the hull's clearance from the box core along it on the way, and beyond the margin and the two shape margins the pair is apart
         * without GJK (oracle hull_box_gjk, the cached direction's test) */
        V3 u0, u1, u2, up; float c0, c1, c2, cp; bool obb_apart;
        {
          const int body = m->col_body[ca];
          const M3 Rw = ldm3(&L.xR[9 * body]);
          const V3 pw = ld3(&L.xp[3 * body]);
          const Xf xc = collider_xf(m, L, cb);
          const V3 tc = xc.p - pw;
          const V3 b0 = col(xc.R, 0), b1 = col(xc.R, 1), b2 = col(xc.R, 2);
          u0 = tmulv(Rw, b0); u1 = tmulv(Rw, b1); u2 = tmulv(Rw, b2);
          c0 = dot(b0, tc); c1 = dot(b1, tc); c2 = dot(b2, tc);
          const Xf xh = collider_xf(m, L, ca);
          const V3 hh = ld3(m->col_he[ca]);
          const V3 a0 = col(xh.R, 0), a1 = col(xh.R, 1), a2 = col(xh.R, 2);
          const int t = l16;
          V3 ax; bool okax = t < 15;
          if (t < 3) ax = pick3(t, a0, a1, a2);
          else if (t < 6) ax = pick3(t - 3, b0, b1, b2);
          else {
            const int e = t - 6, i = e / 3, j = e - 3 * i;
            const V3 cr = cross(pick3(i, a0, a1, a2), pick3(j, b0, b1, b2));
            const float l = norm(cr);
            okax = okax && l > 1e-2f;
            ax = cr * (1.f / fmaxf(l, 1e-2f));
          }
          const float ra = hh.x * fabsf(dot(ax, a0)) + hh.y * fabsf(dot(ax, a1)) + hh.z * fabsf(dot(ax, a2));
          const float rb = hcm.x * fabsf(dot(ax, b0)) + hcm.y * fabsf(dot(ax, b1)) + hcm.z * fabsf(dot(ax, b2));
          const float tl = dot(xh.p - xc.p, ax);
          const float gap = okax ? fabsf(tl) - ra - rb : -1e30f;
          obb_apart = row_max_f(gap) > mg + RP_HULL_MARGIN + 1e-5f;
          up = mk3(0, 0, 0); cp = 0.f;
          if (warm) {
            const V3 vc = mk3(gs1.y, gs1.z, gs1.w);
            const float l = norm(vc);
            if (l > 0.f) {
              const V3 vn = vc * (1.f / l);                 /* (box frame) */
              const V3 hbc0 = mk3(hc0.x - fminf(RP_HULL_MARGIN, hc0.x), hc0.y - fminf(RP_HULL_MARGIN, hc0.y), hc0.z - fminf(RP_HULL_MARGIN, hc0.z));
              up = u0 * vn.x + u1 * vn.y + u2 * vn.z;
              cp = vn.x * c0 + vn.y * c1 + vn.z * c2 + hbc0.x * fabsf(vn.x) + hbc0.y * fabsf(vn.y) + hbc0.z * fabsf(vn.z);      /* the box core's far end along it */
            }
          }
        }
        int out = 0;                                         /* this pair's hf */
        V3 nloc = mk3(0, 0, 0), ploc = mk3(0, 0, 0); float dcon = 0.f;      /* the contact in the BOX's frame: normal (box toward hull), point on the box's surface, distance */
        HCLK_ADD(15, obb_apart ? (1ull << 48) : 0ull)
        PCLK_ADD(29, __builtin_readcyclecounter())
        if (!obb_apart) {
        PCLK_ADD(30, -(long long)__builtin_readcyclecounter())
        const int nn = m->hull_cnt[ca];
        const float4* tv = (const float4*)m->hullv + m->hull_off[ca];
        const float4* __restrict__ cv = (const float4*)m->hcv;
        const int* __restrict__ co = m->hco + m->hcell_first[ca];
        /* the hull's extent along the box's three axes - six support queries - and its clearance along the probe direction, a seventh: two lanes per query (lanes 2 q, 2 q + 1:
         * q = 0 / 1 the lowest / highest l_0, 2 / 3: l_1, 4 / 5: l_2, 6: the lowest coordinate along the probe), each pair of lanes scanning the cell of its direction, two
         * candidates a round (rising vertex numbers in every lane's sequence: a lane's first strict extreme is its lowest-numbered one), one DPP step joins the two lanes */
        float lo0, lo1, lo2, hi0, hi1, hi2, pmin;
        int il0, il1, il2, ih0, ih1, ih2;
        {
          const int q7 = l16 >> 1, h2 = l16 & 1;
          const V3 um = q7 < 2 ? u0 : (q7 < 4 ? u1 : (q7 < 6 ? u2 : up));
          const float cm = q7 < 2 ? c0 : (q7 < 4 ? c1 : (q7 < 6 ? c2 : cp));
          const bool wmax = (q7 & 1) != 0;
          const bool act = q7 < 6 || (q7 == 6 && (up.x != 0.f || up.y != 0.f || up.z != 0.f));
          const int cell = hcell_of(wmax ? um : -um);
          const int o0 = co[cell], o1 = act ? co[cell + 1] : o0;
          float bestv = -1e30f; int besti = 0x7fffffff;
          for (int base = o0 + h2; __any(base < o1); base += 8) {      /* (four loads in flight per lane: a rim circle seen along its axis has sixty candidates, and one load per round was thirty memory latencies in a row) */
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q[u] = cv[base + 2 * u < o1 ? base + 2 * u : o0];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const float l = hull_coord(um, q[u], cm);
              const float val = wmax ? l : -l;
              if (base + 2 * u < o1 && val > bestv) { bestv = val; besti = __float_as_int(q[u].w); }
            }
          }
          const float gm = fmaxf(bestv, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bestv), 0xB1, 0xF, 0xF, true)));      /* (quad_perm [1, 0, 3, 2]: the other lane of the pair) */
          int gi2 = bestv == gm ? besti : 0x7fffffff;
          gi2 = min(gi2, __builtin_amdgcn_update_dpp(0, gi2, 0xB1, 0xF, 0xF, true));
          lo0 = -bcast16<0>(gm); hi0 = bcast16<2>(gm); lo1 = -bcast16<4>(gm); hi1 = bcast16<6>(gm); lo2 = -bcast16<8>(gm); hi2 = bcast16<10>(gm);
          pmin = -bcast16<12>(gm);
          il0 = bcast16i<0>(gi2); ih0 = bcast16i<2>(gi2); il1 = bcast16i<4>(gi2); ih1 = bcast16i<6>(gi2); il2 = bcast16i<8>(gi2); ih2 = bcast16i<10>(gi2);
        }
        PCLK_ADD(30, __builtin_readcyclecounter())
        const float g0 = lo0 - hcm.x, g1 = -hi0 - hcm.x, g2 = lo1 - hcm.y, g3 = -hi1 - hcm.y, g4 = lo2 - hcm.z, g5 = -hi2 - hcm.z;      /* face +k: lowest vertex above it; face -k: highest vertex below it */
        float best = g0; int bf = 0;
        if (g1 > best + K_TIE_EPS) { best = g1; bf = 1; }
        if (g2 > best + K_TIE_EPS) { best = g2; bf = 2; }
        if (g3 > best + K_TIE_EPS) { best = g3; bf = 3; }
        if (g4 > best + K_TIE_EPS) { best = g4; bf = 4; }
        if (g5 > best + K_TIE_EPS) { best = g5; bf = 5; }
        const float d = best - RP_HULL_MARGIN;
        /* the probe direction: cores farther apart than the margin and the two shape margins along it - what GJK's distance phase would end with (oracle
         * hull_box_gjk: "apart").  Only with GJK on: without it such a pair goes to the OBB path, which finds the OBBs apart all the same */
        const bool probe_apart = m->gjk && warm && pmin > mg + 2.f * RP_HULL_MARGIN + 1e-6f;
        HCLK_ADD(15, 1 + ((d > mg) ? 65536 : 0) + ((probe_apart && !(d > mg)) ? (1ull << 32) : 0ull))               /* (profiling build: hull pairs scanned | of them apart << 16) */
        if (!(d > mg)) {
          /* the first vertex (lowest number: the oracle's sequential scan keeps the first strict extreme) whose coordinate along that axis IS the extreme: its lanes kept it */
          const int k = bf >> 1;
          const int iv = (bf & 1) ? (k == 0 ? ih0 : (k == 1 ? ih1 : ih2)) : (k == 0 ? il0 : (k == 1 ? il1 : il2));
          out = -1;
          if (iv != 0x7fffffff) {
            const float4 v = tv[iv];
            const V3 lv = mk3(hull_coord(u0, v, c0), hull_coord(u1, v, c1), hull_coord(u2, v, c2));      /* the vertex in box coordinates */
            const bool beside = (k != 0 && fabsf(lv.x) > hc0.x) || (k != 1 && fabsf(lv.y) > hc0.y) || (k != 2 && fabsf(lv.z) > hc0.z);
            if (!beside) {
              out = 1;
              const float sg = (bf & 1) ? -1.f : 1.f;
              nloc = mk3(k == 0 ? sg : 0.f, k == 1 ? sg : 0.f, k == 2 ? sg : 0.f);
              ploc = lv - nloc * best;                       /* on the box face under the vertex */
              dcon = d;
            } else if (probe_apart) {
              out = 0;                                       /* (only here: a vertex OVER the face within the margin is a contact whatever the probe says - it can be a
                                                              * millimetre beside the box CORE's face and read 'apart' by a fraction of the shape margin.  The cache stays) */
            } else if (m->gjk) {
              /* the deepest vertex lies BESIDE the face (box edges and corners): GJK's distance phase, cores with the 0.001 margin around each (oracle hull_box_gjk;
               * -1 again = the cores touch or overlap: the OBB path keeps that case).  Box frame, seeded with lv against the corner(s) of the box core's nearest feature;
               * support queries = scans of the direction's cell by the row's sixteen lanes */
              HCLK_ADD(27, 1) HCLK_ADD(28, -(long long)__builtin_readcyclecounter()) PCLK_ADD(31, -(long long)__builtin_readcyclecounter())
              V3 hbc = mk3(hc0.x - fminf(RP_HULL_MARGIN, hc0.x), hc0.y - fminf(RP_HULL_MARGIN, hc0.y), hc0.z - fminf(RP_HULL_MARGIN, hc0.z));
              const bool ox = fabsf(lv.x) > hbc.x, oy = fabsf(lv.y) > hbc.y, oz = fabsf(lv.z) > hbc.z;
              const int nout = (ox ? 1 : 0) + (oy ? 1 : 0) + (oz ? 1 : 0);
              /* (the simplex lives in the row's scratch, and so does v while a scan runs and the scan's directions while the simplex is solved: the kernel has no registers for them) */
              double* Z = (double*)&L.npscr[HROW_FLOATS * row];
              float* Y = &L.npscr[HROW_FLOATS * row + 2 * 18];
              static_assert(NPSCR_FLOATS >= 4 * HROW_FLOATS && 64 >= 2 * 18 + 15, "a row's scratch: GJK's simplex (twelve doubles), the weights and v, 8-byte aligned, fifteen floats behind them; the polytope of hull_epa16 from float 64 on");
#define GJK_PARK_DIRS() do { st3(Y, u0); st3(Y + 3, u1); st3(Y + 6, u2); st3(Y + 9, mk3(c0, c1, c2)); st3(Y + 12, hbc); asm volatile("" ::: "memory"); } while (0)
#define GJK_FETCH_DIRS() do { asm volatile("" ::: "memory"); u0 = ld3(Y); u1 = ld3(Y + 3); u2 = ld3(Y + 6); { const V3 t_ = ld3(Y + 9); c0 = t_.x; c1 = t_.y; c2 = t_.z; } hbc = ld3(Y + 12); } while (0)
              GjkSimplex S;
              S.W = Z; S.b1 = S.b2 = S.b3 = 0; S.i1 = S.i2 = S.i3 = 0;
              auto diffd = [](V3 a, V3 b) { return mkd((double)a.x - (double)b.x, (double)a.y - (double)b.y, (double)a.z - (double)b.z); };      /* w = a - b, in double from the fp32 points */
              bool fail = false;
              if (warm) {
                /* the simplex this pair's last call ended with, at today's poses: its hull vertices (by number) against its box-core corners */
                const float4 gs0 = *(const float4*)gslot;
                const int nc = __float_as_int(gs0.y);
                S.n = nc & 15; S.b0 = (nc >> 4) & 7; S.b1 = (nc >> 8) & 7; S.b2 = (nc >> 12) & 7;
                S.i0 = __float_as_int(gs0.z); S.i1 = __float_as_int(gs0.w); S.i2 = __float_as_int(gslot[4]);
                S.i0 = (unsigned)S.i0 < (unsigned)nn ? S.i0 : 0; S.i1 = (unsigned)S.i1 < (unsigned)nn ? S.i1 : 0; S.i2 = (unsigned)S.i2 < (unsigned)nn ? S.i2 : 0;
                S.n = S.n < 1 ? 1 : (S.n > 3 ? 3 : S.n);
                const float4 q0 = tv[S.i0], q1 = tv[S.i1], q2 = tv[S.i2];
                S.put(0, diffd(mk3(hull_coord(u0, q0, c0), hull_coord(u1, q0, c1), hull_coord(u2, q0, c2)), GjkSimplex::corner(S.b0, hbc)));
                S.put(1, diffd(mk3(hull_coord(u0, q1, c0), hull_coord(u1, q1, c1), hull_coord(u2, q1, c2)), GjkSimplex::corner(S.b1, hbc)));
                S.put(2, diffd(mk3(hull_coord(u0, q2, c0), hull_coord(u1, q2, c1), hull_coord(u2, q2, c2)), GjkSimplex::corner(S.b2, hbc)));
              } else {
                const int near = (lv.x >= 0.f ? 1 : 0) | (lv.y >= 0.f ? 2 : 0) | (lv.z >= 0.f ? 4 : 0);
                S.b0 = near; S.n = 1; S.i0 = S.i1 = iv;
                if (nout == 2) {                             /* the two ends of the nearest edge: along the one axis lv lies inside of */
                  const int fb = !ox ? 1 : (!oy ? 2 : 4);
                  S.b0 = near & ~fb; S.b1 = near | fb;
                  S.put(1, diffd(lv, GjkSimplex::corner(S.b1, hbc))); S.n = 2;
                }
                S.put(0, diffd(lv, GjkSimplex::corner(S.b0, hbc)));
                fail = nout == 0;
              }
              WSYNC();
              bool apart = false, epa_ok = false, tetra = false;
              D3 epa_n = mkd(0, 0, 0), epa_b = mkd(0, 0, 0); double epa_depth = 0.0;
              D3 v = mkd(0, 0, 0); double dd = 0.0;
              if (!fail) {
                GJK_PARK_DIRS();
                gjk_closest(S, lane);
                GJK_FETCH_DIRS();
                v = S.closest(); dd = ddot(v, v);
                fail = dd < GJK_ZERO;
              }
              const double far = (double)mg + 2.0 * (double)RP_HULL_MARGIN;
#pragma unroll 1
              for (int it = 0; it < 32 && !fail; it++) {
                HCLK_ADD(26, 1)
                V3 wa; int wi;
                const V3 vf = mk3((float)v.x, (float)v.y, (float)v.z);
                Z[15] = v.x; Z[16] = v.y; Z[17] = v.z;
                asm volatile("" ::: "memory");
                {                                            /* hull: the vertex of largest projection on -v (lowest number among equals) */
                  const V3 dl = -(u0 * vf.x + u1 * vf.y + u2 * vf.z);
                  float bd = -1e30f; int bi = 0x7fffffff; V3 bq = mk3(0, 0, 0);
                  const int cell = hcell_of(dl);
                  const int o0 = co[cell], o1 = co[cell + 1];      /* (the cell of the direction: a few vertices, one round of the row - five for a rim circle seen along its axis) */
                  for (int base = o0 + l16; __any(base - l16 < o1); base += 32) {
                    const float4 qa = cv[base < o1 ? base : o0], qb = cv[base + 16 < o1 ? base + 16 : o0];      /* (two loads in flight) */
                    const float da = __fmaf_rn(dl.z, qa.z, __fmaf_rn(dl.y, qa.y, dl.x * qa.x)), db = __fmaf_rn(dl.z, qb.z, __fmaf_rn(dl.y, qb.y, dl.x * qb.x));
                    if (base < o1 && da > bd) { bd = da; bi = __float_as_int(qa.w); bq = mk3(qa.x, qa.y, qa.z); }
                    if (base + 16 < o1 && db > bd) { bd = db; bi = __float_as_int(qb.w); bq = mk3(qb.x, qb.y, qb.z); }
                  }
                  const float top = row_max_f(bd);
                  const int key = row_min_i(bd == top ? ((bi << 4) | l16) : 0x7fffffff);      /* (the lowest vertex number among equals; the lane that holds it rides along) */
                  wi = key >> 4;
                  const int win = (lane & 48) | (key & 15);
                  const V3 q = mk3(__shfl(bq.x, win), __shfl(bq.y, win), __shfl(bq.z, win));
                  wa = mk3(hull_coord(u0, make_float4(q.x, q.y, q.z, 0.f), c0), hull_coord(u1, make_float4(q.x, q.y, q.z, 0.f), c1), hull_coord(u2, make_float4(q.x, q.y, q.z, 0.f), c2));
                }
                asm volatile("" ::: "memory");
                v = mkd(Z[15], Z[16], Z[17]);
                const int wb = (vf.x >= 0.f ? 1 : 0) | (vf.y >= 0.f ? 2 : 0) | (vf.z >= 0.f ? 4 : 0);      /* box core: the corner of largest projection on v */
                const D3 w = diffd(wa, GjkSimplex::corner(wb, hbc));
                const double vv = ddot(v, v), vw = ddot(v, w);
                /* v . w / |v| is a lower bound of the distance: beyond the pair's margin and the two shape margins the pair is apart whatever the iteration would still find */
                if (vw > 0.0 && vw * vw > far * far * vv) { apart = true; break; }
                bool dup = false;
                { D3 dw = S.pt(0) - w; dup |= ddot(dw, dw) < GJK_DUP;
                  dw = S.pt(1) - w; dup |= S.n > 1 && ddot(dw, dw) < GJK_DUP;
                  dw = S.pt(2) - w; dup |= S.n > 2 && ddot(dw, dw) < GJK_DUP; }
                if (dup || vv - vw <= GJK_REL * vv) break;
                S.put(S.n, w);
                if (S.n == 1) { S.b1 = wb; S.i1 = wi; } else if (S.n == 2) { S.b2 = wb; S.i2 = wi; } else { S.b3 = wb; S.i3 = wi; }
                S.n++;
                WSYNC();
                GJK_PARK_DIRS();
                gjk_closest(S, lane);
                GJK_FETCH_DIRS();
                if (S.n == 4) {                                  /* the origin lies inside the tetrahedron: the cores overlap */
                  fail = true; tetra = true;
                  break;
                }
                v = S.closest();
                const double nd = ddot(v, v);
                if (nd < GJK_ZERO) { fail = true; break; }
                if (nd >= dd * GJK_STALL) { dd = nd; break; }
                dd = nd;
              }
              if (tetra && m->epa) { GJK_PARK_DIRS(); epa_ok = hull_epa16(L, lane, S, cv, co, epa_n, epa_depth, epa_b); }      /* (after the loop: fewer values live across it) */
#undef GJK_PARK_DIRS
#undef GJK_FETCH_DIRS
              HCLK_ADD(28, __builtin_readcyclecounter()) PCLK_ADD(31, __builtin_readcyclecounter())
              const double distd = sqrt(ddot(v, v));
              HCLK_ADD(27, apart ? (1ull << 16) : (fail ? (1ull << 48) : 0ull))
              if (gslot && l16 == 0) {                       /* what the next call of this pair starts from (oracle GAX_STORE / GAX_CLEAR) */
                if (!fail && (apart || distd > GJK_ZERO)) {
                  *(float4*)gslot = make_float4(__int_as_float(pi_u + 1), __int_as_float(S.n | (S.b0 << 4) | ((S.n > 1 ? S.b1 : 0) << 8) | ((S.n > 2 ? S.b2 : 0) << 12)), __int_as_float(S.i0), __int_as_float(S.n > 1 ? S.i1 : 0));
                  *(float4*)(gslot + 4) = make_float4(__int_as_float(S.n > 2 ? S.i2 : 0), (float)v.x, (float)v.y, (float)v.z);
                } else if (warm) gslot[0] = __int_as_float(0);
              }
              if (epa_ok) {                                  /* penetrating cores: the two margins add along the polytope's normal (oracle hull_box_gjk, RPO_RULE_EPA) */
                out = 1;
                nloc = mk3((float)-epa_n.x, (float)-epa_n.y, (float)-epa_n.z);      /* from the box toward the hull */
                ploc = mk3((float)epa_b.x, (float)epa_b.y, (float)epa_b.z) + nloc * RP_HULL_MARGIN;
                dcon = -(float)epa_depth - 2.f * RP_HULL_MARGIN;
              } else if (apart) out = 0;
              else if (!fail && distd > GJK_ZERO) {
                const float dist = (float)distd;
                const float dg = dist - 2.f * RP_HULL_MARGIN;
                if (dg > mg) { out = 0; HCLK_ADD(27, 1ull << 16) }
                else {
                  out = 1;
                  HCLK_ADD(27, 1ull << 32)
                  const double inv = drcp(distd);
                  nloc = mk3((float)(v.x * inv), (float)(v.y * inv), (float)(v.z * inv));
                  const D3 wit = S.witness(hbc);
                  ploc = mk3((float)wit.x, (float)wit.y, (float)wit.z) + nloc * RP_HULL_MARGIN;
                  dcon = dg;
                }
              }
            }
          }
        }
        }
        CPt pt; pt.p = mk3(0, 0, 0); pt.n = mk3(0, 0, 0); pt.dist = 0.f;
        if (out == 1) {                                      /* back to the world; the model's single application point lies halfway along the gap */
          const Xf xc = collider_xf(m, L, cb);
          const V3 nrm = mulv(xc.R, nloc);
          const V3 pB = mulv(xc.R, ploc) + xc.p;
          pt.p = pB + nrm * (0.5f * dcon); pt.n = flip ? -nrm : nrm; pt.dist = dcon;
        }
        if (l16 == 0) {
          L.hout[gi] = out;
          if (out == 1) {                                    /* the point waits for its pair's batch */
            float* q = &L.hpt[gi][0];
            st3(q, pt.p); st3(q + 3, pt.n); q[6] = pt.dist;
          }
        }
  }
}

template <class LDS>
__device__ __forceinline__ int narrowphase_coop(const DevModel* m, LDS& L, int lane, int nact, float* __restrict__ gax) {      /* returns the number of candidate points stored (wave-uniform) */      /* gax: the env's cached GJK results (contact cache row + PMC_AX), nullptr without the cache */
  const int g = lane >> 3, s = lane & 7;
  float* scr = &L.npscr[NPG_SCRATCH * g];
  float* sv = scr;
  float (*poly)[8][3] = (float (*)[8][3])(scr + 16);
  float (*kept)[4] = (float (*)[4])(scr + 64);
  const unsigned below = (1u << s) - 1u;
  /* ---- 1. the HULL PAIRS of the substep, pooled over all active pairs (one lane per active pair decides whether its pair is one), before the eight-lanes-per-pair
   * batches below look at anything.
   * Arm link against a box: the VERTICES of the convex hull of the link's collision mesh (Bullet: btConvexHullShape, margin 0.001) against the
   * box's six faces - the same decisions and arithmetic as the oracle's hull_face.  The vertex deepest along the face of least penetration is the contact
   * if it lies over that face (what GJK / EPA return for a vertex-on-face contact: a link on the ground plate, on the table top); beside the face the
   * pair goes to GJK's distance phase (hull_item), and to the OBB path below when the cores overlap.
   * The pairs that pass the OBB tests are done sixteen lanes each, four at a time (hull_item16), class by class: a CLASS = the pairs that share a slot of the GJK cache
   * (PMC_AXN slots per env, pair index mod PMC_AXN) runs in pair order in one DPP row, as the oracle's sequential loop has them - whichever of its pairs stores last
   * owns the slot afterwards; classes are independent of one another (an env of the literal random-action rollout has 1.5 hull pairs on average and up to 16). */
  {
    const int ai = lane;
    const bool act = ai < nact;
    const int pi = act ? L.act[ai] : 0;
    const int a = m->pair[pi][0], b = m->pair[pi][1];
    const int ta = m->col_type[a], tb = m->col_type[b];
    const float margin0 = fminf(m->col_margin[a], m->col_margin[b]);
    int hf = -1;                                               /* 1: hull contact, 0: hull says apart, -1: no hull pair, or the OBB path */
      const int body_b0 = m->col_body[b], body_a0 = m->col_body[a];
      const bool hswap = act && m->hull_cnt[b] > 0 && ta == 0 && tb == 0 && body_a0 > m->n_arm;
      const int hc = hswap ? b : a, bc = hswap ? a : b;
      const int hn = act ? m->hull_cnt[hc] : 0;
      bool hq = hn > 0 && (hswap || (tb == 0 && body_b0 == 0));
      if (hq) {
        const Xf xa = collider_xf(m, L, hc), xb = collider_xf(m, L, bc);
        const V3 ha = ld3(m->col_he[hc]), hb0 = ld3(m->col_he[bc]);
        const V3 hb = mk3(fmaxf(hb0.x, RP_HULL_MARGIN), fmaxf(hb0.y, RP_HULL_MARGIN), fmaxf(hb0.z, RP_HULL_MARGIN));      /* (the box as the scan sees it: a plate thinner than the margin counts 0.001 thick) */
        /* the link's OBB (it contains the hull) against the same six faces first: if even the OBB stays clear of the box by more than the pair's margin
         * along one of the box's axes, so does every vertex and the scan would end with "apart" - the common case, a long link whose AABB merely overlaps
         * the table's (same outcome as the oracle's full scan; the 1e-5 keeps rounding at the threshold on the scanning side) */
        const V3 tt = xa.p - xb.p;
        const V3 A0 = col(xa.R, 0), A1 = col(xa.R, 1), A2 = col(xa.R, 2), B0 = col(xb.R, 0), B1 = col(xb.R, 1), B2 = col(xb.R, 2);
        const float r0 = ha.x * fabsf(dot(B0, A0)) + ha.y * fabsf(dot(B0, A1)) + ha.z * fabsf(dot(B0, A2));
        const float r1 = ha.x * fabsf(dot(B1, A0)) + ha.y * fabsf(dot(B1, A1)) + ha.z * fabsf(dot(B1, A2));
        const float r2 = ha.x * fabsf(dot(B2, A0)) + ha.y * fabsf(dot(B2, A1)) + ha.z * fabsf(dot(B2, A2));
        const float og = fmaxf(fmaxf(fabsf(dot(B0, tt)) - r0 - hb.x, fabsf(dot(B1, tt)) - r1 - hb.y), fabsf(dot(B2, tt)) - r2 - hb.z);
        /* ... and along the link OBB's own three axes (a lower bound of the hull's distance all the same: fewer pairs reach the scan and the GJK behind it) */
        const float q0 = hb.x * fabsf(dot(A0, B0)) + hb.y * fabsf(dot(A0, B1)) + hb.z * fabsf(dot(A0, B2));
        const float q1 = hb.x * fabsf(dot(A1, B0)) + hb.y * fabsf(dot(A1, B1)) + hb.z * fabsf(dot(A1, B2));
        const float q2 = hb.x * fabsf(dot(A2, B0)) + hb.y * fabsf(dot(A2, B1)) + hb.z * fabsf(dot(A2, B2));
        const float og2 = fmaxf(fmaxf(fabsf(dot(A0, tt)) - q0 - ha.x, fabsf(dot(A1, tt)) - q1 - ha.y), fabsf(dot(A2, tt)) - q2 - ha.z);
        if (fmaxf(og, og2) > margin0 + RP_HULL_MARGIN + 1e-5f) { hq = false; hf = 0; }
      }
      if (act) L.hout[ai] = hf;
      if (hq) { L.hinfo[ai][0] = hc | (bc << 8) | (hswap ? 65536 : 0); L.hinfo[ai][1] = __float_as_int(margin0); L.hinfo[ai][2] = pi; }
      const unsigned long long any = __ballot(hq);
      if (any != 0ull) {                                       /* (wave-uniform, rare in the bench workload: 2 % of the env-substeps) */
        PCLK_H(30, -(long long)__builtin_readcyclecounter())
        const int slot = (gax && m->gjk) ? (pi & (PMC_AXN - 1)) : (ai & (PMC_AXN - 1));      /* (without the cache nothing is shared: any grouping will do) */
        unsigned leaders = 0u;
        unsigned long long mycls = 0ull;
#pragma unroll 1
        for (int c = 0; c < PMC_AXN; c++) {
          const unsigned long long mk = __ballot(hq && slot == c);
          if (mk != 0ull) leaders |= 1u << c;
          if (lane == c) mycls = mk;
        }
        if (lane < PMC_AXN) L.hcls[lane] = mycls;
        WSYNC();
        /* the classes four at a time, one per DPP row (rising class numbers; a row works through its class in pair order, the rows side by side: hull_item16) */
        {
          const int row = lane >> 4;
#pragma unroll 1
          for (unsigned rest = leaders; rest != 0u;) {       /* (wave-uniform) */
            unsigned t = rest;
            if (row >= 1) t &= t - 1u;
            if (row >= 2) t &= t - 1u;
            if (row >= 3) t &= t - 1u;
            unsigned long long cm = t != 0u ? L.hcls[__ffs(t) - 1] : 0ull;
            rest &= rest - 1u; rest &= rest - 1u; rest &= rest - 1u; rest &= rest - 1u;
#pragma unroll 1
            while (__any(cm != 0ull)) {
              const int gi = cm != 0ull ? __ffsll((long long)cm) - 1 : -1;
              cm &= cm - 1ull;
              hull_item16(m, L, lane, gi, gax);
            }
          }
        }
        PCLK_H(30, __builtin_readcyclecounter())
      }
    WSYNC();
    PCLK(19)
  }
  /* ---- 2. everything else, eight lanes per active pair */
  int cbase = 0;                                    /* candidate points stored so far (wave-uniform) */
  for (int base = 0; base < nact; base += 64 / NPG) {      /* wave-uniform trip count; every lane reaches every barrier */
    const int ai = base + g;
    const bool act = ai < nact;
    const int pi = act ? L.act[ai] : 0;
    const int a = m->pair[pi][0], b = m->pair[pi][1];
    const int ta = m->col_type[a], tb = m->col_type[b];
    const float margin0 = fminf(m->col_margin[a], m->col_margin[b]);     /* Bullet: a manifold's breaking threshold is the smaller of its two objects' */
    const bool bbox = act && ta == 0 && tb == 0;
    if (act && s == 0) {
      /* what the pair's contacts will need later, looked up here (the table loads hide behind the axis tests): friction, and the manifold key =
       * object pair, bit 16 "rotation-locked free body against the static world" (the drawer: that manifold keeps only its deepest point),
       * bits 20-21 which halves of the velocity layout the two bodies touch (0 second only, 1 first only, 2 both: DPP row 0 = the arm and the
       * free bodies of free_row0, DPP row 1 = the other free bodies and the scene joints), bit 22 arm link against a movable body - all
       * properties of the two objects, so the same for the whole run of pairs that makes a manifold */
      const int n = m->n_arm, ba = m->col_body[a], bdy = m->col_body[b];
      const int kf = ba - 1 - n;
      const bool single = kf >= 0 && kf < m->n_free && m->free_rot_locked[kf] && bdy == 0;
      auto half0 = [&](int q) { int f = q - 1 - n; return q >= 1 && (q <= n || (f < m->n_free && ((m->free_row0 >> f) & 1))); };
      const bool r0 = half0(ba) || half0(bdy), r1 = (ba >= 1 && !half0(ba)) || (bdy >= 1 && !half0(bdy));
      const bool arm = (ba >= 1 && ba <= n) || (bdy >= 1 && bdy <= n), movable = ba > n || bdy > n;
      L.key[ai] = m->col_obj[a] * 256 + m->col_obj[b] + (single ? 65536 : 0) + ((r0 ? (r1 ? 2 : 1) : 0) << 20) + ((arm && movable) ? (1 << 22) : 0);
      L.pmu[ai] = m->col_friction[a] * m->col_friction[b];
    }
    const int hf = act ? L.hout[ai] : -1;                    /* the hull phase's outcome for this pair: 1 hull contact (staged in L.hpt), 0 the hull says apart, -1 no hull pair / the OBB path */
    int np = 0;
    CPt mine; mine.p = mk3(0, 0, 0); mine.n = mk3(0, 0, 0); mine.dist = 0.f;      /* the point this lane contributes (lane s < np of its group) */
    if (hf == 1 && s == 0) { const float* hp = &L.hpt[ai][0]; mine.p = ld3(hp); mine.n = ld3(hp + 3); mine.dist = hp[6]; np = 1; }
    asm volatile("" ::: "memory");
    const Xf xa = collider_xf(m, L, a), xb = collider_xf(m, L, b);
    const V3 ha = ld3(m->col_he[a]), hb = ld3(m->col_he[b]);
    const bool bb = bbox && hf < 0;                          /* box against box through the SAT + clipping path */
    /* ... whose points, when one of the boxes is an arm link, exist only while the boxes overlap (DevModel.boxbox_margin = 0: btBoxBoxDetector makes none before
     * that; the oracle's RPO_RULE_BOXOVERLAP explains why the arm's pairs and not the resting objects') */
    const int body_a = m->col_body[a], body_b = m->col_body[b];
    const bool arm_pair = (body_a >= 1 && body_a <= m->n_arm) || (body_b >= 1 && body_b <= m->n_arm);
    const float margin = (bb && (arm_pair || m->persist) && m->boxbox_margin >= 0.f) ? m->boxbox_margin : margin0;      /* (with the contact cache every box pair: the cache keeps the points) */
    if (act && !bb && s == 0) {                              /* sphere against box: one lane, closed form */
      if (ta == 0 && tb == 1) np = sphere_box(xb.p, hb.x, xa.p, xa.R, ha, margin, 1, &mine);
      else if (ta == 1 && tb == 0) np = sphere_box(xa.p, ha.x, xb.p, xb.R, hb, margin, 0, &mine);
    }
    /* ---- box against box */
    const V3 ca = xa.p, cb = xb.p;
    const V3 A0 = col(xa.R, 0), A1 = col(xa.R, 1), A2 = col(xa.R, 2), B0 = col(xb.R, 0), B1 = col(xb.R, 1), B2 = col(xb.R, 2);
    const V3 t = ca - cb;
    auto axis = [&](int ax, bool& valid) {                   /* separating-axis candidate number ax: 0-2 A faces, 3-5 B faces, 6-14 edges */
      valid = true;
      if (ax < 3) return pick3(ax, A0, A1, A2);
      if (ax < 6) return pick3(ax - 3, B0, B1, B2);
      int e = ax - 6, i = e / 3, j = e - 3 * i;
      V3 c = cross(pick3(i, A0, A1, A2), pick3(j, B0, B1, B2));
      float l = norm(c);
      valid = !(l < 1e-6f);
      return c * (1.f / l);
    };
#pragma unroll
    for (int round = 0; round < 2; round++) {
      int ax = 8 * round + s;
      if (bb && ax < 15) {
        bool valid;
        V3 Lx = axis(ax, valid);
        float ra = 0.f, rb = 0.f;
        ra += ha.x * fabsf(dot(Lx, A0)); rb += hb.x * fabsf(dot(Lx, B0));
        ra += ha.y * fabsf(dot(Lx, A1)); rb += hb.y * fabsf(dot(Lx, B1));
        ra += ha.z * fabsf(dot(Lx, A2)); rb += hb.z * fabsf(dot(Lx, B2));
        float sval = fabsf(dot(t, Lx)) - ra - rb;
        sv[ax] = valid ? sval : -1e30f;
      }
    }
    WSYNC();
    /* decisions, by every lane of the group alike: reject, best face (first-wins with tolerance), best edge */
    bool reject = !bb;
    float best_s = -1e30f, edge_s = -1e30f; int best_f = 0, ee = 0;
    if (bb) {
#pragma unroll
      for (int f = 0; f < 6; f++) {
        float sf = sv[f];
        reject |= sf > margin;
        if (sf > best_s + (f == 0 ? 0.f : K_TIE_EPS)) { best_s = sf; best_f = f; }
      }
#pragma unroll
      for (int e = 0; e < 9; e++) {
        float se = sv[6 + e];
        reject |= se > margin;
        if (se > edge_s + K_TIE_EPS) { edge_s = se; ee = e; }
      }
    }
    const bool live = bb && !reject;
    const bool edge_case = live && edge_s > best_s + 0.05f * fabsf(best_s) + 1e-6f;
    const bool face_case = live && !edge_case;
    const int best_kind = best_f < 3 ? 0 : 1, best_i = best_f < 3 ? best_f : best_f - 3;
    if (edge_case && s == 0) {
      bool valid;
      V3 n = axis(6 + ee, valid);
      int ei = ee / 3, ej = ee - 3 * ei;
      if (dot(n, t) < 0.f) n = -n;
      V3 pa = ca, pb = cb;
      if (ei != 0) pa = pa + A0 * (dot(n, A0) > 0.f ? -ha.x : ha.x);
      if (ej != 0) pb = pb + B0 * (dot(n, B0) > 0.f ? hb.x : -hb.x);
      if (ei != 1) pa = pa + A1 * (dot(n, A1) > 0.f ? -ha.y : ha.y);
      if (ej != 1) pb = pb + B1 * (dot(n, B1) > 0.f ? hb.y : -hb.y);
      if (ei != 2) pa = pa + A2 * (dot(n, A2) > 0.f ? -ha.z : ha.z);
      if (ej != 2) pb = pb + B2 * (dot(n, B2) > 0.f ? hb.z : -hb.z);
      V3 Ae = pick3(ei, A0, A1, A2), Be = pick3(ej, B0, B1, B2);
      V3 d = pb - pa;
      float ab = dot(Ae, Be), q1 = dot(Ae, d), q2 = -dot(Be, d);
      float den = 1.f - ab * ab, sa = 0.f, sb = 0.f;
      if (den > 1e-9f) { sa = (q1 + ab * q2) / den; sb = (ab * q1 + q2) / den; }
      V3 xA = pa + Ae * sa, xB = pb + Be * sb;
      float dist = dot(xA - xB, n);
      if (dist <= margin) { mine.p = (xA + xB) * 0.5f; mine.n = n; mine.dist = dist; np = 1; }
    }
    /* face contact: reference box X (the one owning the best face), incident box Y */
    const bool kx = best_kind == 0;
    const V3 cX = kx ? ca : cb, cY = kx ? cb : ca;
    const V3 X0 = kx ? A0 : B0, X1 = kx ? A1 : B1, X2 = kx ? A2 : B2, Y0 = kx ? B0 : A0, Y1 = kx ? B1 : A1, Y2 = kx ? B2 : A2;
    const V3 hX = kx ? ha : hb, hY = kx ? hb : ha;
    V3 nref = pick3(best_i, X0, X1, X2);                     /* == best_L */
    if (dot(nref, cY - cX) < 0.f) nref = -nref;
    int j = 0; float bj = -1.f;
    { float v0 = fabsf(dot(nref, Y0)), v1 = fabsf(dot(nref, Y1)), v2 = fabsf(dot(nref, Y2));
      if (v0 > bj + K_TIE_EPS) { bj = v0; j = 0; }
      if (v1 > bj + K_TIE_EPS) { bj = v1; j = 1; }
      if (v2 > bj + K_TIE_EPS) { bj = v2; j = 2; } }
    const V3 Yj = pick3(j, Y0, Y1, Y2);
    const float sj = dot(nref, Yj) > 0.f ? -1.f : 1.f;
    /* The polygon is walked the way btBoxBoxDetector (ODE's dBoxBox2) walks it (oracle RPO_RULE_ODEORDER): the incident face from its (-, -) corner with its
     * two axes in increasing order, the reference rectangle's sides in the order -u1, +u1, -u2, +u2 with u1 < u2 - the order of a pair's points is the order
     * of the solver's rows */
    const int k1 = j == 0 ? 1 : 0, k2 = j == 2 ? 1 : 2;
    const V3 Yk1 = pick3(k1, Y0, Y1, Y2), Yk2 = pick3(k2, Y0, Y1, Y2);
    const float hYj = pick1(j, hY.x, hY.y, hY.z), hYk1 = pick1(k1, hY.x, hY.y, hY.z), hYk2 = pick1(k2, hY.x, hY.y, hY.z);
    const V3 fc = cY + Yj * (sj * hYj);
    if (face_case && s < 4) {
      float sg0 = s < 2 ? -1.f : 1.f, sg1 = (s == 0 || s == 3) ? -1.f : 1.f;
      st3(poly[0][s], fc + Yk1 * (sg0 * hYk1) + Yk2 * (sg1 * hYk2));
    }
    const int u1 = best_i == 0 ? 1 : 0, u2 = best_i == 2 ? 1 : 2;
    const V3 Xu1 = pick3(u1, X0, X1, X2), Xu2 = pick3(u2, X0, X1, X2);
    const float hXu1 = pick1(u1, hX.x, hX.y, hX.z), hXu2 = pick1(u2, hX.x, hX.y, hX.z), hXi = pick1(best_i, hX.x, hX.y, hX.z);
    int n = 4, cur = 0;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      WSYNC();
      const V3 u = pass < 2 ? Xu1 : Xu2;
      const float h = pass < 2 ? hXu1 : hXu2, sign = (pass & 1) ? 1.f : -1.f;
      bool k0 = false, kc = false; V3 va = mk3(0, 0, 0), vb = va; float da = 0.f, db = 0.f;
      if (face_case && s < n) {
        va = ld3(poly[cur][s]);
        vb = ld3(poly[cur][(s + 1 == n) ? 0 : s + 1]);
        da = sign * dot(va - cX, u) - h;
        db = sign * dot(vb - cX, u) - h;
        k0 = da <= 0.f;
        kc = (da < 0.f && db > 0.f) || (da > 0.f && db < 0.f);
      }
      unsigned m0 = (unsigned)(__ballot(k0) >> (8 * g)) & 0xFFu, m1 = (unsigned)(__ballot(kc) >> (8 * g)) & 0xFFu;
      int off = __popc(m0 & below) + __popc(m1 & below);
      if (k0) st3(poly[cur ^ 1][off], va);
      if (kc) { float tt = da / (da - db); st3(poly[cur ^ 1][off + (k0 ? 1 : 0)], va + (vb - va) * tt); }
      n = __popc(m0) + __popc(m1);
      cur ^= 1;
    }
    WSYNC();
    bool keep = false; V3 pv = mk3(0, 0, 0); float dist = 0.f;
    if (face_case && s < n) {
      pv = ld3(poly[cur][s]);
      dist = dot(pv - cX, nref) - hXi;
      keep = !(dist > margin);
    }
    unsigned mk = (unsigned)(__ballot(keep) >> (8 * g)) & 0xFFu;
    const int cnt = __popc(mk);
    if (keep) { float* kp = kept[__popc(mk & below)]; st3(kp, pv - nref * (0.5f * dist)); kp[3] = dist; }
    WSYNC();
    if (face_case) {
      int deepest = 0;
      for (int c = 1; c < cnt; c++) if (kept[c][3] < kept[deepest][3] - K_TIE_EPS) deepest = c;
      const V3 nn = best_kind == 1 ? nref : -nref;
      const int outn = cnt <= 4 ? cnt : 4;
      int src = s;
      if (cnt > 4) {
        /* more than four (rare): the detector's cullPoints2, by every lane of the group alike - the deepest first, then for each of the three directions
         * a quarter turn further around the polygon's centroid (in the reference face's plane) the unused point nearest to it in angle; lane s takes the
         * s-th pick (same arithmetic as the oracle's box_box).  NOTE: `cnt` is the same in all eight lanes of the group, so the barriers below are
         * reached by whole groups */
        /* (few registers on purpose: the polygon's plane coordinates and angles go through the clip scratch, free by now) */
        float (*q2)[3] = poly[0];
        if (s < cnt) {
          const V3 r = ld3(kept[s]) + nref * (0.5f * kept[s][3]) - cX;      /* back to the polygon's vertex */
          q2[s][0] = dot(r, Xu1); q2[s][1] = dot(r, Xu2);
        }
        WSYNC();
        float area = 0.f, cx = 0.f, cy = 0.f;
#pragma unroll 1
        for (int v = 0; v < cnt; v++) {
          const int w = v + 1 == cnt ? 0 : v + 1;
          const float q = q2[v][0] * q2[w][1] - q2[w][0] * q2[v][1];
          area += q; cx += q * (q2[v][0] + q2[w][0]); cy += q * (q2[v][1] + q2[w][1]);
        }
        area = fabsf(area) > 1e-30f ? 1.f / (3.f * area) : 1e30f;
        cx *= area; cy *= area;
        /* "nearest in angle" without angles (oracle box_box): unit directions from the centroid, the largest cosine against the deepest point's direction
         * turned by jj quarter turns; the first of equals wins */
        float mx = 1.f, my = 0.f;
        if (s < cnt) {
          const float x = q2[s][0] - cx, y = q2[s][1] - cy, l = sqrtf(x * x + y * y);
          if (l > 0.f) { mx = x / l; my = y / l; }
        }
        WSYNC();                                             /* every lane has read the coordinates: the scratch takes the directions */
        if (s < cnt) { q2[s][0] = mx; q2[s][1] = my; }
        WSYNC();
        float wx = q2[deepest][0], wy = q2[deepest][1];
        unsigned avail = ((1u << cnt) - 1u) & ~(1u << deepest);
        src = deepest;
#pragma unroll 1
        for (int jj = 1; jj < 4; jj++) {
          const float tq = wx; wx = -wy; wy = tq;            /* a quarter turn further */
          float bestc = -2.f; int pick = deepest;
#pragma unroll 1
          for (int v = 0; v < cnt; v++) {
            const float c = q2[v][0] * wx + q2[v][1] * wy;
            if (((avail >> v) & 1u) && c > bestc) { bestc = c; pick = v; }
          }
          avail &= ~(1u << pick);
          src = s == jj ? pick : src;
        }
      }
      if (s < outn) { mine.p = ld3(kept[src]); mine.n = nn; mine.dist = kept[src][3]; }
      np = outn;
    }
    /* the pair's points go to the compact candidate list, pairs in order; the list ends at CANDMAX points (the oracle's rule) */
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));                     /* (recomputed here, opaque: a 64-bit mask held across the hull scans is a register pair they spill for) */
    const unsigned long long lower_groups = (1ull << (lane_o & ~7)) - 1ull;      /* lanes of the groups before this one */
    const int npl = (act && s == 0) ? np : 0;                /* the group's count, at its first lane */
    int off = cbase, tot = 0;
#pragma unroll
    for (int bit = 0; bit < 3; bit++) {
      const unsigned long long mb = __ballot((npl >> bit) & 1);
      off += __popcll(mb & lower_groups) << bit; tot += __popcll(mb) << bit;
    }
    cbase += tot;
    const int npg = __shfl(np, lane & ~7);                   /* edge / sphere pairs: only the first lane knows */
    const int nst = min(npg, max(0, CANDMAX - off));
    if (act && s < nst) {
      float* c = &L.cand[(off + s) * 8];
      *(float4*)c = make_float4(mine.p.x, mine.p.y, mine.p.z, mine.n.x);
      *(float4*)(c + 4) = make_float4(mine.n.y, mine.n.z, mine.dist, __int_as_float(a | (b << 8) | (ai << 16) | (body_a << 22) | (body_b << 27)));      /* the record carries its two colliders, its pair (6 bits) and the two bodies (5 bits each) */
    }
    if (act && s == 0) L.candn[ai] = nst | (min(off, CANDMAX) << 8);
    WSYNC();       /* the scratch is reused by the next pass */
  }
  return min(cbase, CANDMAX);
}

/* btPersistentManifold::sortCachedPoints on points stored as 8-float records (p3 n3 dist pad) */
__device__ __forceinline__ int manifold_replace_index(const float* c4, const float* pt) {
  int deepest = -1; float maxpen = pt[6];
  for (int i = 0; i < 4; i++) if (c4[8 * i + 6] < maxpen - K_TIE_EPS) { deepest = i; maxpen = c4[8 * i + 6]; }
  float res[4] = {0, 0, 0, 0};
  V3 P = ld3(pt), p0 = ld3(c4), p1 = ld3(c4 + 8), p2 = ld3(c4 + 16), p3 = ld3(c4 + 24), cr;
  if (deepest != 0) { cr = cross(P - p1, p3 - p2); res[0] = dot(cr, cr); }
  if (deepest != 1) { cr = cross(P - p0, p3 - p2); res[1] = dot(cr, cr); }
  if (deepest != 2) { cr = cross(P - p0, p3 - p1); res[2] = dot(cr, cr); }
  if (deepest != 3) { cr = cross(P - p0, p2 - p1); res[3] = dot(cr, cr); }
  int best = 0;
  for (int i = 1; i < 4; i++) if (res[i] > res[best] * (1.f + 1e-4f)) best = i;
  return best;
}

/* broadphase + narrowphase + manifolds -> L.con*, returns ncon (wave-uniform) */
template <class LDS>
__device__ __forceinline__ int collide(const DevModel* m, LDS& L, int lane, int env, const int* npm_early = nullptr) {
  /* 1. AABB sweep over the baked candidate pairs, 64 per pass; keep the first MAXACT overlapping, in order */
  int nact = 0;
  const int npair = m->n_pair;
  const unsigned short* pp = (const unsigned short*)m->pair;      /* two collider indices per entry, low byte first */
  unsigned short pv[RP_MAX_PAIR / 64];
#pragma unroll
  for (int k = 0; k < RP_MAX_PAIR / 64; k++) { int pi = 64 * k + lane; pv[k] = pp[pi < npair ? pi : 0]; }   /* all loads in flight at once */
  unsigned ovbits = 0u;                  /* bit k: pair 64 k + lane overlaps.  All tests first (independent LDS reads pipeline) */
#pragma unroll
  for (int k = 0; k < RP_MAX_PAIR / 64; k++) {
    int pi = 64 * k + lane;
    if (64 * k < npair && pi < npair) {
      int a = pv[k] & 255, b = pv[k] >> 8;
      /* four 16-byte reads, unconditionally, combined without short-circuit: `||` would make every read wait for the
       * comparison before it (98 exec-mask branches, one LDS round trip each) */
      const float4 alo = *(const float4*)&L.aabb[8 * a], ahi = *(const float4*)&L.aabb[8 * a + 4];
      const float4 blo = *(const float4*)&L.aabb[8 * b], bhi = *(const float4*)&L.aabb[8 * b + 4];
      const float margin = fminf(alo.w, blo.w);
      bool sep = (alo.x > bhi.x + margin) | (blo.x > ahi.x + margin) | (alo.y > bhi.y + margin) | (blo.y > ahi.y + margin) | (alo.z > bhi.z + margin) | (blo.z > ahi.z + margin);
      ovbits |= sep ? 0u : (1u << k);
    }
  }
#pragma unroll
  for (int k = 0; k < RP_MAX_PAIR / 64; k++) {      /* then the ordered compaction: ballots and popcounts only */
    if (64 * k >= npair) break;
    bool ov = (ovbits >> k) & 1u;
    unsigned long long mask = __ballot(ov);
    int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (ov && nact + before < MAXACT) L.act[nact + before] = 64 * k + lane;
    nact += __popcll(mask);
    if (nact >= MAXACT) { nact = MAXACT; break; }
  }
  WSYNC();
  PCLK(8)
#if defined(RP_CLOCKS) && RP_CLOCKS == 2
  if (lane == 0) g_clk[32 * (blockIdx.x & 4095) + 12] = nact;
#endif
  /* 2. narrowphase: eight lanes per active pair */
  const int ncand_all = uni(narrowphase_coop<LDS>(m, L, lane, nact, m->persist ? m->pmcache + (size_t)env * PMC_FLOATS + PMC_AX : nullptr));
  WSYNC();
  asm volatile("" : "+v"(lane));                          /* (the lane number once more, opaque: addresses the manifold stage derives from it are computed there, not held across the narrowphase) */
  PCLK(9)
  /* 3. manifolds: one per run of equal object pairs, <= 4 points (1 for a rotation-locked body against the world).  A manifold's size
   * follows from its candidate counts alone, so every manifold knows its place in the contact list before anything is merged; the
   * first lane of a run merges it (sequentially, in candidate order) into L.man at that place - and only if the list still has room
   * for at least one of its points (cap MAXC, in manifold order). */
  const unsigned long long lower = (1ull << lane) - 1ull;
  int kept = 0, pk = 0;
  float* man = L.man;
  if (m->persist) {
    /* 3'. PERSISTENT manifolds (oracle collide_persistent): the env's contact cache - one manifold per object pair in creation order, <= 4 points kept in the
     * two bodies' frames - is staged behind L.man; lane i owns manifold i.  (a) manifolds whose object pair has no AABB-overlapping collider pair any
     * more leave, the rest close ranks; (b) every active object pair without a manifold gets one, empty, at the end (creation order = row order);
     * (c) every candidate enters its manifold: it replaces the cached point within the threshold of it (A's frame), else it is appended, else it takes
     * the place sortCachedPoints picks; (d) every point is refreshed from its two local points and dropped beyond the threshold (distance or sideways
     * drift; the last point takes the slot); (e) the surviving points - the deepest alone for a rotation-locked body against the static world - go to
     * L.man as this substep's records, the cache goes back to memory. */
    float* C = L.npscr + 8 * MANPTS;
    static_assert(8 * MANPTS + PMC_MANIFOLDS <= NPSCR_FLOATS, "the staged contact cache lies behind L.man in the narrowphase scratch");
    float* g = m->pmcache + (size_t)env * PMC_FLOATS;
    /* only the manifolds in use - typically four of eleven - are loaded and stored: their count is the row's first word (k_prep2 has it fetched with the state
     * record by its other wave: npm_early; the one-kernel path reads it here) */
    const int npm_pre = min(max(uni(npm_early ? *npm_early : __float_as_int(g[0])), 0), PM_MAX);
    const int used_in = PMC_HDR + PMC_MAN * npm_pre;         /* (a multiple of four floats) */
    for (int i = lane * 4; i < used_in; i += 256) *(float4*)&C[i] = *(const float4*)&g[i];
    if (lane == 0) C[0] = __int_as_float(npm_pre);
    WSYNC();
    PCLK(21)
    int npm = uni(__float_as_int(C[0]));
    {                                                        /* (a) */
      const int mykey = lane < npm ? __float_as_int(C[PMC_HDR + PMC_MAN * lane]) : -1;
      const int akey = L.key[lane < nact ? lane : 0];      /* pair k's key waits in lane k: the loops below read lanes, not LDS (a dependent LDS round trip per iteration until round 5) */
      bool touched = false; int fl = 0;
      for (int k = 0; k < nact; k++) { const int kk = __builtin_amdgcn_readlane(akey, k); if ((kk & 0xFFFF) == mykey) { touched = true; fl = kk >> 16; } }
      if (touched) C[PMC_HDR + PMC_MAN * lane + 3] = __int_as_float(fl);
      const unsigned long long keep = __ballot(touched);
      WSYNC();
      int nk = 0;
      for (int j = 0; j < npm; j++)
        if ((keep >> j) & 1ull) {
          if (nk != j && lane < PMC_MAN) C[PMC_HDR + PMC_MAN * nk + lane] = C[PMC_HDR + PMC_MAN * j + lane];
          nk++;
        }
      npm = nk;
      WSYNC();
    }
    {                                                        /* (b) the collider pairs of one object pair are neighbours in the pair list */
      const int objk = lane < nact ? (L.key[lane] & 0xFFFF) : -1;
      const bool first = lane < nact && (lane == 0 || (L.key[lane - 1] & 0xFFFF) != objk);
      const int mkey = __float_as_int(C[PMC_HDR + PMC_MAN * (lane < npm ? lane : 0)]);      /* (after the ranks closed) manifold j's key in lane j */
      bool present = false;
      for (int j = 0; j < npm; j++) present |= __builtin_amdgcn_readlane(mkey, j) == objk;
      const bool isnew = first && !present;
      const unsigned long long mnew = __ballot(isnew);
      const int slot = npm + __popcll(mnew & lower);
      if (isnew && slot < PM_MAX) {
        const int pi = L.act[lane];
        float* M = &C[PMC_HDR + PMC_MAN * slot];
        M[0] = __int_as_float(objk); M[1] = __int_as_float(0);
        M[2] = fminf(m->col_margin[m->pair[pi][0]], m->col_margin[m->pair[pi][1]]);
        M[3] = __int_as_float(L.key[lane] >> 16);
        for (int t = 4; t < PMC_MAN; t++) M[t] = 0.f;        /* (the slot was not loaded: what lies there is narrowphase scratch, and the cache is part of the state rows) */
      }
      npm = min(PM_MAX, npm + (int)__popcll(mnew));
      WSYNC();
    }
    PCLK(22)
    /* (c) one lane per CANDIDATE (at most CANDMAX = 64): its manifold, its two points in the bodies' frames, the nearest cached point within the threshold
     * - among the points as they are now, before any candidate of this substep goes in (the oracle's rule: that is what makes this parallel).  A matched
     * candidate replaces that point - of several on one slot the last in pair order (LDS max over the lane numbers) stays; the unmatched ones (a contact
     * in its first substep: rare) go in afterwards, one after the other, in the lane of their manifold */
    unsigned long long unmatched, mine = 0ull;      /* candidates that matched no cached point: all of them | those of manifold `lane` */
    {
      const int ncand = ncand_all;                          /* (the narrowphase's own count: six dependent cross-lane maxima over the pairs' counts until round 5) */
      const bool isc = lane < ncand;
      const float* c = &L.cand[8 * (isc ? lane : 0)];
      const V3 p = ld3(c), nr = mk3(c[3], c[4], c[5]);
      const float dist = c[6];
      const int abw = __float_as_int(c[7]);
      const int objk = L.key[(abw >> 16) & 63] & 0xFFFF;
      const int mkey = __float_as_int(C[PMC_HDR + PMC_MAN * (lane < npm ? lane : 0)]);      /* manifold j's key in lane j (the new ones included) */
      int mi = -1;
      for (int jm = 0; jm < npm; jm++) if (__builtin_amdgcn_readlane(mkey, jm) == objk) mi = jm;
      if (!isc) mi = -1;
      const float* M = &C[PMC_HDR + PMC_MAN * (mi >= 0 ? mi : 0)];
      const float thr = M[2];
      if (dist > thr) mi = -1;
      const int n = __float_as_int(M[1]);
      const int ba = (abw >> 22) & 31, bb = (abw >> 27) & 31;
      const V3 lA = tmulv(ldm3(&L.xR[9 * ba]), p + nr * (0.5f * dist) - ld3(&L.xp[3 * ba]));
      const V3 lB = tmulv(ldm3(&L.xR[9 * bb]), p - nr * (0.5f * dist) - ld3(&L.xp[3 * bb]));
      int sl = -1; float shortest = thr * thr;
      for (int q = 0; q < 4; q++) {
        const V3 d = ld3(&M[8 + PMC_PT * q]) - lA;
        const float dd = dot(d, d);
        if (q < n && dd < shortest) { shortest = dd; sl = q; }
      }
      if (lane < 4 * PM_MAX) C[PMC_HDR + PMC_MAN * (lane >> 2) + 4 + (lane & 3)] = 0.f;      /* slot owners (header pad): candidate number + 1, 0 = none (no NaN patterns in a state row) */
      WSYNC();
      const bool hit = mi >= 0 && sl >= 0;
      int* own = (int*)&C[PMC_HDR + PMC_MAN * (hit ? mi : 0) + 4 + (hit ? sl : 0)];
      if (hit) atomicMax(own, lane + 1);
      WSYNC();
      if (hit && *own == lane + 1) {
        float* P = &C[PMC_HDR + PMC_MAN * mi + 8 + PMC_PT * sl];
        st3(P, lA); st3(P + 3, lB); st3(P + 6, nr); P[9] = dist; P[10] = __int_as_float(abw & (int)0xFFC0FFFF);      /* colliders and bodies (the pair index is of this substep only) */
      }
      unmatched = __ballot(mi >= 0 && sl < 0);
      if (unmatched != 0ull)                                 /* (wave-uniform) every manifold's own list: sequential inside a manifold, the manifolds side by side */
        for (int jm = 0; jm < npm; jm++) { const unsigned long long b = __ballot(mi == jm && sl < 0); if (lane == jm) mine = b; }
      WSYNC();
    }
    if (unmatched != 0ull) {                                 /* (a contact in its first substep) lane i adds the unmatched candidates of manifold i, one after the other in pair
                                                              * order - round r takes every manifold's r-th at once (until round 5 the lanes walked ALL unmatched candidates and
                                                              * skipped the other manifolds': 64 first contacts of an arm falling onto the furniture cost 80 k cycles) */
      float* M = &C[PMC_HDR + PMC_MAN * (lane < npm ? lane : 0)];
      int n = __float_as_int(M[1]);
      const float thr = M[2];
      for (unsigned long long todo = lane < npm ? mine : 0ull; __any(todo != 0ull); todo &= todo - 1ull) {
        if (todo == 0ull) continue;
        const int ci = __ffsll((long long)todo) - 1;
        const float* c = &L.cand[8 * ci];
        const int abw = __float_as_int(c[7]);
        const V3 p = ld3(c), nr = mk3(c[3], c[4], c[5]);
        const float dist = c[6];
        const int ba = (abw >> 22) & 31, bb = (abw >> 27) & 31;
        const V3 lA = tmulv(ldm3(&L.xR[9 * ba]), p + nr * (0.5f * dist) - ld3(&L.xp[3 * ba]));
        const V3 lB = tmulv(ldm3(&L.xR[9 * bb]), p - nr * (0.5f * dist) - ld3(&L.xp[3 * bb]));
        int sl = -1; float shortest = thr * thr;
        for (int q = 0; q < n; q++) { const V3 d = ld3(&M[8 + PMC_PT * q]) - lA; const float dd = dot(d, d); if (dd < shortest) { shortest = dd; sl = q; } }
        if (sl < 0) {
          if (n < 4) sl = n++;
          else {                                             /* btPersistentManifold::sortCachedPoints on the local-A points */
            int deepest = -1; float maxpen = dist;
            for (int q = 0; q < 4; q++) if (M[8 + PMC_PT * q + 9] < maxpen) { deepest = q; maxpen = M[8 + PMC_PT * q + 9]; }
            const V3 q0 = ld3(&M[8]), q1 = ld3(&M[8 + PMC_PT]), q2 = ld3(&M[8 + 2 * PMC_PT]), q3 = ld3(&M[8 + 3 * PMC_PT]);
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f; V3 cr;
            if (deepest != 0) { cr = cross(lA - q1, q3 - q2); r0 = dot(cr, cr); }
            if (deepest != 1) { cr = cross(lA - q0, q3 - q2); r1 = dot(cr, cr); }
            if (deepest != 2) { cr = cross(lA - q0, q3 - q1); r2 = dot(cr, cr); }
            if (deepest != 3) { cr = cross(lA - q0, q2 - q1); r3 = dot(cr, cr); }
            sl = 0; float rb = r0;
            if (r1 > rb) { rb = r1; sl = 1; }
            if (r2 > rb) { rb = r2; sl = 2; }
            if (r3 > rb) { rb = r3; sl = 3; }
          }
        }
        float* P = &M[8 + PMC_PT * sl];
        st3(P, lA); st3(P + 3, lB); st3(P + 6, nr); P[9] = dist; P[10] = __int_as_float(abw & (int)0xFFC0FFFF);
      }
      if (lane < npm) M[1] = __int_as_float(n);
    }
    WSYNC();
    PCLK(23)
    /* (d), (e): one lane per cached POINT (lane = 4 * manifold + slot).  The point is refreshed and judged in its lane; the four lanes of a manifold then
     * replay the sequential removal (slot i dropped: the last point takes it) on the four drop bits - the same arrangement in all of them - and every
     * survivor goes to its new slot of the cache and, if the contact list has room for it, to its record in L.man straight from its registers */
    int cnt = 0;
    {
      const int mi = lane >> 2, q = lane & 3;
      const bool mine = mi < npm;
      const float* M = &C[PMC_HDR + PMC_MAN * (mine ? mi : 0)];
      const int n = mine ? __float_as_int(M[1]) : 0;
      const float thr = M[2];
      const int fl = __float_as_int(M[3]);
      const bool valid = q < n;
      float pt[PMC_PT];
#pragma unroll
      for (int t = 0; t < PMC_PT; t++) pt[t] = M[8 + PMC_PT * q + t];
      const int ab = valid ? __float_as_int(pt[10]) : 0;
      const int ba = (ab >> 22) & 31, bb = (ab >> 27) & 31;
      const V3 nr = mk3(pt[6], pt[7], pt[8]);
      const V3 pA = mulv(ldm3(&L.xR[9 * ba]), mk3(pt[0], pt[1], pt[2])) + ld3(&L.xp[3 * ba]);
      const V3 pB = mulv(ldm3(&L.xR[9 * bb]), mk3(pt[3], pt[4], pt[5])) + ld3(&L.xp[3 * bb]);
      const float dist = dot(pA - pB, nr);
      bool drop = !(dist <= thr);
      if (!drop) { const V3 diff = pB - (pA - nr * dist); drop = dot(diff, diff) > thr * thr; }
      const unsigned d4 = (unsigned)(__ballot(valid && drop) >> (4 * (mi & 15))) & 15u;
      /* the removal loop of the oracle on slot numbers: arr[i] = which original point sits in slot i */
      int a0 = 0, a1 = 1, a2 = 2, a3 = 3, nn = n;
      for (int i = n - 1; i >= 0; i--) {
        const int at = i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3));
        if ((d4 >> at) & 1u) {
          const int last = nn - 1 == 0 ? a0 : (nn - 1 == 1 ? a1 : (nn - 1 == 2 ? a2 : a3));
          if (i == 0) a0 = last; else if (i == 1) a1 = last; else if (i == 2) a2 = last; else a3 = last;
          nn--;
        }
      }
      int slot = -1;                                        /* this point's slot after the removals, -1 = dropped */
      if (valid && !((d4 >> q) & 1u)) slot = (nn > 0 && a0 == q) ? 0 : ((nn > 1 && a1 == q) ? 1 : ((nn > 2 && a2 == q) ? 2 : ((nn > 3 && a3 == q) ? 3 : -1)));
      /* the rotation-locked body against the static world: its deepest point alone (first of equals within K_TIE_EPS, in slot order) */
      const bool single = (fl & 1) != 0;
      bool emit = slot >= 0;
      if (single && nn > 0) {
        const int base = lane & ~3;
        const int s0 = __shfl(slot, base), s1 = __shfl(slot, base + 1), s2 = __shfl(slot, base + 2), s3 = __shfl(slot, base + 3);
        const float e0 = __shfl(dist, base), e1 = __shfl(dist, base + 1), e2 = __shfl(dist, base + 2), e3 = __shfl(dist, base + 3);
        float ds[4] = {0.f, 0.f, 0.f, 0.f}; int who[4] = {0, 0, 0, 0};      /* distance and owner lane of the point in slot k */
        if (s0 >= 0) { ds[s0] = e0; who[s0] = 0; }
        if (s1 >= 0) { ds[s1] = e1; who[s1] = 1; }
        if (s2 >= 0) { ds[s2] = e2; who[s2] = 2; }
        if (s3 >= 0) { ds[s3] = e3; who[s3] = 3; }
        int only = 0;
        for (int k = 1; k < nn; k++) if (ds[k] < ds[only] - K_TIE_EPS) only = k;
        emit = slot >= 0 && who[only] == q;
      }
      cnt = (q == 0 && mine) ? (single ? (nn > 0 ? 1 : 0) : nn) : 0;
      int off = 0;
#pragma unroll
      for (int bit = 0; bit < 3; bit++) off += __popcll(__ballot((cnt >> bit) & 1) & lower) << bit;
      off = __shfl(off, lane & ~3);                          /* the manifold's first record */
      const int room = max(0, MAXC - off);
      if (q == 0 && mine) { kept = min(cnt, room); man = &L.man[8 * (off < MANPTS ? off : 0)]; pk = fl << 16; }
      WSYNC();                                               /* every lane holds its point: the cache slots may be overwritten */
      if (slot >= 0) {
        float* P = &C[PMC_HDR + PMC_MAN * mi + 8 + PMC_PT * slot];
#pragma unroll
        for (int t = 0; t < 9; t++) P[t] = pt[t];
        P[9] = dist; P[10] = pt[10];
      }
      if (q == 0 && mine) C[PMC_HDR + PMC_MAN * mi + 1] = __int_as_float(nn);
      const int ri = single ? 0 : slot;
      if (emit && ri < room) {
        const V3 pm = (pA + pB) * 0.5f;
        float* r = &L.man[8 * (off + ri)];
        r[0] = pm.x; r[1] = pm.y; r[2] = pm.z; r[3] = nr.x; r[4] = nr.y; r[5] = nr.z; r[6] = dist; r[7] = __int_as_float(ab);
      }
    }
    WSYNC();
    PCLK(24)
    if (lane == 0) C[0] = __int_as_float(npm);
    WSYNC();
    for (int i = lane * 4; i < PMC_HDR + PMC_MAN * npm; i += 256) *(float4*)&g[i] = *(const float4*)&C[i];      /* (what lies behind the last manifold in memory is never read) */
    PCLK(25)
  } else {
  int mycnt = 0, run_end = lane;
  bool single = false;
  if (lane < nact) {
    const int mykey = L.key[lane];
    const bool head = lane == 0 || L.key[lane - 1] != mykey;
    if (head) {
      int sum = 0;
      for (run_end = lane; run_end < nact && L.key[run_end] == mykey; run_end++) sum += L.candn[run_end] & 255;
      single = (mykey & 65536) != 0;
      mycnt = min(sum, single ? 1 : 4);
    }
  }
  int off = 0;
#pragma unroll
  for (int bit = 0; bit < 3; bit++) off += __popcll(__ballot((mycnt >> bit) & 1) & lower) << bit;
  kept = min(mycnt, max(0, MAXC - off));      /* (the AABBs / narrowphase scratch that L.man may lie over are dead since the last barrier) */
  man = &L.man[8 * (off < MANPTS ? off : 0)];
  pk = lane < nact ? L.key[lane] : 0;
  if (kept > 0) {
    int cnt = 0;
    for (int j = lane; j < run_end; j++) {
      const int cn = L.candn[j];
      for (int i = 0; i < (cn & 255); i++) {
        const float* c = &L.cand[((cn >> 8) + i) * 8];
        int dst;
        if (single) {
          if (cnt == 0) dst = cnt++;
          else dst = c[6] < man[6] - K_TIE_EPS ? 0 : -1;
        } else if (cnt < 4) dst = cnt++;
        else dst = manifold_replace_index(man, c);
        if (dst >= 0) { const float4 c0 = *(const float4*)c, c1 = *(const float4*)(c + 4); *(float4*)&man[8 * dst] = c0; *(float4*)&man[8 * dst + 4] = c1; }
      }
    }
  }
  WSYNC();
  }
  /* Contacts leave in solver order (stable partition of the manifold order, the oracle's collide() explains it): key 2 * (touches both
   * halves of the velocity layout) + (arm link against a movable body).  A lane's points all belong to one object pair, hence to one
   * key; exclusive prefixes of the per-lane counts (0..4) per key come from ballots. */
  int cls = 0, key = 0;
  if (kept > 0) { cls = (pk >> 20) & 3; key = 2 * (cls == 2 ? 1 : 0) + ((pk >> 22) & 1); }      /* (narrowphase_coop wrote the pair's classes into its key) */
  int before = 0, total = 0;          /* points of smaller keys + points of my key in earlier lanes */
  {
    unsigned long long mk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) mk[k] = __ballot(kept > 0 && key == k);
#pragma unroll
    for (int bit = 0; bit < 3; bit++) {
      const unsigned long long mb = __ballot((kept >> bit) & 1);
      total += __popcll(mb) << bit;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int all = __popcll(mb & mk[k]) << bit, low = __popcll(mb & mk[k] & lower) << bit;
        before += k < key ? all : (k == key ? low : 0);
      }
    }
  }
  for (int i = 0; i < kept; i++) {
    const float4 c0 = *(const float4*)&man[8 * i], c1 = *(const float4*)&man[8 * i + 4];
    const int ab = __float_as_int(c1.w);              /* the colliders may differ from point to point inside a manifold */
    const int o = before + i;
    L.conp[3 * o] = c0.x; L.conp[3 * o + 1] = c0.y; L.conp[3 * o + 2] = c0.z;
    L.conn[3 * o] = c0.w; L.conn[3 * o + 1] = c1.x; L.conn[3 * o + 2] = c1.y;
    L.cond[o] = c1.z;
    L.cona[o] = ab & 255; L.conb[o] = (ab >> 8) & 255; L.conk[o] = cls;
    L.conmu[o] = m->persist ? m->col_friction[ab & 255] * m->col_friction[(ab >> 8) & 255] : L.pmu[(ab >> 16) & 63];      /* (a cached point's pair may not be active now: same product from the table) */
  }
  WSYNC();
  PCLK(10)
  return total;
}

/* ------------------------------------------------------------------ arm dynamics: CRBA mass matrix, RNEA bias, inverse */
template <class LDS>
__device__ __forceinline__ void arm_dynamics(const DevModel* m, LDS& L, int lane) {
  int n = m->n_arm;
  V3 O = ld3(L.O);
  if (lane < n) {      /* own spatial inertia about O: (m, h = m c, Ibar = R Ic R^T - m [c]x^2) */
    M3 R = ldm3(&L.xR[9 * (1 + lane)]);
    V3 c = ld3(&L.xp[3 * (1 + lane)]) + mulv(R, ld3(m->arm_com[lane])) - O;
    float mass = m->arm_mass[lane];
    M3 Ic = ldm3(m->arm_inertia[lane]);
    M3 Rt; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt.m[3 * i + j] = R.m[3 * j + i];
    M3 Iw = mul(mul(R, Ic), Rt);
    float cc = dot(c, c);
    float* I = &L.inert[10 * lane];
    I[0] = mass; st3(I + 1, c * mass);
    I[4] = Iw.m[0] + mass * (cc - c.x * c.x); I[5] = Iw.m[4] + mass * (cc - c.y * c.y); I[6] = Iw.m[8] + mass * (cc - c.z * c.z);
    I[7] = Iw.m[1] - mass * c.x * c.y; I[8] = Iw.m[2] - mass * c.x * c.z; I[9] = Iw.m[5] - mass * c.y * c.z;
  }
  WSYNC();
  if (lane < n) {      /* composite inertia of the subtree; spatial velocity from the ancestors */
    uint32_t sub = m->arm_sub[lane], anc = m->arm_anc[lane];
    float acc[10];
    for (int k = 0; k < 10; k++) acc[k] = 0.f;
    V6 v = zero6();
    for (int j = 0; j < n; j++) {           /* wave-uniform j: LDS broadcasts, read unconditionally and selected (x + 0 is exact) */
      const bool ins = (sub >> j) & 1u, isanc = (anc >> j) & 1u;
      for (int k = 0; k < 10; k++) { float t = L.inert[10 * j + k]; acc[k] += ins ? t : 0.f; }
      V6 sj = ld6(&L.S[6 * j]) * L.st[ST_QD + j];
      v.a = v.a + (isanc ? sj.a : mk3(0, 0, 0)); v.l = v.l + (isanc ? sj.l : mk3(0, 0, 0));
    }
    for (int k = 0; k < 10; k++) L.compI[10 * lane + k] = acc[k];
    st6(&L.vsp[6 * lane], v);
    V6 Si = ld6(&L.S[6 * lane]);
    st6(&L.csp[6 * lane], crm(v, Si * L.st[ST_QD + lane]));
    st6(&L.Fv[6 * lane], inertia_mul(acc, Si));
  }
  WSYNC();
  PCLK(11)
  for (int e = lane; e < n * n; e += 64) {   /* M_ij = S_i . (Ic_j S_j) for i an ancestor-or-self of j, accumulated in fp64 */
    int i = e / n, j = e % n;
    if (i <= j) {
      double val = 0.0;
      if ((m->arm_anc[j] >> i) & 1u) {
        const float* a = &L.S[6 * i];
        const float* b = &L.Fv[6 * j];
        for (int k = 0; k < 6; k++) val += (double)a[k] * (double)b[k];
      }
      L.Md[i * 12 + j] = val; L.Md[j * 12 + i] = val;
    }
  }
  if (lane < n) {      /* bias force of each body: f = I a_bias + v x* (I v), a_bias = -g + sum of ancestors' c */
    uint32_t anc = m->arm_anc[lane];
    V6 a = zero6();
    a.l.z = -K_GRAVITY;
    for (int j = 0; j < n; j++) { V6 cj = ld6(&L.csp[6 * j]); bool on = (anc >> j) & 1u; a.a = a.a + (on ? cj.a : mk3(0, 0, 0)); a.l = a.l + (on ? cj.l : mk3(0, 0, 0)); }
    V6 v = ld6(&L.vsp[6 * lane]);
    const float* I = &L.inert[10 * lane];
    st6(&L.fsp[6 * lane], inertia_mul(I, a) + crf(v, inertia_mul(I, v)));
  }
  WSYNC();
  if (lane < n) {
    uint32_t sub = m->arm_sub[lane];
    V6 f = zero6();
    for (int j = 0; j < n; j++) { V6 fj = ld6(&L.fsp[6 * j]); bool on = (sub >> j) & 1u; f.a = f.a + (on ? fj.a : mk3(0, 0, 0)); f.l = f.l + (on ? fj.l : mk3(0, 0, 0)); }
    L.tau[lane] = dot6(ld6(&L.S[6 * lane]), f);
  }
  PCLK(13)
  /* Cholesky M = L L^T and M^-1 = L^-T L^-1 in fp64, entirely in registers: lane i holds row i of M (then of L), the
   * entries of other rows arrive by v_readlane (k, j are compile-time, so every register index is static) - no LDS
   * traffic and no barriers inside the 12-step elimination.  Pivots enter as reciprocals (1/sqrt once per column). */
  WSYNC();
  {
    const int li = lane < RP_MAX_ARM ? lane : 0;
    double row[RP_MAX_ARM], invd[RP_MAX_ARM];
#pragma unroll
    for (int j = 0; j < RP_MAX_ARM; j++) row[j] = (lane < n && j < n) ? L.Md[li * 12 + j] : (j == lane ? 1.0 : 0.0);   /* identity padding beyond n */
#pragma unroll
    for (int k = 0; k < RP_MAX_ARM; k++) {
      double mkk = readlane_d(row[k], k);
      double rp = rsqrt(mkk);
      invd[k] = rp;
      double lik = row[k] * rp;                 /* lane k: sqrt(m_kk); lanes > k: L_ik; lanes < k: unused upper part */
      row[k] = lik;
#pragma unroll
      for (int j = k + 1; j < RP_MAX_ARM; j++) row[j] -= lik * readlane_d(lik, j);
    }
    /* column `lane` of M^-1: L y = e_lane, L^T x = y */
    double y[RP_MAX_ARM];
#pragma unroll
    for (int i = 0; i < RP_MAX_ARM; i++) {
      double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; k++) sacc -= readlane_d(row[k], i) * y[k];
      y[i] = sacc * invd[i];
    }
#pragma unroll
    for (int i = RP_MAX_ARM - 1; i >= 0; i--) {
      double sacc = y[i];
#pragma unroll
      for (int k = i + 1; k < RP_MAX_ARM; k++) sacc -= readlane_d(row[i], k) * y[k];
      y[i] = sacc * invd[i];
    }
    if (lane < n) {
#pragma unroll
      for (int i = 0; i < RP_MAX_ARM; i++) if (i < n) L.Minv[i * 12 + lane] = (float)y[i];
    }
  }
  WSYNC();
  PCLK(14)
}

/* unconstrained velocities v* = v + dt * a for every dof (lane = dof) */
template <class LDS>
__device__ __forceinline__ void unconstrained_velocities(const DevModel* m, LDS& L, int lane) {
  int n = m->n_arm;
  float vs = 0.f;
  if (lane < n) {
    float qdd = 0.f;
    for (int k = 0; k < n; k++) qdd -= L.Minv[lane * 12 + k] * L.tau[k];
    vs = L.st[ST_QD + lane] + K_DT * qdd;
  } else if (lane < n + 6 * m->n_free) {
    int k = (lane - n) / 6, c = (lane - n) % 6;
    const float* f = &L.st[ST_FREE + 13 * k];
    if (c < 3) {
      V3 v = ld3(f + 7);
      float vn = norm(v);
      vs = comp(v, c) + K_DT * (-(K_LIN_DAMP + K_LIN_DAMP * vn) * comp(v, c)) + (c == 2 ? K_DT * K_GRAVITY : 0.f);
    } else if (!m->free_rot_locked[k]) {
      M3 R = ldm3(&L.xR[9 * (1 + n + k)]);
      V3 w = ld3(f + 10);
      V3 wl = tmulv(R, w);
      V3 I = ld3(m->free_inertia[k]);
      V3 Iw = mk3(I.x * wl.x, I.y * wl.y, I.z * wl.z);
      V3 g = cross(wl, Iw);
      float wn = norm(wl), kd = K_ANG_DAMP + K_ANG_DAMP * wn;
      V3 al = mk3((-g.x - Iw.x * kd) / I.x, (-g.y - Iw.y * kd) / I.y, (-g.z - Iw.z * kd) / I.z);
      V3 aw = mulv(R, al);
      vs = comp(w, c - 3) + K_DT * comp(aw, c - 3);
    }
  } else if (lane < m->nv) {
    int k = lane - n - 6 * m->n_free;
    float qd = L.st[ST_JQD + k];
    if (m->j1_type[k] == 1) {
      M3 R = ldm3(&L.xR[9 * (1 + n + m->n_free + k)]);
      V3 a = mulv(R, ld3(m->j1_axis[k]));
      vs = qd + K_DT * K_GRAVITY * a.z;
    } else {
      vs = qd + K_DT * (-(K_ANG_DAMP + K_ANG_DAMP * fabsf(qd)) * qd);
    }
  }
  if (lane < 32) L.vstar[lane] = lane < m->nv ? vs : 0.f;
  if (lane < m->n_free) {     /* world inverse inertia of free body `lane` */
    float* o = &L.finv[9 * lane];
    if (m->free_rot_locked[lane]) for (int i = 0; i < 9; i++) o[i] = 0.f;
    else {
      M3 R = ldm3(&L.xR[9 * (1 + n + lane)]);
      V3 I = ld3(m->free_inertia[lane]);
      float ii[3] = {1.f / I.x, 1.f / I.y, 1.f / I.z};
      for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) {
        float v = 0.f;
        for (int t = 0; t < 3; t++) v += R.m[3 * r + t] * R.m[3 * s + t] * ii[t];
        o[3 * r + s] = v;
      }
    }
  }
  WSYNC();
}

/* ------------------------------------------------------------------ constraint rows */
#define SR_UNIT 0   /* J = e_dofA on an arm dof                        (motors) */
#define SR_J1 1     /* J = e_dofA on a scene joint                      (door / button / dial motors) */
#define SR_GEAR 2   /* J = e_dofA + ratio * e_dofB on arm dofs          (Panda finger gear) */
#define SR_LIMIT 3  /* J = sign * e_dofA on an arm dof                 (joint limits; solved like SR_UNIT) */

template <class LDS>
__device__ __forceinline__ void put_srow(LDS& L, int r, int type, int dofA, float sign, float rhs, float dinv, float lo, float hi, int dofB) {
  float* s = &L.srow[8 * r];
  s[0] = __int_as_float(type); s[1] = __int_as_float(dofA); s[2] = sign; s[3] = rhs; s[4] = dinv; s[5] = lo; s[6] = hi;
  s[7] = __int_as_float(dofB);
}

/* The non-contact rows in the order btMultiBodyConstraintSolver walks them (creation order in the world: the scene bodies' joint motors - made before
 * the arm -, the arm's joint-limit constraints - added while the URDF tree is converted -, its motors, the gear; the sweeps walk this list in
 * ALTERNATING direction, solve_rows / k_solve2).  A joint-limit row exists only while the limit is violated and pushes back with erp 0.2
 * (btMultiBodyJointLimitConstraint::createConstraintRows).  Same rule as the oracle's build_rows (RPO_RULE_ORDER | RPO_RULE_LIMIT) and the frozen
 * reference step (rp_bullet_ref.c RPB_ORDER, RPB_LIMIT).  Returns the number of small rows (wave-uniform). */
template <class LDS>
__device__ __forceinline__ int build_small_rows(const DevModel* m, LDS& L, int lane) {
  int n = m->n_arm, nr = 0;
  if (lane < m->n_j1) {  /* scene joint motors */
    int d = dof_j1(m, lane);
    float minv = m->j1_minv[lane], dinv = 1.f / minv;
    float des = m->j1_has_pos_motor[lane] ? K_KP * (m->j1_motor_target[lane] - L.st[ST_JQ + lane]) / K_DT : 0.f;
    float mx = m->j1_motor_maximp[lane];
    put_srow(L, lane, SR_J1, d, minv, (des - L.vstar[d]) * dinv, dinv, -mx, mx, 0);
  }
  nr = m->n_j1;
  {                      /* joint limits, dof-major, lower before upper */
    int i = lane >> 1, side = lane & 1;
    bool on = false; float pen = 0.f;
    if (lane < 2 * n && m->arm_limited[i]) {
      float q = L.st[ST_Q + i];
      pen = side == 0 ? q - m->arm_lower[i] : m->arm_upper[i] - q;
      on = m->spec_limits ? !(pen > K_LIMIT_ACTIVATION) : !(pen > 0.f);      /* default: only while the limit is violated (btMultiBodyJointLimitConstraint); RP_CFG_SPECULATIVE_LIMITS: round 2's speculative rows */
    }
    unsigned long long mask = __ballot(on);
    if (on) {
      int r = nr + __popcll(mask & ((1ull << lane) - 1ull));
      float sgn = side == 0 ? 1.f : -1.f;
      float dinv = 1.f / L.Minv[i * 12 + i];
      float relv = sgn * L.vstar[i], pos_err = 0.f, vel_err = -relv;
      if (pen > 0.f) vel_err -= pen / K_DT; else pos_err = -pen * (m->spec_limits ? K_ERP : K_ERP_LIMIT) / K_DT;      /* (pen > 0 only under the speculative rule: the row lets the joint close the gap within the substep) */
      put_srow(L, r, SR_LIMIT, i, sgn, (pos_err + vel_err) * dinv, dinv, 0.f, K_LIMIT_MAXIMP, 0);
    }
    nr += __popcll(mask);
  }
  if (lane < n) {        /* arm motors (btMultiBodyJointMotor) */
    float dinv = 1.f / L.Minv[lane * 12 + lane];
    float mode = L.st[ST_MMODE + lane];
    float des = mode != 0.f ? K_KP * (L.st[ST_MTARGET + lane] - L.st[ST_Q + lane]) / K_DT : 0.f;
    float mx = L.st[ST_MMAXIMP + lane];
    put_srow(L, nr + lane, SR_UNIT, lane, 1.f, (des - L.vstar[lane]) * dinv, dinv, -mx, mx, 0);
  }
  nr += n;
  if (m->arm_type == RP_ARM_PANDA) {   /* finger gear: qd_a + ratio qd_b -> 0 (environments.py:400-405) */
    if (lane == 0) {
      int a = m->d9p, b = m->d10p;
      float ratio = -1.f;
      float diag = L.Minv[a * 12 + a] + 2.f * ratio * L.Minv[a * 12 + b] + ratio * ratio * L.Minv[b * 12 + b];
      float dinv = safe_inv(diag);
      float relv = L.vstar[a] + ratio * L.vstar[b];
      float pos_err = -(L.st[ST_Q + a] + ratio * L.st[ST_Q + b]) * 0.1f / K_DT;
      put_srow(L, nr, SR_GEAR, a, ratio, (pos_err - relv) * dinv, dinv, -50.f * K_DT, 50.f * K_DT, b);
    }
    nr += 1;
  }
  return nr;
}

/* Torsional friction (URDF spinning_friction of the gripper links; oracle RPO_RULE_SPIN): one row per run of contacts of one collider pair, bounded by
 * (spin_a mu_b + spin_b mu_a) times the normal impulse of the run's first contact, at most MAXT of them in contact order.  Fills L.torc / L.tors, returns
 * their number (wave-uniform).  Contacts of one collider pair are neighbours in the contact list (one manifold, one class). */
template <class LDS>
__device__ __forceinline__ int tors_list(const DevModel* m, LDS& L, int lane, int ncon) {
  bool head = false; float spin = 0.f;
  if (lane < ncon) {
    const int a = L.cona[lane], b = L.conb[lane];
    head = lane == 0 || L.cona[lane - 1] != a || L.conb[lane - 1] != b;
    spin = m->col_spin[a] * m->col_friction[b] + m->col_spin[b] * m->col_friction[a];
  }
  const unsigned long long mk = __ballot(head && spin > 0.f);
  const int t = __popcll(mk & ((1ull << lane) - 1ull));
  if (head && spin > 0.f && t < MAXT) { L.torc[t] = lane; L.tors[t] = spin; }
  const int nt = __popcll(mk);
  return nt < MAXT ? nt : MAXT;
}

/* Contact rows (normals first, then two friction rows per point, btPlaneSpace1 directions).
 * A row touches at most two bodies, so it is stored compactly: slot0 = 12 entries starting at dof off0, slot1 = 6
 * entries starting at dof off1 (an empty slot has off = 64).  If the arm is involved it takes slot0 (off0 = 0).
 * Three register-light passes: (A) lane = row: Jacobian entries, the non-arm part of M^-1 J^T, partial diagonal;
 * (B) lane = (row, i): arm part B_i = sum_k Minv[i][k] J_k; (C) lane = row: diagonal, relative velocity, rhs. */
/* The rows of contacts [c0, c0 + nc) are built at local indices [0, 3 nc): normals first, then the friction pairs.  The one-kernel path
 * builds all contacts at once (c0 = 0, nc = ncon: local index = row number); k_prep2 builds PREP_CH contacts at a time and copies each
 * chunk to its rows of the workspace. */
/* ... and nt torsional rows (tors_list) between the normals and the friction rows, local indices [nc, nc + nt): pure torques about the parent contact's
 * normal.  The one-kernel path builds them with the contacts; k_prep2 as a chunk of their own (nc = 0). */
template <class LDS>
__device__ __forceinline__ void contact_rows(const DevModel* m, LDS& L, int lane, int c0, int nc, int nt) {
  const int n = m->n_arm;
  const int nrows = 3 * nc + nt;
  V3 O = ld3(L.O);
  for (int r = lane; r < nrows; r += 64) {      /* pass A */
    int ci, dir;
    if (r < nc) { ci = c0 + r; dir = 0; }
    else if (r < nc + nt) { ci = L.torc[r - nc]; dir = 3; }
    else { ci = c0 + ((r - nc - nt) >> 1); dir = 1 + ((r - nc - nt) & 1); }
    V3 nrm = ld3(&L.conn[3 * ci]);
    const V3 pmid = ld3(&L.conp[3 * ci]);
    /* a contact acts at its point on A on body A and at its point on B on body B (Bullet's positionWorldOnA / B; oracle RPO_RULE_LEVER): the stored point is
     * their midpoint, they lie cond apart along the normal */
    const V3 half = nrm * (0.5f * L.cond[ci]);
    V3 d = nrm;
    const bool tors = dir == 3;
    if (dir == 1 || dir == 2) {            /* btPlaneSpace1 */
      V3 t1, t2;
      if (fabsf(nrm.z) > 0.7071067811865475244f) {
        float a = nrm.y * nrm.y + nrm.z * nrm.z, k = 1.f / sqrtf(a);
        t1 = mk3(0.f, -nrm.z * k, nrm.y * k);
        t2 = mk3(a * k, -nrm.x * t1.z, nrm.x * t1.y);
      } else {
        float a = nrm.x * nrm.x + nrm.y * nrm.y, k = 1.f / sqrtf(a);
        t1 = mk3(-nrm.y * k, nrm.x * k, 0.f);
        t2 = mk3(-nrm.z * t1.y, nrm.z * t1.x, a * k);
      }
      d = dir == 1 ? t1 : t2;
    }
    float* J = &L.J[r * ROWW];
    float* B = &L.B[r * ROWW];
    for (int k = 0; k < ROWW; k++) { J[k] = 0.f; B[k] = 0.f; }
    bool has_arm = false;
    float diag = 0.f, relv = 0.f;
    int off0 = 64, off1 = 64;
    int bodyA = m->col_body[L.cona[ci]], bodyB = m->col_body[L.conb[ci]];
    bool any_arm = (bodyA >= 1 && bodyA <= n) || (bodyB >= 1 && bodyB <= n);
    int nother = 0;
    /* (round 5) the arm's entries accumulate in registers and every loop over the arm's dofs in this function is unrolled to RP_MAX_ARM with its loads unconditional: as rolled
     * loops with a run-time bound each iteration was a dependent LDS round trip - a read-modify-write of J[k] here, two or three loads per term in passes B and C - and the
     * rows after the join are the tail of every k_prep2 block (10 k cycles at the median, 18 - 33 k in the heavy ones).  Same terms, same order. */
    float Jr[RP_MAX_ARM];
#pragma unroll
    for (int k = 0; k < RP_MAX_ARM; k++) Jr[k] = 0.f;
    for (int side = 0; side < 2; side++) {
      int body = side == 0 ? bodyA : bodyB;
      float sign = side == 0 ? 1.f : -1.f;
      if (body == 0) continue;
      const V3 p = side == 0 ? pmid + half : pmid - half;
      if (body <= n) {
        V6 f; f.a = tors ? d : cross(p - O, d); f.l = tors ? mk3(0, 0, 0) : d;
        uint32_t anc = m->arm_anc[body - 1];
#pragma unroll
        for (int k = 0; k < RP_MAX_ARM; k++) {
          const float v = sign * dot6(ld6(&L.S[6 * k]), f);
          Jr[k] = (k < n && ((anc >> k) & 1u)) ? Jr[k] + v : Jr[k];
        }
        has_arm = true;
        continue;
      }
      /* non-arm body: goes to slot1 if the arm holds slot0 or if it is the second such body, else to slot0 */
      int base = (any_arm || nother == 1) ? 12 : 0;
      nother++;
      if (body <= n + m->n_free) {
        int k = body - 1 - n, dd = dof_free(m, k);
        V3 rr = p - ld3(&L.st[ST_FREE + 13 * k]);
        V3 rxn = tors ? d : cross(rr, d);
        float im = 1.f / m->free_mass[k];
        M3 Ii = ldm3(&L.finv[9 * k]);
        V3 w = mulv(Ii, rxn);
        const V3 dl = tors ? mk3(0, 0, 0) : d;
        float jl[3] = {sign * dl.x, sign * dl.y, sign * dl.z}, ja[3] = {sign * rxn.x, sign * rxn.y, sign * rxn.z};
        float ba[3] = {sign * w.x, sign * w.y, sign * w.z};
        for (int i = 0; i < 3; i++) {
          J[base + i] = jl[i]; B[base + i] = jl[i] * im; J[base + 3 + i] = ja[i]; B[base + 3 + i] = ba[i];
          diag += jl[i] * jl[i] * im + ja[i] * ba[i];
          relv += jl[i] * L.vstar[dd + i] + ja[i] * L.vstar[dd + 3 + i];
        }
        if (base == 0) off0 = dd; else off1 = dd;
      } else {
        int k = body - 1 - n - m->n_free, dd = dof_j1(m, k);
        M3 R = ldm3(&L.xR[9 * body]);
        V3 a = mulv(R, ld3(m->j1_axis[k]));
        float j = tors ? (m->j1_type[k] == 1 ? 0.f : sign * dot(a, d)) : (m->j1_type[k] == 1 ? sign * dot(d, a) : sign * dot(a, cross(p - ld3(m->j1_pos[k]), d)));
        float minv = m->j1_minv[k];
        J[base] = j; B[base] = j * minv;
        diag += j * j * minv; relv += j * L.vstar[dd];
        if (base == 0) off0 = dd; else off1 = dd;
      }
    }
    if (has_arm) {
      off0 = 0;
#pragma unroll
      for (int k = 0; k < RP_MAX_ARM; k++) if (k < n) J[k] = Jr[k];
    }
    float* s = &L.rowS[4 * r];
    s[0] = diag; s[1] = relv; s[2] = dir == 0 ? 0.f : (tors ? L.tors[r - nc] : L.conmu[ci]); s[3] = __int_as_float(dir == 0 ? 0 : ci);
    float* t = &L.rowT[4 * r];
    t[0] = __int_as_float(has_arm ? 1 : 0); t[1] = dir == 0 ? 1e10f : 0.f; t[2] = __int_as_float(off0); t[3] = __int_as_float(off1);
  }
  WSYNC();
  for (int e = lane; e < nrows * n; e += 64) {  /* pass B */
    int r = e / n, i = e - r * n;
    if (__float_as_int(L.rowT[4 * r]) != 0) {
      const float* J = &L.J[r * ROWW];
      const float* Mi = &L.Minv[i * 12];
      float b = 0.f;
#pragma unroll
      for (int k = 0; k < RP_MAX_ARM; k++) { const float t = Mi[k] * J[k]; b = k < n ? b + t : b; }
      L.B[r * ROWW + i] = b;
    }
  }
  WSYNC();
  for (int r = lane; r < nrows; r += 64) {      /* pass C */
    float* s = &L.rowS[4 * r];
    float* t = &L.rowT[4 * r];
    float diag = s[0], relv = s[1];
    if (__float_as_int(t[0]) != 0) {
      const float* J = &L.J[r * ROWW];
      const float* B = &L.B[r * ROWW];
#pragma unroll
      for (int i = 0; i < RP_MAX_ARM; i++) { const float jb = J[i] * B[i], jv = J[i] * L.vstar[i]; diag = i < n ? diag + jb : diag; relv = i < n ? relv + jv : relv; }
    }
    float dinv, rhs, cfmr = 0.f;
    if (r < nc) {
      const int ci = c0 + r;
      /* <contact> stiffness / damping of either link (the gripper links) make the normal row soft: cfm and erp as in
       * setupMultiBodyContactConstraint (BT_CONTACT_FLAG_CONTACT_STIFFNESS_DAMPING); same arithmetic as the oracle's build_rows */
      float cfm = 0.f, erp = K_ERP;
      float s0 = m->col_stiff[L.cona[ci]], s1 = m->col_stiff[L.conb[ci]];
      if (s0 > 0.f || s1 > 0.f) {
        float d0 = s0 > 0.f ? m->col_damp[L.cona[ci]] : 0.1f, d1 = s1 > 0.f ? m->col_damp[L.conb[ci]] : 0.1f;
        if (!(s0 > 0.f)) s0 = 1e18f;
        if (!(s1 > 0.f)) s1 = 1e18f;
        float ks = 1.f / (1.f / s0 + 1.f / s1), kd = d0 + d1;
        cfm = 1.f / (K_DT * (K_DT * ks + kd));
        erp = (K_DT * ks) / (K_DT * ks + kd);
      }
      dinv = safe_inv(diag + cfm);
      cfmr = cfm * dinv;
      float pen = L.cond[ci] + K_SLOP, pos_err = 0.f, vel_err = -relv;
      if (pen > 0.f) vel_err -= pen / K_DT; else pos_err = -pen * erp / K_DT;
      rhs = (pos_err + vel_err) * dinv;
    } else { dinv = safe_inv(diag); rhs = -relv * dinv; }
    s[0] = rhs; s[1] = cfmr;          /* [1]: the row's softness cfm * dinv (dinv itself is folded into J below) */
    t[0] = __int_as_float((__float_as_int(t[0]) != 0 && __float_as_int(t[3]) != 64) ? 1 : 0);   /* row spans arm and non-arm dofs */
    float* Jw = &L.J[r * ROWW];          /* fold dinv into the stored row: the sweeps use Jd = J * dinv */
    for (int k = 0; k < ROWW; k++) Jw[k] *= dinv;
  }
}

/* 50 sweeps of sequential impulses; lane l owns dv[l]; returns dv of this lane.
 * Everything per-row is wave-uniform: loop bounds and indices are forced into SGPRs, the clamp is one v_med3,
 * accumulated impulses live in
 * registers one row per lane (v_readlane to fetch, lane-select to store), and the LDS data of row r+1 is fetched
 * while row r's dependent chain (multiply -> DPP reduction -> clamp -> axpy) runs.
 * Friction limits are lo = lo_c - mu*lambda[parent], hi = hi_c + mu*lambda[parent]; normal rows carry mu = 0,
 * lo_c = 0, hi_c = 1e10 and a dummy parent, which reproduces [0, 1e10] exactly without a branch. */
__device__ __forceinline__ float pgs_update(float delta, float lam, float lo, float hi, float& lam_out) {
  /* delta = rhs - (J * dinv) . dv (dinv folded into the stored row): the unclamped step.  Delta form, as Bullet's row
   * solver: the step is clamped to [lo - lam, hi - lam], then lam += d.  Rounding contract shared with k_solve2: the
   * products J_i dv_i are rounded on their own and summed (DPP butterfly; a unit row has one product), and every row
   * ends with the fused dv = fma(B, d, dv). */
  float d = __builtin_amdgcn_fmed3f(delta, lo - lam, hi - lam);
  lam_out = lam + d;
  return d;
}

template <class LDS>
__device__ float solve_rows(const DevModel* m, LDS& L, int lane, int nsmall_, int ncon_, int nt_) {
  const int n = m->n_arm;
  const int nsmall = uni(nsmall_), nrc = uni(3 * ncon_ + nt_);      /* contact rows: normals, torsional rows, friction pairs (contact_rows) */
  float dv = 0.f;
  float lamS = 0.f, lamC0 = 0.f, lamC1 = 0.f;
  for (int it = 0; it < K_NITER; it++) {
    {   /* scene-joint motors, limits, motors, gear (build_small_rows explains the order), walked in alternating direction: J has one or two
         * unit entries, B is a (combination of) column(s) of M^-1 */
      for (int rr = 0; rr < nsmall; rr++) {
        const int r = (it & 1) ? rr : nsmall - 1 - rr;      /* backwards in the even sweeps (the first one), forwards in the odd ones */
        const float4 c0 = *(const float4*)&L.srow[8 * r], c1 = *(const float4*)&L.srow[8 * r + 4];
        const int type_ = uni(__float_as_int(c0.x)), dA = uni(__float_as_int(c0.y)), dB = uni(__float_as_int(c1.w));
        const int type = type_ == SR_LIMIT ? SR_UNIT : type_;
        const float sg = c0.z, jA = c1.x;      /* c1.x = dinv = the folded J entry at dofA */
        float bl = 0.f;
        if (type == SR_J1) bl = lane == lane_pos(m, dA) ? sg : 0.f;
        else if (lane < n) bl = type == SR_UNIT ? sg * L.Minv[lane * 12 + dA] : L.Minv[lane * 12 + dA] + sg * L.Minv[lane * 12 + dB];
        float delta;         /* rounding contract shared with k_solve2: a unit row is one fma, the gear row two products rounded on their own, summed, subtracted */
        if (type == SR_UNIT) delta = __fmaf_rn(-(sg * jA), lane_read(dv, dA), c0.w);
        else if (type == SR_J1) delta = __fmaf_rn(-jA, lane_read(dv, uni(lane_pos(m, dA))), c0.w);
        else { float pa = jA * lane_read(dv, dA); float pb = (sg * jA) * lane_read(dv, dB); float jdv = pa + pb; delta = c0.w - jdv; }
        float lam = lane_read(lamS, r), lnew;
        const float d = pgs_update(delta, lam, c1.y, c1.z, lnew);
        lamS = lane == r ? lnew : lamS;
        dv = fmaf(bl, d, dv);
      }
    }
    if (nrc > 0) {   /* contact normals then frictions: compact rows, scalars two rows ahead, J/B one row ahead */
      float4 sn = *(const float4*)&L.rowS[0], tn = *(const float4*)&L.rowT[0];
      float4 s2 = sn, t2 = tn;
      if (nrc > 1) { s2 = *(const float4*)&L.rowS[4]; t2 = *(const float4*)&L.rowT[4]; }
      float jn = 0.f, bn = 0.f;
      const int mydof = lane_dof(m, lane);
      {
        int i1 = mydof - __float_as_int(tn.w), i0 = mydof - __float_as_int(tn.z);
        int idx = (unsigned)i1 < 6u ? 12 + i1 : ((unsigned)i0 < 12u ? i0 : -1);
        if (idx >= 0 && mydof >= 0) { jn = L.J[idx]; bn = L.B[idx]; }
      }
      for (int r = 0; r < nrc; r++) {
        float jl = jn, bl = bn;
        float rhs = sn.x, cfmr = sn.y, mu = sn.z, lo_c = 0.f, hi_c = tn.y;      /* contact rows: lower bound 0 (tn.x carries a flag) */
        int parent = uni(__float_as_int(sn.w));
        sn = s2; tn = t2;
        if (r + 1 < nrc) {
          int i1 = mydof - __float_as_int(tn.w), i0 = mydof - __float_as_int(tn.z);
          int idx = (unsigned)i1 < 6u ? 12 + i1 : ((unsigned)i0 < 12u ? i0 : -1);
          jn = 0.f; bn = 0.f;
          if (idx >= 0 && mydof >= 0) { jn = L.J[(r + 1) * ROWW + idx]; bn = L.B[(r + 1) * ROWW + idx]; }
          if (r + 2 < nrc) { s2 = *(const float4*)&L.rowS[4 * (r + 2)]; t2 = *(const float4*)&L.rowT[4 * (r + 2)]; }
        }
        float lamv = r < 64 ? lamC0 : lamC1;
        float lam = lane_read(lamv, r & 63);
        const float tot = lane_read(lamC0, parent);             /* parent < ncon <= MAXC */
        float lim = mu * tot;
        float prod = jl * dv;
        float jdv = wave_sum32(prod), lnew;
        float d = pgs_update(__fmaf_rn(-lam, cfmr, rhs) - jdv, lam, lo_c - lim, hi_c + lim, lnew);
        if (r >= uni(ncon_) && !(tot > 0.f)) { d = 0.f; lnew = lam; }      /* a friction (or torsional) row is skipped while its normal impulse is not positive */
        float sel = lane == (r & 63) ? lnew : lamv;
        lamC0 = r < 64 ? sel : lamC0;
        lamC1 = r < 64 ? lamC1 : sel;
        dv = fmaf(bl, d, dv);
      }
    }
  }
  return dv;
}

/* ------------------------------------------------------------------ one stepSimulation() */
/* (defined behind heavy_solve) a heavy env's solve + integration in the residual form, the very code k_solve2 runs: true if this env is one */
__device__ bool substep_heavy(const DevModel* m, EnvLds& L, int lane, int nsmall, int ncon, int nt);
__device__ void substep(const DevModel* m, EnvLds& L, int lane, int env) {
  int n = m->n_arm;
  fk_bodies(m, L, lane);
  __syncthreads();
  joint_subspaces(m, L, lane);
  collider_aabbs(m, L, lane);
  __syncthreads();
  int ncon = collide(m, L, lane, env);
  arm_dynamics(m, L, lane);
  unconstrained_velocities(m, L, lane);
  int nsmall = build_small_rows(m, L, lane);
  const int nt = tors_list(m, L, lane, ncon);
  WSYNC();
  contact_rows(m, L, lane, 0, ncon, nt);
  __syncthreads();
  if (substep_heavy(m, L, lane, nsmall, ncon, nt)) { __syncthreads(); return; }      /* (block-uniform: one wave per block) */
  float dv = solve_rows<EnvLds>(m, L, lane, nsmall, ncon, nt);
  __syncthreads();
  /* apply and integrate (semi-implicit Euler) */
  const int dd = lane_dof(m, lane);
  float vnew = clampf((dd >= 0 ? L.vstar[dd] : 0.f) + dv, -K_MAXVEL, K_MAXVEL);      /* btMultiBody::m_maxCoordinateVelocity */
  if (dd >= 0 && dd < n) {
    L.st[ST_QD + dd] = vnew;
    L.st[ST_Q + dd] += K_DT * vnew;
  } else if (dd >= 0 && dd < n + 6 * m->n_free) {
    int k = (dd - n) / 6, c = (dd - n) % 6;
    L.st[ST_FREE + 13 * k + 7 + c] = vnew;
  } else if (dd >= 0) {
    int k = dd - n - 6 * m->n_free;
    L.st[ST_JQD + k] = vnew;
    L.st[ST_JQ + k] += K_DT * vnew;
  }
  __syncthreads();
  if (lane < m->n_free) {
    float* f = &L.st[ST_FREE + 13 * lane];
    V3 v = ld3(f + 7), w = ld3(f + 10);
    st3(f, ld3(f) + v * K_DT);
    float wn = norm(w);
    if (wn > 0.7853981633974483f / K_DT) wn = 0.7853981633974483f / K_DT;
    V3 ax;
    if (wn < 0.001f) ax = w * (0.5f * K_DT - K_DT * K_DT * K_DT * 0.020833333333f * wn * wn);
    else ax = w * (sinf(0.5f * wn * K_DT) / wn);
    Q4 dq = {ax.x, ax.y, ax.z, cosf(0.5f * wn * K_DT)};
    Q4 q0 = {f[3], f[4], f[5], f[6]};
    Q4 qn = qmul(dq, q0);
    float nr = 1.f / sqrtf(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
    f[3] = qn.x * nr; f[4] = qn.y * nr; f[5] = qn.z * nr; f[6] = qn.w * nr;
  }
  __syncthreads();
}

/* ------------------------------------------------------------------ inverse kinematics (wave-uniform, registers only) */
struct ChainQ { float q[7]; };

/* damped-least-squares IK, the oracle's ik_solve restricted to the EE chain (the other dofs decouple exactly), solved
 * COOPERATIVELY BY THE 16 LANES OF A DPP ROW: lane j < NC owns chain joint j.  Per iteration: every lane forms its
 * joint's local transform (one sincos per lane), a 3-round shifted scan composes the chain (row_shr 1, 2, 4), lane j
 * builds Jacobian column j, lane r builds and eliminates row r of J^T J + damp I (columns / pivot rows arrive by row
 * broadcast), the back-substitution walks the pivots once.  ~700 instructions per iteration instead of ~2 700 for one
 * lane per env, and four times as many waves.  NC = chain length (6 UR5, 7 Panda).  All lanes of a row must call it
 * with the same target; qj = this lane's joint value; returns the new one.  `live` = this row has an env at all. */
/* The IK's arithmetic is written with FUSED multiply-adds in a fixed order (the library is compiled with -ffp-contract=off: nothing fuses by itself), and the oracle's
 * ik_solve has the same fma in the same places (rp_oracle.c ik_* helpers): 1 139 instead of 1 346 VALU instructions per iteration, on a phase every chain of a step starts
 * with (round 4 measured the fused IK against the UNFUSED oracle and had to reject it; mirrored, the two round alike). */
__device__ __forceinline__ float dotF(V3 a, V3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 crossF(V3 a, V3 b) { return mk3(__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x))); }
__device__ __forceinline__ V3 mulvF(const M3& a, V3 v) {
  return mk3(__fmaf_rn(a.m[2], v.z, __fmaf_rn(a.m[1], v.y, a.m[0] * v.x)), __fmaf_rn(a.m[5], v.z, __fmaf_rn(a.m[4], v.y, a.m[3] * v.x)), __fmaf_rn(a.m[8], v.z, __fmaf_rn(a.m[7], v.y, a.m[6] * v.x)));
}
__device__ __forceinline__ V3 mulvaddF(const M3& a, V3 v, V3 p) {      /* p + a v */
  return mk3(__fmaf_rn(a.m[2], v.z, __fmaf_rn(a.m[1], v.y, __fmaf_rn(a.m[0], v.x, p.x))), __fmaf_rn(a.m[5], v.z, __fmaf_rn(a.m[4], v.y, __fmaf_rn(a.m[3], v.x, p.y))),
             __fmaf_rn(a.m[8], v.z, __fmaf_rn(a.m[7], v.y, __fmaf_rn(a.m[6], v.x, p.z))));
}
__device__ __forceinline__ M3 mulF(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[3 * i + j] = __fmaf_rn(a.m[3 * i + 2], b.m[6 + j], __fmaf_rn(a.m[3 * i + 1], b.m[3 + j], a.m[3 * i] * b.m[j]));
  return r;
}
__device__ __forceinline__ M3 axis_angleF(V3 a, float q) {
  float s, c;
  sincosf(q, &s, &c);
  const float t = 1.f - c, tx = t * a.x, ty = t * a.y, tz = t * a.z, sx = s * a.x, sy = s * a.y, sz = s * a.z;
  M3 r = {{__fmaf_rn(tx, a.x, c), __fmaf_rn(tx, a.y, -sz), __fmaf_rn(tx, a.z, sy), __fmaf_rn(tx, a.y, sz), __fmaf_rn(ty, a.y, c), __fmaf_rn(ty, a.z, -sx), __fmaf_rn(tx, a.z, -sy),
           __fmaf_rn(ty, a.z, sx), __fmaf_rn(tz, a.z, c)}};
  return r;
}
__device__ __forceinline__ Q4 qmulF(Q4 a, Q4 b) {
  Q4 r = {__fmaf_rn(-a.z, b.y, __fmaf_rn(a.y, b.z, __fmaf_rn(a.x, b.w, a.w * b.x))), __fmaf_rn(a.z, b.x, __fmaf_rn(a.y, b.w, __fmaf_rn(-a.x, b.z, a.w * b.y))),
          __fmaf_rn(a.z, b.w, __fmaf_rn(-a.y, b.x, __fmaf_rn(a.x, b.y, a.w * b.z))), __fmaf_rn(-a.z, b.z, __fmaf_rn(-a.y, b.y, __fmaf_rn(-a.x, b.x, a.w * b.w)))};
  return r;
}
__device__ __forceinline__ Xf xf_compose(const Xf& a, const Xf& b) { Xf r; r.R = mulF(a.R, b.R); r.p = mulvaddF(a.R, b.p, a.p); return r; }
template <int D>
__device__ __forceinline__ Xf xf_shr(const Xf& x) {
  Xf r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.R.m[k] = shr16<D>(x.R.m[k]);
  r.p = mk3(shr16<D>(x.p.x), shr16<D>(x.p.y), shr16<D>(x.p.z));
  return r;
}

/* chain FK by the 16 lanes of a DPP row: local transforms, then an inclusive scan of compositions over lanes 0..NC-1.
 * Out: this lane's joint origin and world axis, and (replicated) the EE site position and orientation. */
template <int NC>
__device__ __forceinline__ void chain_fk_coop(const M3& R0, V3 p0, V3 ax, bool rev, const Xf& base, V3 sp, const M3& sr, float qj, int l16,
                                              V3& org, V3& axw, V3& pos, M3& Rs) {
  const bool isj = l16 < NC;
  Xf x; x.R = ident3(); x.p = mk3(0, 0, 0);
  if (isj) {
    x.R = R0; x.p = p0;
    if (rev) x.R = mulF(R0, axis_angleF(ax, qj)); else { const V3 d = mulvF(R0, ax); x.p = mk3(__fmaf_rn(d.x, qj, p0.x), __fmaf_rn(d.y, qj, p0.y), __fmaf_rn(d.z, qj, p0.z)); }
  }
  if (l16 == 0) x = xf_compose(base, x);
  { Xf y = xf_shr<1>(x); Xf z = xf_compose(y, x); if (l16 >= 1) x = z; }
  { Xf y = xf_shr<2>(x); Xf z = xf_compose(y, x); if (l16 >= 2) x = z; }
  { Xf y = xf_shr<4>(x); Xf z = xf_compose(y, x); if (l16 >= 4) x = z; }
  org = x.p; axw = mulvF(x.R, ax);
  const V3 posl = mulvaddF(x.R, sp, x.p);
  const M3 Rsl = mulF(x.R, sr);
  pos = mk3(bcast16<NC - 1>(posl.x), bcast16<NC - 1>(posl.y), bcast16<NC - 1>(posl.z));
#pragma unroll
  for (int k = 0; k < 9; k++) Rs.m[k] = bcast16<NC - 1>(Rsl.m[k]);
}

/* measured EE link pose (getLinkState(arm, endEffectorIndex)[0], [1]) by the 16 lanes of a DPP row; qj = lane's joint value */
__device__ __forceinline__ void ee_pose_coop(const DevModel* m, float qj, int l16, V3& pos, Q4& orn) {
  const int nc = m->ee_chain;
  const int jj = l16 < nc ? l16 : 0;
  const M3 R0 = ldm3(m->arm_jrot[jj]);
  const V3 p0 = ld3(m->arm_jpos[jj]), ax = ld3(m->arm_axis[jj]);
  const bool rev = m->arm_jtype[jj] == 0;
  Xf base; base.R = ldm3(m->base_rot); base.p = ld3(m->base_pos);
  const V3 sp = ld3(m->site_pos[RP_SITE_EE]);
  const M3 sr = ldm3(m->site_rot[RP_SITE_EE]);
  V3 org, axw; M3 Rs;
  if (nc == 6) chain_fk_coop<6>(R0, p0, ax, rev, base, sp, sr, qj, l16, org, axw, pos, Rs);
  else chain_fk_coop<7>(R0, p0, ax, rev, base, sp, sr, qj, l16, org, axw, pos, Rs);
  orn = m3_to_quat(Rs);
}

/* action types (environments.py:88-113, 915-981).  Layouts: rpy types [x y z r p y grip], quat types [x y z qx qy qz qw grip],
 * joint types [q0..q(nd-1) grip].  The relative pose types add the action to the measured EE pose (orientation
 * componentwise; relative_rpy after getEulerFromQuaternion), the relative joint type to the measured joints. */
#define RP_ACT_ABS_RPY 0
#define RP_ACT_REL_RPY 1
#define RP_ACT_ABS_QUAT 2
#define RP_ACT_REL_QUAT 3
#define RP_ACT_ABS_JOINTS 4
#define RP_ACT_REL_JOINTS 5
__device__ __forceinline__ void load_action(const DevModel* m, const float* action, int env, float* a8) {
  const int at = m->action_type, na = m->n_action, nd = m->n_target;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    float hi = ((at == RP_ACT_ABS_RPY && k < 6) || (at == RP_ACT_ABS_JOINTS && k < nd)) ? 6.f : 1.f;   /* action_space.high, step()'s clip */
    a8[k] = k < na ? clampf(action[(size_t)env * na + k], -hi, hi) : 0.f;
  }
}
__device__ __forceinline__ void action_target(int at, const float* a8, V3 cp, Q4 cq, V3& tpos, Q4& tq) {
  tpos = mk3(a8[0], a8[1], a8[2]);
  if (at == RP_ACT_ABS_RPY) tq = quat_from_euler(a8[3], a8[4], a8[5]);
  else if (at == RP_ACT_ABS_QUAT) { tq.x = a8[3]; tq.y = a8[4]; tq.z = a8[5]; tq.w = a8[6]; }
  else {
    tpos = mk3(a8[0] + cp.x, a8[1] + cp.y, a8[2] + cp.z);
    if (at == RP_ACT_REL_QUAT) { tq.x = a8[3] + cq.x; tq.y = a8[4] + cq.y; tq.z = a8[5] + cq.z; tq.w = a8[6] + cq.w; }
    else {
      V3 ce = euler_from_quat(cq.x, cq.y, cq.z, cq.w);
      tq = quat_from_euler(a8[3] + ce.x, a8[4] + ce.y, a8[5] + ce.z);
    }
  }
}

template <int NC>
__device__ float ik_coop(const DevModel* m, V3 tpos, Q4 tq, float qj, int max_iter, int l16, bool live, bool* capped = nullptr, bool* marginal = nullptr) {
  const bool isj = l16 < NC;
  const int jj = isj ? l16 : 0;
  const M3 R0 = ldm3(m->arm_jrot[jj]);
  const V3 p0 = ld3(m->arm_jpos[jj]), ax = ld3(m->arm_axis[jj]);
  const bool rev = m->arm_jtype[jj] == 0;
  Xf base; base.R = ldm3(m->base_rot); base.p = ld3(m->base_pos);
  const V3 sp = ld3(m->site_pos[RP_SITE_EE]);
  const M3 sr = ldm3(m->site_rot[RP_SITE_EE]);
  bool done = !live;
  for (int it = 0; it < max_iter; it++) {
    V3 org, axw, pos; M3 Rs;
    chain_fk_coop<NC>(R0, p0, ax, rev, base, sp, sr, qj, l16, org, axw, pos, Rs);
    const V3 ep = tpos - pos;
    const float res = sqrtf(dotF(ep, ep));
    /* a stopping test decided within 0.5 % of the threshold (fp32 rounding of a residual of 1e-4 on positions of order 1: ~0.1 %): another evaluation order
     * of the same arithmetic (the CPU oracle's) may stop an iteration earlier or later and end ~5e-5 rad elsewhere (status bit 16 of the step; the parity
     * tests read it) */
    if (marginal && !done && it > 0 && res > 0.995f * K_IK_RES && res < 1.005f * K_IK_RES) *marginal = true;
    done = done || (it > 0 && res < K_IK_RES);
    if (__ballot(!done) == 0ull) break;                        /* every env of the wave has converged */
    /* pose error, by every lane alike */
    Q4 qc = m3_to_quat(Rs);
    Q4 qi = {-qc.x, -qc.y, -qc.z, qc.w};
    Q4 dq = qmulF(tq, qi);
    float vn = sqrtf(__fmaf_rn(dq.z, dq.z, __fmaf_rn(dq.y, dq.y, dq.x * dq.x)));     /* 2 acos(w) in its fp32-safe atan2 form */
    float angle = 2.f * atan2f(vn, dq.w);
    V3 axis = mk3(1, 0, 0);
    if (vn >= 1e-12f) { float sc = 1.f / vn; axis = mk3(dq.x * sc, dq.y * sc, dq.z * sc); }
    if (angle > RP_PI_F) angle -= 2.f * RP_PI_F;
    const float err[6] = {ep.x, ep.y, ep.z, angle * axis.x, angle * axis.y, angle * axis.z};
    /* Jacobian column of this lane's joint */
    float Jc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (isj) {
      V3 lin, ang = mk3(0, 0, 0);
      if (rev) { lin = crossF(axw, pos - org); ang = axw; } else lin = axw;
      Jc[0] = lin.x; Jc[1] = lin.y; Jc[2] = lin.z; Jc[3] = ang.x; Jc[4] = ang.y; Jc[5] = ang.z;
    }
    /* row l16 of A = J^T J + damp I and of b = J^T err */
    float A[NC], b = Jc[0] * err[0];
#pragma unroll
    for (int k = 1; k < 6; k++) b = __fmaf_rn(Jc[k], err[k], b);
    static_for<0, NC>([&](auto cc) {
      constexpr int c = decltype(cc)::v;
      float sacc = Jc[0] * bcast16<c>(Jc[0]);
#pragma unroll
      for (int k = 1; k < 6; k++) sacc = __fmaf_rn(Jc[k], bcast16<c>(Jc[k]), sacc);
      A[c] = sacc + (l16 == c ? K_IK_DAMP : 0.f);
    });
    /* SPD solve without pivoting: lane r eliminates its row against the broadcast pivot row; the pivots' reciprocals
     * serve the back-substitution too */
    float ipiv[NC];
    static_for<0, NC>([&](auto cc) {
      constexpr int c = decltype(cc)::v;
      float pc[NC];
#pragma unroll
      for (int k = c; k < NC; k++) pc[k] = bcast16<c>(A[k]);
      const float pbv = bcast16<c>(b);
      const float inv = 1.f / pc[c];
      ipiv[c] = inv;
      if (l16 > c) {
        float f = A[c] * inv;
#pragma unroll
        for (int k = c; k < NC; k++) A[k] = __fmaf_rn(-f, pc[k], A[k]);
        b = __fmaf_rn(-f, pbv, b);
      }
    });
    float xs[NC];
    static_for<0, NC>([&](auto rr) {
      constexpr int r = NC - 1 - decltype(rr)::v;
      float sacc = b;
#pragma unroll
      for (int k = r + 1; k < NC; k++) sacc = __fmaf_rn(-A[k], xs[k], sacc);
      xs[r] = bcast16<r>(sacc * ipiv[r]);
    });
    float mx = 0.f, mine = 0.f;
#pragma unroll
    for (int j = 0; j < NC; j++) { mx = fmaxf(mx, fabsf(xs[j])); mine = l16 == j ? xs[j] : mine; }
    const float sc = mx > K_IK_MAXSTEP ? K_IK_MAXSTEP / mx : 1.f;
    if (!done && isj) qj = __fmaf_rn(sc, mine, qj);
  }
  if (capped) *capped = !done;      /* the residual test never passed: this call ran out of iterations (status bit 8 of the step) */
  return qj;
}

/* replicated-value front end: every lane passes the same start vector and gets the same solution back */
__device__ __forceinline__ ChainQ ik_solve(const DevModel* m, V3 tpos, Q4 tq, ChainQ q, int max_iter, int l16, bool* capped = nullptr, bool* marginal = nullptr) {
  float qj = 0.f;
#pragma unroll
  for (int j = 0; j < 7; j++) qj = l16 == j ? q.q[j] : qj;
  qj = m->ee_chain == 6 ? ik_coop<6>(m, tpos, tq, qj, max_iter, l16, true, capped, marginal) : ik_coop<7>(m, tpos, tq, qj, max_iter, l16, true, capped, marginal);
  ChainQ r;
  static_for<0, 7>([&](auto jc) { constexpr int j = decltype(jc)::v; r.q[j] = bcast16<j>(qj); });
  return r;
}

/* perform_action('absolute_rpy') .. close_gripper (environments.py:915-1073); every lane computes the same values,
 * lane 0 writes the motor commands into the state record.  Returns the target poses. */
__device__ ChainQ perform_action(const DevModel* m, EnvLds& L, int lane, const float* a8) {
  const int nd = m->n_target, at = m->action_type, l16 = lane & 15;
  bool ik_capped = false, ik_marginal = false;
  ChainQ cur;
#pragma unroll
  for (int j = 0; j < 7; j++) cur.q[j] = j < m->ee_chain ? L.st[ST_Q + j] : 0.f;
  ChainQ sol;
  if (at == RP_ACT_ABS_JOINTS || at == RP_ACT_REL_JOINTS) {
#pragma unroll
    for (int j = 0; j < 7; j++) sol.q[j] = j < nd ? (at == RP_ACT_REL_JOINTS ? a8[j] + L.st[ST_Q + j] : a8[j]) : 0.f;
  } else {
    V3 cp = mk3(0, 0, 0); Q4 cq = {0.f, 0.f, 0.f, 1.f};
    if (at == RP_ACT_REL_RPY || at == RP_ACT_REL_QUAT) {
      float qj = 0.f;
#pragma unroll
      for (int j = 0; j < 7; j++) qj = l16 == j ? cur.q[j] : qj;
      ee_pose_coop(m, qj, l16, cp, cq);
    }
    V3 tpos; Q4 tq;
    action_target(at, a8, cp, cq, tpos, tq);
    if (m->arm_type == RP_ARM_PANDA) sol = ik_solve(m, tpos, tq, cur, 200, l16, &ik_capped, &ik_marginal);
    else {   /* InverseKinematicsSolver.calc_angles: 4 chained default IK solves from the measured joints */
      sol = cur;
      for (int rep = 0; rep < 4; rep++) sol = ik_solve(m, tpos, tq, sol, 20, l16, &ik_capped, &ik_marginal);
    }
  }
  ChainQ tp;
#pragma unroll
  for (int j = 0; j < 7; j++) {
    tp.q[j] = 0.f;
    if (j < nd) {
      float t = clampf(sol.q[j], m->ll[j], m->ul[j]);
      float c = L.st[ST_Q + j];
      tp.q[j] = clampf(t, c - m->inc[j], c + m->inc[j]);
    }
  }
  __syncthreads();
  if (lane == 0) {
    L.st[ST_STATUS] = __int_as_float((ik_capped ? 8 : 0) | (ik_marginal ? 16 : 0));
    for (int j = 0; j < nd; j++) { L.st[ST_MMODE + j] = 1.f; L.st[ST_MTARGET + j] = tp.q[j]; L.st[ST_MMAXIMP + j] = 240.f * K_DT; }
    float g = a8[m->n_action - 1];
    if (m->arm_type == RP_ARM_PANDA) {
      float amt = 0.04f - g / 25.f;
      int ds[2] = {m->d9p, m->d10p};
      for (int i = 0; i < 2; i++) { L.st[ST_MMODE + ds[i]] = 1.f; L.st[ST_MTARGET + ds[i]] = amt; L.st[ST_MMAXIMP + ds[i]] = 100.f * K_DT; }
    } else {
      float amt = g - 0.2f;
      float left = L.st[ST_Q + m->d18];
      int ds[6] = {m->d18, m->d20, m->d12, m->d15, m->d10, m->d13};
      float tg[6] = {amt * 0.055f, left, amt * 0.5f, amt * 0.5f, amt * 0.8f, amt * 0.8f};
      float fo[6] = {100.f, 1000.f, 100.f, 100.f, 100.f, 100.f};
      for (int i = 0; i < 6; i++) { L.st[ST_MMODE + ds[i]] = 1.f; L.st[ST_MTARGET + ds[i]] = tg[i]; L.st[ST_MMAXIMP + ds[i]] = fo[i] * K_DT; }
    }
  }
  __syncthreads();
  return tp;
}

/* ------------------------------------------------------------------ observation / reward */
__device__ __forceinline__ float dial01(float x) { float mod = x - 2.f * floorf(x * 0.5f); return (mod * RP_PI_F) / (2.2f * RP_PI_F); }

template <class LDS>
__device__ __forceinline__ V3 site_pos_world(const DevModel* m, const LDS& L, int s) {
  int b = m->site_body[s];
  return ld3(&L.xp[3 * b]) + mulv(ldm3(&L.xR[9 * b]), ld3(m->site_pos[s]));
}

__device__ float success_func(const float* ag, const float* g) {
  for (int k = 0; k < 3; k++) if (fabsf(g[k] - ag[k]) > 0.05f) return -1.f;
  V3 eg = euler_from_quat(g[3], g[4], g[5], g[6]), ea = euler_from_quat(ag[3], ag[4], ag[5], ag[6]);
  if (fabsf(eg.x - ea.x) > RP_PI_F / 4 || fabsf(eg.y - ea.y) > RP_PI_F / 4 || fabsf(eg.z - ea.z) > RP_PI_F / 4) return -1.f;
  if (fabsf(g[7] - ag[7]) > 0.025f) return -1.f;
  if (fabsf(g[8] - ag[8]) > 0.04f) return -1.f;
  if (fabsf(g[9] - ag[9]) > 0.01f) return -1.f;
  if (fabsf(g[10] - ag[10]) > 0.3f) return -1.f;
  return 0.f;
}

/* compute_reward_sparse (environments.py:278-304) or, with sparse=False, the dense compute_reward = -||ag - dg|| over the whole
 * goal vector (environments.py:169-170, 273-275; calc_target_distance takes the norm of the full difference) */
__device__ float compute_reward(const DevModel* m, const float* ag, const float* dg, bool force_sparse = false) {
  if (m->dense_reward && !force_sparse) {
    float s = 0.f;
    for (int k = 0; k < m->n_ag; k++) { float d = ag[k] - dg[k]; s += d * d; }
    return -sqrtf(s);
  }
  if (m->play) return success_func(ag, dg);
  float dx = ag[0] - dg[0], dy = ag[1] - dg[1], dz = ag[2] - dg[2];
  float dist = sqrtf(dx * dx + dy * dy + dz * dz);
  return dist > m->rew_thresh ? -1.f : -dist;
}

__device__ __forceinline__ void flip_quat(float* v, const float* last) {
  bool all = true;
  for (int i = 0; i < 4; i++) {
    int s = (v[i] > 0.f) - (v[i] < 0.f), l = (last[i] > 0.f) - (last[i] < 0.f);
    if (s != -l) all = false;
  }
  if (all) for (int i = 0; i < 4; i++) v[i] = -v[i];
}

/* out layout in L.out: obs_quat @0 (19) | ag @19 (11) | dg @30 (11) | cag @41 (4) | fps @45 (19) | joints @64 (8) |
 * velocity @72 (6) | observation @78 (18) | proprio @96 | reward @97 | is_success @98 | status @99 */
#ifndef RP_WIDE
#define O_OBS 0
#define O_AG 19
#define O_DG 30
#define O_CAG 41
#define O_FPS 45
#define O_JOINTS 64
#define O_VEL 72
#define O_OBSV 78
#define O_PROP 96
#define O_REW 97
#define O_SUCC 98
#define O_STATUS 99
#define OBS_MAX 19
#define AG_MAX 11
#else      /* two objects: obs_quat 26 | ag 18 | dg 18 | cag 4 | fps 26 | joints 8 | velocity 6 | observation 25 | flags */
#define O_OBS 0
#define O_AG 26
#define O_DG 44
#define O_CAG 62
#define O_FPS 66
#define O_JOINTS 92
#define O_VEL 100
#define O_OBSV 106
#define O_PROP 131
#define O_REW 132
#define O_SUCC 133
#define O_STATUS 134
#define OBS_MAX 26
#define AG_MAX 18
#endif

/* calc_state (environments.py:799-864): transforms must be current (fk_bodies). Stateful in play mode. */
template <class LDS>
__device__ void calc_state(const DevModel* m, LDS& L, int lane) {
  fk_bodies(m, L, lane);
  __syncthreads();
  joint_subspaces(m, L, lane);
  __syncthreads();
  /* gripper_proprioception ray (environments.py:720-743): lane c tests collider c */
  int prop = -1;
  if (m->arm_type != RP_ARM_PANDA) {
    V3 g1 = site_pos_world(m, L, RP_SITE_PADL), g2 = site_pos_world(m, L, RP_SITE_PADR);
    V3 ee = site_pos_world(m, L, RP_SITE_EE), wr = site_pos_world(m, L, RP_SITE_WRIST);
    V3 p1 = ee - (ee - wr) * 0.5f, p2 = (g1 + g2) * 0.5f + (ee - wr) * 0.2f, d = p2 - p1;
    float t = 2.f; int link = -1;
    bool hull_cand = false;      /* an arm link whose box the segment meets: its HULL is the collider (round 6; rounds 1 - 5: the box) - clipped against its face planes below */
    if (lane < m->n_col) {
      Xf x = collider_xf(m, L, lane);
      V3 he = ld3(m->col_he[lane]);
      if (m->col_type[lane] == 0) {
        const bool is_hull = m->hpl != nullptr && m->hpl_cnt[lane] > 0;
        V3 ol = tmulv(x.R, p1 - x.p), dl = tmulv(x.R, d);
        float o3[3] = {ol.x, ol.y, ol.z}, d3[3] = {dl.x, dl.y, dl.z}, h3[3] = {he.x, he.y, he.z};
        bool inside = fabsf(o3[0]) <= h3[0] && fabsf(o3[1]) <= h3[1] && fabsf(o3[2]) <= h3[2];
        bool hit = !inside || is_hull;      /* (the box around a hull is a cull only: a start inside it says nothing about the hull) */
        float tmin = 0.f, tmax = 1.f;
        for (int k = 0; k < 3 && hit; k++) {
          if (fabsf(d3[k]) < 1e-12f) { if (fabsf(o3[k]) > h3[k]) hit = false; continue; }
          float t1 = (-h3[k] - o3[k]) / d3[k], t2 = (h3[k] - o3[k]) / d3[k];
          if (t1 > t2) { float s = t1; t1 = t2; t2 = s; }
          tmin = fmaxf(tmin, t1); tmax = fminf(tmax, t2);
          if (tmin > tmax) hit = false;
        }
        if (is_hull) hull_cand = hit;
        else if (hit) { t = tmin; link = m->col_link[lane]; }
      } else {
        V3 oc = p1 - x.p;
        float a = dot(d, d), b = 2.f * dot(oc, d), cc = dot(oc, oc) - he.x * he.x;
        float disc = b * b - 4.f * a * cc;
        if (cc >= 0.f && disc >= 0.f) {
          float tt = (-b - sqrtf(disc)) / (2.f * a);
          if (tt >= 0.f && tt <= 1.f) { t = tt; link = m->col_link[lane]; }
        }
      }
    }
    /* the hull candidates, one after the other, by the whole wave: lane k clips against planes k, k + 64, ... (the oracle's ray_hull: enter = the latest plane crossed inwards at
     * a positive parameter, exit = the earliest crossed outwards, a ray outside a plane it runs parallel to misses) */
    {
      unsigned long long cand = __ballot(hull_cand);
      while (cand != 0ull) {
        const int c = __builtin_ctzll(cand);
        cand &= cand - 1ull;
        const Xf x = collider_xf(m, L, c);
        const V3 ol = tmulv(x.R, p1 - x.p), dl = tmulv(x.R, d);
        const float4* pl = (const float4*)m->hpl + m->hpl_off[c];
        const int npl = m->hpl_cnt[c];
        float t_in = 0.f, t_lim = 1.f; bool miss = false;
        for (int k = lane; k < npl; k += 64) {
          const float4 p = pl[k];
          const float den = p.x * dl.x + p.y * dl.y + p.z * dl.z, num = -(p.x * ol.x + p.y * ol.y + p.z * ol.z + p.w);
          if (fabsf(den) < 1e-12f) { if (num < 0.f) miss = true; continue; }
          const float tt = num / den;
          if (den < 0.f) t_in = fmaxf(t_in, tt); else t_lim = fminf(t_lim, tt);
        }
        for (int off = 32; off > 0; off >>= 1) { t_in = fmaxf(t_in, __shfl_xor(t_in, off)); t_lim = fminf(t_lim, __shfl_xor(t_lim, off)); }
        const bool hit = __ballot(miss) == 0ull && t_in > 0.f && t_in <= t_lim;
        if (hit && lane == c) { t = t_in; link = m->col_link[c]; }
      }
    }
    /* min over lanes, lowest collider index wins ties (the oracle scans colliders in order with <) */
    float best = t; int bl = link, bi = lane;
    for (int off = 32; off > 0; off >>= 1) {
      float ot = __shfl_xor(best, off); int ol = __shfl_xor(bl, off); int oi = __shfl_xor(bi, off);
      if (ot < best || (ot == best && oi < bi)) { best = ot; bl = ol; bi = oi; }
    }
    prop = (best >= 1.f || bl == 18 || bl == 20) ? 0 : 1;
  }
  if (lane == 0) {
    int n = m->n_arm;
    int eb = m->site_body[RP_SITE_EE];
    M3 Rb = ldm3(&L.xR[9 * eb]);
    V3 pos = ld3(&L.xp[3 * eb]) + mulv(Rb, ld3(m->site_pos[RP_SITE_EE]));
    Q4 orn = m3_to_quat(mul(Rb, ldm3(m->site_rot[RP_SITE_EE])));
    V6 v = zero6();
    uint32_t anc = m->arm_anc[eb - 1];
    for (int j = 0; j < n; j++) if ((anc >> j) & 1u) v = v + ld6(&L.S[6 * j]) * L.st[ST_QD + j];
    V3 lin = v.l + cross(v.a, pos - ld3(L.O)), ang = v.a;
    float grip = L.st[ST_Q + m->d_grip_obs] * (m->arm_type == RP_ARM_PANDA ? 1.f : 23.f);
    float* o = L.out;
    float *st = &o[O_OBS], *ag = &o[O_AG];      /* assembled in place, in LDS: as private arrays their running indices (ns++, nag++) sent them to scratch memory (80 B a lane) */
    int ns = 0, nag = 0, nf = 0;
    st[ns++] = pos.x; st[ns++] = pos.y; st[ns++] = pos.z;
    if (m->return_velocity) { st[ns++] = lin.x; st[ns++] = lin.y; st[ns++] = lin.z; }
    if (m->use_orientation) { st[ns++] = orn.x; st[ns++] = orn.y; st[ns++] = orn.z; st[ns++] = orn.w; }
    st[ns++] = grip;
    if (m->num_objects > 0) {
      for (int b = 0; b < m->num_objects; b++) {
        const float* f = &L.st[ST_FREE + 13 * b];
        for (int k = 0; k < 3; k++) { st[ns++] = f[k]; ag[nag++] = f[k]; }
        if (m->use_orientation) for (int k = 0; k < 4; k++) { st[ns++] = f[3 + k]; ag[nag++] = f[3 + k]; }
        if (m->return_velocity) for (int k = 0; k < 3; k++) st[ns++] = f[7 + k];
      }
      if (m->play) {
        float ex[4] = {L.st[ST_FREE + 13 * m->drawer_free + 1], L.st[ST_JQ + 0], L.st[ST_JQ + 1], dial01(L.st[ST_JQ + 2])};
        for (int k = 0; k < 4; k++) { st[ns++] = ex[k]; ag[nag++] = ex[k]; }
      }
    } else {
      ag[nag++] = pos.x; ag[nag++] = pos.y; ag[nag++] = pos.z;
    }
    if (m->play) {
      if (L.st[ST_HAVE_LAST] != 0.f) {
        flip_quat(&st[3], &L.st[ST_LAST_EE_Q]);
        flip_quat(&st[11], &L.st[ST_LAST_BLK_Q]);
#ifdef RP_WIDE
        flip_quat(&st[19], &L.st[ST_LAST_OBS19]);       /* (19, 23) as written in the reference: one past the second quaternion's start */
#endif
        flip_quat(&ag[3], &L.st[ST_LAST_AG_Q]);
#ifdef RP_WIDE
        flip_quat(&ag[10], &L.st[ST_LAST_AG10]);
#endif
      }
      for (int k = 0; k < 4; k++) { L.st[ST_LAST_EE_Q + k] = st[3 + k]; L.st[ST_LAST_BLK_Q + k] = st[11 + k]; L.st[ST_LAST_AG_Q + k] = ag[3 + k]; }
#ifdef RP_WIDE
      for (int k = 0; k < 4; k++) { L.st[ST_LAST_OBS19 + k] = st[19 + k]; L.st[ST_LAST_AG10 + k] = ag[10 + k]; }
#endif
      L.st[ST_HAVE_LAST] = 1.f;
    }
    int ng = __float_as_int(L.st[ST_NGOAL]);
    for (int k = 0; k < ng; k++) o[O_DG + k] = L.st[ST_GOAL + k];
    o[O_CAG] = pos.x; o[O_CAG + 1] = pos.y; o[O_CAG + 2] = pos.z; o[O_CAG + 3] = grip;
    o[O_FPS + nf++] = pos.x; o[O_FPS + nf++] = pos.y; o[O_FPS + nf++] = pos.z;
    if (m->num_objects > 0 && m->use_orientation) { o[O_FPS + nf++] = orn.x; o[O_FPS + nf++] = orn.y; o[O_FPS + nf++] = orn.z; o[O_FPS + nf++] = orn.w; }
    o[O_FPS + nf++] = grip;
    if (m->num_objects > 0) for (int k = 0; k < nag; k++) o[O_FPS + nf++] = ag[k];
    for (int j = 0; j < 8; j++) o[O_JOINTS + j] = m->joints_dof[j] >= 0 ? L.st[ST_Q + m->joints_dof[j]] : 0.f;
    o[O_VEL] = lin.x; o[O_VEL + 1] = lin.y; o[O_VEL + 2] = lin.z; o[O_VEL + 3] = ang.x; o[O_VEL + 4] = ang.y; o[O_VEL + 5] = ang.z;
    V3 eul = euler_from_quat(st[3], st[4], st[5], st[6]);
    int no = 0;
    o[O_OBSV + no++] = st[0]; o[O_OBSV + no++] = st[1]; o[O_OBSV + no++] = st[2];
    o[O_OBSV + no++] = eul.x; o[O_OBSV + no++] = eul.y; o[O_OBSV + no++] = eul.z;
    for (int k = 7; k < ns; k++) o[O_OBSV + no++] = st[k];
    o[O_PROP] = __int_as_float(prop);
    float r = compute_reward(m, &o[O_AG], &o[O_DG]);
    o[O_REW] = r;
    o[O_SUCC] = __int_as_float(r < 0.f ? 0 : 1);
    bool bad = false, fell = false;
    for (int k = 0; k < ST_MMODE; k++) if (!isfinite(L.st[k])) bad = true;
    for (int b = 0; b < m->num_objects; b++) if (L.st[ST_FREE + 13 * b + 2] < m->floor_z) fell = true;   /* below the lowest static collider */
    o[O_STATUS] = __int_as_float((bad ? 1 : 0) | (fell ? 2 : 0) | (__float_as_int(L.st[ST_STATUS]) & 24));
  }
  __syncthreads();
}

struct OutPtrs {
  float *obs_quat, *achieved_goal, *desired_goal, *cag, *fps, *joints, *velocity, *observation;
  int *proprio; float* reward; int *is_success; float* target_poses; int* status;
  float* pack;            /* [N, n_obs + n_ag + 2]: obs_quat | achieved_goal | reward | is_success (rp_out.pack) */
};

template <class LDS>
__device__ void write_outputs(const DevModel* m, const LDS& L, int lane, int env, const OutPtrs& o) {
  const float* s = L.out;
  if (o.obs_quat && lane < m->n_obs) o.obs_quat[(size_t)env * m->n_obs + lane] = s[O_OBS + lane];
  if (o.achieved_goal && lane < m->n_ag) o.achieved_goal[(size_t)env * m->n_ag + lane] = s[O_AG + lane];
  if (o.desired_goal && lane < m->n_ag) o.desired_goal[(size_t)env * m->n_ag + lane] = s[O_DG + lane];
  if (o.cag && lane < 4) o.cag[(size_t)env * 4 + lane] = s[O_CAG + lane];
  if (o.fps && lane < m->n_fps) o.fps[(size_t)env * m->n_fps + lane] = s[O_FPS + lane];
  if (o.joints && lane < 8) o.joints[(size_t)env * 8 + lane] = s[O_JOINTS + lane];
  if (o.velocity && lane < 6) o.velocity[(size_t)env * 6 + lane] = s[O_VEL + lane];
  if (o.observation && lane < m->n_observation) o.observation[(size_t)env * m->n_observation + lane] = s[O_OBSV + lane];
  if (o.pack) {
    const int no = m->n_obs, na = m->n_ag, w = no + na + 2;
    float* p = o.pack + (size_t)env * w;
    if (lane < no) p[lane] = s[O_OBS + lane];
    else if (lane < no + na) p[lane] = s[O_AG + lane - no];
    else if (lane == no + na) p[lane] = s[O_REW];
    else if (lane == no + na + 1) p[lane] = (float)__float_as_int(s[O_SUCC]);
  }
  if (lane == 0) {
    if (o.proprio) o.proprio[env] = __float_as_int(s[O_PROP]);
    if (o.reward) o.reward[env] = s[O_REW];
    if (o.is_success) o.is_success[env] = __float_as_int(s[O_SUCC]);
    if (o.status) o.status[env] = __float_as_int(s[O_STATUS]);
  }
}

template <class LDS>
__device__ __forceinline__ void load_state(LDS& L, const float* state, int env, int lane) {
  const float* r = state + (size_t)env * RP_REC_FLOATS;
  L.st[lane] = r[lane];
  L.st[lane + 64] = r[lane + 64];
  __syncthreads();
}
template <class LDS>
__device__ __forceinline__ void store_state(const LDS& L, float* state, int env, int lane) {
  __syncthreads();
  float* r = state + (size_t)env * RP_REC_FLOATS;
  r[lane] = L.st[lane];
  r[lane + 64] = L.st[lane + 64];
}

/* ------------------------------------------------------------------ kernels */
/* playEnv.step for env = blockIdx.x */
__global__ void __launch_bounds__(64, RP_WAVES_PER_EU) k_step(const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ action,
                                            OutPtrs out, int N) {
  __shared__ EnvLds L;
  int env = blockIdx.x, lane = threadIdx.x;
  if (env >= N) return;
  load_state(L, state, env, lane);
  float a8[8];
  load_action(m, action, env, a8);
  ChainQ tp = perform_action(m, L, lane, a8);
  for (int s = 0; s < K_NSUB; s++) substep(m, L, lane, env);
  calc_state(m, L, lane);
  write_outputs(m, L, lane, env, out);
  if (out.target_poses && lane == 0)
    for (int j = 0; j < m->n_target; j++) out.target_poses[(size_t)env * m->n_target + j] = tp.q[j];
  store_state(L, state, env, lane);
}

__global__ void __launch_bounds__(64) k_calc_state(const DevModel* __restrict__ m, float* __restrict__ state, OutPtrs out, int env0, int N,
                                                   const int* __restrict__ member) {
  __shared__ ObsLds L;
  int env = env0 + blockIdx.x, lane = threadIdx.x;
  if (env >= N) return;
  if (member) env = member[env];          /* place in the group -> env (groups are cut by load, see k_member) */
  load_state(L, state, env, lane);
  calc_state(m, L, lane);
  write_outputs(m, L, lane, env, out);
  store_state(L, state, env, lane);
}

/* uniform draw for this env from the counter RNG; counter lives in the state record */
__device__ __forceinline__ float next_u(EnvLds& L, uint64_t seed, uint32_t genv) {
  uint32_t c = (uint32_t)__float_as_int(L.st[ST_RNG]);
  float u = rng_uniform(seed, genv, c);
  L.st[ST_RNG] = __int_as_float((int)(c + 1));
  return u;
}

/* reset_goal_pos (environments.py:492-516); goal == nullptr draws a random goal; all lanes call, lane 0 mutates */
__device__ void reset_goal_pos(const DevModel* m, EnvLds& L, int lane, const float* goal, uint64_t seed, uint32_t genv) {
  if (lane == 0) {
    if (!goal) {
      const int ng = m->num_objects > 1 ? m->num_objects : 1;        /* num_goals = max(num_objects, 1) draws of 3 (environments.py:78, 495-498) */
      for (int g = 0; g < ng; g++)
        for (int k = 0; k < 3; k++) L.st[ST_GOAL + 3 * g + k] = m->goal_lo[k] + (m->goal_hi[k] - m->goal_lo[k]) * next_u(L, seed, genv);
      L.st[ST_NGOAL] = __int_as_float(3 * ng);
    } else {
      int ng = __float_as_int(L.st[ST_NGOAL]);
      for (int k = 0; k < ng; k++) L.st[ST_GOAL + k] = goal[k];
    }
  }
  __syncthreads();
  if (m->play) {
    calc_state(m, L, lane);
    if (lane == 0) {
      int n = m->n_ag;
      int idx = (int)(next_u(L, seed, genv) * n);
      if (idx >= n) idx = n - 1;
      float bump = next_u(L, seed, genv);
      for (int k = 0; k < n; k++) L.st[ST_GOAL + k] = L.out[O_AG + k];
      L.st[ST_GOAL + idx] = L.st[ST_GOAL + idx] + bump;
      L.st[ST_NGOAL] = __int_as_float(n);
    }
    __syncthreads();
  }
}

/* reset_object_pos, o = None (environments.py:519-540): drawer and scene joints to their defaults, every object to a uniform
 * draw from the spawn range (+3 cm per object), yaw 90 deg.  Lane 0 mutates the record; the 100 settle substeps follow. */
__device__ void reset_sample_objects(const DevModel* m, EnvLds& L, int lane, uint64_t seed, uint32_t genv) {
  if (lane == 0) {
    if (m->play) {
      float* d = &L.st[ST_FREE + 13 * m->drawer_free];
      for (int k = 0; k < 3; k++) d[k] = m->free_pos0[m->drawer_free][k];
      for (int k = 0; k < 4; k++) d[3 + k] = m->free_quat0[m->drawer_free][k];
      for (int k = 7; k < 13; k++) d[k] = 0.f;
      for (int k = 0; k < m->n_j1; k++) { L.st[ST_JQ + k] = 0.f; L.st[ST_JQD + k] = 0.f; }
    }
    float height = 0.03f;
    for (int b = 0; b < m->num_objects; b++) {
      float* f = &L.st[ST_FREE + 13 * b];
      for (int k = 0; k < 3; k++) f[k] = m->obj_lo[k] + (m->obj_hi[k] - m->obj_lo[k]) * next_u(L, seed, genv);
      f[2] += height;
      f[3] = 0.f; f[4] = 0.f; f[5] = 0.7071f; f[6] = 0.7071f;
      for (int k = 7; k < 13; k++) f[k] = 0.f;
      height += 0.03f;
    }
  }
  __syncthreads();
}
/* an object ended above / beyond the env's upper bound after settling (environments.py:537-540): sample again */
__device__ __forceinline__ bool reset_objects_out_of_bounds(const DevModel* m, const EnvLds& L) {
  bool outb = false;
  for (int b = 0; b < m->num_objects; b++)
    for (int k = 0; k < 3; k++) if (L.st[ST_FREE + 13 * b + k] > m->env_hi[k]) outb = true;
  return outb;
}
/* reset_arm's target when o = None (environments.py:575-581): uniform in the goal range (+0.2 z for the UR5) */
__device__ __forceinline__ void reset_sample_arm_target(const DevModel* m, EnvLds& L, int lane, uint64_t seed, uint32_t genv, float* tx) {
  float u0 = 0.f, u1 = 0.f, u2 = 0.f;
  if (lane == 0) { u0 = next_u(L, seed, genv); u1 = next_u(L, seed, genv); u2 = next_u(L, seed, genv); }
  u0 = unif(u0); u1 = unif(u1); u2 = unif(u2);
  tx[0] = m->goal_lo[0] + (m->goal_hi[0] - m->goal_lo[0]) * u0;
  tx[1] = m->goal_lo[1] + (m->goal_hi[1] - m->goal_lo[1]) * u1;
  tx[2] = m->goal_lo[2] + (m->goal_hi[2] - m->goal_lo[2]) * u2;
  if (m->arm_type != RP_ARM_PANDA) tx[2] += 0.2f;
}
/* the rest of one reset attempt (environments.py:590-603, 173-187): rest pose, one IK on the live arm, first 6 joints only
 * (quirk F5), a new goal, the observation; returns the reward of the fresh state (reset repeats while it is > -1) */
__device__ float reset_arm_goal_obs(const DevModel* m, EnvLds& L, int lane, uint64_t seed, uint32_t genv, const float* tx, Q4 torn) {
  __syncthreads();
  if (lane == 0) {
    int nrest = m->arm_type == RP_ARM_PANDA ? 8 : 6;
    for (int i = 0; i < nrest; i++) { L.st[ST_Q + i] = m->rest[i]; L.st[ST_QD + i] = 0.f; }
    L.st[ST_STATUS] = 0.f;               /* (the "IK capped" note of the last step does not outlive a reset) */
  }
  __syncthreads();
  ChainQ cur;
#pragma unroll
  for (int j = 0; j < 7; j++) cur.q[j] = j < m->ee_chain ? L.st[ST_Q + j] : 0.f;
  ChainQ sol = ik_solve(m, mk3(tx[0], tx[1], tx[2]), torn, cur, 20, lane & 15);
  __syncthreads();
  if (lane == 0) for (int i = 0; i < 6; i++) { L.st[ST_Q + i] = sol.q[i]; L.st[ST_QD + i] = 0.f; }
  __syncthreads();
  reset_goal_pos(m, L, lane, nullptr, seed, genv);
  calc_state(m, L, lane);
  /* sparse=False: the reward is -distance, never <= -1 inside the scene - the reference's `while r > -1` would not end there (environments.py:176-186
   * with 169-170), so there is no behaviour to match: a dense env keeps its first draw (the oracle does the same) */
  float r = m->dense_reward ? -1.f : L.out[O_REW];
  __syncthreads();
  return r;
}

/* playEnv.reset(o=None) and reset(o) (environments.py:173-187, 519-603) in one kernel, one wave per env, the settle substeps
 * through the fused substep().  rp_reset_to uses it (no settling there), and rp_reset when the fused path is selected; the
 * default rp_reset runs the same sequence through the split pipeline (k_reset_* below). */
__global__ void __launch_bounds__(64, RP_WAVES_PER_EU) k_reset(const DevModel* __restrict__ m, float* __restrict__ state, const uint8_t* __restrict__ mask,
                                             OutPtrs out, int N, uint64_t seed, uint32_t env_offset, const float* __restrict__ obs_o, int n_o) {
  __shared__ EnvLds L;
  int env = blockIdx.x, lane = threadIdx.x;
  if (env >= N) return;
  if (mask && !mask[env]) return;
  uint32_t genv = env_offset + (uint32_t)env;
  load_state(L, state, env, lane);
  float r = 0.f;
  for (int attempt = 0; attempt < 64 && r > -1.f; attempt++) {
    float tx[3];
    Q4 torn = {0.f, 0.f, 0.f, 1.f};
    if (obs_o) {
      /* reset(o) (environments.py:519-525, 542-556, 575-590): drawer / scene joints to their defaults, the object block read
       * from o[11:18] (use_orientation) or o[7:10], the arm's IK target from o[0:3] (+ o[3:7] / o[6:10]); no settling */
      const float* o = obs_o + (size_t)env * n_o;
      if (lane == 0) {
        if (m->play) {
          float* d = &L.st[ST_FREE + 13 * m->drawer_free];
          for (int k = 0; k < 3; k++) d[k] = m->free_pos0[m->drawer_free][k];
          for (int k = 0; k < 4; k++) d[3 + k] = m->free_quat0[m->drawer_free][k];
          for (int k = 7; k < 13; k++) d[k] = 0.f;
          for (int k = 0; k < m->n_j1; k++) { L.st[ST_JQ + k] = 0.f; L.st[ST_JQD + k] = 0.f; }
        }
        int index = m->use_orientation ? 11 : 7, inc = m->use_orientation ? 10 : 6;
        for (int b = 0; b < m->num_objects; b++) {
          float* f = &L.st[ST_FREE + 13 * b];
          for (int k = 0; k < 3; k++) f[k] = o[index + k];
          if (m->use_orientation) for (int k = 0; k < 4; k++) f[3 + k] = o[index + 3 + k];
          else { f[3] = 0.f; f[4] = 0.f; f[5] = 0.f; f[6] = 1.f; }
          for (int k = 7; k < 13; k++) f[k] = 0.f;
          index += inc;
        }
      }
      tx[0] = o[0]; tx[1] = o[1]; tx[2] = o[2];
      if (m->use_orientation) { int q0 = m->return_velocity ? 6 : 3; torn.x = o[q0]; torn.y = o[q0 + 1]; torn.z = o[q0 + 2]; torn.w = o[q0 + 3]; }
    } else {
      for (int depth = 0; depth < 9; depth++) {
        reset_sample_objects(m, L, lane, seed, genv);
        for (int i = 0; i < K_NSETTLE; i++) substep(m, L, lane, env);
        if (!reset_objects_out_of_bounds(m, L)) break;
      }
      reset_sample_arm_target(m, L, lane, seed, genv, tx);
    }
    r = reset_arm_goal_obs(m, L, lane, seed, genv, tx, torn);
  }
  write_outputs(m, L, lane, env, out);
  store_state(L, state, env, lane);
}

/* ---- rp_reset through the split pipeline.  The host runs rounds; in a round every env that still has a settle phase ahead
 * ("pending") is gathered into a dense scratch range, gets its objects sampled (k_reset_sample), runs the 100 settle substeps as
 * 100 x (k_prep2, k_solve2) over that range - the same kernels as rp_step - and then either finishes its reset, samples the
 * objects again (out of bounds: depth + 1) or starts its next attempt (fresh state already rewarded: attempt + 1); records go
 * back to the env's slot after every round (k_reset_finish).  Same draws, same order, same arithmetic as k_reset. */
__global__ void k_reset_mark(const uint8_t* __restrict__ mask, int4* __restrict__ meta, int N) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env < N) meta[env] = make_int4((!mask || mask[env]) ? 1 : 0, 0, 0, 0);        /* pending, attempt, depth */
}
__global__ void k_reset_flag_pending(const int4* __restrict__ meta, int* __restrict__ status, int N) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env < N && status && meta[env].x != 0) status[env] |= 4;
}
/* pending envs in index order -> idx[0 .. count) (one wave) */
__global__ void __launch_bounds__(64) k_reset_list(const int4* __restrict__ meta, int* __restrict__ idx, int* __restrict__ count, int N) {
  const int lane = threadIdx.x;
  int total = 0;
  for (int base = 0; base < N; base += 64) {
    const int env = base + lane;
    const bool p = env < N && meta[env].x != 0;
    const unsigned long long bal = __ballot(p);
    if (p) idx[total + __popcll(bal & ((1ull << lane) - 1ull))] = env;
    total += __popcll(bal);
  }
  if (lane == 0) *count = total;
}
__global__ void __launch_bounds__(64) k_reset_sample(const DevModel* __restrict__ m, const float* __restrict__ state, float* __restrict__ scratch,
                                                     const int* __restrict__ idx, int M, uint64_t seed, uint32_t env_offset) {
  __shared__ EnvLds L;
  const int slot = blockIdx.x, lane = threadIdx.x;
  if (slot >= M) return;
  const int env = idx[slot];
  load_state(L, state, env, lane);
  reset_sample_objects(m, L, lane, seed, env_offset + (uint32_t)env);
  store_state(L, scratch, slot, lane);
}
__global__ void __launch_bounds__(64, RP_WAVES_PER_EU) k_reset_finish(const DevModel* __restrict__ m, const float* __restrict__ scratch, float* __restrict__ state,
                                                                       const int* __restrict__ idx, int4* __restrict__ meta, OutPtrs out, int M, uint64_t seed,
                                                                       uint32_t env_offset) {
  __shared__ EnvLds L;
  const int slot = blockIdx.x, lane = threadIdx.x;
  if (slot >= M) return;
  const int env = idx[slot];
  const uint32_t genv = env_offset + (uint32_t)env;
  int4 mt = meta[env];
  load_state(L, scratch, slot, lane);
  if (reset_objects_out_of_bounds(m, L) && mt.z < 8) {
    mt.z++;                                        /* sample the objects again, settle again */
  } else {
    float tx[3];
    const Q4 torn = {0.f, 0.f, 0.f, 1.f};
    reset_sample_arm_target(m, L, lane, seed, genv, tx);
    const float r = reset_arm_goal_obs(m, L, lane, seed, genv, tx, torn);
    /* already solved: the whole reset again.  (sparse=False: the reward is -distance, never <= -1 inside the scene - the reference's `while r > -1` would not end,
     * environments.py:176-186 with 169-170; there is no behaviour to match, and a dense env keeps its first draw like the oracle) */
    if (!m->dense_reward && r > -1.f && mt.y + 1 < 64) { mt.y++; mt.z = 0; }
    else { mt.x = 0; write_outputs(m, L, lane, env, out); }
  }
  if (lane == 0) meta[env] = mt;
  store_state(L, state, env, lane);
}

__global__ void __launch_bounds__(64) k_reset_goal(const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ goal,
                                                  const uint8_t* __restrict__ mask, int N, uint64_t seed, uint32_t env_offset) {
  __shared__ EnvLds L;
  int env = blockIdx.x, lane = threadIdx.x;
  if (env >= N) return;
  if (mask && !mask[env]) return;
  load_state(L, state, env, lane);
  reset_goal_pos(m, L, lane, goal ? goal + (size_t)env * m->n_ag : nullptr, seed, env_offset + (uint32_t)env);
  store_state(L, state, env, lane);
}

/* initial records: everything as loaded (arm at q = 0, bodies at creation poses, default velocity motors) */
__global__ void k_init(const DevModel* __restrict__ m, float* __restrict__ state, int N) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= N) return;
  float* r = state + (size_t)env * RP_REC_FLOATS;
  for (int k = 0; k < RP_REC_FLOATS; k++) r[k] = 0.f;
  for (int f = 0; f < m->n_free; f++) {
    for (int k = 0; k < 3; k++) r[ST_FREE + 13 * f + k] = m->free_pos0[f][k];
    for (int k = 0; k < 4; k++) r[ST_FREE + 13 * f + 3 + k] = m->free_quat0[f][k];
  }
  for (int f = m->n_free; f < ST_NFREE; f++) r[ST_FREE + 13 * f + 6] = 1.f;
  for (int i = 0; i < ST_NARM; i++) r[ST_MMAXIMP + i] = K_DEFMOTOR;
  r[ST_NGOAL] = __int_as_float(m->n_goal_init);
}

__global__ void k_reward(const DevModel* __restrict__ m, const float* __restrict__ ag, const float* __restrict__ dg, float* __restrict__ r, int M, int force_sparse) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  r[i] = compute_reward(m, ag + (size_t)i * m->n_ag, dg + (size_t)i * m->n_ag, force_sparse != 0);
}

/* state copies (rp_get_state / rp_set_state with broadcast): a row of the caller's buffer is the env's record followed by its contact cache (nc floats, 0 under
 * RP_CFG_STATELESS_CONTACTS) */
__global__ void k_copy_state(float* __restrict__ rec, float* __restrict__ cache, const float* __restrict__ src, int N, int src_count, int nc) {
  const size_t W = RP_REC_FLOATS + (size_t)nc;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * W) return;
  const size_t env = i / W, k = i % W;
  const float v = src[(src_count == 1 ? 0 : env) * W + k];
  if (k < RP_REC_FLOATS) rec[env * RP_REC_FLOATS + k] = v; else cache[env * nc + (k - RP_REC_FLOATS)] = v;
}
__global__ void k_read_state(float* __restrict__ dst, const float* __restrict__ rec, const float* __restrict__ cache, int N, int nc) {
  const size_t W = RP_REC_FLOATS + (size_t)nc;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * W) return;
  const size_t env = i / W, k = i % W;
  dst[i] = k < RP_REC_FLOATS ? rec[env * RP_REC_FLOATS + k] : cache[env * nc + (k - RP_REC_FLOATS)];
}

/* ------------------------------------------------------------------ split pipeline for rp_step
 * The fused substep() above needs > 256 VGPRs in its cold phases (IK, narrowphase, row build) although the hot PGS
 * loop needs ~60, so rp_step runs right-sized kernels instead:
 *   k_action  (16 lanes per env) clip + action type -> IK target + IK + motor targets -> state records (one launch with the first k_prep2: k_action_prep)
 *   12 x { k_prep2 (wave per env: FK, collision, dynamics, constraint rows -> per-env workspace, L2/MALL resident)
 *          k_solve2 (two envs per wave: PGS sweeps on register-resident rows + integration, state record in/out) }
 *   k_calc_state (wave per env) calc_state + reward + outputs */
/* perform_action (environments.py:915-1073), 16 lanes (one DPP row) per env, four envs per wave: cooperative IK */
__device__ __forceinline__ void action_body(const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ action,
                                            float* __restrict__ target_poses, int env0, int N, const int* __restrict__ member, const int bid) {
  const int l16 = threadIdx.x & 15;
  const int env_raw = env0 + bid * (blockDim.x >> 4) + (threadIdx.x >> 4);      /* this launch covers places [env0, N) of its group; one env per DPP row */
  const bool live = env_raw < N;
  const int place = live ? env_raw : env0;
  const int env = member ? member[place] : place;
  float* st = state + (size_t)env * RP_REC_FLOATS;
  float a8[8];
  load_action(m, action, env, a8);
  const int nd = m->n_target, nc = m->ee_chain, at = m->action_type;
  const float q0 = st[ST_Q + (l16 < RP_MAX_ARM ? l16 : 0)];            /* measured joint value of dof l16 */
  float qj = l16 < nc ? q0 : 0.f;
  bool ik_capped = false, ik_marginal = false;
  if (at == RP_ACT_ABS_JOINTS || at == RP_ACT_REL_JOINTS) {
    float aj = 0.f;
#pragma unroll
    for (int j = 0; j < 7; j++) aj = l16 == j ? a8[j] : aj;
    qj = at == RP_ACT_REL_JOINTS ? aj + q0 : aj;
  } else {
    V3 cp = mk3(0, 0, 0); Q4 cq = {0.f, 0.f, 0.f, 1.f};
    if (at == RP_ACT_REL_RPY || at == RP_ACT_REL_QUAT) ee_pose_coop(m, qj, l16, cp, cq);
    V3 tpos; Q4 tq;
    action_target(at, a8, cp, cq, tpos, tq);
    if (m->arm_type == RP_ARM_PANDA) qj = ik_coop<7>(m, tpos, tq, qj, 200, l16, live, &ik_capped, &ik_marginal);
    else if (nc == 6) { for (int rep = 0; rep < 4; rep++) qj = ik_coop<6>(m, tpos, tq, qj, 20, l16, live, &ik_capped, &ik_marginal); }
    else { for (int rep = 0; rep < 4; rep++) qj = ik_coop<7>(m, tpos, tq, qj, 20, l16, live, &ik_capped, &ik_marginal); }
  }
  if (!live) return;
  if (l16 < nd) {
    float t = clampf(qj, m->ll[l16], m->ul[l16]);
    t = clampf(t, q0 - m->inc[l16], q0 + m->inc[l16]);
    st[ST_MMODE + l16] = 1.f; st[ST_MTARGET + l16] = t; st[ST_MMAXIMP + l16] = 240.f * K_DT;
    if (target_poses) target_poses[(size_t)env * nd + l16] = t;
  }
  if (l16 == 0) {
    st[ST_STATUS] = __int_as_float((ik_capped ? 8 : 0) | (ik_marginal ? 16 : 0));      /* the last IK call of this step ran out of iterations / a stopping test was marginal: k_calc_state passes them on (status bits 8, 16) */
    float g = a8[m->n_action - 1];
    if (m->arm_type == RP_ARM_PANDA) {
      float amt = 0.04f - g / 25.f;
      int ds[2] = {m->d9p, m->d10p};
      for (int i = 0; i < 2; i++) { st[ST_MMODE + ds[i]] = 1.f; st[ST_MTARGET + ds[i]] = amt; st[ST_MMAXIMP + ds[i]] = 100.f * K_DT; }
    } else {
      float amt = g - 0.2f;
      float left = st[ST_Q + m->d18];
      int ds[6] = {m->d18, m->d20, m->d12, m->d15, m->d10, m->d13};
      float tg[6] = {amt * 0.055f, left, amt * 0.5f, amt * 0.5f, amt * 0.8f, amt * 0.8f};
      float fo[6] = {100.f, 1000.f, 100.f, 100.f, 100.f, 100.f};
      for (int i = 0; i < 6; i++) { st[ST_MMODE + ds[i]] = 1.f; st[ST_MTARGET + ds[i]] = tg[i]; st[ST_MMAXIMP + ds[i]] = fo[i] * K_DT; }
    }
  }
}

__global__ void __launch_bounds__(64) k_action(const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ action,
                                              float* __restrict__ target_poses, int env0, int N, const int* __restrict__ member) {
  action_body(m, state, action, target_poses, env0, N, member, blockIdx.x);
}

__device__ __forceinline__ void copy_out(float* __restrict__ dst, const float* src, int nfloat, int lane) {
  for (int i = lane * 4; i < nfloat; i += 256) *(float4*)(dst + i) = *(const float4*)(src + i);
}

/* ------------------------------------------------------------------ split pipeline: register-resident rows, 2 envs/wave,
 * two concurrent row streams per env.
 *
 * Velocity component d sits at lane position lane_pos(d) of its env's 32-lane half.  DPP row 0: arm dof i at lane i and the three
 * translation components of the rotation-locked drawer behind them (rp_model.free_row0); DPP row 1: scene joints and the other free
 * bodies.  The rows of a substep:
 *   unit rows    motor i, lower / upper limit of arm dof i (J = +-e_i, B = +-M^-1[:, i]; the sign is folded into rhs and the
 *                bounds, which is exact) in row 0; the motor of scene joint t in row 1, solved by the same instructions as arm motor t
 *   gear         the Panda finger gear (two entries, row 0)
 *   contacts     every contact touches row 0 only (arm against the world, drawer on its rails, arm against the drawer), row 1 only
 *                (objects and scene-joint bodies against the world) or both (arm or drawer against an object / a scene-joint body)
 * Rows of different DPP rows act on disjoint velocity components and COMMUTE exactly, so a register SLOT holds the s-th row-0
 * contact in DPP row 0 and the s-th row-1 contact in DPP row 1 and one instruction sequence solves both; contacts that touch
 * both rows come after all of them in the solver order (collide(), shared with the oracle), one per slot from the top end, with
 * the two DPP rows' partial dot products folded.  If the slots of the two envs of a wave do not fit the 21 registers, every
 * contact takes a folded slot of its own (same code, same results).  Results are bit-identical to the sequential order.
 *
 * Instruction budget (measured, tools/ubench): a dependent VALU op costs ~4.7 cycles, a DPP op that reads the previous result
 * 12, and a DPP read needs two wait states after the VALU write of its source - so the sweep loops hold nothing but the
 * dependent chain:
 *   - J, B of every contact row live in registers (lane = dof), expanded once in the prologue;
 *   - per-row scalars live LANE-DISTRIBUTED in "planes" (lane k of a plane register = the value of the row labelled k in that
 *     plane): rhs, lo, hi, accumulated impulse lam.  Once per sweep and plane, vector ops form the bounds of the step lo - lam,
 *     hi - lam (delta form of the row update, see pgs_update) and fold the sweep's steps into lam;
 *   - a row's step forms in every lane from that lane's own plane entries - lane k's are the row's - and leaves lane k by one
 *     row broadcast fused into dv += B * step (v_fmac_f32_dpp row_newbcast:k); the owner lanes are an SGPR-pair mask;
 *   - a unit row needs no dot product at all (one fma, med3, the broadcast-fmac).
 * Planes and labels (label & 15 = lane, label >> 4 = register):
 *   M / L / U  motor, lower limit, upper limit of arm dof i at lane i        (DPP row 0; U lane 12 = gear row); plane M's DPP
 *              row 1 holds the scene-joint motors at lanes 0..2
 *   N0, N1     normal of the contact in slot s at lane s & 15 of register s >> 4
 *   F0x, F1x   friction direction 0 / 1 of that contact at the lane of its normal, so that mu * lam_normal is one vector
 *              multiply per sweep. */
#define NBJ 3                         /* scene-joint motor rows: labels [0, NBJ) of plane B0 */
#define LBL_N NBJ                     /* normal row of contact c -> label LBL_N + c */
#define NLBL (LBL_N + MAXC)           /* 24 */
#define GEAR_LANE 12
#define W3_HDR 0                      /* ints: maskL, maskU, nj, ncon, n arm-only contacts, n non-arm contacts, gear, n spanning contacts */
#define W3_VSTAR 8                    /* 32, dof-indexed */
#define W3_MU 40                      /* 32, contact-indexed */
#define W3_MINV 72                    /* 144 */
#define W3_A 216                      /* 10 arrays of 16, dof-indexed: dinv | M rhs lo hi | L rhs lo hi | U rhs lo hi */
#define W3_GEAR (W3_A + 160)          /* dofA, dofB, ratio, dinv, rhs, lo, hi, - */
#define W3_ZERO (W3_GEAR + 7)         /* always 0.0f: the address unconditional loads fall back to */
#define W3_BJ (W3_GEAR + 8)           /* 5 arrays of 4, scene joint k: J*dinv, rhs, lo, hi, 1/m */
#define W3_ROWS (W3_BJ + 20)          /* contact rows as built: rhs, dinv, mu, parent */
#define W3_ROWT (W3_ROWS + 4 * MAXROWC)   /* fold flag, hi_c, off0, off1 */
#define W3_ROFF (W3_ROWT + 4 * MAXROWC)   /* 64 ints: off0 | off1 << 8 of compact row r; entry 63 = 0.0f */
#define W3_SLOT (W3_ROFF + 64)        /* 64 ints: contact index of the s-th non-arm | arm-only | spanning contact (21 each), -1 = none */
#define W3_J (W3_SLOT + 64)           /* compact contact rows, ROWW floats each */
#define W3_B (W3_J + ROWREG)
#define W3_SROW (W3_B + ROWREG)       /* the non-contact rows as build_small_rows made them (typed list, signs not folded): [0] their number, [4 + 8 r ..] row r.  Written for the envs
                                       * super_solve takes (more than HV_MAXC contacts; every env under debug flag 1) */
#define W3_FLOATS (W3_SROW + 4 + 8 * MAXSMALL)
#define STAGE_FLOATS (128 + 2 * ROWREG)   /* ROFF | SLOT | J | B: contiguous, staged through LDS by k_solve2 */
#define AOUT_FLOATS (160 + 8 + 20)
static_assert(W3_A % 4 == 0 && W3_ROWS % 4 == 0 && W3_ROFF % 4 == 0 && AOUT_FLOATS % 4 == 0, "16-byte copies");

#define HV_MAXC 14                    /* contacts of an env on k_solve2's heavy path (heavy_solve; oracle: RES_MAX_CON) */
#ifndef S4_SLOTS0
#define S4_SLOTS0 8                   /* the four-env path's contact slots: row-0 stream (arm, drawer: wave 0; oracle: RES_SLOTS0) */
#endif
#ifndef S4_SLOTS1
#define S4_SLOTS1 8                   /* ... row-1 stream (objects, scene-joint bodies: wave 1; oracle: RES_SLOTS1) */
#endif
/* which path of k_solve2 solves an env with nc contacts - nA of them touch the first half of the velocity layout only, nB the second half only, nC both.  0: the four-env
 * path (dv form); 1: the heavy path (one env per wave, residual form); 2: the heavy path with more than HV_MAXC contacts (a second lane register: heavy_solve<true>) */
__device__ __forceinline__ int hv_class(int nc, int nA, int nB, int nC) {
  const bool heavy = nC != 0 || nA > S4_SLOTS0 || nB > S4_SLOTS1;
  return !heavy ? 0 : (nc <= HV_MAXC ? 1 : 2);
}
#define PREP_THREADS 128
/* one env's preparation by the two waves of a block: L = the block's PrepLds, env = the env, cenv = the env whose contact cache it uses (rp_reset settles in a dense
 * scratch range: the cache stays the env's own), pair_tab[pair_idx] = this env's entry of the pairing table (env | contact count << 24; stored by wave 1, lane 0).
 * Ends without a barrier: the caller synchronises before L is used again. */
__device__ __forceinline__ void prep2_core(PrepLds& L, const DevModel* __restrict__ m, const float* __restrict__ state, float* __restrict__ ws, const int env, const int cenv,
                                           int* __restrict__ pair_tab, const int pair_idx, int* __restrict__ hv_cnt = nullptr, int* __restrict__ hv_list = nullptr, const int prep_flags = 0) {      /* hv_cnt, hv_list: k_solve2's list of heavy envs (its worker blocks), nullptr = none; prep_flags bit 0: every env's typed row list */      /* (table and index apart: a per-thread pointer held across the whole kernel costs two registers, and this kernel spills for less) */
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  PCLK(6) PCLK(0) PCLK_ZERO(15) PCLK_ZERO(19) PCLK_ZERO(26) PCLK_ZERO(27) PCLK_ZERO(28) PCLK_ZERO(29) PCLK_ZERO(30) PCLK_ZERO(31)
  static_assert(RP_REC_FLOATS == PREP_THREADS, "one float of the record per thread");
  L.st[tid] = state[(size_t)env * RP_REC_FLOATS + tid];
  if (tid == 64) L.hdr[3] = pair_idx;                        /* (waits in LDS for the end of the kernel: a register held across both phases is one the narrowphase spills for) */
  if (tid == 64 && m->persist) L.hdr[2] = __float_as_int(m->pmcache[(size_t)cenv * PMC_FLOATS]);      /* the cache's manifold count, for collide() */
  __syncthreads();
  PCLK(16)
  if (wid == 0) fk_bodies(m, L, lane);
  __syncthreads();
  PCLK(17)
  float* w = ws + (size_t)env * W3_FLOATS;
  if (wid == 0) {
    /* ---- wave 0: collision detection -> the contact list */
    collider_aabbs(m, L, lane);
    WSYNC();
    PCLK(1)
    int ncon = collide<PrepLds>(m, L, lane, cenv, &L.hdr[2]);
    ncon = uni(ncon);
#ifdef RP_ABL_MAXCON    /* timing ablation: drop contacts beyond RP_ABL_MAXCON to expose the tail effect in k_solve2 */
    if (ncon > RP_ABL_MAXCON) ncon = RP_ABL_MAXCON;
#endif
    if (lane == 0) L.hdr[0] = ncon;
    PCLK(2)
  } else {
    /* ---- wave 1: the arm's dynamics, v*, the unit rows (motors, limits, gear, scene-joint motors) in the solver's dof-indexed form */
    joint_subspaces(m, L, lane);
    PCLK(18)
    arm_dynamics(m, L, lane);
    unconstrained_velocities(m, L, lane);
    PCLK(3)
    int nsmall = build_small_rows(m, L, lane);
    nsmall = uni(nsmall);
    for (int i = lane; i < AOUT_FLOATS; i += 64) L.aout[i] = 0.f;
    if (lane < 2) L.amask[lane] = 0u;
    if (lane == 2) L.amask[2] = (unsigned)nsmall;      /* (waits in LDS for the end of the kernel, like the pairing index) */
    WSYNC();
    bool gear = false;
    if (lane < nsmall) {      /* signs folded into rhs and bounds: exact */
      const float* s = &L.srow[8 * lane];
      int type = __float_as_int(s[0]), dA = __float_as_int(s[1]);
      float sg = s[2], rhs = s[3], dinv = s[4], lo = s[5], hi = s[6];
      if (type == SR_UNIT) {
        L.aout[dA] = dinv; L.aout[16 + dA] = rhs; L.aout[32 + dA] = lo; L.aout[48 + dA] = hi;
      } else if (type == SR_LIMIT) {
        int pl = sg > 0.f ? 64 : 112;
        L.aout[pl + dA] = sg * rhs; L.aout[pl + 16 + dA] = sg * lo; L.aout[pl + 32 + dA] = sg * hi;
        atomicOr(&L.amask[sg > 0.f ? 0 : 1], 1u << dA);
      } else if (type == SR_J1) {
        int k = lane;
        if (k < NBJ) { float* bq = &L.aout[168]; bq[k] = dinv; bq[4 + k] = rhs; bq[8 + k] = lo; bq[12 + k] = hi; bq[16 + k] = sg; }
      } else {
        float* g = &L.aout[160];
        g[0] = s[1]; g[1] = s[7]; g[2] = sg; g[3] = dinv; g[4] = rhs; g[5] = lo; g[6] = hi;
        gear = true;
      }
    }
    const bool anygear = __ballot(gear) != 0ull;
    if (lane == 0) L.hdr[1] = anygear ? 1 : 0;
    WSYNC();
    if (lane < 32) w[W3_VSTAR + lane] = L.vstar[lane];
    copy_out(w + W3_MINV, L.Minv, 144, lane);
    copy_out(w + W3_A, L.aout, AOUT_FLOATS, lane);
  }
  __syncthreads();            /* the join: contacts (wave 0) and M^-1, v*, joint subspaces (wave 1) are there; aout and the dynamics scratch are dead */
  const int ncon = L.hdr[0];
  int tid_j = threadIdx.x;
  asm volatile("" : "+v"(tid_j));      /* (the thread's numbers once more, opaque to the compiler: what the phases above derived from them need not survive those phases) */
  const int wid_j = tid_j >> 6;
  if (wid_j == 0) {
    const int lane = tid_j & 63;
    float* w = ws + (size_t)env * W3_FLOATS;
    /* contact classes (collide() ordered them): rank inside the class -> slot tables for k_solve2 */
    const int cls = lane < ncon ? L.conk[lane] : 3;
    const unsigned long long mB = __ballot(cls == 0), mA = __ballot(cls == 1), mC = __ballot(cls == 2);
    const unsigned long long lower = (1ull << lane) - 1ull;
    L.slot[lane] = -1;
    L.roff[lane] = 0;
    const int nt = tors_list(m, L, lane, ncon);               /* torsional rows */
    WSYNC();
    if (cls == 0) L.slot[__popcll(mB & lower)] = lane;             /* s-th contact of the second half (DPP row 1) */
    else if (cls == 1) L.slot[21 + __popcll(mA & lower)] = lane;   /* s-th contact of the first half (DPP row 0) */
    else if (cls == 2) L.slot[42 + __popcll(mC & lower)] = lane;   /* j-th contact that touches both */
    /* contact rows, PREP_CH contacts at a time: built in LDS, then copied to their places in the workspace (row number: normals first,
     * then the friction pairs - the order the one-kernel path builds them in - and behind them, compact rows 3 ncon + t, the torsional rows).
     * The torsional rows ride in the last chunk when it has room for them (its lanes are idle anyway), else in a chunk of their own; k_solve2
     * finds a torsional row's parent - the normal impulse that bounds it - through the parent's class and its rank inside it (= its slot),
     * packed beside the parent's index */
    bool tors_done = nt == 0;
    for (int c0 = 0; c0 < ncon || !tors_done; c0 += PREP_CH) {
      const int nc = max(0, min(PREP_CH, ncon - c0));
      const int ntl = (!tors_done && c0 + PREP_CH >= ncon && 3 * nc + nt <= 3 * PREP_CH) ? nt : 0;      /* (wave-uniform) */
      contact_rows(m, L, lane, c0, nc, ntl);
      WSYNC();
      auto global_row = [&](int lr) { return lr < nc ? c0 + lr : (lr < nc + ntl ? 3 * ncon + (lr - nc) : ncon + 2 * c0 + (lr - nc - ntl)); };
      for (int e = lane; e < (3 * nc + ntl) * ROWW; e += 64) {
        const int lr = e / ROWW, k = e - lr * ROWW;
        const int gr = global_row(lr);
        w[W3_J + gr * ROWW + k] = L.J[e]; w[W3_B + gr * ROWW + k] = L.B[e];
      }
      for (int e = lane; e < (3 * nc + ntl) * 4; e += 64) {
        const int lr = e >> 2, k = e & 3;
        const int gr = global_row(lr);
        float v = L.rowS[e];
        if (k == 3 && lr >= nc && lr < nc + ntl) {
          const int c = L.torc[lr - nc], kc = L.conk[c];
          const unsigned long long mk = kc == 1 ? mA : (kc == 2 ? mC : mB);
          v = __int_as_float(c | (kc << 8) | (__popcll(mk & ((1ull << c) - 1ull)) << 12));
        }
        w[W3_ROWS + gr * 4 + k] = v; w[W3_ROWT + gr * 4 + k] = L.rowT[e];
      }
      if (lane < 3 * nc + ntl && !(lane >= nc && lane < nc + ntl)) {
        const int gr = global_row(lane);
        L.roff[gr] = __float_as_int(L.rowT[4 * lane + 2]) | (__float_as_int(L.rowT[4 * lane + 3]) << 8);
      }
      if (ntl > 0) tors_done = true;
      WSYNC();
    }
    PCLK(4)
    if (lane == 0) {
      int nj = m->n_j1 < NBJ ? m->n_j1 : NBJ;
      w[W3_HDR] = __int_as_float((int)L.amask[0]); w[W3_HDR + 1] = __int_as_float((int)L.amask[1]);
      w[W3_HDR + 2] = __int_as_float(nj); w[W3_HDR + 3] = __int_as_float(ncon);
      w[W3_HDR + 4] = __int_as_float(__popcll(mA)); w[W3_HDR + 5] = __int_as_float(__popcll(mB));
      w[W3_HDR + 6] = __int_as_float(L.hdr[1] | (nt << 8)); w[W3_HDR + 7] = __int_as_float(__popcll(mC));      /* gear present | torsional rows << 8 */
      if (hv_list != nullptr && (hv_class(ncon, __popcll(mA), __popcll(mB), __popcll(mC)) != 0 || (prep_flags & 1))) hv_list[atomicAdd(hv_cnt, 1)] = env | (ncon << 24);      /* a heavy env: one of k_solve2's worker blocks to itself */
    }
    if (lane < 32) w[W3_MU + lane] = lane < ncon ? L.conmu[lane] : 0.f;
    copy_out(w + W3_ROFF, (const float*)L.roff, 64, lane);
    copy_out(w + W3_SLOT, (const float*)L.slot, 64, lane);
  } else {
    if ((tid_j & 63) == 0) pair_tab[L.hdr[3]] = env | (ncon << 24);     /* + its contact count: k_solve2 sizes its row copy without waiting for the header */
    if (prep_flags & 1) {      /* (debug flag 1) super_solve's envs: the typed row list too (it walks the rows the way the one-kernel path does) */
      const int lane = tid_j & 63;
      float* w = ws + (size_t)env * W3_FLOATS;
      const int nsm = (int)L.amask[2];
      if (lane == 0) w[W3_SROW] = __int_as_float(nsm);
      for (int i = lane; i < 8 * nsm; i += 64) w[W3_SROW + 4 + i] = L.srow[i];
    }
  }
  PCLK(5) PCLK(7)
}
/* k_prep2's block: its env from its place in the group, its place in the pairing table of the k_solve2 after this launch */
__device__ __forceinline__ void prep2_body(const DevModel* __restrict__ m, const float* __restrict__ state, float* __restrict__ ws, int env0, int N,
                                           const int* __restrict__ sort_cnt, int* __restrict__ sort_cnt_next, const int* __restrict__ sort_slot, int* __restrict__ pair_env,
                                           const int* __restrict__ member, const int bid, const int* __restrict__ cache_env = nullptr, int* __restrict__ hv_cnt = nullptr, int* __restrict__ hv_cnt_next = nullptr,
                                           int* __restrict__ hv_list = nullptr, const int prep_flags = 0) {
  __shared__ PrepLds L;
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  int env = env0 + bid;
  if (env >= N) return;
  if (member) env = member[env];     /* this block's place in its group -> env (groups are cut by load, see k_member) */
  if (bid == 0) {             /* the histogram that the k_solve2 after this launch fills (for the substep after it) starts at zero */
    for (int i = tid; i < SORT_BINS; i += PREP_THREADS) sort_cnt_next[i] = 0;
    if (tid == 0 && hv_cnt_next) *hv_cnt_next = 0;      /* ... and so does the list of heavy envs that the NEXT preparation fills (this launch's own counter: zeroed by the launch before it / by k_member) */
  }
  /* pairing table for the k_solve2 after this launch: this env's place among the envs of its group sorted by load class,
   * heaviest first = envs in heavier (class, replica) bins + its rank inside its bin (both from the previous k_solve2) */
  int pair_place = 0;
  if (wid == 1) {      /* (summed right away: this wave has ~10 k cycles of slack against the collision wave, and eight more registers live across the whole kernel - they are
                        * allocated in the collision wave's code too - are what pushed the narrowphase into scratch memory) */
    const int my_slot = sort_slot[env];
    int cnt8[8];
#pragma unroll
    for (int t = 0; t < 8; t++) cnt8[t] = sort_cnt[8 * lane + t];
    const int mybin = my_slot >> SORT_RANK_BITS;
    int above = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) above += (8 * lane + t > mybin) ? cnt8[t] : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) above += __shfl_xor(above, d);
    pair_place = above + (my_slot & SORT_RANK_MASK);
  }
  prep2_core(L, m, state, ws, env, cache_env ? cache_env[env] : env, pair_env, env0 + pair_place, hv_cnt, hv_list ? hv_list + env0 : nullptr, prep_flags);
}
/* two entry points on the same body: rp_step's substeps, and the settle substeps of rp_reset under their own name so that
 * profiles keep the two apart */
#define PREP2_ARGS const DevModel* __restrict__ m, const float* __restrict__ state, float* __restrict__ ws, int env0, int N, \
                   const int* __restrict__ sort_cnt, int* __restrict__ sort_cnt_next, const int* __restrict__ sort_slot, int* __restrict__ pair_env, \
                   const int* __restrict__ member
#define HV_ARGS int* __restrict__ hv_cnt, int* __restrict__ hv_cnt_next, int* __restrict__ hv_list, int prep_flags      /* the heavy envs of this substep (k_solve2's worker blocks): counter, the next substep's counter, list; prep2_core's flags */
__global__ void __launch_bounds__(PREP_THREADS, RP_PREP_WAVES) k_prep2(PREP2_ARGS, HV_ARGS) { prep2_body(m, state, ws, env0, N, sort_cnt, sort_cnt_next, sort_slot, pair_env, member, blockIdx.x, nullptr, hv_cnt, hv_cnt_next, hv_list, prep_flags); }
__global__ void __launch_bounds__(PREP_THREADS, RP_PREP_WAVES) k_settle_prep(PREP2_ARGS, const int* __restrict__ cache_env, int prep_flags) { prep2_body(m, state, ws, env0, N, sort_cnt, sort_cnt_next, sort_slot, pair_env, member, blockIdx.x, cache_env, nullptr, nullptr, nullptr, prep_flags); }
/* First substep of a step: the action kernel and the first k_prep2 in ONE launch.  Nothing k_prep2 builds depends on the new motor
 * targets except the motor rows themselves (v*, M^-1, contacts and limit rows see q and qd only), so the nab action blocks (first in
 * the grid: they are the long pole, ~80 dependent IK iterations) and the prep blocks of the same envs run side by side instead of
 * one after the other, and the k_solve2 that follows rebuilds the motor rows from the record (debug/flag bit 1) with the formula
 * of build_small_rows.  The prep blocks may read motor fields that an action block is writing: those values only reach the rows
 * that are rebuilt. */
__global__ void __launch_bounds__(PREP_THREADS, RP_PREP_WAVES) k_action_prep(const DevModel* __restrict__ m, float* __restrict__ state, float* __restrict__ ws, int env0, int N,
                                                                   const int* __restrict__ sort_cnt, int* __restrict__ sort_cnt_next, const int* __restrict__ sort_slot,
                                                                   int* __restrict__ pair_env, const int* __restrict__ member, const float* __restrict__ action,
                                                                   float* __restrict__ target_poses, int nab, HV_ARGS) {
  if ((int)blockIdx.x < nab) {
    /* the IK is the launch's long pole (80 dependent iterations, one wave per SIMD) and the preparation blocks beside it have 150 us of slack: its waves go first
     * wherever both want the same issue slot */
    __builtin_amdgcn_s_setprio(3);
    action_body(m, state, action, target_poses, env0, N, member, blockIdx.x);
  }
  else prep2_body(m, state, ws, env0, N, sort_cnt, sort_cnt_next, sort_slot, pair_env, member, blockIdx.x - nab, nullptr, hv_cnt, hv_cnt_next, hv_list, prep_flags);
}


typedef float f16v __attribute__((ext_vector_type(16)));
/* DPP butterfly inside each 16-lane row: every lane ends with the sum over its row */
__device__ __forceinline__ float row16_sum(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));
  return v;
}
/* sum of the two DPP rows of each half, delivered to every lane of that half */
__device__ __forceinline__ float fold_rows(float v) {
  unsigned u = __float_as_uint(v);
  auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);     /* [r0 r0 r2 r2], [r1 r1 r3 r3] */
  return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
/* one plane register set: lane k holds the scalars of the row labelled k */
struct Plane { float rhs, lo, hi, lam, dacc, loP, hiP; };
struct PlaneN : Plane { float cfm, rhsE; };      /* contact normals: softness cfm * dinv of the row, rhsE = rhs - lam cfm per sweep */
/* the row bodies below are inline asm (the compiler inserts no hazard nops inside): a DPP read needs two wait states
 * after the VALU write of its source, so the bounds written here are fenced once per sweep instead */
#define PLANE_FENCE4(a, b, c, d) asm volatile("s_nop 1" : "+v"((a).loP), "+v"((a).hiP), "+v"((b).loP), "+v"((b).hiP), "+v"((c).loP), "+v"((c).hiP), "+v"((d).loP), "+v"((d).hiP))
#define PLANE_FENCE_N(a, b, c) asm volatile("s_nop 1" : "+v"((a).loP), "+v"((a).hiP), "+v"((b).rhsE), "+v"((c).rhsE))
__device__ __forceinline__ void plane_begin(Plane& p) { p.loP = p.lo - p.lam; p.hiP = p.hi - p.lam; p.dacc = 0.f; }
__device__ __forceinline__ void plane_end(Plane& p) { p.lam += p.dacc; }
/* friction plane: bounds -+ mu * (normal impulse) from the normals' plane of the same lanes */
__device__ __forceinline__ void nplane_begin(PlaneN& p) { p.loP = p.lo - p.lam; p.hiP = p.hi - p.lam; p.dacc = 0.f; p.rhsE = __fmaf_rn(-p.lam, p.cfm, p.rhs); }
/* ... while the normal impulse is not positive the friction rows of that contact are skipped (Bullet's `if (totalImpulse > 0)`): step
 * bounds [0, 0] */
__device__ __forceinline__ void fplane_begin(Plane& p, float lim, float tot) {
  const bool on = tot > 0.f;
  p.loP = on ? (0.f - lim) - p.lam : 0.f; p.hiP = on ? (0.f + lim) - p.lam : 0.f; p.dacc = 0.f;
}

#define DPP_ALL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"

/* unit row of the dof at lane I of its DPP row: the step forms in that lane from its own plane entries (delta form:
 * d = clamp(rhs - Jd dv, lo - lam, hi - lam)), one broadcast spreads it, col = B column of the row.
 * 4 VALU instructions; dependent chain fma, med3, (2 wait states), broadcast-fmac */
template <int I>
__device__ __forceinline__ void unit_row(float jd, float col, float& dv, Plane& p, int l16) {
  float t;
  const unsigned long long own = 0x0001000100010001ull << (I & 15);      /* the lanes that own label I (one per DPP row): an SGPR pair, no v_cmp */
  asm volatile(
      "v_fma_f32 %[t], -%[jd], %[dv], %[rhs]\n"        /* rhs - Jd dv, one rounding (the one-kernel path's unit rows: the same fma) */
      "v_med3_f32 %[t], %[t], %[lo], %[hi]\n"
      "v_cndmask_b32_e64 %[dacc], %[dacc], %[t], %[own]\n"      /* the two wait states between med3 and the DPP read: this and the nop */
      "s_nop 0\n"
      "v_fmac_f32_dpp %[dv], %[t], %[col] row_newbcast:%[k]" DPP_ALL
      : [dv] "+v"(dv), [dacc] "+v"(p.dacc), [t] "=&v"(t)
      : [jd] "v"(jd), [col] "v"(col), [rhs] "v"(p.rhs), [lo] "v"(p.loP), [hi] "v"(p.hiP), [own] "s"(own), [k] "n"(I & 15));
  (void)l16;
}
/* general row labelled K in plane p: 16-lane dot product (DPP butterfly), every lane ends with the sum; then - like the unit row - the
 * step forms in EVERY lane from that lane's own plane entries (lane K's are the row's), and the last instruction takes lane K's step by
 * row broadcast: no broadcast of rhs / lo / hi (10 VALU instructions instead of 13; with two waves per SIMD the sweeps are issue-bound).
 * FOLD rows (SEQ path) add the other DPP row's sum, which is what a row that spans arm and non-arm dofs needs and an exact no-op (+0)
 * for the others: 4 more instructions, cheaper than a scalar branch around them (a not-taken s_cbranch costs ~13 cycles in this chain,
 * a taken one ~27).  A DPP read needs two wait states after the VALU write of its source: the s_nops. */
template <int K, bool FOLDABLE>
__device__ __forceinline__ void generic_row(float Jr, float Br, float& dv, Plane& p, int l16, float prhs) {
  float t, u;
  const unsigned long long own = 0x0001000100010001ull << (K & 15);
  (void)l16;
  if (!FOLDABLE)
    asm volatile(
        "v_mul_f32 %[t], %[J], %[dv]\n"
        "s_nop 1\n"
        "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2]" DPP_ALL
        "s_nop 1\n"
        "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1]" DPP_ALL
        "s_nop 1\n"
        "v_add_f32_dpp %[t], %[t], %[t] row_half_mirror" DPP_ALL
        "s_nop 1\n"
        "v_add_f32_dpp %[t], %[t], %[t] row_mirror" DPP_ALL
        "v_sub_f32 %[t], %[rhs], %[t]\n"
        "v_med3_f32 %[t], %[t], %[lo], %[hi]\n"
        "v_cndmask_b32_e64 %[dacc], %[dacc], %[t], %[own]\n"
        "s_nop 0\n"
        "v_fmac_f32_dpp %[dv], %[t], %[B] row_newbcast:%[k]" DPP_ALL
        : [dv] "+v"(dv), [dacc] "+v"(p.dacc), [t] "=&v"(t)
        : [J] "v"(Jr), [B] "v"(Br), [rhs] "v"(prhs), [lo] "v"(p.loP), [hi] "v"(p.hiP), [own] "s"(own), [k] "n"(K & 15));
  else
    asm volatile(
        "v_mul_f32 %[t], %[J], %[dv]\n"
        "s_nop 1\n"
        "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2]" DPP_ALL
        "s_nop 1\n"
        "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1]" DPP_ALL
        "s_nop 1\n"
        "v_add_f32_dpp %[t], %[t], %[t] row_half_mirror" DPP_ALL
        "s_nop 1\n"
        "v_add_f32_dpp %[t], %[t], %[t] row_mirror" DPP_ALL
        "v_mov_b32 %[u], %[t]\n"
        "s_nop 1\n"
        "v_permlane16_swap_b32 %[t], %[u]\n"
        "v_add_f32 %[t], %[t], %[u]\n"
        "v_sub_f32 %[t], %[rhs], %[t]\n"
        "v_med3_f32 %[t], %[t], %[lo], %[hi]\n"
        "v_cndmask_b32_e64 %[dacc], %[dacc], %[t], %[own]\n"
        "s_nop 0\n"
        "v_fmac_f32_dpp %[dv], %[t], %[B] row_newbcast:%[k]" DPP_ALL
        : [dv] "+v"(dv), [dacc] "+v"(p.dacc), [t] "=&v"(t), [u] "=&v"(u)
        : [J] "v"(Jr), [B] "v"(Br), [rhs] "v"(prhs), [lo] "v"(p.loP), [hi] "v"(p.hiP), [own] "s"(own), [k] "n"(K & 15));
}

#define SOLVE_WAVES 2                 /* k_solve2 waves per block */
#ifndef RP_SOLVE_WAVES_PER_EU
#define RP_SOLVE_WAVES_PER_EU 3       /* k_solve2's register budget: 168 VGPRs (round 6: the two-env register path is gone; the block's 24 KB of LDS allow six blocks = three waves per SIMD) */
#endif
#ifdef RP_PROLOGUE_CLOCKS      /* profiling build: where the two-env path's prologue spends its time (tools/gpu_prologue_clocks.py) */
__device__ unsigned long long g_pclk[8 * 4096];
#define PRO_MARK(i) if (lane == 0) g_pclk[8 * (wb & 4095) + (i)] = __builtin_readcyclecounter();
#else
#define PRO_MARK(i)
#endif
/* a pair_env entry carries its env's contact count in the top byte.  The mask is inline asm on purpose: written in C, hipcc (ROCm 7.2) can drop it - in
 * solve4_eligible it turned (pe & 0xFFFFFF) * W3_FLOATS into a 24-bit multiply and then widened that to v_mad_u64_u32 on the unmasked register: wild
 * address, aperture violation */
__device__ __forceinline__ int pair_env_id(int pe) {
  int e;
  asm volatile("v_and_b32 %0, 0xffffff, %1" : "=v"(e) : "v"(pe));
  return e;
}
/* ------------------------------------------------------------------ k_solve2, uncoupled envs: FOUR envs per pair of waves.
 * When no contact of an env touches both halves of the velocity layout (the usual case: block on table, drawer on its rails), the rows of
 * DPP row 0 (arm motors, limits, gear, the drawer's contacts) and those of DPP row 1 (scene-joint motors, the objects' contacts) never
 * meet: two independent streams.  solve2_body above zips them in one wave (motor t beside scene joint t, slot s of row 0 beside slot s of
 * row 1) and half of its lanes idle most of the time - the limit rows have no partner, the block has more contacts than the drawer.  Here
 * a block's two waves take four such envs and one STREAM each: wave 0 solves the row-0 streams of all four (one env per DPP row), wave 1
 * their row-1 streams.  Same row bodies, same order inside each stream, hence the same bits - with half the instructions per env and a
 * shorter chain (row-0 stream: ~20 unit rows + the drawer's 6; row-1 stream: 3 unit rows + the block's 12).  A block takes this path if
 * all its four envs are uncoupled and their contacts fit the slots below; otherwise its waves run solve2_body on two envs each. */
template <int T>
__device__ __forceinline__ void solve4_body(const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ ws, int env0, int N,
                                            const int* __restrict__ pair_env, int* __restrict__ sort_cnt_next, int* __restrict__ sort_slot, int debug_flags, float* __restrict__ stl, const int bq,
                                            const unsigned gmask) {
  constexpr int NSL = T == 0 ? S4_SLOTS0 : S4_SLOTS1;
  const int lane = threadIdx.x & 63, g = lane >> 4, l16 = lane & 15;
  const int place = bq * 4 + g;
  const int pe = (place < N - env0 && ((gmask >> g) & 1u)) ? pair_env[env0 + place] : -1;
  const int env = pe < 0 ? -1 : pair_env_id(pe);
  const bool valid = env >= 0;
  const float* w = ws + (size_t)(valid ? env : 0) * W3_FLOATS;
  const int n = m->n_arm;
  const float4 h0 = *(const float4*)&w[W3_HDR], h1 = *(const float4*)&w[W3_HDR + 4];
  const int my_mL = valid ? __float_as_int(h0.x) : 0, my_mU = valid ? __float_as_int(h0.y) : 0;
  const int my_nj = valid ? __float_as_int(h0.z) : 0, my_nc = valid ? __float_as_int(h0.w) : 0;
  const int my_nA = valid ? __float_as_int(h1.x) : 0, my_nB = valid ? __float_as_int(h1.y) : 0;
  const int my_gr = valid ? (__float_as_int(h1.z) & 255) : 0;
  const int my_nt = (T == 0 && valid) ? (__float_as_int(h1.z) >> 8) : 0;       /* torsional rows: all in the row-0 stream (their parents are the arm's contacts) */
#define W4_OR(x) (__builtin_amdgcn_readlane(x, 0) | __builtin_amdgcn_readlane(x, 16) | __builtin_amdgcn_readlane(x, 32) | __builtin_amdgcn_readlane(x, 48))
#define W4_MAX(x) max(max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)), max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)))
  const int maskL = T == 0 ? W4_OR(my_mL) : 0, maskU = T == 0 ? W4_OR(my_mU) : 0, gear = T == 0 ? W4_OR(my_gr) : 0;
  const int my_ns = T == 0 ? my_nA : my_nB;                  /* this stream's contacts */
  const int nS = W4_MAX(my_ns), nJ = T == 0 ? 0 : W4_MAX(my_nj), nT = T == 0 ? W4_MAX(my_nt) : 0;
  const int dd = lane_dof(m, T == 0 ? l16 : 16 + l16);      /* velocity component owned by this lane, -1 if none */
  /* the four state records: one DPP row each */
  float* st = stl + RP_REC_FLOATS * g;
  {
    const float* r = state + (size_t)(valid ? env : 0) * RP_REC_FLOATS;
#pragma unroll
    for (int k = 0; k < RP_REC_FLOATS / 16; k++) st[l16 + 16 * k] = r[l16 + 16 * k];
  }
  const float* wzero = w + W3_ZERO;
  auto ldz = [&](const float* q, bool c) { return *(c ? q : wzero); };
  const float vstar = ldz(&w[W3_VSTAR + (dd >= 0 ? dd : 0)], valid && dd >= 0);
  /* unit rows (the planes of solve2_body, one DPP row's worth) */
  const float* wa = w + W3_A;
  const float* bj = w + W3_BJ;
  const bool arm_lane = T == 0 && valid && l16 < n;
  const int ia = arm_lane ? l16 : 0;
  const bool jl = T == 1 && valid && l16 < my_nj;
  const int kj = jl ? l16 : 0;
  const float dinvX = T == 0 ? ldz(&wa[ia], arm_lane) : ldz(&bj[kj], jl);
  Plane X0, PL, PU;
  X0.rhs = T == 0 ? ldz(&wa[16 + ia], arm_lane) : ldz(&bj[4 + kj], jl);
  X0.lo = T == 0 ? ldz(&wa[32 + ia], arm_lane) : ldz(&bj[8 + kj], jl);
  X0.hi = T == 0 ? ldz(&wa[48 + ia], arm_lane) : ldz(&bj[12 + kj], jl);
  PL.rhs = PL.lo = PL.hi = PU.rhs = PU.lo = PU.hi = 0.f;
  PL.loP = PL.hiP = PL.dacc = PU.loP = PU.hiP = PU.dacc = 0.f;
  float Jg = 0.f, Bg = 0.f;
  float Bm[12];
  if (T == 0) {
    PL.rhs = ldz(&wa[64 + ia], arm_lane); PL.lo = ldz(&wa[80 + ia], arm_lane); PL.hi = ldz(&wa[96 + ia], arm_lane);
    PU.rhs = ldz(&wa[112 + ia], arm_lane); PU.lo = ldz(&wa[128 + ia], arm_lane); PU.hi = ldz(&wa[144 + ia], arm_lane);
    if (debug_flags & 2) {      /* first substep after k_action_prep: the motor rows from the record's fresh targets (build_small_rows' formula) */
      const float* r = state + (size_t)(valid ? env : 0) * RP_REC_FLOATS;
      const float mode = r[ST_MMODE + ia], tgt = r[ST_MTARGET + ia], mx = r[ST_MMAXIMP + ia], qi = r[ST_Q + ia];
      const float des = mode != 0.f ? K_KP * (tgt - qi) / K_DT : 0.f;
      const float rhs = (des - vstar) * dinvX;
      if (arm_lane) { X0.rhs = rhs; X0.lo = -mx; X0.hi = mx; }
    }
    {
      const float* gq = w + W3_GEAR;
      float g0 = gq[0], g1 = gq[1], ratio = gq[2], gd = gq[3], g4 = gq[4], g5 = gq[5], g6 = gq[6];
      int a = __float_as_int(g0) & 15, b = __float_as_int(g1) & 15;
      float ma = w[W3_MINV + ia * 12 + (a < 12 ? a : 0)], mb = w[W3_MINV + ia * 12 + (b < 12 ? b : 0)];
      bool on = arm_lane && my_gr != 0;
      Jg = on ? (l16 == a ? gd : (l16 == b ? ratio * gd : 0.f)) : 0.f;
      Bg = on ? ma + ratio * mb : 0.f;
      bool gl = valid && l16 == GEAR_LANE && my_gr != 0;
      PU.rhs = gl ? g4 : PU.rhs; PU.lo = gl ? g5 : PU.lo; PU.hi = gl ? g6 : PU.hi;
    }
    const float4 m0 = *(const float4*)&w[W3_MINV + ia * 12], m1 = *(const float4*)&w[W3_MINV + ia * 12 + 4], m2 = *(const float4*)&w[W3_MINV + ia * 12 + 8];      /* (as in solve2_body) */
    const float mr[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
#pragma unroll
    for (int t = 0; t < 12; t++) Bm[t] = (arm_lane && t < n) ? mr[t] : 0.f;
  } else {
    const float colJ = ldz(&bj[16 + kj], jl);
#pragma unroll
    for (int t = 0; t < 12; t++) Bm[t] = (t < LBL_N && l16 == t) ? colJ : 0.f;
  }
  /* contact slots: slot s = this stream's s-th contact, its scalars at lane s of the planes */
  const int* slot_tab = (const int*)(w + W3_SLOT) + (T == 0 ? 21 : 0);
  PlaneN PN; Plane PF[2];
  float muN;
  {
    const bool on = valid && l16 < NSL && l16 < my_ns;
    const int cc = on ? slot_tab[l16] : 0;
    PN.rhs = ldz(&w[W3_ROWS + 4 * cc], on); PN.lo = 0.f; PN.hi = ldz(&w[W3_ROWT + 4 * cc + 1], on);
    PN.cfm = ldz(&w[W3_ROWS + 4 * cc + 1], on); PN.rhsE = PN.rhs;
    muN = ldz(&w[W3_MU + cc], on);
#pragma unroll
    for (int d = 0; d < 2; d++) { PF[d].rhs = ldz(&w[W3_ROWS + 4 * ((on ? my_nc : 0) + 2 * cc + d)], on); PF[d].lo = 0.f; PF[d].hi = 0.f; }
  }
  float JN[NSL], BN[NSL], JF[2][NSL], BF[2][NSL];
#pragma unroll
  for (int s = 0; s < NSL; s++) {
    if (s >= nS) { JN[s] = BN[s] = JF[0][s] = BF[0][s] = JF[1][s] = BF[1][s] = 0.f; continue; }      /* (wave-uniform) no env of the four has a contact here: no loads */
    const bool used = valid && s < my_ns;
    const int c = used ? slot_tab[s] : 0;
    const int off = used ? __float_as_int(w[W3_ROFF + c]) : 0;
    const int i1 = dd - (off >> 8), i0 = dd - (off & 255);
    const int idx = (unsigned)i1 < 6u ? 12 + i1 : ((unsigned)i0 < 12u ? i0 : -1);
    const bool ok = used && dd >= 0 && idx >= 0;
    const int rn = ROWW * c + idx, rf = ROWW * (my_nc + 2 * c) + idx;
    JN[s] = ldz(&w[W3_J + (ok ? rn : 0)], ok); BN[s] = ldz(&w[W3_B + (ok ? rn : 0)], ok);
    JF[0][s] = ldz(&w[W3_J + (ok ? rf : 0)], ok); BF[0][s] = ldz(&w[W3_B + (ok ? rf : 0)], ok);
    JF[1][s] = ldz(&w[W3_J + (ok ? rf + ROWW : 0)], ok); BF[1][s] = ldz(&w[W3_B + (ok ? rf + ROWW : 0)], ok);
  }
  PN.lam = 0.f; PF[0].lam = 0.f; PF[1].lam = 0.f; X0.lam = 0.f; PL.lam = 0.f; PU.lam = 0.f;
  /* torsional rows (solve2_body explains them): here the parent is this stream's contact number `rank`, at lane `rank` of the normals' plane */
  float JT[MAXT], BT[MAXT], spinT = 0.f; Plane PT; int tsrc = lane;
  PT.rhs = PT.lo = PT.hi = PT.lam = PT.dacc = PT.loP = PT.hiP = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; t++) { JT[t] = 0.f; BT[t] = 0.f; }
  if (T == 0 && nT > 0) {
    const bool on = l16 < my_nt;
    const int rt = 3 * my_nc + (on ? l16 : 0);
    PT.rhs = ldz(&w[W3_ROWS + 4 * rt], on);
    spinT = ldz(&w[W3_ROWS + 4 * rt + 2], on);
    const int pk = on ? __float_as_int(w[W3_ROWS + 4 * rt + 3]) : 0;
    tsrc = (lane & 48) + ((pk >> 12) & 15);
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
      const bool used = t < my_nt;
      const int r = 3 * my_nc + (used ? t : 0);
      const int i1 = dd - (used ? __float_as_int(w[W3_ROWT + 4 * r + 3]) : 64), i0 = dd - (used ? __float_as_int(w[W3_ROWT + 4 * r + 2]) : 64);
      const int idx = (unsigned)i1 < 6u ? 12 + i1 : ((unsigned)i0 < 12u ? i0 : -1);
      const bool ok = used && dd >= 0 && idx >= 0;
      JT[t] = ldz(&w[W3_J + ROWW * r + (ok ? idx : 0)], ok); BT[t] = ldz(&w[W3_B + ROWW * r + (ok ? idx : 0)], ok);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);    /* vmcnt(0): all row registers have landed before the sweep loop */
#if defined(RP_CLOCKS) && RP_CLOCKS != 2
  if (lane == 0) { g_clk[8 * (blockIdx.x * SOLVE_WAVES + T) + 1] = wall_clock64();      /* (profiling build: the four-env wave's rows are loaded; its stream, slots in use, limit rows) */
    g_clk[8 * (blockIdx.x * SOLVE_WAVES + T) + 6] = (1ull << 30) | (unsigned long long)T | ((unsigned long long)nS << 8) | ((unsigned long long)(__popc(maskL) + __popc(maskU)) << 16) | ((unsigned long long)nT << 24); }
#endif
  /* counting sort by load class for the next substep's pairing: one atomic per env, from the row-1 wave */
  int sort_pos = 0, sort_bin = 0;
  if (T == 1 && l16 == 0 && valid && !(debug_flags & 4)) {
    const int my_nS = max(my_nA, my_nB);
    const int key = my_nS < 1 ? 0 : (my_nS > 14 ? 7 : (my_nS - 1) >> 1);
    sort_bin = key * SORT_REPS + (place & (SORT_REPS - 1));
    sort_pos = atomicAdd(&sort_cnt_next[sort_bin], 1);
  }
  float dv = 0.f;
#define REP3(M) M(0) M(1) M(2)
#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define REP12(M) REP8(M) M(8) M(9) M(10) M(11)
#define REP12R(M) M(11) M(10) M(9) M(8) M(7) M(6) M(5) M(4) M(3) M(2) M(1) M(0)
#define REP16(M) REP12(M) M(12) M(13) M(14) M(15)
  /* one macro call per contact slot of the stream (literal indices: they name labels) */
#if S4_SLOTS0 == 3
#define REPS0(M) M(0) M(1) M(2)
#elif S4_SLOTS0 == 4
#define REPS0(M) M(0) M(1) M(2) M(3)
#elif S4_SLOTS0 == 8
#define REPS0(M) REP8(M)
#else
#error "S4_SLOTS0: 3, 4 or 8"
#endif
#if S4_SLOTS1 == 4
#define REPS1(M) M(0) M(1) M(2) M(3)
#elif S4_SLOTS1 == 6
#define REPS1(M) M(0) M(1) M(2) M(3) M(4) M(5)
#elif S4_SLOTS1 == 8
#define REPS1(M) REP8(M)
#elif S4_SLOTS1 == 16
#define REPS1(M) REP16(M)
#else
#error "S4_SLOTS1: 4, 6, 8 or 16"
#endif
#pragma unroll 1
  for (int it = 0; it < K_NITER; it++) {
    /* in-loop copies of the guards, re-read every sweep so that they stay s_cmp + s_cbranch */
    int nS_it = __builtin_amdgcn_readfirstlane(nS), mL_it = __builtin_amdgcn_readfirstlane(maskL), mU_it = __builtin_amdgcn_readfirstlane(maskU);
    int gr_it = __builtin_amdgcn_readfirstlane(gear), nJ_it = __builtin_amdgcn_readfirstlane(nJ), nT_it = __builtin_amdgcn_readfirstlane(nT);
    asm volatile("" : "+s"(nS_it), "+s"(mL_it), "+s"(mU_it), "+s"(gr_it), "+s"(nJ_it), "+s"(nT_it));
    plane_begin(X0); nplane_begin(PN);
    if (T == 0) { plane_begin(PL); plane_begin(PU); }
    asm volatile("s_nop 1" : "+v"(X0.loP), "+v"(X0.hiP), "+v"(PL.loP), "+v"(PL.hiP), "+v"(PU.loP), "+v"(PU.hiP), "+v"(PN.loP), "+v"(PN.hiP), "+v"(PN.rhsE));
#define UNIT_M(t) unit_row<(t)>(dinvX, Bm[t], dv, X0, l16);
#define UNIT_L(i) unit_row<(i)>(dinvX, Bm[i], dv, PL, l16); unit_row<(i)>(dinvX, Bm[i], dv, PU, l16);
#define UNIT_LR(i) unit_row<(i)>(dinvX, Bm[i], dv, PU, l16); unit_row<(i)>(dinvX, Bm[i], dv, PL, l16);
#define UNIT_LO(i) unit_row<(i)>(dinvX, Bm[i], dv, PL, l16);
    if (T == 0) {      /* the unit rows in Bullet's order, walked in alternating direction: the sequence and the guards of solve2_body */
      if (it & 1) {
        if ((mL_it | mU_it) & 0x03F) { if (mU_it & 0x03F) { UNIT_L(0) UNIT_L(1) UNIT_L(2) UNIT_L(3) UNIT_L(4) UNIT_L(5) } else { UNIT_LO(0) UNIT_LO(1) UNIT_LO(2) UNIT_LO(3) UNIT_LO(4) UNIT_LO(5) } }
        if ((mL_it | mU_it) & 0xFC0) { if (mU_it & 0xFC0) { UNIT_L(6) UNIT_L(7) UNIT_L(8) UNIT_L(9) UNIT_L(10) UNIT_L(11) } else { UNIT_LO(6) UNIT_LO(7) UNIT_LO(8) UNIT_LO(9) UNIT_LO(10) UNIT_LO(11) } }
        REP12(UNIT_M)
        if (gr_it) generic_row<GEAR_LANE, false>(Jg, Bg, dv, PU, l16, PU.rhs);
      } else {
        if (gr_it) generic_row<GEAR_LANE, false>(Jg, Bg, dv, PU, l16, PU.rhs);
        REP12R(UNIT_M)
        if ((mL_it | mU_it) & 0xFC0) { if (mU_it & 0xFC0) { UNIT_LR(11) UNIT_LR(10) UNIT_LR(9) UNIT_LR(8) UNIT_LR(7) UNIT_LR(6) } else { UNIT_LO(11) UNIT_LO(10) UNIT_LO(9) UNIT_LO(8) UNIT_LO(7) UNIT_LO(6) } }
        if ((mL_it | mU_it) & 0x03F) { if (mU_it & 0x03F) { UNIT_LR(5) UNIT_LR(4) UNIT_LR(3) UNIT_LR(2) UNIT_LR(1) UNIT_LR(0) } else { UNIT_LO(5) UNIT_LO(4) UNIT_LO(3) UNIT_LO(2) UNIT_LO(1) UNIT_LO(0) } }
      }
      plane_end(X0); plane_end(PL); plane_end(PU);
    } else {
      if (nJ_it > 0) { REP3(UNIT_M) }      /* the scene-joint motors act on different dofs with diagonal responses: they commute exactly, any order gives the same bits */
      plane_end(X0);
    }
#undef UNIT_M
#undef UNIT_L
#undef UNIT_LR
#undef UNIT_LO
#define NRM4(s) if (nS_it <= (s)) goto nrm_done; generic_row<(s), false>(JN[s], BN[s], dv, PN, l16, PN.rhsE);
    if (T == 0) { REPS0(NRM4) } else { REPS1(NRM4) }
#undef NRM4
  nrm_done:
    plane_end(PN);
    if (T == 0 && nT_it > 0) {
      const float lp = __shfl(PN.lam, tsrc);
      fplane_begin(PT, spinT * lp, lp);
      asm volatile("s_nop 1" : "+v"(PT.loP), "+v"(PT.hiP));
      generic_row<0, false>(JT[0], BT[0], dv, PT, l16, PT.rhs);
      if (nT_it > 1) generic_row<1, false>(JT[1], BT[1], dv, PT, l16, PT.rhs);
      if (nT_it > 2) generic_row<2, false>(JT[2], BT[2], dv, PT, l16, PT.rhs);
      if (nT_it > 3) generic_row<3, false>(JT[3], BT[3], dv, PT, l16, PT.rhs);
      plane_end(PT);
    }
    if (nS_it > 0) {
      fplane_begin(PF[0], muN * PN.lam, PN.lam); fplane_begin(PF[1], muN * PN.lam, PN.lam);
      asm volatile("s_nop 1" : "+v"(PF[0].loP), "+v"(PF[0].hiP), "+v"(PF[1].loP), "+v"(PF[1].hiP));
#define FRC4(s) if (nS_it <= (s)) goto frc_done; generic_row<(s), false>(JF[0][s], BF[0][s], dv, PF[0], l16, PF[0].rhs); generic_row<(s), false>(JF[1][s], BF[1][s], dv, PF[1], l16, PF[1].rhs);
      if (T == 0) { REPS0(FRC4) } else { REPS1(FRC4) }
#undef FRC4
    frc_done:
      plane_end(PF[0]); plane_end(PF[1]);
    }
  }
#undef REP3
#undef REP8
#undef REP12
#undef REP12R
#undef REP16
#undef REPS0
#undef REPS1
#undef W4_OR
#undef W4_MAX
  /* integrate this stream's components and bodies; write this stream's fields of the records */
  const float vnew = clampf(vstar + dv, -K_MAXVEL, K_MAXVEL);
  const int nfree = m->n_free;
  WSYNC();
  if (dd >= 0) {
    if (dd < n) { st[ST_QD + dd] = vnew; st[ST_Q + dd] += K_DT * vnew; }
    else if (dd < n + 6 * nfree) { int k = (dd - n) / 6, c = (dd - n) % 6; st[ST_FREE + 13 * k + 7 + c] = vnew; }
    else { int k = dd - n - 6 * nfree; st[ST_JQD + k] = vnew; st[ST_JQ + k] += K_DT * vnew; }
  }
  WSYNC();
  const bool mine = l16 < nfree && (((m->free_row0 >> l16) & 1) != 0) == (T == 0);      /* free body l16 belongs to this stream */
  if (mine) {
    float* f = &st[ST_FREE + 13 * l16];
    V3 v = ld3(f + 7), wv = ld3(f + 10);
    st3(f, ld3(f) + v * K_DT);
    float wn = norm(wv);
    if (wn > 0.7853981633974483f / K_DT) wn = 0.7853981633974483f / K_DT;
    V3 ax;
    if (wn < 0.001f) ax = wv * (0.5f * K_DT - K_DT * K_DT * K_DT * 0.020833333333f * wn * wn);
    else ax = wv * (sinf(0.5f * wn * K_DT) / wn);
    Q4 dq = {ax.x, ax.y, ax.z, cosf(0.5f * wn * K_DT)};
    Q4 q0 = {f[3], f[4], f[5], f[6]};
    Q4 qn = qmul(dq, q0);
    float nr = 1.f / sqrtf(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
    f[3] = qn.x * nr; f[4] = qn.y * nr; f[5] = qn.z * nr; f[6] = qn.w * nr;
  }
  WSYNC();
  if (valid) {
    float* r = state + (size_t)env * RP_REC_FLOATS;
    if (T == 0) {
      if (l16 < n) { r[ST_Q + l16] = st[ST_Q + l16]; r[ST_QD + l16] = st[ST_QD + l16]; }
    } else {
      if (l16 < m->n_j1) { r[ST_JQ + l16] = st[ST_JQ + l16]; r[ST_JQD + l16] = st[ST_JQD + l16]; }
      if (l16 == 0 && !(debug_flags & 4)) {
        int sp = sort_pos;
        asm volatile("" : "+v"(sp));
        sort_slot[env] = (sort_bin << SORT_RANK_BITS) | sp;
      }
    }
    for (int k = 0; k < nfree; k++)
      if ((((m->free_row0 >> k) & 1) != 0) == (T == 0) && l16 < 13) r[ST_FREE + 13 * k + l16] = st[ST_FREE + 13 * k + l16];
  }
}

/* ------------------------------------------------------------------ k_solve2, HEAVY envs: ONE env per wave, the sweeps in RESIDUAL (Delassus) form (round 6).
 * Which envs: the ones the four-env path cannot take - a contact that spans the two halves of the velocity layout (arm or drawer against the block or a scene-joint body:
 * a grasp, a push, a drawer pulled by the gripper) or more contacts of one half than that path has slots (hv_class; the oracle's residual_form, rule bit
 * RPO_RULE_RESIDUAL).  Until round 5 they took a two-envs-per-wave path where every contact row is a 16- or 32-lane dot product inside the dependent chain (13 - 17
 * instructions and six wait states per row step, 185 cycles for a lone wave): 1 - 2 % of the envs, and their waves WERE every launch's second half (75 us at the median,
 * 125 at the most, while 98 % of the waves had finished after 43).  Here a lane carries ONE NUMBER through the sweeps, the row's unclamped step
 *     r = rhs - lambda cfm - Jd . dv      (Bullet's deltaImpulse before its clamp),
 * kept up to date instead of being summed anew:
 *   lanes  0 .. 11   arm dof i: s = -Jd . dv, shared by the dof's motor row and its limit rows (same Jd, signs folded): each steps from s plus its own rhs (hv_row2o)
 *   lanes 12 .. 14   scene joint k: r of its motor row;        lane 15: the Panda finger gear
 *   lanes 16 .. 29   normal of contact c (c < 14);             30, 31, 46, 47: the torsional rows
 *   lanes 32 .. 45 / 48 .. 61   its two friction rows
 *   (BIG, more than HV_MAXC contacts) a SECOND number per lane: lanes 0 .. 6 / 8 .. 14 / 16 .. 22 = normal / friction rows of contacts 14 .. 20
 * and a row step is  d = med3(r, lo - lambda, hi - lambda) in every lane at once, ONE v_readlane of the row's lane, and r = fma(-C[:, row], d, r): four instructions, none
 * of them a reduction.  C[l][r] = X_l . B_r (X_l = Jd_l for a row lane, the motor row's folded entry times e_d for a dof lane; plus the row's own softness on the
 * diagonal), summed over the dofs in ascending order with fused multiply-adds from zero, is built once per launch: the two dense tables X, B (labels x 28 dofs) in LDS,
 * lane l keeps X_l in registers and reads B_r by broadcast; the columns end in 64 registers (BIG: 21 more, and the second number's columns in LDS, fetched one row step
 * ahead).  No velocity is carried: after the sweeps dv = sum of B_r lambda_r over all rows in a fixed order.  Same rows, same order, same clamps as everywhere else
 * (build_small_rows' order walked in alternating direction, normals, torsional rows, friction pairs): in exact arithmetic the dv form line by line, in fp32 another
 * rounding - which is why the form is a property of the ENV'S OWN contact list (never of who shares a wave with whom) and why the oracle has it too
 * (solve_rows_residual). */
#define RP_DBG_SEQ 1          /* debug flag 1 (tests): the dv-form envs take super_solve (the one-kernel path's solver on the workspace's rows) instead of the four-env path */
#define RP_DBG_MOTOR 2        /* first substep after k_action_prep: the motor rows are rebuilt from the record */
#define RP_DBG_NOSORT 4       /* k_chain's substeps before the last: nobody reads their load classes */
#define RP_DBG_WORKERS 8      /* the heavy envs (residual form) of this launch are solved by the worker blocks at the head of the grid (k_solve2), not inside their own blocks */
#ifdef RP_WIDE
#define HV_NQ 8                       /* groups of four dofs in the dot products: nv = 30 (wide), <= 27 */
#else
#define HV_NQ 7
#endif
#define HV_STRIDE (4 * HV_NQ)         /* row stride of the dense tables */
#define HV_L_J1 12
#define HV_L_GEAR 15
#define HV_L_N 16
#define HV_L_F0 32
#define HV_L_F1 48
#define HV_L2 64                      /* first label of the second register (BIG): contact 14 + c: normal 64 + c, friction rows 72 + c, 80 + c */
#define HV_NLAB 88                    /* labels of a BIG env (64 + 24) */
#define HV_A2W 24                     /* lanes of the second register */
#define HV_STAGE 16                   /* columns staged through LDS at a time on their way into registers */
/* LDS of one heavy env: X table (after the build: the second register's columns, 88 x 24; after the sweeps: the final sum's sequence) | B table | column staging | label flags | record */
#define HV_OFF_Y (HV_NLAB * HV_STRIDE)
#define HV_OFF_ST (2 * HV_NLAB * HV_STRIDE)
#define HV_OFF_FL (HV_OFF_ST + HV_STAGE * 64)
#define HV_OFF_REC (HV_OFF_FL + HV_NLAB)
#define HV_LDS_FLOATS (((HV_OFF_REC + RP_REC_FLOATS) + 3) & ~3)
static_assert(HV_NLAB * HV_STRIDE >= HV_NLAB * HV_A2W && HV_NLAB * HV_STRIDE >= 256, "the X table's place holds the second register's columns, then the final sum's sequence");
__device__ __forceinline__ int hv_label_tors(int t) { return t < 2 ? 30 + t : 44 + t; }      /* 30, 31, 46, 47 */
/* label of workspace row gr of an env with nc contacts: normals [0, nc), friction pairs nc + 2 c + d, torsional rows 3 nc + t */
__device__ __forceinline__ int hv_label_row(int gr, int nc) {
  if (gr < nc) return gr < HV_MAXC ? HV_L_N + gr : HV_L2 + gr - HV_MAXC;
  if (gr < 3 * nc) {
    const int f = gr - nc, c = f >> 1;
    return c < HV_MAXC ? ((f & 1) ? HV_L_F1 : HV_L_F0) + c : HV_L2 + ((f & 1) ? 16 : 8) + c - HV_MAXC;
  }
  return hv_label_tors(gr - 3 * nc);
}
/* Row steps, inline asm like the other row bodies (the compiler pads no hazards inside): r = the lanes' numbers, na = MINUS this lane's entry of the row's column.
 * Wait states (gfx940 family): a v_readlane needs one after the VALU write of its source, a VALU read of the SGPR it wrote needs two.  Rows come in pairs where they
 * can: the first row's v_writelane (the step into the impulse register) is the second row's wait state.  The steps (SGPRs) are returned for the second register (BIG). */
template <int K>
__device__ __forceinline__ int hv_row1(float& r, float& dacc, const float loP, const float hiP, const float na) {
  float t; int s;
  asm volatile(
      "v_med3_f32 %[t], %[r], %[lo], %[hi]\n"
      "s_nop 0\n"
      "v_readlane_b32 %[s], %[t], %[k]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s], %[a]\n"
      "v_writelane_b32 %[dacc], %[s], %[k]\n"
      : [r] "+v"(r), [dacc] "+v"(dacc), [t] "=&v"(t), [s] "=&s"(s)
      : [lo] "v"(loP), [hi] "v"(hiP), [a] "v"(na), [k] "n"(K));
  return s;
}
struct HvS2 { int a, b; };
template <int K1, int K2>
__device__ __forceinline__ HvS2 hv_row2(float& r, float& dacc, const float loP, const float hiP, const float na1, const float na2) {
  float t; int s1, s2;
  asm volatile(
      "v_med3_f32 %[t], %[r], %[lo], %[hi]\n"
      "s_nop 0\n"
      "v_readlane_b32 %[s1], %[t], %[k1]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s1], %[a1]\n"
      "v_med3_f32 %[t], %[r], %[lo], %[hi]\n"
      "v_writelane_b32 %[dacc], %[s1], %[k1]\n"
      "v_readlane_b32 %[s2], %[t], %[k2]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s2], %[a2]\n"
      "v_writelane_b32 %[dacc], %[s2], %[k2]\n"
      : [r] "+v"(r), [dacc] "+v"(dacc), [t] "=&v"(t), [s1] "=&s"(s1), [s2] "=&s"(s2)
      : [lo] "v"(loP), [hi] "v"(hiP), [a1] "v"(na1), [a2] "v"(na2), [k1] "n"(K1), [k2] "n"(K2));
  HvS2 o = {s1, s2};
  return o;
}
/* the motor rows of arm dofs K1, K2: an arm dof's lane carries s = -Jd . dv alone and each of the dof's rows - motor, lower limit, upper limit - steps from s plus its
 * own right-hand side (off).  (With the motor row's rhs inside the lane, as until the middle of round 6, the limit rows read (rhs_motor + s) + (rhs_limit - rhs_motor):
 * under far targets, |rhs_motor| ~ 1e3, that costs fp32 its last three digits - 0.4 rad/s of joint velocity in ONE substep against fp64, measured on the oracle's fp32
 * build under distribution A, tests/test_oracle_dist_a.py's env 11.) */
template <int K1, int K2>
__device__ __forceinline__ HvS2 hv_row2o(float& r, float& dacc, const float off, const float loP, const float hiP, const float na1, const float na2) {
  float t; int s1, s2;
  asm volatile(
      "v_add_f32 %[t], %[r], %[o]\n"
      "v_med3_f32 %[t], %[t], %[lo], %[hi]\n"
      "s_nop 0\n"
      "v_readlane_b32 %[s1], %[t], %[k1]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s1], %[a1]\n"
      "v_add_f32 %[t], %[r], %[o]\n"
      "v_med3_f32 %[t], %[t], %[lo], %[hi]\n"
      "v_writelane_b32 %[dacc], %[s1], %[k1]\n"
      "v_readlane_b32 %[s2], %[t], %[k2]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s2], %[a2]\n"
      "v_writelane_b32 %[dacc], %[s2], %[k2]\n"
      : [r] "+v"(r), [dacc] "+v"(dacc), [t] "=&v"(t), [s1] "=&s"(s1), [s2] "=&s"(s2)
      : [o] "v"(off), [lo] "v"(loP), [hi] "v"(hiP), [a1] "v"(na1), [a2] "v"(na2), [k1] "n"(K1), [k2] "n"(K2));
  HvS2 o = {s1, s2};
  return o;
}
/* the two limit rows of arm dof K (its lane holds s: a row's number is that plus its right-hand side); A first, then B */
template <int K>
__device__ __forceinline__ HvS2 hv_rowLU(float& r, float& daccA, const float offA, const float loA, const float hiA, float& daccB, const float offB, const float loB, const float hiB, const float na) {
  float t; int s1, s2;
  asm volatile(
      "v_add_f32 %[t], %[r], %[oa]\n"
      "v_med3_f32 %[t], %[t], %[loa], %[hia]\n"
      "s_nop 0\n"
      "v_readlane_b32 %[s1], %[t], %[k]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s1], %[a]\n"
      "v_add_f32 %[t], %[r], %[ob]\n"
      "v_med3_f32 %[t], %[t], %[lob], %[hib]\n"
      "v_writelane_b32 %[da], %[s1], %[k]\n"
      "v_readlane_b32 %[s2], %[t], %[k]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s2], %[a]\n"
      "v_writelane_b32 %[db], %[s2], %[k]\n"
      : [r] "+v"(r), [da] "+v"(daccA), [db] "+v"(daccB), [t] "=&v"(t), [s1] "=&s"(s1), [s2] "=&s"(s2)
      : [oa] "v"(offA), [loa] "v"(loA), [hia] "v"(hiA), [ob] "v"(offB), [lob] "v"(loB), [hib] "v"(hiB), [a] "v"(na), [k] "n"(K));
  HvS2 o = {s1, s2};
  return o;
}
template <int K>
__device__ __forceinline__ int hv_rowL(float& r, float& dacc, const float off, const float loP, const float hiP, const float na) {
  float t; int s;
  asm volatile(
      "v_add_f32 %[t], %[r], %[o]\n"
      "v_med3_f32 %[t], %[t], %[lo], %[hi]\n"
      "s_nop 0\n"
      "v_readlane_b32 %[s], %[t], %[k]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s], %[a]\n"
      "v_writelane_b32 %[dacc], %[s], %[k]\n"
      : [r] "+v"(r), [dacc] "+v"(dacc), [t] "=&v"(t), [s] "=&s"(s)
      : [o] "v"(off), [lo] "v"(loP), [hi] "v"(hiP), [a] "v"(na), [k] "n"(K));
  return s;
}
/* a row of the SECOND register (lane K of it): its step goes into both registers; na = the first register's entry of its column, na2 the second's */
template <int K>
__device__ __forceinline__ void hv_rowB(float& r, float& r2, float& dacc2, const float loP2, const float hiP2, const float na, const float na2) {
  float t; int s;
  asm volatile(
      "v_med3_f32 %[t], %[r2], %[lo], %[hi]\n"
      "s_nop 0\n"
      "v_readlane_b32 %[s], %[t], %[k]\n"
      "s_nop 1\n"
      "v_fmac_f32 %[r], %[s], %[a]\n"
      "v_fmac_f32 %[r2], %[s], %[a2]\n"
      "v_writelane_b32 %[dacc], %[s], %[k]\n"
      : [r] "+v"(r), [r2] "+v"(r2), [dacc] "+v"(dacc2), [t] "=&v"(t), [s] "=&s"(s)
      : [lo] "v"(loP2), [hi] "v"(hiP2), [a] "v"(na), [a2] "v"(na2), [k] "n"(K));
}
struct HvPlane { float lo, hi, lam, dacc, loP, hiP; };
/* st_lds: the env's state record in LDS if the caller keeps it there (the one-kernel twins), else nullptr: the record is read from and written to `state`.
 * w: the env's workspace row (W3_*).  lds: HV_LDS_FLOATS floats of this wave's own.  BIG: the env has more than HV_MAXC contacts. */
template <bool BIG>
__device__ __forceinline__ void heavy_solve(const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ w, const int env, int* __restrict__ sort_cnt_next,
                                            int* __restrict__ sort_slot, const int debug_flags, float* __restrict__ lds, float* st_lds, const int sort_salt, const int clk_wave = -1, const int nc_hint = -1) {
  /* nc_hint: the env's contact count if the caller has it already (the list entry carries it): then the header need not arrive before the row and plane loads go out */
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));      /* (opaque: the callers loop over envs, and everything derived from the lane number - dozens of table indices - would be hoisted out of that loop and held in registers across the whole solve) */
  const int lane = lane_;
  const int n = m->n_arm;
#if defined(RP_CLOCKS) && RP_CLOCKS == 1      /* profiling build (tools/gpu_clocks6.py): 0 start, 1 tables and columns built, 2 sweeps done, 3 end (shader clock), 4 / 5 start / end (100 MHz wall clock), 6 rows, 7 where */
#define HV_CLK(i) if (lane == 0 && clk_wave >= 0) g_clk[8 * clk_wave + (i)] = __builtin_readcyclecounter();
#define HV_CLK2(i) if (lane == 0 && clk_wave >= 0) g_clk[8 * (clk_wave + 1) + (i)] = __builtin_readcyclecounter();      /* (the block's idle second wave's slots: header there, tables scattered, X rows loaded, sort atomic issued) */
  if (lane == 0 && clk_wave >= 0) g_clk[8 * clk_wave + 4] = wall_clock64();
#else
#define HV_CLK(i)
#define HV_CLK2(i)
#endif
  HV_CLK(0)
  float* Xd = lds; float* Yd = lds + HV_OFF_Y; float* Cst = lds + HV_OFF_ST; float* stl = st_lds ? st_lds : lds + HV_OFF_REC;
  constexpr int NLAB = BIG ? HV_NLAB : 64;
  constexpr int MAXCON = BIG ? MAXC : HV_MAXC;
  const float4 h0 = *(const float4*)&w[W3_HDR], h1 = *(const float4*)&w[W3_HDR + 4];
  const int nc = nc_hint >= 0 ? nc_hint : uni(__float_as_int(h0.w));
  const int nj = m->n_j1 < NBJ ? m->n_j1 : NBJ;      /* (= the header's: prep2_core) */
  /* the state record: into registers now, into LDS when the tables are built */
  float st_v0 = 0.f, st_v1 = 0.f;
  if (!st_lds) { const float* r = state + (size_t)env * RP_REC_FLOATS; st_v0 = r[lane]; st_v1 = r[lane + 64]; }
  /* the rows, compact (two body slots), all loads first: one latency, not one per pass (beyond the env's rows they read stale rows of its own workspace: never used) */
  constexpr int HV_NIT = ((3 * MAXCON + MAXT) * ROWW + 63) / 64;
  float jv[HV_NIT], bv[HV_NIT]; int of[HV_NIT];
#pragma unroll
  for (int i = 0; i < HV_NIT; i++) {
    const int e = lane + 64 * i;
    const int g2 = e / ROWW, k = e - g2 * ROWW;
    jv[i] = w[W3_J + e]; bv[i] = w[W3_B + e];
    of[i] = __float_as_int(w[W3_ROWT + 4 * g2 + (k < 12 ? 2 : 3)]);
  }
  float mv[3];
#pragma unroll
  for (int i = 0; i < 3; i++) mv[i] = w[W3_MINV + (lane + 64 * i < 144 ? lane + 64 * i : 0)];
  /* ---- the planes: this lane's row(s) */
  const float* wzero = w + W3_ZERO;
  auto ldz = [&](const float* q, bool c) { return *(c ? q : wzero); };
  const bool arm_lane = lane < n;
  const int ia = arm_lane ? lane : 0;
  const float* wa = w + W3_A;
  const float* bj = w + W3_BJ;
  const bool jl = lane >= HV_L_J1 && lane < HV_L_J1 + nj;
  const int kj = jl ? lane - HV_L_J1 : 0;
  int gr = -1, kind = 0, cpar = 0;                         /* this lane's workspace row; 1 friction, 2 torsional; the contact whose normal impulse bounds it */
  if (lane >= HV_L_N && lane < HV_L_N + HV_MAXC) { if (lane - HV_L_N < nc) gr = lane - HV_L_N; }
  else if (lane >= HV_L_F0 && lane < HV_L_F0 + HV_MAXC) { if (lane - HV_L_F0 < nc) { gr = nc + 2 * (lane - HV_L_F0); kind = 1; cpar = lane - HV_L_F0; } }
  else if (lane >= HV_L_F1 && lane < HV_L_F1 + HV_MAXC) { if (lane - HV_L_F1 < nc) { gr = nc + 2 * (lane - HV_L_F1) + 1; kind = 1; cpar = lane - HV_L_F1; } }
  else {
    const int t = lane == 30 ? 0 : (lane == 31 ? 1 : (lane == 46 ? 2 : (lane == 47 ? 3 : -1)));
    if (t >= 0) { gr = 3 * nc + t; kind = 2; }           /* (whether the env has that many torsional rows: when the header is there) */
  }
  const bool rl = gr >= 0;
  const int grc = rl ? gr : 0;
  HvPlane P0, PL, PU;
  float cfm = 0.f, mu = 0.f;
  float rhs0 = ldz(&w[W3_ROWS + 4 * grc], rl);
  if (kind == 0) { cfm = ldz(&w[W3_ROWS + 4 * grc + 1], rl); P0.lo = 0.f; P0.hi = ldz(&w[W3_ROWT + 4 * grc + 1], rl); }
  else { P0.lo = 0.f; P0.hi = 0.f; }
  if (kind == 1) mu = w[W3_MU + cpar];
  int tsrc = lane;
  if (kind == 2) {
    mu = w[W3_ROWS + 4 * grc + 2];
    const int cp = __float_as_int(w[W3_ROWS + 4 * grc + 3]) & 255;
    tsrc = cp < HV_MAXC ? HV_L_N + cp : 64 + cp - HV_MAXC;      /* (BIG: a parent in the second register: 64 + its lane there) */
  }
  const float vstar_a = ldz(&w[W3_VSTAR + ia], arm_lane);
  const float jd = arm_lane ? wa[ia] : (jl ? bj[kj] : 0.f);      /* the folded entry of a dof lane's unit rows */
  if (arm_lane || jl) { rhs0 = arm_lane ? wa[16 + ia] : bj[4 + kj]; P0.lo = arm_lane ? wa[32 + ia] : bj[8 + kj]; P0.hi = arm_lane ? wa[48 + ia] : bj[12 + kj]; }
  const float rhsL = ldz(&wa[64 + ia], arm_lane), rhsU = ldz(&wa[112 + ia], arm_lane);
  PL.lo = ldz(&wa[80 + ia], arm_lane); PL.hi = ldz(&wa[96 + ia], arm_lane);
  PU.lo = ldz(&wa[128 + ia], arm_lane); PU.hi = ldz(&wa[144 + ia], arm_lane);
  /* (BIG) the second register's rows: lanes 0 .. 6 normals, 8 .. 14 / 16 .. 22 friction rows of contacts 14 + (lane & 7) */
  HvPlane P2; float cfm2 = 0.f, mu2 = 0.f, rhs2 = 0.f; int gr2 = -1, kind2 = 0;
  P2.lo = P2.hi = P2.lam = P2.dacc = P2.loP = P2.hiP = 0.f;
  if (BIG) {
    const int c2 = HV_MAXC + (lane & 7), sec = lane >> 3;
    if (lane < HV_A2W && (lane & 7) < MAXC - HV_MAXC && c2 < nc) { gr2 = sec == 0 ? c2 : nc + 2 * c2 + (sec - 1); kind2 = sec == 0 ? 0 : 1; }
    const bool r2l = gr2 >= 0;
    const int g2c = r2l ? gr2 : 0;
    rhs2 = ldz(&w[W3_ROWS + 4 * g2c], r2l);
    if (kind2 == 0) { cfm2 = ldz(&w[W3_ROWS + 4 * g2c + 1], r2l); P2.hi = ldz(&w[W3_ROWT + 4 * g2c + 1], r2l); }
    else mu2 = ldz(&w[W3_MU + c2], r2l);
  }
  if (debug_flags & RP_DBG_MOTOR) {      /* first substep after k_action_prep: the motor rows from the record's fresh targets (build_small_rows' formula) */
    const float* r = st_lds ? st_lds : state + (size_t)env * RP_REC_FLOATS;
    const float mode = r[ST_MMODE + ia], tgt = r[ST_MTARGET + ia], mx = r[ST_MMAXIMP + ia], qi = r[ST_Q + ia];
    const float des = mode != 0.f ? K_KP * (tgt - qi) / K_DT : 0.f;
    const float rhs = (des - vstar_a) * jd;
    if (arm_lane) { rhs0 = rhs; P0.lo = -mx; P0.hi = mx; }
  }
  /* ---- the dense tables */
  {
    constexpr int NZ = (2 * NLAB * HV_STRIDE + 255) / 256;      /* (X and B tables are neighbours; the staging area behind them soaks up the last store's overshoot) */
    static_assert(2 * HV_NLAB * HV_STRIDE + 256 <= HV_OFF_FL + 256, "zeroing");
#pragma unroll
    for (int i = 0; i < NZ; i++) {
      const int a = 4 * lane + 256 * i;
      if (BIG) { if (a < 2 * HV_NLAB * HV_STRIDE) *(float4*)&lds[a] = make_float4(0.f, 0.f, 0.f, 0.f); }
      else { if (a < 64 * HV_STRIDE) *(float4*)&Xd[a] = make_float4(0.f, 0.f, 0.f, 0.f); if (a < 64 * HV_STRIDE) *(float4*)&Yd[a] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
  }
  /* the header (every load above is in flight by now) */
  const int maskL = uni(__float_as_int(h0.x)), maskU = uni(__float_as_int(h0.y));
  const int nA = uni(__float_as_int(h1.x)), nB = uni(__float_as_int(h1.y)), gear = uni(__float_as_int(h1.z)) & 255, nt = uni(__float_as_int(h1.z)) >> 8, nC = uni(__float_as_int(h1.w));
  const int nrc = 3 * nc + nt;
  HV_CLK2(0)
  if (kind == 2 && gr - 3 * nc >= nt) { gr = -1; kind = 0; rhs0 = 0.f; mu = 0.f; tsrc = lane; }      /* a torsional lane without a row */
  const bool gl = lane == HV_L_GEAR && gear != 0;
  if (gl) { const float* g = w + W3_GEAR; rhs0 = g[4]; P0.lo = g[5]; P0.hi = g[6]; }
  WSYNC();
#pragma unroll
  for (int i = 0; i < 3; i++) {                            /* arm dof t: X = jd e_t (the motor row's folded entry), B = M^-1[:, t] */
    const int e = lane + 64 * i, d = e / 12, t = e - 12 * d;
    if (e < 144 && d < n && t < n) Yd[t * HV_STRIDE + d] = mv[i];
  }
  if (arm_lane) Xd[lane * HV_STRIDE + lane] = jd;
  if (jl) { const int d = dof_j1(m, kj); Xd[lane * HV_STRIDE + d] = jd; Yd[lane * HV_STRIDE + d] = bj[16 + kj]; }
  if (gear != 0 && lane < n) {                             /* the gear: Jd = gd (e_a + ratio e_b), B = M^-1[:, a] + ratio M^-1[:, b] */
    const float* g = w + W3_GEAR;
    const int a = __float_as_int(g[0]) & 15, b = __float_as_int(g[1]) & 15;
    const float ratio = g[2], gd = g[3];
    Yd[HV_L_GEAR * HV_STRIDE + lane] = w[W3_MINV + lane * 12 + (a < 12 ? a : 0)] + ratio * w[W3_MINV + lane * 12 + (b < 12 ? b : 0)];
    if (lane == a) Xd[HV_L_GEAR * HV_STRIDE + lane] = gd;
    if (lane == b) Xd[HV_L_GEAR * HV_STRIDE + lane] = ratio * gd;
  }
#pragma unroll
  for (int i = 0; i < HV_NIT; i++) {                       /* contact, friction and torsional rows: compact (two body slots) -> dense */
    const int e = lane + 64 * i;
    const bool ok = e < nrc * ROWW;
    const int ec = ok ? e : 0, g2 = ec / ROWW, k = ec - g2 * ROWW;
    const int d = of[i] + (k < 12 ? k : k - 12);
    const int lab = hv_label_row(g2, nc);
    if (ok && d < HV_STRIDE) {                             /* (an empty slot has offset 64; entries past a body's dofs are zeros and must not overwrite a neighbour's) */
      if (jv[i] != 0.f) Xd[lab * HV_STRIDE + d] = jv[i];
      if (bv[i] != 0.f) Yd[lab * HV_STRIDE + d] = bv[i];
    }
  }
  /* ---- the columns: -C[lane][s] for every label s in use */
  WSYNC();
  HV_CLK2(1)
  float X[4 * HV_NQ], X2[BIG ? 4 * HV_NQ : 1];
#pragma unroll
  for (int q = 0; q < HV_NQ; q++) { const float4 v = *(const float4*)&Xd[lane * HV_STRIDE + 4 * q]; X[4 * q] = v.x; X[4 * q + 1] = v.y; X[4 * q + 2] = v.z; X[4 * q + 3] = v.w; }
  if (BIG) {
    const int l2 = lane < HV_A2W ? HV_L2 + lane : 0;
#pragma unroll
    for (int q = 0; q < HV_NQ; q++) { const float4 v = *(const float4*)&Xd[l2 * HV_STRIDE + 4 * q]; const bool on = lane < HV_A2W; X2[4 * q] = on ? v.x : 0.f; X2[4 * q + 1] = on ? v.y : 0.f; X2[4 * q + 2] = on ? v.z : 0.f; X2[4 * q + 3] = on ? v.w : 0.f; }
  }
  /* (a column leaves its loop through LDS - HV_STAGE columns at a time, one float per lane - and the registers are filled with constant indices afterwards: a dynamically
   * indexed write into 16-wide register vectors costs a copy of all of them per column.  BIG: the second register's entries go to their final place, A2, where the X table was) */
  HV_CLK2(2)
  f16v AC[4]; f16v AC2a; float AC2b[8];
  float* A2 = Xd;
  {
    const unsigned long long mc = nc >= 14 ? 0x3FFFull : ((1ull << nc) - 1ull);
    const unsigned long long mt = (1ull << nt) - 1ull;
    const unsigned long long use = ((1ull << n) - 1ull) | (((1ull << nj) - 1ull) << HV_L_J1) | (gear != 0 ? 1ull << HV_L_GEAR : 0ull)
                                   | (mc << HV_L_N) | ((mt & 3ull) << 30) | (mc << HV_L_F0) | (((mt >> 2) & 3ull) << 46) | (mc << HV_L_F1);
    const unsigned m2 = BIG ? (nc > HV_MAXC ? (1u << (nc - HV_MAXC)) - 1u : 0u) : 0u;
    const unsigned use2 = m2 | (m2 << 8) | (m2 << 16);      /* labels 64 + j */
    if (BIG) WSYNC();                                      /* (every lane has its X rows: A2 may overwrite the table) */
    constexpr int NCH = BIG ? 6 : 4;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      WSYNC();
#pragma unroll
      for (int i = 0; i < HV_STAGE * 64 / 256; i++) *(float4*)&Cst[4 * lane + 256 * i] = make_float4(0.f, 0.f, 0.f, 0.f);      /* (labels not in use: zero columns - their rows' steps are exact zeros, but 0 x garbage is not) */
      WSYNC();
      const unsigned um = ch < 4 ? (unsigned)((use >> (16 * ch)) & 0xFFFFull) : (unsigned)((use2 >> (16 * (ch - 4))) & 0xFFFFu);
      int u = __builtin_amdgcn_readfirstlane((int)um);
      /* two columns per pass, the B row of the one after next on its way while a column's dot product runs (a lone wave: nobody else hides an LDS round trip).  Every term
       * is taken, zeros included: they are exact, and a branch around them costs an LDS round trip of its own */
#define HV_LOADY(YV, sl) { const float* Y_ = &Yd[(16 * ch + (sl)) * HV_STRIDE]; _Pragma("unroll") for (int q = 0; q < HV_NQ; q++) YV[q] = *(const float4*)&Y_[4 * q]; }      /* the same address in every lane: broadcast reads */
#define HV_COLUMN(YV, sl) { \
        float acc = 0.f, acc2 = 0.f; \
        _Pragma("unroll") for (int q = 0; q < HV_NQ; q++) { \
          const float4 y4 = YV[q]; \
          acc = __fmaf_rn(X[4 * q], y4.x, acc); acc = __fmaf_rn(X[4 * q + 1], y4.y, acc); acc = __fmaf_rn(X[4 * q + 2], y4.z, acc); acc = __fmaf_rn(X[4 * q + 3], y4.w, acc); \
          if (BIG) { acc2 = __fmaf_rn(X2[4 * q], y4.x, acc2); acc2 = __fmaf_rn(X2[4 * q + 1], y4.y, acc2); acc2 = __fmaf_rn(X2[4 * q + 2], y4.z, acc2); acc2 = __fmaf_rn(X2[4 * q + 3], y4.w, acc2); } \
        } \
        const int s_ = 16 * ch + (sl); \
        Cst[(sl) * 64 + lane] = -(lane == s_ ? acc + cfm : acc);      /* a soft normal row's own column carries its softness */ \
        if (BIG) { if (lane < HV_A2W) A2[s_ * HV_A2W + lane] = -(HV_L2 + lane == s_ ? acc2 + cfm2 : acc2); } }
      if (u != 0) {
        float4 ya[HV_NQ], yb[HV_NQ];
        int sa = __builtin_ctz((unsigned)u);
        u &= u - 1;
        HV_LOADY(ya, sa)
#pragma unroll 1
        while (true) {
          const bool more_b = u != 0;
          const int sb = more_b ? __builtin_ctz((unsigned)u) : sa;
          u &= u - 1;
          HV_LOADY(yb, sb)                                 /* (no column left: the same row once more, unused) */
          HV_COLUMN(ya, sa)
          if (!more_b) break;
          const bool more_a = u != 0;
          sa = more_a ? __builtin_ctz((unsigned)u) : sb;
          u &= u - 1;
          HV_LOADY(ya, sa)
          HV_COLUMN(yb, sb)
          if (!more_a) break;
        }
      }
#undef HV_LOADY
#undef HV_COLUMN
      WSYNC();
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const float v = Cst[j * 64 + lane];                 /* (no select: sixteen conditional loads are sixteen jumps into cold code) */
        if (ch < 4) AC[ch < 4 ? ch : 0][j] = v;
        else if (ch == 4) AC2a[j] = v;
        else if (j < 8) AC2b[j] = v;
      }
    }
  }
  /* (the first register's entry of the column of second-register row j: contact 14 + (j & 7), normal / friction rows j >> 3) */
#define HV_AC2(j) ((j) < 16 ? AC2a[(j) & 15] : AC2b[((j) - 16) & 7])
  HV_CLK2(3)
  if (!st_lds) { stl[lane] = st_v0; stl[lane + 64] = st_v1; }
  __builtin_amdgcn_s_waitcnt(0x0F70);    /* vmcnt(0): every plane value has landed before the sweep loop */
  WSYNC();
  /* counting sort by load class for the next substep's pairing (solve4_body explains it): coupled envs first */
  int sort_pos = 0, sort_bin = 0;
  if (lane == 0 && !(debug_flags & RP_DBG_NOSORT)) {
    const int my_nS = max(nA, nB);
    const int key = 8 * (nC < 7 ? nC : 7) + (my_nS < 1 ? 0 : (my_nS > 14 ? 7 : (my_nS - 1) >> 1));
    sort_bin = key * SORT_REPS + (sort_salt & (SORT_REPS - 1));
    sort_pos = atomicAdd(&sort_cnt_next[sort_bin], 1);
  }
  HV_CLK(1)
  /* ---- the sweeps */
#ifdef RP_NO_SLANE      /* (debugging builds, with the oracle's RPO_NO_SLANE=1: the motor row's number in the lane, as until the middle of round 6) */
  float rr = rhs0, rr2 = rhs2;
  const float offM = 0.f, offL = rhsL - rhs0, offU = rhsU - rhs0;
#else
  float rr = arm_lane ? 0.f : rhs0, rr2 = rhs2;      /* an arm dof's lane: s = -Jd . dv; its rows add their own right-hand sides (hv_row2o) */
  const float offM = arm_lane ? rhs0 : 0.f, offL = rhsL, offU = rhsU;
#endif
  P0.lam = 0.f; PL.lam = 0.f; PU.lam = 0.f;
  const float* a2p = A2 + (lane < HV_A2W ? lane : 0);      /* this lane's entries of the second register's columns: a2p[s * HV_A2W] */
  /* BIG: a first-register row's step goes into the second register too; the column entry is fetched before the row's chain starts */
#define HV_B1(lab, sexpr) { if (BIG) { const float a2_ = a2p[(lab) * HV_A2W]; const int s_ = (sexpr); rr2 = __fmaf_rn(a2_, __int_as_float(s_), rr2); } else { (void)(sexpr); } }
#define HV_B2(labA, labB, sexpr) { if (BIG) { const float a2a_ = a2p[(labA) * HV_A2W], a2b_ = a2p[(labB) * HV_A2W]; const HvS2 s_ = (sexpr); rr2 = __fmaf_rn(a2a_, __int_as_float(s_.a), rr2); rr2 = __fmaf_rn(a2b_, __int_as_float(s_.b), rr2); } else { (void)(sexpr); } }
#pragma unroll 1
  for (int it = 0; it < K_NITER; it++) {
    int nc_it = uni(nc), nt_it = uni(nt), nj_it = uni(nj), mL_it = uni(maskL), mU_it = uni(maskU), gr_it = uni(gear), n_it = uni(n);
    asm volatile("" : "+s"(nc_it), "+s"(nt_it), "+s"(nj_it), "+s"(mL_it), "+s"(mU_it), "+s"(gr_it), "+s"(n_it));
    P0.loP = P0.lo - P0.lam; P0.hiP = P0.hi - P0.lam; P0.dacc = 0.f;
    PL.loP = PL.lo - PL.lam; PL.hiP = PL.hi - PL.lam; PL.dacc = 0.f;
    PU.loP = PU.lo - PU.lam; PU.hiP = PU.hi - PU.lam; PU.dacc = 0.f;
    if (BIG) { P2.loP = P2.lo - P2.lam; P2.hiP = P2.hi - P2.lam; P2.dacc = 0.f; }
#define HV_M2(i, j) HV_B2(i, j, (hv_row2<(i), (j)>(rr, P0.dacc, P0.loP, P0.hiP, AC[0][i], AC[0][j])))
#define HV_A2(i, j) HV_B2(i, j, (hv_row2o<(i), (j)>(rr, P0.dacc, offM, P0.loP, P0.hiP, AC[0][i], AC[0][j])))
#define HV_M1(i) HV_B1(i, (hv_row1<(i)>(rr, P0.dacc, P0.loP, P0.hiP, AC[0][i])))
#define HV_LO(i) HV_B1(i, (hv_rowL<(i)>(rr, PL.dacc, offL, PL.loP, PL.hiP, AC[0][i])))
#define HV_L(i) HV_B2(i, i, (hv_rowLU<(i)>(rr, PL.dacc, offL, PL.loP, PL.hiP, PU.dacc, offU, PU.loP, PU.hiP, AC[0][i])))
#define HV_LR(i) HV_B2(i, i, (hv_rowLU<(i)>(rr, PU.dacc, offU, PU.loP, PU.hiP, PL.dacc, offL, PL.loP, PL.hiP, AC[0][i])))
    /* the non-contact rows in build_small_rows' order - scene-joint motors, limits (dof-major, lower before upper; only while violated), arm motors, gear - forwards in
     * the odd sweeps, backwards in the even ones (the first).  A row that is absent inside a pair or a group (a limit that holds, a motor beyond n_arm) is all zeros:
     * its step is an exact zero */
    if (it & 1) {
      if (nj_it > 0) { HV_M2(12, 13) if (nj_it > 2) { HV_M1(14) } }
      if ((mL_it | mU_it) & 0x03F) { if (mU_it & 0x03F) { HV_L(0) HV_L(1) HV_L(2) HV_L(3) HV_L(4) HV_L(5) } else { HV_LO(0) HV_LO(1) HV_LO(2) HV_LO(3) HV_LO(4) HV_LO(5) } }
      if ((mL_it | mU_it) & 0xFC0) { if (mU_it & 0xFC0) { HV_L(6) HV_L(7) HV_L(8) HV_L(9) HV_L(10) HV_L(11) } else { HV_LO(6) HV_LO(7) HV_LO(8) HV_LO(9) HV_LO(10) HV_LO(11) } }
      HV_A2(0, 1) HV_A2(2, 3) HV_A2(4, 5) HV_A2(6, 7)
      if (n_it > 8) { HV_A2(8, 9) if (n_it > 10) { HV_A2(10, 11) } }
      if (gr_it) { HV_M1(15) }
    } else {
      if (gr_it) { HV_M1(15) }
      if (n_it > 8) { if (n_it > 10) { HV_A2(11, 10) } HV_A2(9, 8) }
      HV_A2(7, 6) HV_A2(5, 4) HV_A2(3, 2) HV_A2(1, 0)
      if ((mL_it | mU_it) & 0xFC0) { if (mU_it & 0xFC0) { HV_LR(11) HV_LR(10) HV_LR(9) HV_LR(8) HV_LR(7) HV_LR(6) } else { HV_LO(11) HV_LO(10) HV_LO(9) HV_LO(8) HV_LO(7) HV_LO(6) } }
      if ((mL_it | mU_it) & 0x03F) { if (mU_it & 0x03F) { HV_LR(5) HV_LR(4) HV_LR(3) HV_LR(2) HV_LR(1) HV_LR(0) } else { HV_LO(5) HV_LO(4) HV_LO(3) HV_LO(2) HV_LO(1) HV_LO(0) } }
      if (nj_it > 0) { if (nj_it > 2) { HV_M1(14) } HV_M2(13, 12) }
    }
#undef HV_L
#undef HV_LR
#undef HV_LO
#undef HV_M2
#undef HV_A2
    /* contact normals, in contact order, two at a time (an absent second one: an exact zero step) */
#define HV_N2(c) if (nc_it <= (c)) goto hv_ndone; HV_B2(HV_L_N + (c), HV_L_N + (c) + 1, (hv_row2<HV_L_N + (c), HV_L_N + (c) + 1>(rr, P0.dacc, P0.loP, P0.hiP, AC[1][c], AC[1][(c) + 1])))
    HV_N2(0) HV_N2(2) HV_N2(4) HV_N2(6) HV_N2(8) HV_N2(10) HV_N2(12)
#undef HV_N2
    if (BIG) {
#define HV_NB(j) if (nc_it <= HV_MAXC + (j)) goto hv_ndone; hv_rowB<(j)>(rr, rr2, P2.dacc, P2.loP, P2.hiP, HV_AC2(j), a2p[(HV_L2 + (j)) * HV_A2W]);
      HV_NB(0) HV_NB(1) HV_NB(2) HV_NB(3) HV_NB(4) HV_NB(5) HV_NB(6)
#undef HV_NB
    }
  hv_ndone:
    P0.lam += P0.dacc; P0.dacc = 0.f;
    PL.lam += PL.dacc; PU.lam += PU.dacc;
    if (BIG) { P2.lam += P2.dacc; P2.dacc = 0.f; }
    if (nc_it > 0) {
      /* bounds of the friction and torsional rows: -+ mu * (normal impulse of their contact) - the normals' DPP row (lanes 16 .. 31) copied into every DPP row, the
       * torsional rows' parents by a lane permute; while that impulse is not positive the row is skipped (step bounds [0, 0], Bullet's `if (totalImpulse > 0)`) */
      const unsigned lu = __float_as_uint(P0.lam);
      const auto s32 = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);        /* [r0 r1 r0 r1] */
      const auto s16 = __builtin_amdgcn_permlane16_swap(s32[0], s32[0], false, false);  /* ..., [r1 r1 r1 r1] */
      float src = __uint_as_float(s16[1]);
      if (nt_it > 0) {
        float lp = __shfl(P0.lam, tsrc & 63);
        if (BIG) { const float lp2 = __shfl(P2.lam, tsrc & 63); lp = tsrc >= 64 ? lp2 : lp; }
        src = kind == 2 ? lp : src;
      }
      const float lim = mu * src;
      const bool on = src > 0.f;
      if (kind != 0) { P0.loP = on ? (0.f - lim) - P0.lam : 0.f; P0.hiP = on ? (0.f + lim) - P0.lam : 0.f; }
      if (BIG) {
        const float src2 = __shfl(P2.lam, lane & 7);
        const float lim2 = mu2 * src2;
        const bool on2 = src2 > 0.f;
        if (kind2 != 0) { P2.loP = on2 ? (0.f - lim2) - P2.lam : 0.f; P2.hiP = on2 ? (0.f + lim2) - P2.lam : 0.f; }
      }
      if (nt_it > 0) {
        HV_B1(30, (hv_row1<30>(rr, P0.dacc, P0.loP, P0.hiP, AC[1][14])))
        if (nt_it > 1) HV_B1(31, (hv_row1<31>(rr, P0.dacc, P0.loP, P0.hiP, AC[1][15])))
        if (nt_it > 2) HV_B1(46, (hv_row1<46>(rr, P0.dacc, P0.loP, P0.hiP, AC[2][14])))
        if (nt_it > 3) HV_B1(47, (hv_row1<47>(rr, P0.dacc, P0.loP, P0.hiP, AC[2][15])))
      }
#define HV_F(c) if (nc_it <= (c)) goto hv_fdone; HV_B2(HV_L_F0 + (c), HV_L_F1 + (c), (hv_row2<HV_L_F0 + (c), HV_L_F1 + (c)>(rr, P0.dacc, P0.loP, P0.hiP, AC[2][c], AC[3][c])))
      HV_F(0) HV_F(1) HV_F(2) HV_F(3) HV_F(4) HV_F(5) HV_F(6) HV_F(7) HV_F(8) HV_F(9) HV_F(10) HV_F(11) HV_F(12) HV_F(13)
#undef HV_F
      if (BIG) {
#define HV_FB(j) if (nc_it <= HV_MAXC + (j)) goto hv_fdone; hv_rowB<8 + (j)>(rr, rr2, P2.dacc, P2.loP, P2.hiP, HV_AC2(8 + (j)), a2p[(HV_L2 + 8 + (j)) * HV_A2W]); \
                                                            hv_rowB<16 + (j)>(rr, rr2, P2.dacc, P2.loP, P2.hiP, HV_AC2(16 + (j)), a2p[(HV_L2 + 16 + (j)) * HV_A2W]);
        HV_FB(0) HV_FB(1) HV_FB(2) HV_FB(3) HV_FB(4) HV_FB(5) HV_FB(6)
#undef HV_FB
      }
    hv_fdone:
      P0.lam += P0.dacc;
      if (BIG) P2.lam += P2.dacc;
    }
#undef HV_M1
  }
#undef HV_B1
#undef HV_B2
#undef HV_AC2
  HV_CLK(2)
  /* ---- the velocity change: dv = sum of B_r lambda_r over all rows - motors, lower limits, upper limits (dof by dof each), scene-joint motors, gear, then the contact
   * rows in the workspace's order -, lane l < 32 <-> component lane_dof(l) as on the four-env path */
  const int l = lane & 31;
  const int dd = lane < 32 ? lane_dof(m, l) : -1;
  float dv = 0.f;
  {
    /* the terms in their order, through LDS (where the X table / the second register's columns were): every lane files the impulses of its rows at their positions, then lane <-> dof sums them */
    float* lamseq = Xd; int* offseq = (int*)(Xd + 128);
    const int nlim = (maskL | maskU) != 0 ? 2 * n : 0;
    const int base_j = n + nlim, base_g = base_j + nj, base_r = base_g + (gear != 0 ? 1 : 0), nterm = base_r + nrc;
    WSYNC();
    if (arm_lane) {
      lamseq[lane] = P0.lam; offseq[lane] = lane * HV_STRIDE;
      if (nlim) { lamseq[n + lane] = PL.lam; offseq[n + lane] = lane * HV_STRIDE; lamseq[2 * n + lane] = PU.lam; offseq[2 * n + lane] = lane * HV_STRIDE; }
    }
    if (jl) { lamseq[base_j + kj] = P0.lam; offseq[base_j + kj] = lane * HV_STRIDE; }
    if (gl) { lamseq[base_g] = P0.lam; offseq[base_g] = lane * HV_STRIDE; }
    if (gr >= 0) { lamseq[base_r + gr] = P0.lam; offseq[base_r + gr] = lane * HV_STRIDE; }
    if (BIG) { if (gr2 >= 0) { lamseq[base_r + gr2] = P2.lam; offseq[base_r + gr2] = (HV_L2 + lane) * HV_STRIDE; } }
    WSYNC();
    const int ddc = (dd >= 0 && dd < HV_STRIDE) ? dd : 0;
    const float* Yc = Yd + ddc;
#pragma unroll 4
    for (int k = 0; k < nterm; k++) dv = __fmaf_rn(Yc[offseq[k]], lamseq[k], dv);
    if (dd < 0) dv = 0.f;
  }
  const float vstar = ldz(&w[W3_VSTAR + (dd >= 0 ? dd : 0)], dd >= 0);
  const float vnew = clampf(vstar + dv, -K_MAXVEL, K_MAXVEL);
  float* st = stl;
  WSYNC();
  if (dd >= 0) {
    if (dd < n) { st[ST_QD + dd] = vnew; st[ST_Q + dd] += K_DT * vnew; }
    else if (dd < n + 6 * m->n_free) { const int k = (dd - n) / 6, c = (dd - n) % 6; st[ST_FREE + 13 * k + 7 + c] = vnew; }
    else { const int k = dd - n - 6 * m->n_free; st[ST_JQD + k] = vnew; st[ST_JQ + k] += K_DT * vnew; }
  }
  WSYNC();
  if (lane < m->n_free) {
    float* f = &st[ST_FREE + 13 * lane];
    V3 v = ld3(f + 7), wv = ld3(f + 10);
    st3(f, ld3(f) + v * K_DT);
    float wn = norm(wv);
    if (wn > 0.7853981633974483f / K_DT) wn = 0.7853981633974483f / K_DT;
    V3 ax;
    if (wn < 0.001f) ax = wv * (0.5f * K_DT - K_DT * K_DT * K_DT * 0.020833333333f * wn * wn);
    else ax = wv * (sinf(0.5f * wn * K_DT) / wn);
    Q4 dq = {ax.x, ax.y, ax.z, cosf(0.5f * wn * K_DT)};
    Q4 q0 = {f[3], f[4], f[5], f[6]};
    Q4 qn = qmul(dq, q0);
    float nr = 1.f / sqrtf(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
    f[3] = qn.x * nr; f[4] = qn.y * nr; f[5] = qn.z * nr; f[6] = qn.w * nr;
  }
  WSYNC();
  if (!st_lds) {
    float* r = state + (size_t)env * RP_REC_FLOATS;
    r[lane] = st[lane]; r[lane + 64] = st[lane + 64];
  }
  if (lane == 0 && !(debug_flags & RP_DBG_NOSORT)) {
    int sp = sort_pos;
    asm volatile("" : "+v"(sp));
    sort_slot[env] = (sort_bin << SORT_RANK_BITS) | sp;
  }
  WSYNC();
#if defined(RP_CLOCKS) && RP_CLOCKS == 1
  HV_CLK(3)
  if (lane == 0 && clk_wave >= 0) {
    g_clk[8 * clk_wave + 5] = wall_clock64();
    const int nlim = __popc(maskL) + __popc(maskU);
    g_clk[8 * clk_wave + 6] = (unsigned long long)(n + nj + nlim + (gear ? 1 : 0)) | ((unsigned long long)nc << 8) | ((unsigned long long)nt << 16) | (1ull << 29);      /* bit 29: the heavy path */
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_clk[8 * clk_wave + 7] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
#undef HV_CLK
#undef HV_CLK2
}

/* the second-register variant (inlined: as a real call it would make k_solve2's register count the callee's unlimited one - 248, one wave per SIMD - measured) */
__device__ __forceinline__ void heavy_solve_big(const DevModel* m, float* state, const float* w, int env, int* sort_cnt_next, int* sort_slot, int debug_flags, float* lds, float* st_lds, int sort_salt, int clk_wave, int nc_hint) {
#ifdef RP_NO_BIG      /* register experiments only */
  heavy_solve<false>(m, state, w, env, sort_cnt_next, sort_slot, debug_flags, lds, st_lds, sort_salt, clk_wave, nc_hint);
#else
  heavy_solve<true>(m, state, w, env, sort_cnt_next, sort_slot, debug_flags, lds, st_lds, sort_salt, clk_wave, nc_hint);
#endif
}
/* ------------------------------------------------------------------ k_solve2 under debug flag 1 (tests): the dv-form envs by the one-kernel path's own sweeps instead of
 * the four-env path.  solve_rows (k_step's solver: rows streamed from LDS, dot products by wave_sum32) is the bitwise twin of the four-env path's row bodies - this is the
 * test that holds it inside the split pipeline.  The rows come from the workspace (under the flag k_prep2 also files the typed non-contact row list: W3_SROW) in the order
 * contact_rows built them: normals, torsional rows, friction pairs. */
struct __align__(16) SuperLds {
  float st[RP_REC_FLOATS];
  alignas(16) float Minv[144];
  float vstar[32];
  alignas(16) float srow[MAXSMALL * 8];
  alignas(16) float rowS[MAXROWC * 4];
  alignas(16) float rowT[MAXROWC * 4];
  alignas(16) float J[ROWREG];
  alignas(16) float B[ROWREG];
};
__device__ __forceinline__ void super_solve(const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ w, const int env, int* __restrict__ sort_cnt_next,
                                            int* __restrict__ sort_slot, const int debug_flags, SuperLds& L, const int sort_salt) {
  int lane_ = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_));      /* (as in heavy_solve) */
  const int lane = lane_;
  const int n = m->n_arm;
  const float4 h0 = *(const float4*)&w[W3_HDR], h1 = *(const float4*)&w[W3_HDR + 4];
  const int nc = uni(__float_as_int(h0.w)), nt = uni(__float_as_int(h1.z)) >> 8;
  const int nA = uni(__float_as_int(h1.x)), nB = uni(__float_as_int(h1.y)), nC = uni(__float_as_int(h1.w));
  const int nsmall = uni(__float_as_int(w[W3_SROW]));
  const int nrc = 3 * nc + nt;
  {
    const float* r = state + (size_t)env * RP_REC_FLOATS;
    L.st[lane] = r[lane]; L.st[lane + 64] = r[lane + 64];
  }
  for (int i = lane; i < 144; i += 64) L.Minv[i] = w[W3_MINV + i];
  if (lane < 32) L.vstar[lane] = w[W3_VSTAR + lane];
  for (int i = lane; i < 8 * nsmall; i += 64) L.srow[i] = w[W3_SROW + 4 + i];
  auto global_row = [&](int lr) { return lr < nc ? lr : (lr < nc + nt ? 3 * nc + (lr - nc) : nc + (lr - nc - nt)); };
  for (int e = lane; e < nrc * ROWW; e += 64) {
    const int lr = e / ROWW, k = e - lr * ROWW, gr = global_row(lr);
    L.J[e] = w[W3_J + gr * ROWW + k]; L.B[e] = w[W3_B + gr * ROWW + k];
  }
  for (int e = lane; e < nrc * 4; e += 64) {
    const int lr = e >> 2, k = e & 3, gr = global_row(lr);
    float v = w[W3_ROWS + gr * 4 + k];
    if (k == 3 && lr >= nc && lr < nc + nt) v = __int_as_float(__float_as_int(v) & 255);      /* a torsional row's parent contact (packed with its class and rank for the four-env path) */
    L.rowS[e] = v; L.rowT[e] = w[W3_ROWT + gr * 4 + k];
  }
  WSYNC();
  if ((debug_flags & RP_DBG_MOTOR) && lane < nsmall) {      /* first substep after k_action_prep: the motor rows from the record's fresh targets (build_small_rows' formula) */
    float* sr = &L.srow[8 * lane];
    if (__float_as_int(sr[0]) == SR_UNIT) {
      const int dA = __float_as_int(sr[1]);
      const float mode = L.st[ST_MMODE + dA], tgt = L.st[ST_MTARGET + dA], mx = L.st[ST_MMAXIMP + dA], qi = L.st[ST_Q + dA];
      const float des = mode != 0.f ? K_KP * (tgt - qi) / K_DT : 0.f;
      sr[3] = (des - L.vstar[dA]) * sr[4]; sr[5] = -mx; sr[6] = mx;
    }
  }
  WSYNC();
  int sort_pos = 0, sort_bin = 0;
  if (lane == 0 && !(debug_flags & RP_DBG_NOSORT)) {
    const int my_nS = max(nA, nB);
    const int key = 8 * (nC < 7 ? nC : 7) + (my_nS < 1 ? 0 : (my_nS > 14 ? 7 : (my_nS - 1) >> 1));
    sort_bin = key * SORT_REPS + (sort_salt & (SORT_REPS - 1));
    sort_pos = atomicAdd(&sort_cnt_next[sort_bin], 1);
  }
  const float dv = solve_rows<SuperLds>(m, L, lane, nsmall, nc, nt);
  WSYNC();
  /* apply and integrate: substep()'s lines */
  const int dd = lane_dof(m, lane);
  const float vnew = clampf((dd >= 0 ? L.vstar[dd] : 0.f) + dv, -K_MAXVEL, K_MAXVEL);
  if (dd >= 0 && dd < n) { L.st[ST_QD + dd] = vnew; L.st[ST_Q + dd] += K_DT * vnew; }
  else if (dd >= 0 && dd < n + 6 * m->n_free) { const int k = (dd - n) / 6, c = (dd - n) % 6; L.st[ST_FREE + 13 * k + 7 + c] = vnew; }
  else if (dd >= 0) { const int k = dd - n - 6 * m->n_free; L.st[ST_JQD + k] = vnew; L.st[ST_JQ + k] += K_DT * vnew; }
  WSYNC();
  if (lane < m->n_free) {
    float* f = &L.st[ST_FREE + 13 * lane];
    V3 v = ld3(f + 7), wv = ld3(f + 10);
    st3(f, ld3(f) + v * K_DT);
    float wn = norm(wv);
    if (wn > 0.7853981633974483f / K_DT) wn = 0.7853981633974483f / K_DT;
    V3 ax;
    if (wn < 0.001f) ax = wv * (0.5f * K_DT - K_DT * K_DT * K_DT * 0.020833333333f * wn * wn);
    else ax = wv * (sinf(0.5f * wn * K_DT) / wn);
    Q4 dq = {ax.x, ax.y, ax.z, cosf(0.5f * wn * K_DT)};
    Q4 q0 = {f[3], f[4], f[5], f[6]};
    Q4 qn = qmul(dq, q0);
    float nr = 1.f / sqrtf(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
    f[3] = qn.x * nr; f[4] = qn.y * nr; f[5] = qn.z * nr; f[6] = qn.w * nr;
  }
  WSYNC();
  {
    float* r = state + (size_t)env * RP_REC_FLOATS;
    r[lane] = L.st[lane]; r[lane + 64] = L.st[lane + 64];
  }
  if (lane == 0 && !(debug_flags & RP_DBG_NOSORT)) {
    int sp = sort_pos;
    asm volatile("" : "+v"(sp));
    sort_slot[env] = (sort_bin << SORT_RANK_BITS) | sp;
  }
  WSYNC();
}
/* a block's LDS: four state records per wave (the four-env path), or ONE heavy env's tables / one super env's rows (worker blocks; the in-block fallback) */
union __align__(16) BlockLds {
  float rec[SOLVE_WAVES][4 * RP_REC_FLOATS];
  float hv[HV_LDS_FLOATS];
  SuperLds sup;
};

/* The one-kernel twins (k_step, k_reset: substep()) on a heavy env: the rows they built in LDS are laid out as a workspace row - what k_prep2 hands to k_solve2 - and
 * heavy_solve runs on it, integrating the record in L.st.  One implementation of the residual form on the device: the twins check the split pipeline's plumbing around
 * it, the oracle checks the form. */
__device__ bool substep_heavy(const DevModel* m, EnvLds& L, int lane, int nsmall, int ncon, int nt) {
  const int cls = lane < ncon ? L.conk[lane] : 3;
  const int nB = __popcll(__ballot(cls == 0)), nA = __popcll(__ballot(cls == 1)), nC = __popcll(__ballot(cls == 2));
  const int hcls = hv_class(ncon, nA, nB, nC);
  if (hcls == 0) return false;
  __shared__ __align__(16) float w3[W3_FLOATS];
  __shared__ __align__(16) float hv[HV_LDS_FLOATS];
  nsmall = uni(nsmall);
  float* aout = w3 + W3_A;
  for (int i = lane; i < AOUT_FLOATS; i += 64) aout[i] = 0.f;
  WSYNC();
  bool gear = false, lim_lo = false, lim_up = false; int dAl = 0;
  if (lane < nsmall) {      /* the unit rows in the solver's dof-indexed form (prep2_core's wave 1, line for line: signs folded into rhs and bounds) */
    const float* sr = &L.srow[8 * lane];
    const int type = __float_as_int(sr[0]), dA = __float_as_int(sr[1]);
    const float sg = sr[2], rhs = sr[3], dinv = sr[4], lo = sr[5], hi = sr[6];
    dAl = dA;
    if (type == SR_UNIT) { aout[dA] = dinv; aout[16 + dA] = rhs; aout[32 + dA] = lo; aout[48 + dA] = hi; }
    else if (type == SR_LIMIT) {
      const int pl = sg > 0.f ? 64 : 112;
      aout[pl + dA] = sg * rhs; aout[pl + 16 + dA] = sg * lo; aout[pl + 32 + dA] = sg * hi;
      lim_lo = sg > 0.f; lim_up = !(sg > 0.f);
    } else if (type == SR_J1) {
      const int k = lane;
      if (k < NBJ) { float* bq = &aout[168]; bq[k] = dinv; bq[4 + k] = rhs; bq[8 + k] = lo; bq[12 + k] = hi; bq[16 + k] = sg; }
    } else {
      float* g = &aout[160];
      g[0] = sr[1]; g[1] = sr[7]; g[2] = sg; g[3] = dinv; g[4] = rhs; g[5] = lo; g[6] = hi;
      gear = true;
    }
  }
  unsigned mL = 0u, mU = 0u;
  for (int i = 0; i < RP_MAX_ARM; i++) { if (__ballot(lim_lo && dAl == i) != 0ull) mL |= 1u << i; if (__ballot(lim_up && dAl == i) != 0ull) mU |= 1u << i; }
  const bool anygear = __ballot(gear) != 0ull;
  if (lane == 0) {
    const int nj = m->n_j1 < NBJ ? m->n_j1 : NBJ;
    w3[W3_HDR] = __int_as_float((int)mL); w3[W3_HDR + 1] = __int_as_float((int)mU); w3[W3_HDR + 2] = __int_as_float(nj); w3[W3_HDR + 3] = __int_as_float(ncon);
    w3[W3_HDR + 4] = __int_as_float(nA); w3[W3_HDR + 5] = __int_as_float(nB); w3[W3_HDR + 6] = __int_as_float((anygear ? 1 : 0) | (nt << 8)); w3[W3_HDR + 7] = __int_as_float(nC);
  }
  if (lane < 32) { w3[W3_VSTAR + lane] = L.vstar[lane]; w3[W3_MU + lane] = lane < ncon ? L.conmu[lane] : 0.f; }
  for (int i = lane; i < 144; i += 64) w3[W3_MINV + i] = L.Minv[i];
  /* contact rows: built normals, torsional rows, friction pairs (contact_rows); the workspace keeps normals, friction pairs, torsional rows */
  const int nrc = 3 * ncon + nt;
  auto global_row = [&](int lr) { return lr < ncon ? lr : (lr < ncon + nt ? 3 * ncon + (lr - ncon) : ncon + (lr - ncon - nt)); };
  for (int e = lane; e < nrc * ROWW; e += 64) {
    const int lr = e / ROWW, k = e - lr * ROWW, gr = global_row(lr);
    w3[W3_J + gr * ROWW + k] = L.J[e]; w3[W3_B + gr * ROWW + k] = L.B[e];
  }
  for (int e = lane; e < nrc * 4; e += 64) {
    const int lr = e >> 2, k = e & 3, gr = global_row(lr);
    float v = L.rowS[e];
    if (k == 3 && lr >= ncon && lr < ncon + nt) v = __int_as_float(L.torc[lr - ncon]);      /* a torsional row's parent contact (heavy_solve reads the low byte) */
    w3[W3_ROWS + gr * 4 + k] = v; w3[W3_ROWT + gr * 4 + k] = L.rowT[e];
  }
  WSYNC();
  if (hcls == 1) heavy_solve<false>(m, nullptr, w3, 0, nullptr, nullptr, RP_DBG_NOSORT, hv, L.st, 0);
  else heavy_solve_big(m, nullptr, w3, 0, nullptr, nullptr, RP_DBG_NOSORT, hv, L.st, 0, -1, -1);
  return true;
}

/* The classes of a block's four envs (hv_class), decided from the same headers by both waves alike: masks of the places 4 bq + k whose env the four-env path takes,
 * the heavy path (residual form), the heavy path with its second lane register (more than HV_MAXC contacts) */
__device__ __forceinline__ void block_classes(const float* __restrict__ ws, int env0, int N, const int* __restrict__ pair_env, const int bq, unsigned& light, unsigned& res, unsigned& super, int& env_g) {
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const int place = bq * 4 + g;
  const int pe = place < N - env0 ? pair_env[env0 + place] : -1;
  const int envm = pair_env_id(pe);
  const bool valid = pe >= 0;
  const float* w = ws + (size_t)(valid ? envm : 0) * W3_FLOATS;
  const float4 h0 = *(const float4*)&w[W3_HDR], h1 = *(const float4*)&w[W3_HDR + 4];
  const int cls = valid ? hv_class(__float_as_int(h0.w), __float_as_int(h1.x), __float_as_int(h1.y), __float_as_int(h1.w)) : -1;
  auto pack = [](unsigned long long b) { return (unsigned)((b & 1ull) | ((b >> 15) & 2ull) | ((b >> 30) & 4ull) | ((b >> 45) & 8ull)); };
#ifdef RP_S4_OFF            /* timing ablation: no block takes the four-env path */
  light = 0u; super = pack(__ballot(cls == 0 || cls == 2));
#else
  light = pack(__ballot(cls == 0)); super = pack(__ballot(cls == 2));
#endif
  res = pack(__ballot(cls == 1));
  env_g = valid ? envm : -1;
}

#define SOLVE2_ARGS const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ ws, int env0, int N, \
                    const int* __restrict__ pair_env, int* __restrict__ sort_cnt_next, int* __restrict__ sort_slot, int debug_flags
#if defined(RP_CLOCKS) && RP_CLOCKS != 2      /* profiling build: wall clock at both ends of a four-env wave, bit 30 of word 6 = "took the four-env path" */
#define S4_CLK(i) if ((threadIdx.x & 63) == 0) { const int wb_ = blockIdx.x * SOLVE_WAVES + (threadIdx.x >> 6); g_clk[8 * wb_ + (i)] = wall_clock64(); g_clk[8 * wb_ + 6] |= 1ull << 30; }
#else
#define S4_CLK(i)
#endif
/* one block's solve: bq = its number among the blocks of the launch (its four envs: places 4 bq .. 4 bq + 3 of the pairing table), Ls = its LDS */
__device__ __forceinline__ void solve_block(const DevModel* __restrict__ m, float* __restrict__ state, const float* __restrict__ ws, int env0, int N,
                                            const int* __restrict__ pair_env, int* __restrict__ sort_cnt_next, int* __restrict__ sort_slot, int debug_flags, BlockLds& Ls, const int bq) {
  unsigned light, res, super; int env_g;
  block_classes(ws, env0, N, pair_env, bq, light, res, super, env_g);
  const int wid = threadIdx.x >> 6;
  const unsigned big = super;                              /* class 2: the heavy path with its second lane register */
  super = 0u;
  if (debug_flags & RP_DBG_SEQ) { super = light; light = 0u; }      /* (tests) the dv-form envs by the other implementation of the dv form: the one-kernel path's solver on the workspace's rows */
  if (light != 0u) {
    S4_CLK(4)
    if (wid == 0) solve4_body<0>(m, state, ws, env0, N, pair_env, sort_cnt_next, sort_slot, debug_flags, Ls.rec[0], bq, light);
    else solve4_body<1>(m, state, ws, env0, N, pair_env, sort_cnt_next, sort_slot, debug_flags, Ls.rec[1], bq, light);
    S4_CLK(5)
  }
  if (debug_flags & RP_DBG_WORKERS) return;                /* the launch's worker blocks take the other envs (k_solve2) */
  /* (k_chain, the settle substeps of rp_reset: no worker blocks) this block's other envs, one after the other, by its first wave */
  const unsigned rest = res | big | super;
  if (rest == 0u) return;                                  /* (block-uniform) */
  __syncthreads();                                         /* the state records of both waves' four-env pass are done with */
  if (wid == 0) {
#pragma unroll 1
    for (int g = 0; g < 4; g++) {
      if (!((rest >> g) & 1u)) continue;
      const int env = __builtin_amdgcn_readlane(env_g, 16 * g);
      if ((res >> g) & 1u) heavy_solve<false>(m, state, ws + (size_t)env * W3_FLOATS, env, sort_cnt_next, sort_slot, debug_flags, Ls.hv, nullptr, 4 * bq + g);
      else if ((big >> g) & 1u) heavy_solve_big(m, state, ws + (size_t)env * W3_FLOATS, env, sort_cnt_next, sort_slot, debug_flags, Ls.hv, nullptr, 4 * bq + g, -1, -1);
      else super_solve(m, state, ws + (size_t)env * W3_FLOATS, env, sort_cnt_next, sort_slot, debug_flags, Ls.sup, 4 * bq + g);
      WSYNC();
    }
  }
}
/* k_solve2's grid: HB worker blocks in front - each takes heavy envs of the launch off the list the k_prep2 before it appended them to (hv_list: env | contact count << 24 at
 * places env0 .. of its group, *hv_cnt of them; under debug flag 1 every env is on it), ONE ENV PER BLOCK at a time (its first wave; the block's LDS is one env's tables),
 * block i the entries i, i + HB, ... - and behind them the blocks of four places each.  The heavy envs are the launch's long poles: they start first, and nobody waits
 * in line behind a partner. */
__global__ void __launch_bounds__(64 * SOLVE_WAVES, RP_SOLVE_WAVES_PER_EU) k_solve2(SOLVE2_ARGS, const int* __restrict__ hv_cnt, const int* __restrict__ hv_list, int HB) {
  __shared__ BlockLds Ls;
  if ((int)blockIdx.x < HB) {
    if ((threadIdx.x >> 6) != 0) return;
    const int nH = __builtin_amdgcn_readfirstlane(*hv_cnt);
#if defined(RP_CLOCKS) && RP_CLOCKS == 1
    if ((threadIdx.x & 63) == 0) { const int wb_ = blockIdx.x * SOLVE_WAVES; for (int q = 0; q < 8; q++) g_clk[8 * wb_ + q] = 0ull; g_clk[8 * wb_ + 6] = (1ull << 28) | (unsigned long long)nH; }      /* bit 28: a worker wave (without bit 29: it found no env) */
#endif
#pragma unroll 1
    for (int i = blockIdx.x; i < nH; i += HB) {
      const int pe = __builtin_amdgcn_readfirstlane(hv_list[env0 + i]);
      const int env = __builtin_amdgcn_readfirstlane(pair_env_id(pe)), pnc = pe >> 24;
      const float* w = ws + (size_t)env * W3_FLOATS;
      const float4 h1 = *(const float4*)&w[W3_HDR + 4];
      const int cls = hv_class(pnc, uni(__float_as_int(h1.x)), uni(__float_as_int(h1.y)), uni(__float_as_int(h1.w)));
      if (cls == 1) heavy_solve<false>(m, state, w, env, sort_cnt_next, sort_slot, debug_flags, Ls.hv, nullptr, i, blockIdx.x * SOLVE_WAVES, pnc);
      else if (cls == 2) heavy_solve_big(m, state, w, env, sort_cnt_next, sort_slot, debug_flags, Ls.hv, nullptr, i, blockIdx.x * SOLVE_WAVES, pnc);
      else super_solve(m, state, w, env, sort_cnt_next, sort_slot, debug_flags, Ls.sup, i);      /* (debug flag 1: a dv-form env) */
      WSYNC();
    }
    return;
  }
  solve_block(m, state, ws, env0, N, pair_env, sort_cnt_next, sort_slot, debug_flags, Ls, blockIdx.x - HB);
}
__global__ void __launch_bounds__(64 * SOLVE_WAVES, RP_SOLVE_WAVES_PER_EU) k_settle_solve(SOLVE2_ARGS) {
  __shared__ BlockLds Ls;
  solve_block(m, state, ws, env0, N, pair_env, sort_cnt_next, sort_slot, debug_flags, Ls, blockIdx.x);
}

/* ------------------------------------------------------------------ k_chain: all substeps of a step in ONE launch (round 4's experiment, rp_set_fused(h, 2)).
 * A block of two waves owns the same four envs (places 4 q .. 4 q + 3 of the load ranking `member`) for all nsub substeps and alternates inside itself between their
 * preparation - prep2_core, one env after the other: both waves work on one env as in k_prep2 - and their solve - solve_block, the four-env path or the two-env path as
 * their contacts demand.  The rows go through the same workspace as in the split pipeline (written and read by the same CU: L2 / L1 resident) and the pairing table is
 * the identity on the ranking.  No launch boundary between substeps, so no launch-wide barrier: a block's heavy substeps add to its own chain and to nobody else's.
 * What it costs: one register / LDS footprint for both phases - the solve's (217 VGPRs, 40 KB: two waves per SIMD) - where k_prep2 runs at four waves per SIMD.
 * Same row bodies on the same rows in the same order: the same bits as the split pipeline (tests). */
struct __align__(16) ChainLds { union { PrepLds P; BlockLds S; }; };
#ifdef RP_CHAIN_CLOCKS
__device__ long long g_chain_clk[4 * 4096];
#endif
/* the two phases as REAL calls: inlined into one loop body, everything they derive from the thread index and the model - hundreds of values - is hoisted in front of the
 * substep loop and spilled (256 VGPRs + 580 bytes of scratch); a call keeps each phase's registers its own */
/* (a called function gets its arguments in VGPRs; they are wave-uniform, and the bodies keep their guards in SGPRs: made scalar again here) */
template <class T> __device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long v = (unsigned long long)p;
  return (T*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
__device__ __attribute__((noinline)) void chain_prep(PrepLds* L, const DevModel* m, const float* state, float* ws, int env, int* pair_tab, int place, int prep_flags) {
  prep2_core(*uniform_ptr(L), uniform_ptr(m), uniform_ptr(state), uniform_ptr(ws), uni(env), uni(env), uniform_ptr(pair_tab), uni(place), nullptr, nullptr, uni(prep_flags));
}
__device__ __attribute__((noinline)) void chain_solve(const DevModel* m, float* state, const float* ws, int N, const int* pair_tab, int* sort_cnt_next, int* sort_slot, int flags, BlockLds* Ls, int q) {
  solve_block(uniform_ptr(m), uniform_ptr(state), uniform_ptr(ws), 0, uni(N), uniform_ptr(pair_tab), uniform_ptr(sort_cnt_next), uniform_ptr(sort_slot), uni(flags), *uniform_ptr(Ls), uni(q));
}
__global__ void __launch_bounds__(64 * SOLVE_WAVES, 2) k_chain(const DevModel* __restrict__ m, float* state, float* ws, int N, const int* __restrict__ member,
                                                              int* pair_tab, int* __restrict__ sort_cnt_next, int* __restrict__ sort_slot, int nsub, int debug_flags) {
  __shared__ ChainLds L;
  static_assert(PREP_THREADS == 64 * SOLVE_WAVES, "one block shape for both phases");
  const int nq = (N + 3) >> 2;
#ifdef RP_CHAIN_CLOCKS      /* profiling build: wall clock (100 MHz) spent in the two phases, per block: g_chain_clk[4 b] = prep, [4 b + 1] = solve, [4 b + 2] = start, [4 b + 3] = end */
#define CHCLK(i, sign) if (threadIdx.x == 0) g_chain_clk[4 * blockIdx.x + (i)] += (sign) * (long long)wall_clock64();
  if (threadIdx.x == 0) { g_chain_clk[4 * blockIdx.x] = 0; g_chain_clk[4 * blockIdx.x + 1] = 0; g_chain_clk[4 * blockIdx.x + 2] = (long long)wall_clock64(); }
#else
#define CHCLK(i, sign)
#endif
  for (int sub = 0; sub < nsub; sub++) {
    const int flags = debug_flags | (sub + 1 < nsub ? 4 : 0);      /* load classes for the next step's ranking: from the last substep only */
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
      CHCLK(0, -1)
      for (int k = 0; k < 4; k++) {
        const int place = 4 * q + k;
        if (place < N) {                                           /* (block-uniform) */
          const int env = member ? member[place] : place;
          chain_prep(&L.P, m, state, ws, env, pair_tab, place, debug_flags & 1);
        }
        __syncthreads();                                           /* the next env's preparation (or the solve) takes the LDS over; this env's rows and its table entry are visible to the block */
      }
      CHCLK(0, 1) CHCLK(1, -1)
      chain_solve(m, state, ws, N, pair_tab, sort_cnt_next, sort_slot, flags, &L.S, q);
      __syncthreads();                                             /* the records are written: the next substep's preparation may read them */
      CHCLK(1, 1)
    }
  }
#ifdef RP_CHAIN_CLOCKS
  if (threadIdx.x == 0) g_chain_clk[4 * blockIdx.x + 3] = (long long)wall_clock64();
#endif
}


/* first pairing of a group's envs (before any load class is known): everything in the lightest class, in index order */
__global__ void k_sort_init(int* __restrict__ cnt, int* __restrict__ slot, int env0, int ng) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < SORT_BINS) cnt[i] = i < SORT_REPS ? (ng + i) / SORT_REPS : 0;              /* envs with (e & 7) == 7 - i */
  if (i < ng) slot[env0 + i] = ((SORT_REPS - 1 - (i & (SORT_REPS - 1))) << SORT_RANK_BITS) | (i >> 3);      /* bins 7..0 <-> i & 7 = 0..7: keeps index order */
}

/* Env groups by load.  rp_step cuts the envs into G groups that run their kernel chains on separate streams; a chain lasts as
 * long as its heaviest env keeps k_solve2 busy, so the groups are cut by load: before a step, all envs are ranked by the load
 * class the last k_solve2 gave them (the per-group counting sorts, merged: heavier bins first, inside a bin group by group, then
 * the rank inside the group's bin), place p of the ranking -> member[p], and group g owns places [N g / G, N (g + 1) / G): the
 * heaviest envs share group 0, whose chain is the critical path, while the lighter groups finish early and leave the machine to
 * it.  Inside a group the ranking order is kept for the first pairing (k_sort_init's layout).  One block; results never depend
 * on membership or pairing. */
struct GroupBounds { int b[RP_MAX_GROUPS + 1]; };     /* group g owns places [b[g], b[g + 1]) */
__global__ void __launch_bounds__(1024) k_member(const int* __restrict__ member_old, int* __restrict__ member_new, int* __restrict__ cnt,
                                                 int* __restrict__ sort_slot, int N, int G_old, GroupBounds bo, int G_new, GroupBounds bn, int* __restrict__ hv_cnt_all) {
  __shared__ int tot[SORT_BINS], above[SORT_BINS], gpre[RP_MAX_GROUPS * SORT_BINS];
  const int t = threadIdx.x;
  if (t < 2 * RP_MAX_GROUPS) hv_cnt_all[t] = 0;      /* every group's two heavy-env counters start the step at zero */
  for (int b = t; b < SORT_BINS; b += 1024) {
    int sum = 0;
    for (int g = 0; g < G_old; g++) { gpre[g * SORT_BINS + b] = sum; sum += cnt[g * SORT_BINS + b]; }
    tot[b] = sum;
  }
  __syncthreads();
  if (t < 64) {                          /* envs in heavier bins: suffix sums over the 512 bins, 8 bins per lane of one wave */
    int loc[8], sum = 0;
#pragma unroll
    for (int k = 7; k >= 0; k--) { loc[k] = sum; sum += tot[8 * t + k]; }
    int suf = sum;                        /* inclusive suffix over lanes: sum of lanes >= t */
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_down(suf, d); if (t + d < 64) suf += o; }
#pragma unroll
    for (int k = 0; k < 8; k++) above[8 * t + k] = loc[k] + suf - sum;
  }
  __syncthreads();
  for (int p = t; p < N; p += 1024) {
    const int env = member_old[p];
    int g = 0;
    while (g + 1 < G_old && p >= bo.b[g + 1]) g++;
    const int v = sort_slot[env], b = v >> SORT_RANK_BITS;
    member_new[above[b] + gpre[g * SORT_BINS + b] + (v & SORT_RANK_MASK)] = env;
  }
  __threadfence_block();
  __syncthreads();
  for (int i = t; i < G_new * SORT_BINS; i += 1024) {
    const int g = i / SORT_BINS, b = i % SORT_BINS;
    const int ng = bn.b[g + 1] - bn.b[g];
    cnt[i] = b < SORT_REPS ? (ng + b) / SORT_REPS : 0;
  }
  for (int p = t; p < N; p += 1024) {
    int g = 0;
    while (g + 1 < G_new && p >= bn.b[g + 1]) g++;
    const int i = p - bn.b[g];
    sort_slot[member_new[p]] = ((SORT_REPS - 1 - (i & (SORT_REPS - 1))) << SORT_RANK_BITS) | (i >> 3);
  }
}
__global__ void k_member_identity(int* __restrict__ member, int N, int* __restrict__ hv_cnt_all) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * RP_MAX_GROUPS) hv_cnt_all[i] = 0;
  if (i < N) member[i] = i;
}

/* debug: one substep for every env, dumping intermediates of env `dbg_env` (tests only) */
__global__ void __launch_bounds__(64) k_debug_substep(const DevModel* __restrict__ m, float* __restrict__ state, float* __restrict__ dbg, int N, int dbg_env) {
  __shared__ EnvLds L;
  int env = blockIdx.x, lane = threadIdx.x;
  if (env >= N) return;
  load_state(L, state, env, lane);
  int n = m->n_arm;
  fk_bodies(m, L, lane);
  __syncthreads();
  joint_subspaces(m, L, lane);
  collider_aabbs(m, L, lane);
  __syncthreads();
  int ncon = collide(m, L, lane, env);
  if (env == dbg_env && lane == 0) {
    dbg[0] = (float)ncon;
    for (int c = 0; c < ncon; c++) {
      float* o = dbg + 16 + 9 * c;
      o[0] = (float)L.cona[c]; o[1] = (float)L.conb[c];
      for (int k = 0; k < 3; k++) { o[2 + k] = L.conp[3 * c + k]; o[5 + k] = L.conn[3 * c + k]; }
      o[8] = L.cond[c];
    }
  }
  arm_dynamics(m, L, lane);
  unconstrained_velocities(m, L, lane);
  if (env == dbg_env && lane == 0) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) dbg[320 + i * 12 + j] = L.Minv[i * 12 + j];
    for (int i = 0; i < 32; i++) dbg[480 + i] = L.vstar[i];
    for (int i = 0; i < n; i++) dbg[512 + i] = L.tau[i];
  }
  int nsmall = build_small_rows(m, L, lane);
  const int nt = tors_list(m, L, lane, ncon);
  WSYNC();
  contact_rows(m, L, lane, 0, ncon, nt);
  __syncthreads();
  float dv = solve_rows(m, L, lane, nsmall, ncon, nt);
  if (env == dbg_env) {
    if (lane == 0) { dbg[1] = (float)nsmall; }
    int dd = lane_dof(m, lane);
    if (dd >= 0) dbg[544 + dd] = dv;
  }
  __syncthreads();
}
