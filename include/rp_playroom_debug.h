/* rp_playroom_debug.h — test / tuning hooks exported by librp_playroom_hip.so next to the C ABI of rp_playroom.h.
 *
 * Nothing here stands in for a reference interface: these exist so that the parity tests can compare the library's two step
 * pipelines with each other and with the oracle at intermediate points, and so that profiling scripts can pin the launch
 * layout.  None of them changes a result (both pipelines and every group / slot layout are bit-identical; tests/ check it).
 * tests/test_abi.py checks that every rp_* symbol the library exports is declared in one of the two headers. */
#ifndef RP_PLAYROOM_DEBUG_H
#define RP_PLAYROOM_DEBUG_H
#include "rp_playroom.h"

#ifdef __cplusplus
extern "C" {
#endif

/* step pipeline: 0 = split kernels (default), 1 = one fused kernel per env step (k_step; the in-library cross-check), 2 = the twelve substeps of a step in one launch
 * (k_chain, round 4's experiment: blocks of two waves own four envs for the whole step; grid = RP_CHAIN_BLOCKS read at rp_create, default 1024).  All three produce the
 * same bits (tests/).  With fused = 1 and timers enabled rp_step synchronises on its own kernel to read the timer; fused = 2 has no per-launch timers: rp_set_fused(h, 2)
 * with timers on, and rp_enable_timers(h, n > 0) under fused = 2, return RP_ERR_ARG. */
int rp_set_fused(rp_handle h, int32_t fused);
/* number of env groups (streams) of the split pipeline, 1 .. 16 */
int rp_set_groups(rp_handle h, int32_t groups);
/* bit 0: the solver gives every contact its own folded slot (its fallback layout) instead of solving arm-only and non-arm
 * contacts side by side */
int rp_set_debug_flags(rp_handle h, int32_t flags);
/* one fused substep on every env on the NULL stream; intermediates of env `env` into host_buf[4096] (layout: k_debug_substep) */
int rp_debug_substep(rp_handle h, int32_t env, float* host_buf);
/* per env of the most recent k_prep2: host_buf[2 e] = unit rows, host_buf[2 e + 1] = contacts + 1000 * (arm contact) + 100000 *
 * spanning contacts; synchronises the device */
int rp_debug_row_counts(rp_handle h, int32_t* host_buf);
/* the joints [num_envs][8] of the ghost arm(s) drawn by the latest rp_render_ex with a ghost_arm (environments.py:575-590: rest pose, one IK call, joints [0:6]) */
int rp_debug_ghost_joints(rp_handle h, float* host_buf, int32_t num_envs);
/* rounds the most recent rp_reset took */
int rp_debug_reset_rounds(rp_handle h);
/* PROFILING BUILDS ONLY (tools/build_profiling_libs.sh: -DRP_CLOCKS=1|2, -DRP_PROLOGUE_CLOCKS, -DRP_CHAIN_CLOCKS).  The shipped library exports none of these and
 * tests/test_abi.py does not expect them; such a build also exports, for the tools that load it through RP_PLAYROOM_LIB:
 *     rp_debug_clocks(rp_handle, uint64_t* host_buf, int32_t nwaves)              s_memtime marks per wave of k_solve2 (RP_CLOCKS=1) / per block of k_prep2 (=2)
 *     rp_debug_prologue_clocks(rp_handle, uint64_t* host_buf, int32_t nwaves)     marks of k_solve2's prologue
 *     rp_debug_chain_clocks(rp_handle, int64_t* host_buf, int32_t nblocks)        per block of k_chain: prepare / solve totals
 * each copies the marks of the most recent launch to the host after a device synchronise. */

#ifdef __cplusplus
}
#endif
#endif
