// Micro-benchmark: candidate instruction sequences for one PGS row update (gfx950); cycles per row, 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define LOOPS 200
#define DPPM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// operands: %0 dv  %1 p/tmp  %2 lamreg  %3 rhsreg  %4 J  %5 B  %6 s  %7 lam  %8 lim  %9 limreg  %10 hi  (s[10:11] = lane mask, set up front)
#define KERNEL(name, body)                                                                  \
  __global__ void __launch_bounds__(64) name(float* out, unsigned long long* clk, float seed) { \
    float dv = seed + threadIdx.x, p = 0.f, lamreg = seed * 0.5f, rhsreg = seed * 0.25f, J = 0.001f, B = 0.002f;   \
    float s = 0.f, lam = 0.f, lim = 0.f, limreg = 0.125f, hi = 0.f;                          \
    asm volatile("s_mov_b32 s10, 0x00010001\n s_mov_b32 s11, 0x00010001" ::: "s10", "s11"); \
    unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < LOOPS; i++) { REP64(asm volatile(body : "+v"(dv), "+v"(p), "+v"(lamreg), "+v"(rhsreg), "+v"(J), "+v"(B), "+v"(s), "+v"(lam), "+v"(lim), "+v"(limreg), "+v"(hi));) } \
    unsigned long long t1 = __builtin_readcyclecounter();                                    \
    out[blockIdx.x * 64 + threadIdx.x] = dv + p + lamreg + s + lam + lim + hi;               \
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                         \
  }

/* R0: bare chain (no impulse bookkeeping) */
KERNEL(r0_bare,
  "v_mul_f32 %1, %0, %4\n"
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2]" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1]" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_half_mirror" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_mirror" DPPM
  "v_sub_f32 %1, %6, %1\n v_med3_f32 %1, %1, %8, %10\n v_sub_f32 %1, %1, %7\n v_mul_f32 %1, %1, %5\n v_add_f32 %0, %0, %1\n")
/* R1: as the compiler emits it today (friction row): fused dpp subrev on the chain */
KERNEL(r1_compiler,
  "v_mul_f32 %1, %4, %0\n"
  "v_mov_b32_dpp %7, %3 row_newbcast:13" DPPM
  "s_nop 0\n"
  "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2]" DPPM
  "v_add_f32_dpp %6, %2, %7 row_newbcast:13" DPPM
  "s_nop 0\n"
  "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1]" DPPM
  "s_nop 1\n"
  "v_add_f32_dpp %1, %1, %1 row_half_mirror" DPPM
  "s_nop 1\n"
  "v_add_f32_dpp %1, %1, %1 row_mirror" DPPM
  "v_sub_f32 %1, %6, %1\n"
  "v_med3_f32 %1, %1, -%8, %10\n"
  "v_cndmask_b32_e64 %2, %2, %1, s[10:11]\n"
  "v_mov_b32_dpp %8, %9 row_newbcast:6" DPPM
  "v_subrev_f32_dpp %1, %2, %1 row_newbcast:13" DPPM
  "v_mul_f32 %1, %5, %1\n"
  "v_add_f32 %0, %0, %1\n"
  "v_add_f32 %10, 0, %8\n")
/* R2: hand order: fillers sit in the DPP hazard slots, lam kept in a register, nothing but plain ops after the reduction */
KERNEL(r2_hand,
  "v_mul_f32 %1, %4, %0\n"
  "v_mov_b32_dpp %7, %2 row_newbcast:13" DPPM
  "v_mov_b32_dpp %8, %9 row_newbcast:6" DPPM
  "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2]" DPPM
  "v_add_f32_dpp %6, %3, %7 row_newbcast:13" DPPM
  "v_add_f32 %10, 0, %8\n"
  "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1]" DPPM
  "s_nop 1\n"
  "v_add_f32_dpp %1, %1, %1 row_half_mirror" DPPM
  "s_nop 1\n"
  "v_add_f32_dpp %1, %1, %1 row_mirror" DPPM
  "v_sub_f32 %1, %6, %1\n"
  "v_med3_f32 %1, %1, -%8, %10\n"
  "v_cndmask_b32_e64 %2, %2, %1, s[10:11]\n"
  "v_sub_f32 %1, %1, %7\n"
  "v_mul_f32 %1, %5, %1\n"
  "v_add_f32 %0, %0, %1\n")
/* R3: R2 without any bookkeeping fillers but with the nops (lower bound for this chain shape) == R0; R3 = R0 with s_nop 0 only (invalid hazards, timing only) */
KERNEL(r3_nop0,
  "v_mul_f32 %1, %0, %4\n"
  "s_nop 0\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2]" DPPM
  "s_nop 0\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1]" DPPM
  "s_nop 0\n v_add_f32_dpp %1, %1, %1 row_half_mirror" DPPM
  "s_nop 0\n v_add_f32_dpp %1, %1, %1 row_mirror" DPPM
  "v_sub_f32 %1, %6, %1\n v_med3_f32 %1, %1, %8, %10\n v_sub_f32 %1, %1, %7\n v_mul_f32 %1, %1, %5\n v_add_f32 %0, %0, %1\n")
/* R4: unit row (one non-zero): the dot product is a single broadcast multiply */
KERNEL(r4_unit,
  "s_nop 1\n v_mul_f32_dpp %1, %0, %4 row_newbcast:5" DPPM
  "v_sub_f32 %1, %6, %1\n v_med3_f32 %1, %1, %8, %10\n v_sub_f32 %1, %1, %7\n v_mul_f32 %1, %1, %5\n v_add_f32 %0, %0, %1\n")
/* R5: 8-lane reduction (3 butterfly steps) */
KERNEL(r5_red8,
  "v_mul_f32 %1, %0, %4\n"
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2]" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1]" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_half_mirror" DPPM
  "v_sub_f32 %1, %6, %1\n v_med3_f32 %1, %1, %8, %10\n v_sub_f32 %1, %1, %7\n v_mul_f32 %1, %1, %5\n v_add_f32 %0, %0, %1\n")
/* R6: chain tail only (no reduction): sub med3 sub mul add mul */
KERNEL(r6_tail,
  "v_mul_f32 %1, %0, %4\n v_sub_f32 %1, %6, %1\n v_med3_f32 %1, %1, %8, %10\n v_sub_f32 %1, %1, %7\n v_mul_f32 %1, %1, %5\n v_add_f32 %0, %0, %1\n")
/* R7: fused tail: dv += B*(lnew-lam) as v_fma (needs contraction; timing only) */
KERNEL(r7_fma,
  "v_mul_f32 %1, %0, %4\n"
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2]" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1]" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_half_mirror" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_mirror" DPPM
  "v_sub_f32 %1, %6, %1\n v_med3_f32 %1, %1, %8, %10\n v_sub_f32 %1, %1, %7\n v_fma_f32 %0, %1, %5, %0\n")
/* R8: SEQ row: 32-lane fold */
KERNEL(r8_seq,
  "v_mul_f32 %1, %0, %4\n"
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2]" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1]" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_half_mirror" DPPM
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_mirror" DPPM
  "v_mov_b32 %6, %1\n s_nop 1\n v_permlane16_swap_b32 %1, %6\n v_add_f32 %1, %1, %6\n"
  "v_sub_f32 %1, %6, %1\n v_med3_f32 %1, %1, %8, %10\n v_sub_f32 %1, %1, %7\n v_mul_f32 %1, %1, %5\n v_add_f32 %0, %0, %1\n")

struct K { const char* name; void (*fn)(float*, unsigned long long*, float); };
int main() {
  K ks[] = {{"R0 bare chain (4 dpp + nops)", r0_bare}, {"R1 compiler order (friction row)", r1_compiler}, {"R2 hand order (fillers in hazard slots)", r2_hand},
            {"R3 bare chain, s_nop 0 (invalid)", r3_nop0}, {"R4 unit row (bcast multiply)", r4_unit}, {"R5 8-lane reduction", r5_red8},
            {"R6 tail only (no reduction)", r6_tail}, {"R7 fma tail", r7_fma}, {"R8 SEQ 32-lane fold", r8_seq}};
  float* out; unsigned long long* clk;
  hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&clk, 8192 * 8);
  std::vector<unsigned long long> h(8192);
  for (int wps : {1, 2, 4}) {
    int blocks = 1024 * wps;
    printf("== %d wave(s) per SIMD ==\n", wps);
    for (auto& k : ks) {
      for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; unsigned long long mx = 0;
      for (int i = 0; i < blocks; i++) { s += h[i]; if (h[i] > mx) mx = h[i]; }
      printf("%-44s mean %.1f cyc/row (max wave %.1f)\n", k.name, s / blocks / (double)(LOOPS * 64), mx / (double)(LOOPS * 64));
    }
  }
  return 0;
}
