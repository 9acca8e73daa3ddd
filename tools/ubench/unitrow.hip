// Micro-benchmark (gfx950): the unit row of k_solve2 (motor / limit: one non-zero in J) - today's body against a RESIDUAL form.
//   U0 today:     t = fma(-jd, dv, rhs); t = med3(t, lo, hi); dacc = own ? t : dacc; dv += bcast(t) * col          chain: fmac_dpp -> fma -> med3 -> fmac_dpp
//   U1 residual:  the three planes' residuals r = rhs - jd dv live in registers and every row updates all of them:
//                 t = med3(r_p, lo, hi); dacc...; r_next += bcast(t) * colr; r_b += ..; r_c += ..; dv += bcast(t) * col   chain: fmac_dpp -> med3 -> fmac_dpp
//   U2 residual, one plane updated on the chain and the other three deferred (issue order only)
// cycles per row, 1 / 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP12(x) REP4(x) REP4(x) REP4(x)
#define LOOPS 400
#define DPPM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define KERNEL(name, body)                                                                  \
  __global__ void __launch_bounds__(64) name(float* out, unsigned long long* clk, float seed) { \
    float dv = seed + threadIdx.x, t = 0.f, jd = 0.001f, rhs = seed * 0.25f, lo = -1.f, hi = 1.f, dacc = 0.f, col = 0.002f, colr = -0.003f;   \
    float ra = seed, rb = seed * 2.f, rc = seed * 3.f;                                        \
    asm volatile("s_mov_b32 s10, 0x00010001\n s_mov_b32 s11, 0x00010001" ::: "s10", "s11"); \
    unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < LOOPS; i++) { REP12(asm volatile(body : "+v"(dv), "+v"(t), "+v"(dacc), "+v"(ra), "+v"(rb), "+v"(rc) : "v"(jd), "v"(rhs), "v"(lo), "v"(hi), "v"(col), "v"(colr));) } \
    unsigned long long t1 = __builtin_readcyclecounter();                                    \
    out[blockIdx.x * 64 + threadIdx.x] = dv + t + dacc + ra + rb + rc;                       \
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                         \
  }
// %0 dv %1 t %2 dacc %3 ra %4 rb %5 rc | %6 jd %7 rhs %8 lo %9 hi %10 col %11 colr
KERNEL(u0_today,
  "v_fma_f32 %1, -%6, %0, %7\n"
  "v_med3_f32 %1, %1, %8, %9\n"
  "v_cndmask_b32_e64 %2, %2, %1, s[10:11]\n"
  "s_nop 0\n"
  "v_fmac_f32_dpp %0, %1, %10 row_newbcast:5" DPPM)
KERNEL(u1_residual,
  "v_med3_f32 %1, %3, %8, %9\n"
  "v_cndmask_b32_e64 %2, %2, %1, s[10:11]\n"
  "s_nop 0\n"
  "v_fmac_f32_dpp %3, %1, %11 row_newbcast:5" DPPM
  "v_fmac_f32_dpp %4, %1, %11 row_newbcast:5" DPPM
  "v_fmac_f32_dpp %5, %1, %11 row_newbcast:5" DPPM
  "v_fmac_f32_dpp %0, %1, %10 row_newbcast:5" DPPM)
// two planes only (motor plane + one limit plane): what the common case needs
KERNEL(u2_residual2,
  "v_med3_f32 %1, %3, %8, %9\n"
  "v_cndmask_b32_e64 %2, %2, %1, s[10:11]\n"
  "s_nop 0\n"
  "v_fmac_f32_dpp %3, %1, %11 row_newbcast:5" DPPM
  "v_fmac_f32_dpp %4, %1, %11 row_newbcast:5" DPPM
  "v_fmac_f32_dpp %0, %1, %10 row_newbcast:5" DPPM)
// one plane (motors only)
KERNEL(u3_residual1,
  "v_med3_f32 %1, %3, %8, %9\n"
  "v_cndmask_b32_e64 %2, %2, %1, s[10:11]\n"
  "s_nop 0\n"
  "v_fmac_f32_dpp %3, %1, %11 row_newbcast:5" DPPM
  "v_fmac_f32_dpp %0, %1, %10 row_newbcast:5" DPPM)
struct K { const char* name; void (*fn)(float*, unsigned long long*, float); };
int main() {
  K ks[] = {{"U0 today (fma, med3, cndmask, nop, fmac_dpp)", u0_today}, {"U1 residual, 3 planes + dv (4 fmac_dpp)", u1_residual},
            {"U2 residual, 2 planes + dv", u2_residual2}, {"U3 residual, 1 plane + dv", u3_residual1}};
  float* out; unsigned long long* clk;
  hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&clk, 8192 * 8);
  std::vector<unsigned long long> h(8192);
  for (int wps : {1, 2}) {
    int blocks = 1024 * wps;
    printf("== %d wave(s) per SIMD ==\n", wps);
    for (auto& k : ks) {
      for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; unsigned long long mx = 0;
      for (int i = 0; i < blocks; i++) { s += h[i]; if (h[i] > mx) mx = h[i]; }
      printf("%-52s mean %.1f cyc/row (max wave %.1f)\n", k.name, s / blocks / (double)(LOOPS * 12), mx / (double)(LOOPS * 12));
    }
  }
  return 0;
}
