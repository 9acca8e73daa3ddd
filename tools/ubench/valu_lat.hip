// Micro-benchmark: issue/latency cost of the instructions on the PGS row chain (gfx950). Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define NINST 64
#define LOOPS 200

#define KERNEL(name, body)                                                                  \
  __global__ void __launch_bounds__(64) name(float* out, unsigned long long* clk, float seed) { \
    float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = 1.0f;                  \
    float a2 = a + 1.f, a3 = a + 2.f, a4 = a + 3.f;                                          \
    unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < LOOPS; i++) { REP64(body) }                                          \
    unsigned long long t1 = __builtin_readcyclecounter();                                    \
    out[blockIdx.x * 64 + threadIdx.x] = a + a2 + a3 + a4 + b + c + d;                       \
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                         \
  }

KERNEL(k_add_dep, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(k_add_indep4, asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b));)
KERNEL(k_mul_dep, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(d));)
KERNEL(k_med3_dep, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
typedef float f2 __attribute__((ext_vector_type(2)));
#define PKBODY { f2 x; x.x = a; x.y = a2; f2 y; y.x = b; y.y = c; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y)); a = x.x; a2 = x.y; }
KERNEL(k_pkadd_dep, PKBODY)
KERNEL(k_dpp_quad_dep, asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)
KERNEL(k_dpp_quad_dep_nonop, asm volatile("v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(a2) : "v"(b));)
KERNEL(k_dpp_mirror_dep, asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)
KERNEL(k_dpp_hmirror_dep, asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)
KERNEL(k_dpp_shr_dep, asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)
KERNEL(k_movdpp_dep, asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)
KERNEL(k_nop0, asm volatile("s_nop 0");)
KERNEL(k_nop1, asm volatile("s_nop 1");)
KERNEL(k_swap_dep, asm volatile("s_nop 1\n v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(a2));)
KERNEL(k_swap_add_dep, asm volatile("v_permlane16_swap_b32 %0, %1\n v_add_f32 %0, %0, %1\n v_mov_b32 %1, %0" : "+v"(a), "+v"(a2));)
KERNEL(k_readlane_dep, { int s; asm volatile("v_readlane_b32 %0, %1, 5\n s_nop 0\n v_add_f32 %1, %0, %1" : "=s"(s), "+v"(a)); })
KERNEL(k_fma_dep, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(d), "v"(b));)
KERNEL(k_fmac_dpp_dep, asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(d));)
/* full row step as in k_solve2 PAR path (register impulses): 1 row per body */
KERNEL(k_rowstep, asm volatile(
  "v_mul_f32 %1, %0, %3\n"
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
  "s_nop 1\n v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
  "v_sub_f32 %1, %2, %1\n"
  "v_med3_f32 %1, %1, %4, %5\n"
  "v_sub_f32 %1, %1, %2\n"
  "v_mul_f32 %1, %1, %3\n"
  "v_add_f32 %0, %0, %1\n" : "+v"(a), "+v"(a2) : "v"(a3), "v"(d), "v"(b), "v"(c));)

struct K { const char* name; void (*fn)(float*, unsigned long long*, float); int ninst; };
int main() {
  K ks[] = {{"v_add_f32 dependent", k_add_dep, 1}, {"v_add_f32 4 independent chains (per instr)", k_add_indep4, 4}, {"v_mul_f32 dependent", k_mul_dep, 1},
            {"v_fma_f32 dependent", k_fma_dep, 1}, {"v_med3_f32 dependent", k_med3_dep, 1}, {"v_pk_add_f32 dependent", k_pkadd_dep, 1},
            {"s_nop1 + v_add_dpp quad_perm dependent", k_dpp_quad_dep, 1}, {"2 indep add + v_add_dpp quad (no nop) (per group)", k_dpp_quad_dep_nonop, 1},
            {"s_nop1 + v_add_dpp row_mirror dependent", k_dpp_mirror_dep, 1},
            {"s_nop1 + v_add_dpp row_half_mirror dependent", k_dpp_hmirror_dep, 1}, {"s_nop1 + v_add_dpp row_shr:1 dependent", k_dpp_shr_dep, 1},
            {"s_nop1 + v_mov_dpp dependent", k_movdpp_dep, 1}, {"s_nop1 + v_fmac_dpp dependent", k_fmac_dpp_dep, 1}, {"s_nop 0", k_nop0, 1}, {"s_nop 1", k_nop1, 1},
            {"s_nop1 + v_permlane16_swap dependent", k_swap_dep, 1}, {"swap + add + mov (per group)", k_swap_add_dep, 1},
            {"v_readlane + s_nop0 + v_add(sgpr) dependent (per group)", k_readlane_dep, 1}, {"full PGS row step (per row)", k_rowstep, 1}};
  float* out; unsigned long long* clk;
  hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&clk, 8192 * 8);
  std::vector<unsigned long long> h(8192);
  for (int wps : {1, 2, 4}) {
    int blocks = 1024 * wps;   /* 256 CUs x 4 SIMDs x wps */
    printf("== %d wave(s) per SIMD (%d blocks of 64) ==\n", wps, blocks);
    for (auto& k : ks) {
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f);
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; unsigned long long mx = 0;
      for (int i = 0; i < blocks; i++) { s += h[i]; if (h[i] > mx) mx = h[i]; }
      double per = s / blocks / (double)(LOOPS * NINST * k.ninst);
      printf("%-62s mean %.2f cyc  (max wave %.2f)\n", k.name, per, mx / (double)(LOOPS * NINST * k.ninst));
    }
  }
  return 0;
}
