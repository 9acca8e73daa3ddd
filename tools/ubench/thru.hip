// Micro-benchmark: VALU pipe throughput of independent instruction streams (gfx950): cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define LOOPS 400
#define DPPM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define KERNEL(name, body)                                                                  \
  __global__ void __launch_bounds__(64) name(float* out, unsigned long long* clk, float seed) { \
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = seed; \
    asm volatile("s_mov_b32 s10, 0x00010001\n s_mov_b32 s11, 0x00010001" ::: "s10", "s11"); \
    unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < LOOPS; i++) { REP16(asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) } \
    unsigned long long t1 = __builtin_readcyclecounter();                                    \
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;              \
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                         \
  }
#define X8(op, sfx) op " %0, %0, %8" sfx op " %1, %1, %8" sfx op " %2, %2, %8" sfx op " %3, %3, %8" sfx op " %4, %4, %8" sfx op " %5, %5, %8" sfx op " %6, %6, %8" sfx op " %7, %7, %8" sfx
KERNEL(t_add, X8("v_add_f32", "\n"))
KERNEL(t_add_dpp_quad, X8("v_add_f32_dpp", " quad_perm:[1,0,3,2]" DPPM))
KERNEL(t_add_dpp_mirror, X8("v_add_f32_dpp", " row_mirror" DPPM))
KERNEL(t_add_dpp_bcast, X8("v_add_f32_dpp", " row_newbcast:5" DPPM))
KERNEL(t_med3, "v_med3_f32 %0, %0, %8, %1\n v_med3_f32 %1, %1, %8, %2\n v_med3_f32 %2, %2, %8, %3\n v_med3_f32 %3, %3, %8, %4\n v_med3_f32 %4, %4, %8, %5\n v_med3_f32 %5, %5, %8, %6\n v_med3_f32 %6, %6, %8, %7\n v_med3_f32 %7, %7, %8, %0\n")
KERNEL(t_cndmask, "v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n")
KERNEL(t_readlane, "v_readlane_b32 s12, %0, 3\n v_readlane_b32 s13, %1, 3\n v_readlane_b32 s14, %2, 3\n v_readlane_b32 s15, %3, 3\n v_readlane_b32 s16, %4, 3\n v_readlane_b32 s17, %5, 3\n v_readlane_b32 s18, %6, 3\n v_readlane_b32 s19, %7, 3\n")
struct K { const char* name; void (*fn)(float*, unsigned long long*, float); };
int main() {
  K ks[] = {{"v_add_f32", t_add}, {"v_add_f32_dpp quad_perm", t_add_dpp_quad}, {"v_add_f32_dpp row_mirror", t_add_dpp_mirror}, {"v_add_f32_dpp row_newbcast", t_add_dpp_bcast},
            {"v_med3_f32", t_med3}, {"v_cndmask_b32 (sgpr mask)", t_cndmask}, {"v_readlane_b32", t_readlane}};
  float* out; unsigned long long* clk;
  hipMalloc(&out, 16384 * 64 * 4); hipMalloc(&clk, 16384 * 8);
  std::vector<unsigned long long> h(16384);
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 1024 * wps;
    printf("== %d wave(s) per SIMD: 8 independent chains per wave; cycles per instruction per wave | per SIMD ==\n", wps);
    for (auto& k : ks) {
      for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < blocks; i++) s += h[i];
      double per = s / blocks / (double)(LOOPS * 16 * 8);
      printf("%-50s %.2f | %.2f\n", k.name, per, per / wps);
    }
  }
  return 0;
}
