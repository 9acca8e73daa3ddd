// Micro-benchmark: cost of scalar branches inside a dependent VALU chain (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define LOOPS 200
#define KERNEL(name, body)                                                                  \
  __global__ void __launch_bounds__(64) name(float* out, unsigned long long* clk, float seed, int flag) { \
    float a = seed + threadIdx.x, b = seed;                                                  \
    int f = __builtin_amdgcn_readfirstlane(flag);                                            \
    unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < LOOPS; i++) { REP64(asm volatile(body : "+v"(a) : "v"(b), "s"(f) : "scc");) } \
    unsigned long long t1 = __builtin_readcyclecounter();                                    \
    out[blockIdx.x * 64 + threadIdx.x] = a;                                                  \
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                         \
  }
KERNEL(b_none, "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n")
KERNEL(b_nottaken, "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n s_bitcmp1_b32 %2, 0\n s_cbranch_scc1 .Lx%=\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n .Lx%=:\n")
KERNEL(b_taken, "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n s_bitcmp1_b32 %2, 0\n s_cbranch_scc0 .Lx%=\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n .Lx%=:\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n")
KERNEL(b_cmp_only, "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n s_bitcmp1_b32 %2, 0\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n")
struct K { const char* name; void (*fn)(float*, unsigned long long*, float, int); };
int main() {
  K ks[] = {{"4 dependent adds", b_none}, {"2 adds + cmp + branch NOT taken + 2 adds", b_nottaken}, {"2 adds + cmp + branch TAKEN (skips 2) + 2 adds", b_taken}, {"2 adds + cmp + 2 adds", b_cmp_only}};
  float* out; unsigned long long* clk;
  hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&clk, 8192 * 8);
  std::vector<unsigned long long> h(8192);
  for (int wps : {1, 2}) {
    int blocks = 1024 * wps;
    printf("== %d wave(s) per SIMD: cycles per group ==\n", wps);
    for (auto& k : ks) {
      for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f, 0);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < blocks; i++) s += h[i];
      printf("%-50s %.2f\n", k.name, s / blocks / (double)(LOOPS * 64));
    }
  }
  return 0;
}
