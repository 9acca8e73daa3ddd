// Micro-benchmark: straight-line loop bodies of growing code size (instruction-cache reach), gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))
#define REP1K(x) REP4(REP256(x))
#define I8 asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a) : "v"(b));   /* 8-byte encoding */
#define KERNEL(name, BODY, NI)                                                               \
  __global__ void __launch_bounds__(64) name(float* out, unsigned long long* clk, float seed, int loops) { \
    float a = seed + threadIdx.x, b = seed * 2.f;                                            \
    unsigned long long t0 = __builtin_readcyclecounter();                                    \
    for (int i = 0; i < loops; i++) { BODY }                                                 \
    unsigned long long t1 = __builtin_readcyclecounter();                                    \
    out[blockIdx.x * 64 + threadIdx.x] = a;                                                  \
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                         \
  }
KERNEL(k_2k, REP256(I8), 256)
KERNEL(k_4k, REP256(I8) REP256(I8), 512)
KERNEL(k_8k, REP1K(I8), 1024)
KERNEL(k_16k, REP1K(I8) REP1K(I8), 2048)
KERNEL(k_24k, REP1K(I8) REP1K(I8) REP1K(I8), 3072)
KERNEL(k_32k, REP4(REP1K(I8)), 4096)
KERNEL(k_48k, REP4(REP1K(I8)) REP1K(I8) REP1K(I8), 6144)
KERNEL(k_64k, REP4(REP1K(I8)) REP4(REP1K(I8)), 8192)
KERNEL(k_96k, REP4(REP1K(I8)) REP4(REP1K(I8)) REP4(REP1K(I8)), 12288)
struct K { const char* name; void (*fn)(float*, unsigned long long*, float, int); int ninst; };
int main() {
  K ks[] = {{"2 KB", k_2k, 256}, {"4 KB", k_4k, 512}, {"8 KB", k_8k, 1024}, {"16 KB", k_16k, 2048}, {"24 KB", k_24k, 3072}, {"32 KB", k_32k, 4096},
            {"48 KB", k_48k, 6144}, {"64 KB", k_64k, 8192}, {"96 KB", k_96k, 12288}};
  float* out; unsigned long long* clk;
  hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&clk, 8192 * 8);
  std::vector<unsigned long long> h(8192);
  for (int wps : {1, 2}) {
    int blocks = 1024 * wps;
    printf("== %d wave(s) per SIMD: dependent v_add_f32 (8-byte encoding), cycles per instruction by loop-body size ==\n", wps);
    for (auto& k : ks) {
      int loops = 200000 / k.ninst + 2;
      for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, out, clk, 1.0f, loops);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; unsigned long long mx = 0;
      for (int i = 0; i < blocks; i++) { s += h[i]; if (h[i] > mx) mx = h[i]; }
      printf("%-8s mean %.2f cyc  (max wave %.2f)\n", k.name, s / blocks / ((double)loops * k.ninst), mx / ((double)loops * k.ninst));
    }
  }
  return 0;
}
