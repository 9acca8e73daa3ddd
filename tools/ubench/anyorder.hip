// Micro-benchmark (gfx950): do two kernels of ONE stream overlap when the second is launched with hipExtAnyOrderLaunch (no barrier bit)?
//   A: 16 blocks spinning ~100 us;  B: 2048 blocks spinning ~20 us;  C: 1 block (a dependent successor, launched normally)
//   prints the time of A, B, C enqueued back to back: (a) all normal, (b) B any-order, (c) A and B on two streams with an event join
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void __launch_bounds__(128) k_spin(float* buf, long long ticks) {
  const long long t0 = wall_clock64();
  buf[blockIdx.x * 128 + threadIdx.x] += 1.f;
  while (wall_clock64() - t0 < ticks) { }
}
int main() {
  float* buf; CHK(hipMalloc(&buf, 4096 * 128 * 4)); CHK(hipMemset(buf, 0, 4096 * 128 * 4));
  hipStream_t s, s2; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, ef, ej; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  const long long TA = 10000, TB = 2000, TC = 100;      /* 100 MHz ticks */
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9f;
    for (int r = 0; r < 30; r++) {
      CHK(hipEventRecord(e0, s));
      for (int k = 0; k < 8; k++) {
        if (mode == 2) {
          CHK(hipEventRecord(ef, s)); CHK(hipStreamWaitEvent(s2, ef, 0));
          hipLaunchKernelGGL(k_spin, dim3(16), dim3(128), 0, s2, buf, TA);
          CHK(hipEventRecord(ej, s2));
          hipLaunchKernelGGL(k_spin, dim3(2048), dim3(128), 0, s, buf + 16 * 128, TB);
          CHK(hipStreamWaitEvent(s, ej, 0));
        } else {
          hipLaunchKernelGGL(k_spin, dim3(16), dim3(128), 0, s, buf, TA);
          if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(2048), dim3(128), 0, s, buf + 16 * 128, TB);
          else hipExtLaunchKernelGGL(k_spin, dim3(2048), dim3(128), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, buf + 16 * 128, TB);
        }
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(128), 0, s, buf, TC);
      }
      CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%s: %.1f us per (A 100 us x 16 blocks, B 20 us x 2048 blocks, C 1 us) triple\n", mode == 0 ? "all normal launches      " : (mode == 1 ? "B with hipExtAnyOrderLaunch" : "A on a second stream + join"), best * 1e3f / 8);
  }
  return 0;
}
