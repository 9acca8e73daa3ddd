// Micro-benchmark (gfx950): what one boundary between two DEPENDENT kernels of a stream costs - the step is a chain of 25 of them per env group.
//   chain of K kernels (each: `blocks` blocks of 128 threads spinning ~`us` microseconds, touching `kb` KB each) enqueued (a) as stream launches,
//   (b) as one hipGraph captured from the same launches.  Prints (time of the chain - K x the kernel's own duration) / K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void __launch_bounds__(128) k_spin(float* buf, int floats_per_block, long long cycles) {
  const long long t0 = __builtin_readcyclecounter();
  float* p = buf + (size_t)blockIdx.x * floats_per_block;
  for (int i = threadIdx.x; i < floats_per_block; i += 128) p[i] = p[i] * 1.0001f + 1.f;
  while (__builtin_readcyclecounter() - t0 < cycles) { }
}
int main(int argc, char** argv) {
  const int K = 24, reps = 50;
  for (int cfg = 0; cfg < 3; cfg++) {
    const int blocks = cfg == 0 ? 256 : 4096, kb = cfg == 2 ? 5 : 0;
    const int fpb = kb * 256 + 128;
    float* buf; CHK(hipMalloc(&buf, (size_t)blocks * fpb * 4)); CHK(hipMemset(buf, 0, (size_t)blocks * fpb * 4));
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const long long cyc = 2000;      /* ~20 us at the 100 MHz shader clock counter */
    auto chain = [&]() { for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, buf, fpb, cyc); };
    float one = 0.f;
    { for (int w = 0; w < 3; w++) { hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, buf, fpb, cyc); } CHK(hipStreamSynchronize(s));
      float best = 1e9f;
      for (int r = 0; r < 20; r++) { CHK(hipEventRecord(e0, s)); hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, buf, fpb, cyc); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
      one = best; }
    float t_stream = 0.f, t_graph = 0.f;
    { chain(); CHK(hipStreamSynchronize(s)); CHK(hipEventRecord(e0, s)); for (int r = 0; r < reps; r++) chain(); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&t_stream, e0, e1)); }
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(); CHK(hipStreamEndCapture(s, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    { CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s)); CHK(hipEventRecord(e0, s)); for (int r = 0; r < reps; r++) CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&t_graph, e0, e1)); }
    printf("blocks %d, %d KB per block: kernel alone (event pair, launch included) %.1f us; per kernel in a chain of %d: stream launches %.1f us, hipGraph %.1f us\n",
           blocks, kb, one * 1e3f, K, t_stream * 1e3f / (reps * K), t_graph * 1e3f / (reps * K));
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g)); CHK(hipFree(buf));
  }
  return 0;
}
