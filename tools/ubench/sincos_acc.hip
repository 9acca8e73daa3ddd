// accuracy of the hardware sin/cos (v_sin_f32 / v_cos_f32 via __sinf / __cosf) against fp64 on [-2 pi, 2 pi]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(float* es, float* ec, float* es2, float* ec2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float x = -6.2831853f + 12.5663706f * (i / 1048576.f);
  double xs = sin((double)x), xc = cos((double)x);
  es[i] = fabs((double)__sinf(x) - xs); ec[i] = fabs((double)__cosf(x) - xc);
  float s, c; sincosf(x, &s, &c);
  es2[i] = fabs((double)s - xs); ec2[i] = fabs((double)c - xc);
}
int main() {
  const int n = 1 << 20; float *d[4]; for (auto& p : d) hipMalloc(&p, n * 4);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d[0], d[1], d[2], d[3]);
  float* h = new float[n]; const char* nm[4] = {"__sinf", "__cosf", "sincosf sin", "sincosf cos"};
  for (int j = 0; j < 4; j++) { hipMemcpy(h, d[j], n * 4, hipMemcpyDeviceToHost); float mx = 0; for (int i = 0; i < n; i++) mx = fmaxf(mx, h[i]); printf("%-12s max abs error %.3e\n", nm[j], mx); }
  return 0;
}
