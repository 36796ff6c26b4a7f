#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(float* out, long long iters) {
  float x = threadIdx.x;
  for (long long i = 0; i < iters; ++i) x = x * 1.0000001f + 0.5f;
  if (x == 12345.f) out[0] = x;
}
int main() {
  float* d; hipMalloc(&d, 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a, s);
      for (int k = 0; k < 8; ++k) {
        unsigned flags = (mode == 1 && (k & 1)) ? hipExtAnyOrderLaunch : 0;
        if (mode == 2) flags = hipExtAnyOrderLaunch;
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, flags, d, 200000LL);
      }
      hipEventRecord(b, s);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("mode %d (0 = ordered, 1 = every 2nd any-order, 2 = all any-order): 8 launches of 64 WGs: %.3f ms\n", mode, ms);
    }
  }
  return 0;
}
