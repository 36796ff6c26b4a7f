// which workgroups of a 512 x 256-thread, 72 KB-LDS launch share a CU, and in which wave slots (HW_ID / XCC_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  extern __shared__ float sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
  sm[threadIdx.x] = 1.f;
  __builtin_amdgcn_s_sleep(100);
}
int main() {
  unsigned* d; hipMalloc(&d, 512 * 4 * 2 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipLaunchKernelGGL(k, dim3(512), dim3(256), 72 * 1024, 0, d);
  hipDeviceSynchronize();
  static unsigned h[512 * 4 * 2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 40; ++b) {
    unsigned hw = h[b * 8], x = h[b * 8 + 1];
    printf("block %3d: xcc %u se %u sh %u cu %u | wave slots/simd:", b, x & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
    for (int w = 0; w < 4; ++w) printf(" %u/%u", h[(b * 4 + w) * 2] & 15, (h[(b * 4 + w) * 2] >> 4) & 3);
    printf("\n");
  }
  // pairs sharing a CU
  int same_parity = 0, pairs = 0;
  for (int a = 0; a < 512; ++a) for (int b = a + 1; b < 512; ++b) {
    unsigned ka = (h[a * 8 + 1] & 15) << 16 | (h[a * 8] & 0xff00), kb = (h[b * 8 + 1] & 15) << 16 | (h[b * 8] & 0xff00);
    if (ka == kb) { ++pairs; if ((a & 1) == (b & 1)) ++same_parity; if (pairs <= 12) printf("share a CU: %d %d (diff %d) slots %u %u\n", a, b, b - a, h[a * 8] & 15, h[b * 8] & 15); }
  }
  printf("pairs %d, same blockIdx parity %d\n", pairs, same_parity);
}
