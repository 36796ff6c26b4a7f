// Does the LDS-DMA load (global_load_lds_dwordx4) accept a source address that is only 8-byte aligned?
// (A planar 3-channel conv1 gather at stride 2 would start its 16-byte requests at 8 * ox bytes.)
// hipcc --offload-arch=gfx950 -O2 -o dma_align tools/probes/dma_align.hip && ./dma_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* src, float* out, int shift) {
  __shared__ __attribute__((aligned(16))) float sm[256];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds(src + shift + 4 * lane, (lds_ptr_t)sm, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int j = 0; j < 4; ++j) out[4 * lane + j] = sm[4 * lane + j];
}
int main() {
  float *s, *o;
  hipMalloc(&s, 4096); hipMalloc(&o, 1024);
  std::vector<float> h(1024), r(256);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipMemset(o, 0, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o, shift);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + shift);
    printf("shift %d floats (%2d-byte aligned): %s, mismatches %d, first values %g %g %g %g\n", shift, (shift * 4) % 16 ? (shift * 4) % 16 : 16,
           hipGetErrorString(e), bad, r[0], r[1], r[2], r[3]);
  }
  return 0;
}
