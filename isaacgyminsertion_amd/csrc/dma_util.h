// Small pieces shared by the LDS-DMA kernels (gemm_dma.h, env_mlp.h, rowblock.h): the LDS-DMA instruction, scalar
// pointers, the process-wide bf16-input switch.  Kept apart from gemm_dma.h so that a kernel header (and its probe under
// tools/probes) can be compiled without instantiating every GEMM configuration.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace igi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One wave-instruction of LDS-DMA: 64 lanes x 16 B, LDS destination = lds_base + lane*16.
__device__ __forceinline__ void dma16(const float* gsrc, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(gsrc, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

// opt-in bf16-input mode of the large products: IGI_GEMM_BF16=1 in the environment, or igi_gemm_set_bf16_inputs()
static inline int& bf16_mode_ref() {
  static int m = -1;
  if (m < 0) { const char* e = getenv("IGI_GEMM_BF16"); m = e ? (atoi(e) != 0) : 0; }
  return m;
}
static inline int bf16_mode() { return bf16_mode_ref(); }

// EXPERIMENT (off by default): fp32 products on the bf16 matrix pipe by an EXACT three-plane split -- every operand element
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (each difference is exact, 3 x 8
// significant bits = fp32's 24), every cross product of two planes is exact in fp32, the MFMA accumulates in fp32.
// IGI_GEMM_X3 / igi_gemm_set_bf16x3: 0 off, 9 all nine cross products (nothing dropped), 6 without the three products
// below 2^-24 of the leading one (mid*lo, lo*mid, lo*lo).  Applies to the large k-contiguous forward products only.
static inline int& x3_mode_ref() {
  static int m = -1;
  if (m < 0) { const char* e = getenv("IGI_GEMM_X3"); const int v = e ? atoi(e) : 0; m = (v == 6 || v == 9) ? v : 0; }
  return m;
}
static inline int x3_mode() { return x3_mode_ref(); }

}  // namespace igi
