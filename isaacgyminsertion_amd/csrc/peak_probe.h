// What this box's matrix pipes deliver when nothing else is in the way: a register-only loop of
// v_mfma_f32_32x32x2_f32 (no LDS, no memory) on every SIMD of the chip -- the measured counterpart of the 157.3 TFLOP/s
// spec figure bench.py prices the kernels against ("re-measure on the box with a microbench and state
// both").  Operands are lane- and iteration-dependent non-trivial values (a loop on zeros holds a higher clock than real
// data does: MI355X_MICROARCH.md, DVFS give-back); the accumulators are folded into one store nobody reads so the loop
// cannot be removed.  The first wave of every block also stamps the shader clock (s_memtime) against the 100 MHz
// real-time counter (s_memrealtime): clock = d(memtime) / d(memrealtime) x 100 MHz, the in-kernel clock under this load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f32.h"

namespace igi {

constexpr int PEAK_MFMA_PER_ITER = 16;          // four independent accumulators x four MFMAs each

typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: v_mfma_f32_32x32x2_f32 (4096 flop, 64 cycles) -- the instruction every fp32 product of this library runs on;
//       1: v_mfma_f32_16x16x4_f32 (2048 flop, 32 cycles) -- same flop per cycle, another shape (the clock a chip holds under
//          load can depend on the shape: MI355X_MICROARCH.md, DVFS give-back (7));
//       2: v_mfma_f32_32x32x16_bf16 (32768 flop, 32 cycles) -- the pipe the bf16x3 experiment runs on.
// out[2 * block] = shader cycles, out[2 * block + 1] = 100 MHz ticks spent in the loop (wave 0 of the block)
template <int SHAPE>
__global__ __launch_bounds__(256) void k_mfma_peak(int iters, unsigned long long* __restrict__ out, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  // |values| around 1 with every mantissa bit in use; they change every iteration
  float a[4], b[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    a[u] = 0.37f + 0.0131f * (float)((lane * 7 + u * 13) % 61);
    b[u] = -0.91f + 0.0173f * (float)((lane * 11 + u * 5 + blockIdx.x) % 53);
  }
  float s = 0.f;
  unsigned long long c0, r0, c1, r1;
  if constexpr (SHAPE == 0) {
    f32x16 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 3], b[u], acc[u], 0, 0, 0);
      // (keeps the operands moving without a vector instruction per MFMA: one rotation per 16 MFMAs)
      const float t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = -t;
    }
    c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[u][r];
  } else if constexpr (SHAPE == 1) {
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[u][r] = 0.f;
    c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 3], b[u], acc[u], 0, 0, 0);
      const float t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = -t;
    }
    c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[u][r];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    pk_bf16x8 av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        av[u][j] = (__bf16)(a[u] * (1.0f + 0.07f * j) * 0.05f);
        bv[u][j] = (__bf16)(b[(u + j) & 3] * (1.0f - 0.05f * j));
      }
    c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(u + j) & 3], bv[u], acc[u], 0, 0, 0);
      const pk_bf16x8 t = av[0]; av[0] = av[1]; av[1] = av[2]; av[2] = av[3]; av[3] = t;
    }
    c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[u][r];
  }
  if (s == 12345.678f) sink[0] = s;             // never true in practice: the loop stays
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

// One launch of `blocks` x 256 threads; flops = blocks * 4 waves * iters * 16 MFMAs * {4096, 2048, 32768} by shape.
// `out`: 2 * blocks uint64 (device), `sink`: one float (device).
static int mfma_peak_probe(int shape, int blocks, int iters, unsigned long long* out, float* sink, hipStream_t s) {
  if (blocks < 1 || iters < 1 || !out || !sink || shape < 0 || shape > 2) return IGI_E_BADARG;
  if (shape == 0) hipLaunchKernelGGL(k_mfma_peak<0>, dim3(blocks), dim3(256), 0, s, iters, out, sink);
  else if (shape == 1) hipLaunchKernelGGL(k_mfma_peak<1>, dim3(blocks), dim3(256), 0, s, iters, out, sink);
  else hipLaunchKernelGGL(k_mfma_peak<2>, dim3(blocks), dim3(256), 0, s, iters, out, sink);
  return (int)hipGetLastError();
}

}  // namespace igi
