// Segmented point-cloud encoder (algo/models/transformer/pointnets.py:12-42):
//   y[b][c] = max_n ( W2 . gelu(W1 . x[b][n] + b1) + b2 )[c],  W1 (64,3), W2 (256,64), erf-GELU;
// forward (+ argmax) and backward.
//
// The reference materialises (B,N,64) and (B,N,256) activations in HBM (512 KB per object per
// sample at N = 400).  Here one persistent workgroup streams samples: W2 is staged once in LDS
// (reduction-major, 64 KB), the 64-wide hidden rows of 32 points are produced on the VALU straight into
// an LDS MFMA-operand tile, the 32x256 second-layer block comes from exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32, each wave owns 64 output columns) and only a running (max, argmax) per
// column survives in registers: HBM traffic is the 12 B/point input and 2 KB/sample output.
// Backward routes dy through the argmax rows only (<= 256 of the N points), recomputing those hidden
// rows; weight gradients accumulate in LDS / registers per workgroup and leave as per-workgroup
// partials that k_slab_reduce sums in fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/igi_ppo.h"
#include "teacher.h"

namespace igi {

constexpr int PN_H = 64, PN_OUT = 256, PN_IN = 3;
constexpr int PN_P = PN_H * PN_IN + PN_H + PN_OUT * PN_H + PN_OUT;  // 16896
constexpr int PN_OW1 = 0, PN_OB1 = PN_H * PN_IN, PN_OW2 = PN_OB1 + PN_H, PN_OB2 = PN_OW2 + PN_OUT * PN_H;
constexpr int PN_BLOCKS = 512;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * expf(-0.5f * x * x) * 0.39894228040143267794f;
}

__global__ __launch_bounds__(256) void k_pointnet_fwd(const float* __restrict__ x, int B, int N,
                                                      const float* __restrict__ params, float* __restrict__ y,
                                                      int* __restrict__ argmax) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* W2t = sm;                       // [64 k][256 c]
  float* Hs = sm + PN_H * PN_OUT;        // [64 k][32 m]
  float* W1s = Hs + PN_H * 32;           // [64][3] + b1[64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  {  // stage W2 transposed: thread c owns row c of W2 (64 floats)
    const float* w2 = params + PN_OW2 + tid * PN_H;
#pragma unroll
    for (int k4 = 0; k4 < PN_H / 4; ++k4) {
      const float4 v = *reinterpret_cast<const float4*>(w2 + 4 * k4);
      W2t[(4 * k4 + 0) * PN_OUT + tid] = v.x;
      W2t[(4 * k4 + 1) * PN_OUT + tid] = v.y;
      W2t[(4 * k4 + 2) * PN_OUT + tid] = v.z;
      W2t[(4 * k4 + 3) * PN_OUT + tid] = v.w;
    }
    if (tid < PN_H * PN_IN + PN_H) W1s[tid] = params[tid];  // W1 then b1 are contiguous
  }
  __syncthreads();
  const int m_h = tid & 31, kq = tid >> 5;  // hidden-tile production: point m_h, hidden units kq*8..+7
  const int c0 = wave * 64 + l31, c1 = c0 + 32;
  const float b2_0 = params[PN_OB2 + c0], b2_1 = params[PN_OB2 + c1];
  const int ntiles = (N + 31) / 32;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float* xb = x + (long long)b * N * PN_IN;
    float best0 = -INFINITY, best1 = -INFINITY;
    int bi0 = 0, bi1 = 0;
    for (int rt = 0; rt < ntiles; ++rt) {
      {
        const int n = rt * 32 + m_h;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (n < N) { px = xb[n * 3]; py = xb[n * 3 + 1]; pz = xb[n * 3 + 2]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kq * 8 + j;
          const float pre = ((W1s[k * 3] * px + W1s[k * 3 + 1] * py) + W1s[k * 3 + 2] * pz) + W1s[PN_H * PN_IN + k];
          Hs[k * 32 + m_h] = gelu_erf(pre);
        }
      }
      __syncthreads();
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll 8
      for (int k2 = 0; k2 < PN_H / 2; ++k2) {
        const int k = 2 * k2 + h;
        const float a = Hs[k * 32 + l31];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, W2t[k * PN_OUT + c0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, W2t[k * PN_OUT + c1], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // rows ascend with r: strict '>' keeps the first maximum
        const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < N) {
          if (acc0[r] > best0) { best0 = acc0[r]; bi0 = row; }
          if (acc1[r] > best1) { best1 = acc1[r]; bi1 = row; }
        }
      }
      __syncthreads();
    }
    // the two lane halves hold interleaved row groups of the same column
    const float o0 = __shfl_xor(best0, 32, 64), o1 = __shfl_xor(best1, 32, 64);
    const int oi0 = __shfl_xor(bi0, 32, 64), oi1 = __shfl_xor(bi1, 32, 64);
    if (o0 > best0 || (o0 == best0 && oi0 < bi0)) { best0 = o0; bi0 = oi0; }
    if (o1 > best1 || (o1 == best1 && oi1 < bi1)) { best1 = o1; bi1 = oi1; }
    if (h == 0) {
      y[(long long)b * PN_OUT + c0] = best0 + b2_0;
      y[(long long)b * PN_OUT + c1] = best1 + b2_1;
      if (argmax) {
        argmax[(long long)b * PN_OUT + c0] = bi0;
        argmax[(long long)b * PN_OUT + c1] = bi1;
      }
    }
  }
}

// thread c = output column.  Per sample: recompute the hidden row of its argmax point, accumulate
// dW2[c][:] (LDS, [k][c] so lanes hit consecutive banks) and db2, form t[k] = dy*W2[c][k]*gelu'(pre)
// and reduce t[k] * (x, y, z, 1) over the 256 columns into dW1 / db1 (16 k at a time through LDS).
__global__ __launch_bounds__(256) void k_pointnet_bwd(const float* __restrict__ x, int B, int N,
                                                      const float* __restrict__ params,
                                                      const float* __restrict__ dy, const int* __restrict__ argmax,
                                                      float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* dW2s = sm;                          // [64 k][256 c]
  float* Ts = dW2s + PN_H * PN_OUT;          // [16 k][256 c]
  float* Xs = Ts + 16 * PN_OUT;              // [256 c][4]  (x, y, z, 1) of the argmax point
  float* W1s = Xs + PN_OUT * 4;              // W1 [64][3] + b1 [64]
  const int tid = threadIdx.x;
  for (int e = tid; e < PN_H * PN_OUT; e += 256) dW2s[e] = 0.f;
  if (tid < PN_H * PN_IN + PN_H) W1s[tid] = params[tid];
  __syncthreads();
  const float* w2row = params + PN_OW2 + tid * PN_H;
  float db2 = 0.f;
  // reduction role: k_local = tid / 16, part = tid % 16; part 0 keeps dW1 / db1 for k = chunk*16 + k_local
  const int kl = tid >> 4, part = tid & 15;
  float acc1[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[q][j] = 0.f;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float g = dy[(long long)b * PN_OUT + tid];
    const int n = argmax[(long long)b * PN_OUT + tid];
    const float* xp = x + ((long long)b * N + n) * PN_IN;
    const float px = xp[0], py = xp[1], pz = xp[2];
    db2 += g;
    *reinterpret_cast<float4*>(Xs + tid * 4) = make_float4(px, py, pz, 1.0f);
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = chunk * 16 + j;
        const float pre = ((W1s[k * 3] * px + W1s[k * 3 + 1] * py) + W1s[k * 3 + 2] * pz) + W1s[PN_H * PN_IN + k];
        dW2s[k * PN_OUT + tid] += g * gelu_erf(pre);
        Ts[j * PN_OUT + tid] = (g * w2row[k]) * gelu_erf_grad(pre);
      }
      __syncthreads();
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int c = part + 16 * q;
        const float t = Ts[kl * PN_OUT + c];
        const float4 xv = *reinterpret_cast<const float4*>(Xs + c * 4);
        s0 += t * xv.x; s1 += t * xv.y; s2 += t * xv.z; s3 += t * xv.w;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {  // the 16 `part` lanes of one k are adjacent lanes
        s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64); s3 += __shfl_xor(s3, o, 64);
      }
      acc1[chunk][0] += s0; acc1[chunk][1] += s1; acc1[chunk][2] += s2; acc1[chunk][3] += s3;
      __syncthreads();
    }
  }
  // per-workgroup partial in parameter layout
  float* out = partial + (long long)blockIdx.x * PN_P;
  if (part == 0) {
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {
      const int k = chunk * 16 + kl;
      out[PN_OW1 + k * 3 + 0] = acc1[chunk][0];
      out[PN_OW1 + k * 3 + 1] = acc1[chunk][1];
      out[PN_OW1 + k * 3 + 2] = acc1[chunk][2];
      out[PN_OB1 + k] = acc1[chunk][3];
    }
  }
  out[PN_OB2 + tid] = db2;
  for (int k = 0; k < PN_H; ++k) out[PN_OW2 + tid * PN_H + k] = dW2s[k * PN_OUT + tid];
}

static inline int pn_blocks(int64_t B) { return (int)(B < PN_BLOCKS ? B : PN_BLOCKS); }
static size_t pointnet_workspace_bytes(int64_t B) { return sizeof(float) * (size_t)PN_P * pn_blocks(B); }

static int pointnet_forward(const float* x, int64_t B, int N, const float* params, float* y, int* argmax,
                            hipStream_t s) {
  if (!x || !params || !y || B < 1 || N < 1 || B > (1 << 30)) return IGI_E_BADARG;
  const size_t shm = sizeof(float) * (PN_H * PN_OUT + PN_H * 32 + PN_H * PN_IN + PN_H);
  static bool attr = false;
  if (!attr) {
    IGI_HIP_TRY(hipFuncSetAttribute((const void*)k_pointnet_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr = true;
  }
  {
    // algorithmic: 2 * (3*64 + 64*256) flop per point; 12 B/point in, 256 values + 256 indices per cloud out
    ProfScope ps(PC_POINTNET_FWD, s, 2.0 * (PN_IN * PN_H + PN_H * PN_OUT) * (double)B * N,
                 12.0 * (double)B * N + 8.0 * PN_OUT * (double)B);
    IGI_LAUNCH(k_pointnet_fwd, dim3(pn_blocks(B)), dim3(256), shm, s, x, (int)B, N, params, y, argmax);
  }
  return (int)hipGetLastError();
}

static int pointnet_backward(const float* x, int64_t B, int N, const float* params, const float* dy,
                             const int* argmax, float* grads, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!x || !params || !dy || !argmax || !grads || !ws || B < 1 || N < 1) return IGI_E_BADARG;
  if (ws_bytes < pointnet_workspace_bytes(B)) return IGI_E_WORKSPACE;
  const int nb = pn_blocks(B);
  const size_t shm = sizeof(float) * (PN_H * PN_OUT + 16 * PN_OUT + PN_OUT * 4 + PN_H * PN_IN + PN_H);
  static bool attr = false;
  if (!attr) {
    IGI_HIP_TRY(hipFuncSetAttribute((const void*)k_pointnet_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr = true;
  }
  float* partial = reinterpret_cast<float*>(ws);
  {
    // executed work: only the <= 256 arg-max points of a cloud carry gradient: hidden rows recomputed (2*3*64), the
    // second layer's weight gradient and the data gradient into the hidden layer (2 * 2*64 per selected output),
    // the first layer's weight gradient (2*3*64)
    ProfScope ps(PC_POINTNET_BWD, s, (double)B * PN_OUT * (4.0 * PN_IN * PN_H + 4.0 * PN_H),
                 (double)B * (8.0 * PN_OUT + 12.0 * PN_OUT) + 4.0 * PN_P * nb);
    IGI_LAUNCH(k_pointnet_bwd, dim3(nb), dim3(256), shm, s, x, (int)B, N, params, dy, argmax, partial);
  }
  SegTable t;
  t.n = 1;
  Segment& sg = t.s[0];
  sg.dst = 0; sg.src = partial; sg.stride = PN_P; sg.count = PN_P; sg.cols = PN_P; sg.src_ld = 0; sg.nparts = nb;
  hipLaunchKernelGGL(k_slab_reduce, dim3(64, 1), dim3(RED_THREADS), 0, s, t, grads);
  return (int)hipGetLastError();
}

}  // namespace igi
