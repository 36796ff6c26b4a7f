// Stand-alone RunningMeanStd forward (algo/models/running_mean_std.py:60-93) for the normalisers
// that live outside the fused minibatch kernels (value de-normalisation in model_act,
// stud_obs_mean_std / pcl_mean_std at ingest, ext_adapt.py:404-420).
//   pass 1: per-block fp64 column sums (each thread owns one column; a block covers
//           blockDim/D rows per sweep so loads stay row-contiguous)
//   pass 2: one block merges the batch moments into the fp64 state (Chan)
//   pass 3: elementwise normalise / de-normalise with fp32 coefficients
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "teacher.h"

namespace igi {

constexpr int RMS_BLOCKS_MAX = 256;

static inline int rms_blocks(int64_t rows, int D) {
  const int rpb = (D <= 256) ? (256 / D) : 1;     // rows per sweep of one block
  int64_t nb = (rows + (int64_t)rpb * 64 - 1) / ((int64_t)rpb * 64);  // >= 64 sweeps per block
  if (nb < 1) nb = 1;
  if (nb > RMS_BLOCKS_MAX) nb = RMS_BLOCKS_MAX;
  return (int)nb;
}

__global__ __launch_bounds__(256) void k_rms_partial(const float* __restrict__ x, long long rows, int D,
                                                     double* __restrict__ part) {
  extern __shared__ double sh[];  // [2][threads]
  const int tpr = (D <= 256) ? D : 256;           // threads per row
  const int rpb = (D <= 256) ? (256 / D) : 1;
  const int active = tpr * rpb;
  const int tid = threadIdx.x;
  for (int c0 = 0; c0 < D; c0 += tpr) {
    double s = 0, s2 = 0;
    const int c = c0 + tid % tpr;
    if (tid < active && c < D) {
      for (long long r = (long long)blockIdx.x * rpb + tid / tpr; r < rows; r += (long long)gridDim.x * rpb) {
        const double v = x[r * D + c];
        s += v;
        s2 += v * v;
      }
    }
    sh[tid] = s;
    sh[256 + tid] = s2;
    __syncthreads();
    if (tid < tpr && c < D) {
      double a = 0, b = 0;
      for (int q = 0; q < rpb; ++q) { a += sh[q * tpr + tid]; b += sh[256 + q * tpr + tid]; }
      part[((long long)blockIdx.x * D + c) * 2 + 0] = a;
      part[((long long)blockIdx.x * D + c) * 2 + 1] = b;
    }
    __syncthreads();
  }
}

// coef[2c] = mean, coef[2c+1] = sqrt(var+eps)
__global__ void k_rms_merge(const double* __restrict__ part, int nblocks, long long rows, int D,
                            double* __restrict__ state, float eps, int train, float* __restrict__ coef) {
  __shared__ double cnt;
  if (threadIdx.x == 0) cnt = state[2 * D];
  __syncthreads();
  const double n = (double)rows;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    double mean = state[c], var = state[D + c];
    if (train) {
      double s = 0, s2 = 0;
      for (int b = 0; b < nblocks; ++b) {
        s += part[((long long)b * D + c) * 2 + 0];
        s2 += part[((long long)b * D + c) * 2 + 1];
      }
      const double m = s / n;
      double v = (s2 - n * m * m) / (n - 1.0);
      if (v < 0) v = 0;
      double count = cnt;
      chan_merge(mean, var, count, (float)m, (float)v, n);
      state[c] = mean;
      state[D + c] = var;
    }
    coef[2 * c + 0] = (float)mean;
    coef[2 * c + 1] = sqrtf((float)var + eps);
  }
  __syncthreads();
  if (train && threadIdx.x == 0) state[2 * D] = cnt + n;
}

__global__ __launch_bounds__(256) void k_rms_apply(const float* __restrict__ x, float* __restrict__ y,
                                                   long long total, int D, const float* __restrict__ coef,
                                                   int unnorm) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    const float m = coef[2 * c], d = coef[2 * c + 1];
    const float v = x[e];
    y[e] = unnorm ? (d * clamp5(v) + m) : clamp5((v - m) / d);
  }
}

static size_t rms_workspace_bytes(int64_t rows, int D) {
  return (size_t)ru64(sizeof(double) * 2 * (size_t)D * rms_blocks(rows, D)) + (size_t)ru64(sizeof(float) * 2 * D);
}

static int rms_forward(const float* x, float* y, int64_t rows, int D, double* state, float eps, int train,
                       int unnorm, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!x || !y || !state || rows < 1 || D < 1 || !ws) return IGI_E_BADARG;
  if (train && rows < 2) return IGI_E_BADARG;
  if (ws_bytes < rms_workspace_bytes(rows, D)) return IGI_E_WORKSPACE;
  const int nb = rms_blocks(rows, D);
  double* part = reinterpret_cast<double*>(ws);
  float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ru64(sizeof(double) * 2 * (size_t)D * nb));
  if (train)
    hipLaunchKernelGGL(k_rms_partial, dim3(nb), dim3(256), sizeof(double) * 512, s, x, (long long)rows, D, part);
  hipLaunchKernelGGL(k_rms_merge, dim3(1), dim3(256), 0, s, part, nb, (long long)rows, D, state, eps, train, coef);
  const long long total = (long long)rows * D;
  int ab = (int)((total + 255) / 256);
  if (ab > 2048) ab = 2048;
  hipLaunchKernelGGL(k_rms_apply, dim3(ab), dim3(256), 0, s, x, y, total, D, coef, unnorm);
  return (int)hipGetLastError();
}

}  // namespace igi
