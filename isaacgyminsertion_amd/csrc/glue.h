// The student step's data movement between the native blocks, as launches of this library instead of ATen's
// (ext_adapt.py:785-828 around the model; experience.py:117-139; tact.py:542-571):
//   k_gather_rows : the minibatch rows of up to eight arenas in ONE launch (experience.py:117-139 gathers every key with
//                   its own index kernel: four launches per optimizer step for the keys a distillation step reads)
//   k_cat_cols    : out[b] = [part_0[b] | part_1[b] | ...] (+ a broadcast row, the positional encoding): the token
//                   concatenation and the concatenation of the point-cloud encodings
//   k_split_cols  : the reverse, for the backward pass -- every part's gradient dense, one launch instead of one strided
//                   copy per part
// Pure copies (and one add): bit-exact by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f32.h"

namespace igi {

constexpr int GLUE_MAX = 8;

struct GatherRowsArgs {
  const float* src[GLUE_MAX];
  float* dst[GLUE_MAX];
  long long width[GLUE_MAX];    // floats per row
  int vec[GLUE_MAX];            // 1: 16-byte accesses (width % 4 == 0, both bases 16-byte aligned)
  int n;
  const long long* rows;        // [nrows] row numbers into the arenas
  long long nrows, rows_total;
};

// grid (row chunks, arenas): a block moves whole rows; rows outside [0, rows_total) come back as NaN (index_select
// asserts on the device; a NaN row is as loud and leaves the stream alive)
__global__ __launch_bounds__(256) void k_gather_rows(const GatherRowsArgs a) {
  const int k = blockIdx.y;
  const float* __restrict__ src = a.src[k];
  float* __restrict__ dst = a.dst[k];
  const long long w = a.width[k];
  if (a.vec[k]) {
    const long long w4 = w >> 2;
    // rows per block pass: wide rows one at a time, narrow rows several per block
    const int rpb = w4 >= 256 ? 1 : (int)(256 / (w4 > 0 ? w4 : 1));
    const int lr = rpb > 1 ? (int)(threadIdx.x / w4) : 0;
    const long long c0 = rpb > 1 ? (long long)(threadIdx.x % w4) : threadIdx.x;
    if (rpb > 1 && lr >= rpb) return;
    for (long long r = (long long)blockIdx.x * rpb + lr; r < a.nrows; r += (long long)gridDim.x * rpb) {
      const long long sr = a.rows[r];
      const bool ok = sr >= 0 && sr < a.rows_total;
      const float4* s4 = reinterpret_cast<const float4*>(src + (ok ? sr : 0) * w);
      float4* d4 = reinterpret_cast<float4*>(dst + r * w);
      const float qn = __builtin_nanf("");
      for (long long c = c0; c < w4; c += (rpb > 1 ? w4 : 256)) d4[c] = ok ? s4[c] : make_float4(qn, qn, qn, qn);
    }
    return;
  }
  const int rpb = w >= 256 ? 1 : (int)(256 / (w > 0 ? w : 1));
  const int lr = rpb > 1 ? (int)(threadIdx.x / w) : 0;
  const long long c0 = rpb > 1 ? (long long)(threadIdx.x % w) : threadIdx.x;
  if (rpb > 1 && lr >= rpb) return;
  for (long long r = (long long)blockIdx.x * rpb + lr; r < a.nrows; r += (long long)gridDim.x * rpb) {
    const long long sr = a.rows[r];
    const bool ok = sr >= 0 && sr < a.rows_total;
    const float* s1 = src + (ok ? sr : 0) * w;
    for (long long c = c0; c < w; c += (rpb > 1 ? w : 256)) dst[r * w + c] = ok ? s1[c] : __builtin_nanf("");
  }
}

static int gather_rows(int n, const float* const* src, const int64_t* width, float* const* dst, const int64_t* rows,
                       int64_t nrows, int64_t rows_total, hipStream_t s) {
  if (n < 1 || n > GLUE_MAX || !src || !width || !dst || !rows || nrows < 1 || rows_total < 1) return IGI_E_BADARG;
  GatherRowsArgs a;
  a.n = n; a.rows = (const long long*)rows; a.nrows = nrows; a.rows_total = rows_total;
  for (int k = 0; k < n; ++k) {
    if (!src[k] || !dst[k] || width[k] < 1) return IGI_E_BADARG;
    a.src[k] = src[k]; a.dst[k] = dst[k]; a.width[k] = width[k];
    a.vec[k] = (width[k] % 4 == 0) && aligned16(src[k]) && aligned16(dst[k]);
  }
  long long gx = nrows < 4096 ? nrows : 4096;
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)gx, n), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

struct CatColsArgs {
  float* part[GLUE_MAX];        // dense [rows][width[k]]
  long long width[GLUE_MAX], off[GLUE_MAX];
  int n;
  float* cat;                   // dense [rows][total]
  const float* add;             // [total] or null: a row added to every row of the concatenation (forward only)
  long long rows, total;
};

// SPLIT = false: cat[b][off_k + c] = part_k[b][c] (+ add[off_k + c]);  SPLIT = true: part_k[b][c] = cat[b][off_k + c]
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_cat_cols(const CatColsArgs a) {
  const int k = blockIdx.y;
  const long long w = a.width[k], off = a.off[k];
  float* __restrict__ part = a.part[k];
  const long long n = a.rows * w;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const long long b = e / w, c = e - b * w;
    if (SPLIT) part[e] = a.cat[b * a.total + off + c];
    else a.cat[b * a.total + off + c] = a.add ? part[e] + a.add[off + c] : part[e];
  }
}

static int cat_cols(int n, float* const* part, const int64_t* width, float* cat, const float* add, int64_t rows, bool split,
                    hipStream_t s) {
  if (n < 1 || n > GLUE_MAX || !part || !width || !cat || rows < 1 || (split && add)) return IGI_E_BADARG;
  CatColsArgs a;
  a.n = n; a.cat = cat; a.add = add; a.rows = rows;
  long long off = 0, wmax = 0;
  for (int k = 0; k < n; ++k) {
    if (!part[k] || width[k] < 1) return IGI_E_BADARG;
    a.part[k] = part[k]; a.width[k] = width[k]; a.off[k] = off;
    off += width[k];
    if (width[k] > wmax) wmax = width[k];
  }
  a.total = off;
  long long gx = (rows * wmax + 255) / 256;
  if (gx > 2048) gx = 2048;
  if (split) hipLaunchKernelGGL(k_cat_cols<true>, dim3((unsigned)gx, n), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_cat_cols<false>, dim3((unsigned)gx, n), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace igi
