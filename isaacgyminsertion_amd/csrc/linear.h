// nn.Linear (+ fused activation) forward / backward on the fp32 MFMA GEMMs, for the small student MLPs
// (lin encoder, point-cloud compress, decoders, action head: tact.py:137-212, 337-339, 367-369,
// 407-410) and the offline supervised loop (runner.py:194-304).  Activations: 0 none, 1 tanh, 2 relu.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_dma.h"
#include "gemm_f32.h"
#include "teacher.h"  // IGI_HIP_TRY

namespace igi {

constexpr int LIN_NONE = 0, LIN_TANH = 1, LIN_RELU = 2;

// dz = dy * act'(y), written densely [rows][out]
__global__ __launch_bounds__(256) void k_act_grad(const float* __restrict__ dy, int lddy,
                                                  const float* __restrict__ y, int ldy, float* __restrict__ dz,
                                                  long long rows, int out, int act) {
  const long long n = rows * out;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / out;
    const int c = (int)(i - r * out);
    const float g = dy[r * lddy + c], a = y[r * ldy + c];
    dz[i] = (act == LIN_TANH) ? g * (1.0f - a * a) : (a > 0.f ? g : 0.f);
  }
}

// dst[i] = sum over parts of src[part * stride + i], in a fixed order: thread (e, grp) of a block adds
// parts grp, grp+G, ... with four independent accumulators, the G group sums are combined through LDS in
// group order.  G grows with the number of parts so that long part lists are not one serial chain of
// dependent loads (a single-block serial version took 27 us for 512 parts of 32 floats).
__global__ __launch_bounds__(256) void k_split_sum_g(float* __restrict__ dst, const float* __restrict__ src,
                                                     long long n, int parts, long long stride, int G) {
  __shared__ float sh[256];
  const int epb = 256 / G;
  const int el = threadIdx.x % epb, grp = threadIdx.x / epb;
  const long long e = (long long)blockIdx.x * epb + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n) {
    const float* p = src + e;
    const long long st = stride * G;
    int k = grp;
    for (; k + 3 * G < parts; k += 4 * G) {
      const float* q = p + (long long)k * stride;
      s0 += q[0]; s1 += q[st]; s2 += q[2 * st]; s3 += q[3 * st];
    }
    for (; k < parts; k += G) s0 += p[(long long)k * stride];
  }
  float v = (s0 + s1) + (s2 + s3);
  if (G > 1) {
    sh[threadIdx.x] = v;
    __syncthreads();
    if (grp == 0) {
      v = 0.f;
      for (int q = 0; q < G; ++q) v += sh[q * epb + el];
    }
  }
  if (grp == 0 && e < n) dst[e] = v;
}

// the weight and the bias partials of one Linear in ONE launch: blocks [0, blocks0) sum segment 0 exactly as
// k_split_sum_g would, the remaining blocks segment 1 (same group structure, same order of additions)
__global__ __launch_bounds__(256) void k_split_sum2_g(float* __restrict__ dst0, const float* __restrict__ src0,
                                                      long long n0, long long stride0, int blocks0,
                                                      float* __restrict__ dst1, const float* __restrict__ src1,
                                                      long long n1, long long stride1, int parts, int G) {
  __shared__ float sh[256];
  const bool second = (int)blockIdx.x >= blocks0;
  float* __restrict__ dst = second ? dst1 : dst0;
  const float* __restrict__ src = second ? src1 : src0;
  const long long n = second ? n1 : n0, stride = second ? stride1 : stride0;
  const int bid = second ? (int)blockIdx.x - blocks0 : (int)blockIdx.x;
  const int epb = 256 / G;
  const int el = threadIdx.x % epb, grp = threadIdx.x / epb;
  const long long e = (long long)bid * epb + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n) {
    const float* p = src + e;
    const long long st = stride * G;
    int k = grp;
    for (; k + 3 * G < parts; k += 4 * G) {
      const float* q = p + (long long)k * stride;
      s0 += q[0]; s1 += q[st]; s2 += q[2 * st]; s3 += q[3 * st];
    }
    for (; k < parts; k += G) s0 += p[(long long)k * stride];
  }
  float v = (s0 + s1) + (s2 + s3);
  if (G > 1) {
    sh[threadIdx.x] = v;
    __syncthreads();
    if (grp == 0) {
      v = 0.f;
      for (int q = 0; q < G; ++q) v += sh[q * epb + el];
    }
  }
  if (grp == 0 && e < n) dst[e] = v;
}

static inline void split_sum2(float* dst0, const float* src0, long long n0, long long stride0, float* dst1,
                              const float* src1, long long n1, long long stride1, int parts, hipStream_t s) {
  const int G = parts >= 64 ? 16 : (parts >= 8 ? 4 : 1);
  const int epb = 256 / G;
  const int b0 = (int)((n0 + epb - 1) / epb), b1 = (int)((n1 + epb - 1) / epb);
  hipLaunchKernelGGL(k_split_sum2_g, dim3((unsigned)(b0 + b1)), dim3(256), 0, s, dst0, src0, n0, stride0, b0, dst1, src1,
                     n1, stride1, parts, G);
}

static inline void split_sum(float* dst, const float* src, long long n, int parts, long long stride, hipStream_t s) {
  const int G = parts >= 64 ? 16 : (parts >= 8 ? 4 : 1);
  const int epb = 256 / G;
  hipLaunchKernelGGL(k_split_sum_g, dim3((unsigned)((n + epb - 1) / epb)), dim3(256), 0, s, dst, src, n, parts,
                     stride, G);
}

// every layer's weight / bias partials in ONE launch: segment q covers blocks [block_begin[q], block_begin[q + 1]) and
// is summed exactly as k_split_sum_g sums it (same groups, same order of additions)
constexpr int MLP_SUM_SEGS = 24;   // a chain of eight layers (16) | the token encoder's two layers (8 Linears + 4 layer norms: 20)
struct SplitSumTable {
  float* dst[MLP_SUM_SEGS];
  const float* src[MLP_SUM_SEGS];
  long long n[MLP_SUM_SEGS], stride[MLP_SUM_SEGS];
  int parts[MLP_SUM_SEGS], G[MLP_SUM_SEGS], block_begin[MLP_SUM_SEGS + 1];
  int nseg = 0, blocks = 0;
};
// queue one segment (summed exactly as split_sum / split_sum2 would sum it); false: the table is full
static inline bool split_sum_defer(SplitSumTable& t, float* dst, const float* src, long long cnt, long long stride,
                                   int parts) {
  if (t.nseg >= MLP_SUM_SEGS) return false;
  const int G = parts >= 64 ? 16 : (parts >= 8 ? 4 : 1);
  const int q = t.nseg++;
  t.dst[q] = dst; t.src[q] = src; t.n[q] = cnt; t.stride[q] = stride; t.parts[q] = parts; t.G[q] = G;
  t.block_begin[q] = t.blocks;
  t.blocks += (int)((cnt + 256 / G - 1) / (256 / G));
  return true;
}
__global__ __launch_bounds__(256) void k_split_sum_multi(const SplitSumTable t) {
  __shared__ float sh[256];
  int q = 0;
#pragma unroll
  for (int i = 1; i < MLP_SUM_SEGS; ++i)
    if (i < t.nseg && (int)blockIdx.x >= t.block_begin[i]) q = i;
  q = __builtin_amdgcn_readfirstlane(q);
  const int G = t.G[q], parts = t.parts[q];
  const long long n = t.n[q], stride = t.stride[q];
  const float* __restrict__ src = t.src[q];
  float* __restrict__ dst = t.dst[q];
  const int bid = (int)blockIdx.x - t.block_begin[q];
  const int epb = 256 / G;
  const int el = threadIdx.x % epb, grp = threadIdx.x / epb;
  const long long e = (long long)bid * epb + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n) {
    const float* p = src + e;
    const long long st = stride * G;
    int k = grp;
    for (; k + 3 * G < parts; k += 4 * G) {
      const float* qq = p + (long long)k * stride;
      s0 += qq[0]; s1 += qq[st]; s2 += qq[2 * st]; s3 += qq[3 * st];
    }
    for (; k < parts; k += G) s0 += p[(long long)k * stride];
  }
  float v = (s0 + s1) + (s2 + s3);
  if (G > 1) {   // (block-uniform)
    sh[threadIdx.x] = v;
    __syncthreads();
    if (grp == 0) {
      v = 0.f;
      for (int i = 0; i < G; ++i) v += sh[i * epb + el];
    }
  }
  if (grp == 0 && e < n) dst[e] = v;
}

static inline void split_sum_flush(SplitSumTable& t, hipStream_t s) {
  if (t.nseg == 0) return;
  t.block_begin[t.nseg] = t.blocks;
  hipLaunchKernelGGL(k_split_sum_multi, dim3((unsigned)t.blocks), dim3(256), 0, s, t);
  t.nseg = 0; t.blocks = 0;
}

static inline int linear_splitk(long long rows, int in, int out) {
  // enough k-splits to put a workgroup on every CU, at least 256 rows per split
  const long long tiles = (long long)((out + DMA_BM - 1) / DMA_BM) * ((in + 127) / 128);
  long long sk = (256 + tiles - 1) / tiles;
  // rows per split: these products are one or two tiles over a few thousand rows -- with 256 rows per split (8 k-tiles)
  // a chain of dependent k-tiles (DMA wait, 32 MFMAs, barrier) on 8 - 32 CUs.  64 rows per split (IGI_LIN_SK_ROWS):
  // 14.1 -> ~9 us per weight gradient, student step 3.85 -> 3.70 ms at the configs[3] share, 8.61 -> 8.52 at configs[2]
  // (sweep 256 / 128 / 64 / 32, two rounds on one box)
  static int rps = -1;
  if (rps < 0) { const char* e = getenv("IGI_LIN_SK_ROWS"); rps = e ? atoi(e) : 64; if (rps < 32) rps = 32; rps &= ~31; }
  const long long maxsk = rows / rps > 1 ? rows / rps : 1;
  if (sk > maxsk) sk = maxsk;
  if (sk > 64) sk = 64;
  return (int)(sk < 1 ? 1 : sk);
}

// scratch for linear_backward: dz [rows][out] + split-k slabs of dW and db
static inline size_t linear_workspace_bytes(long long rows, int in, int out) {
  if (rows < 1 || in < 1 || out < 1) return 0;
  const int sk = linear_splitk(rows, in, out);
  size_t f = (size_t)rows * out;
  f = (f + 3) / 4 * 4;
  if (sk > 1) f += (size_t)sk * ((size_t)out * in + out);
  return f * sizeof(float) + 16;
}

static int linear_forward(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, long long rows,
                          int in, int out, int act, hipStream_t s) {
  if (!x || !W || !y || rows < 0 || in < 1 || out < 1 || ldx < in || ldy < out || act < 0 || act > 2 ||
      rows > (1LL << 30))
    return IGI_E_BADARG;
  if (act != LIN_NONE && !b) return IGI_E_BADARG;
  if (rows == 0) return 0;
  GemmArgs g;
  g.A = x; g.lda = ldx;
  g.B = W; g.ldb = in;
  g.bias = b;
  g.M = (int)rows; g.N = out; g.K = in;
  g.C = y; g.ldc = ldy;
  g.epilogue = act == LIN_TANH ? EPI_BIAS_TANH : (act == LIN_RELU ? EPI_BIAS_RELU : (b ? EPI_BIAS : EPI_STORE));
  return (int)gemm(g, true, true, s);
}

// One backward level in one grid: a weight-gradient product (reduction-major operands, split-K slab store, bias sums)
// and a data-gradient product (k-contiguous dz, reduction-major W, any epilogue of the generic body: plain store,
// ReLU', tanh').  Kinds: 0 / 1 = weight gradient on 128 x 128 / 128 x 64 tiles, 4 / 5 = data gradient on 128 x 128 /
// 128 x 64 tiles.  The weight-gradient tiles (the longer reductions) lead the grid.
__global__ __launch_bounds__(DMA_THREADS, 4) void gemm_dma_mlp_level_kernel(const GemmMulti table_in_kernarg) {
  (void)table_in_kernarg;
  gemm_multi_cptr gr = (gemm_multi_cptr)__builtin_amdgcn_kernarg_segment_ptr();
  const int bid = blockIdx.x;
  int p = (gr->n > 1 && bid >= gr->tile_end[0]) ? 1 : 0;
  p = __builtin_amdgcn_readfirstlane(p);
  const int start = (p > 0 ? gr->tile_end[0] : 0);
  int local = bid - start;
  if ((start & 7) == 0) local = xcd_remap(local, gr->tile_end[p] - start);
  const GemmArgs& g = *(const GemmArgs*)&gr->g[p];
  const int kind = gr->kind[p];
  if (kind == 0) gemm_dma_body<128, false, false, 0, 2, DMA_BM, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else if (kind == 1) gemm_dma_body<64, false, false, 0, 2, DMA_BM, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else if (kind == 4) gemm_dma_body<128, true, false, 0, 2>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else gemm_dma_body<64, true, false, 0, 2>(g, gr->n_tiles[p], gr->m_tiles[p], local);
}

// wg (may be NULL) and dg (may be NULL): launched together when both fit the LDS-DMA kernel, else one by one
static int mlp_level(GemmArgs* wg, GemmArgs* dg, hipStream_t s) {
  auto prep_w = [&](GemmArgs& g) {
    if (g.splitk < 1) g.splitk = 1;
    return dma_eligible(g, false, false) && !g.gather;
  };
  auto prep_d = [&](GemmArgs& g) {
    if (g.splitk < 1) g.splitk = 1;
    if (!dma_eligible(g, true, false) || g.gather) return false;
    g.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.N & 3) == 0 && (!g.aux || (aligned16(g.aux) && (g.ldaux & 3) == 0));
    return true;
  };
  const bool wok = wg && prep_w(*wg), dok = dg && prep_d(*dg);
  if (wok && dok) {
    GemmMulti mt_;
    auto put = [&](const GemmArgs& g, bool wgrad) {
      const int bn = g.N <= 64 ? 64 : 128;
      const int nt = (g.N + bn - 1) / bn, mtl = (g.M + DMA_BM - 1) / DMA_BM;
      GemmArgs gg = g;
      if (wgrad) gg.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.sCsplit & 3) == 0 && (g.N & 3) == 0;
      dma_set_divs(gg, nt, mtl);
      const int k = mt_.n++;
      mt_.g[k] = gg; mt_.n_tiles[k] = nt; mt_.m_tiles[k] = mtl;
      mt_.kind[k] = wgrad ? (bn == 64 ? 1 : 0) : (bn == 64 ? 5 : 4);
      mt_.tile_end[k] = (k > 0 ? mt_.tile_end[k - 1] : 0) + nt * mtl * g.nbatch * g.splitk;
    };
    put(*wg, true);
    put(*dg, false);
    constexpr size_t ring = sizeof(float) * 2 * (DMA_BM + 128) * DMA_BK, epi = sizeof(float) * DMA_WAVES * 64 * (32 + 4);
    constexpr size_t shm = ring > epi ? ring : epi;
    static bool attr = false;
    if (!attr) {
      IGI_HIP_TRY(hipFuncSetAttribute((const void*)gemm_dma_mlp_level_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)shm));
      attr = true;
    }
    IGI_LAUNCH(gemm_dma_mlp_level_kernel, dim3(mt_.tile_end[mt_.n - 1]), dim3(DMA_THREADS), shm, s, mt_);
    return (int)hipGetLastError();
  }
  if (dg) IGI_HIP_TRY(gemm(*dg, true, false, s));
  if (wg) IGI_HIP_TRY(gemm(*wg, false, false, s));
  return (int)hipGetLastError();
}

// dy: gradient w.r.t. the layer OUTPUT y (after the activation); dx may be NULL (first layer), db may be NULL,
// dW may be NULL (frozen layer: only the data gradient is wanted; db must be NULL too then).
static int linear_backward(const float* x, int ldx, const float* W, const float* y, int ldy, const float* dy, int lddy,
                           float* dx, int lddx, float* dW, float* db, long long rows, int in, int out, int act,
                           void* workspace, size_t workspace_bytes, hipStream_t s, SplitSumTable* defer = nullptr) {
  // defer: queue the sums of the split-row partials instead of launching them (the caller flushes the table once, e.g.
  // after the token encoder's eight Linears); the workspace must then stay untouched until that flush
  if (!x || !W || !dy || (!dW && db) || rows < 0 || in < 1 || out < 1 || ldx < in || lddy < out || act < 0 || act > 2 ||
      rows > (1LL << 30) || (dx && lddx < in))
    return IGI_E_BADARG;
  if (act != LIN_NONE && (!y || ldy < out)) return IGI_E_BADARG;
  if (rows == 0) {
    if (dW) IGI_HIP_TRY(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)out * in, s));
    if (db) IGI_HIP_TRY(hipMemsetAsync(db, 0, sizeof(float) * out, s));
    return 0;
  }
  if (!workspace || workspace_bytes < linear_workspace_bytes(rows, in, out)) return IGI_E_WORKSPACE;
  float* ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15);
  const float* dz = dy;
  int lddz = lddy;
  size_t dz_f = ((size_t)rows * out + 3) / 4 * 4;
  if (act != LIN_NONE) {
    long long nb = (rows * out + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_act_grad, dim3((unsigned)nb), dim3(256), 0, s, dy, lddy, y, ldy, ws, rows, out, act);
    dz = ws;
    lddz = out;
  }
  GemmArgs gdx;
  if (dx) {  // dx[rows][in] = dz . W
    gdx.A = dz; gdx.lda = lddz;
    gdx.B = W; gdx.ldb = in;
    gdx.M = (int)rows; gdx.N = in; gdx.K = out;
    gdx.C = dx; gdx.ldc = lddx;
    if (!dW) IGI_HIP_TRY(gemm(gdx, true, false, s));
  }
  if (dW) {  // dW[out][in] = dz^T x, db = column sums of dz; split over the rows, summed in fixed order
    int sk = linear_splitk(rows, in, out);
    int kchunk = 0;
    if (sk > 1) {  // whole 32-row k-tiles per split, and no empty split
      kchunk = (int)(((rows + sk - 1) / sk + DMA_BK - 1) / DMA_BK * DMA_BK);
      sk = (int)((rows + kchunk - 1) / kchunk);
    }
    float* slabW = ws + dz_f;
    float* slabB = slabW + (size_t)sk * out * in;
    GemmArgs g;
    g.A = dz; g.lda = lddz;
    g.B = x; g.ldb = ldx;
    g.M = out; g.N = in; g.K = (int)rows;
    if (sk > 1) {
      g.C = slabW; g.ldc = in;
      g.Cbias = db ? slabB : nullptr;
      g.splitk = sk; g.kchunk = kchunk;
      g.sCsplit = (long long)out * in; g.sCbiasSplit = out;
    } else {
      g.C = dW; g.ldc = in;
      g.Cbias = db;
    }
    // the two products read the same dz and do not depend on each other: one grid (mlp_level) when both fit the
    // LDS-DMA kernel -- one launch and one fill / drain less per layer (the token encoder's eight Linears, standalone
    // HipLinear layers); same tiles' arithmetic, same bits
    {
      const int rc = mlp_level(&g, dx ? &gdx : nullptr, s);
      if (rc) return rc;
    }
    if (sk > 1) {
      const long long nW = (long long)out * in;
      // weight and bias partials in one launch (as two segments of k_slab_reduce they took 24 us against 2 x 4.5 us: its
      // 16-byte path walks the partials of four elements as one dependent chain; parallelism over elements wins here)
      // both segments or neither: with room for the weight segment only it would be summed twice (flush, then directly)
      const bool room = defer && defer->nseg + (db ? 2 : 1) <= MLP_SUM_SEGS;
      const bool queued = room && split_sum_defer(*defer, dW, slabW, nW, nW, sk) &&
                          (!db || split_sum_defer(*defer, db, slabB, (long long)out, (long long)out, sk));
      if (queued) return (int)hipGetLastError();
      if (defer) { split_sum_flush(*defer, s); }   // table full: everything queued so far, then this layer directly
      if (db) split_sum2(dW, slabW, nW, nW, db, slabB, (long long)out, (long long)out, sk, s);
      else split_sum(dW, slabW, nW, sk, nW, s);
    }
  }
  return (int)hipGetLastError();
}

}  // namespace igi
