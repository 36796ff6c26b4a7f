// Token decoder of the student: a stack of nn.TransformerEncoderLayer(d_model=32, nhead=2,
// dim_feedforward=128, activation="gelu", batch_first=True, norm_first=True) over S <= 8 tokens
// (algo/models/transformer/tact.py:137-158), forward and backward.
//
//   per layer:  x1 = x  + drop(out_proj(attn(in_proj(LN1(x)))))          (sa block,  norm_first)
//               x2 = x1 + drop(linear2(drop(gelu(linear1(LN2(x1))))))    (ff block)
//
// Rows are tokens, sample-major (row = b * S + s); every Linear is a row-wise product and runs on the
// MFMA GEMMs (linear.h: igi_linear_forward / backward with deterministic split sums for the weight
// gradients); everything else is a handful of small fused kernels:
//   k_resid_ln_fwd : residual add (+ dropout of the branch) fused with the NEXT LayerNorm; a 32-lane half
//                    wave owns a row (lane = feature), statistics by DPP row sums
//   k_attn_fwd/bwd : softmax(q k^T / sqrt(16)) v for an S x S block per (sample, head); 16 lanes own a
//                    (sample, head) pair, lane = head dimension, dot products are 16-lane DPP row sums;
//                    backward recomputes the probabilities (nothing S x S is ever stored)
//   k_gelu_fwd/bwd : exact (erf) GELU + dropout
//   k_ln_bwd       : LayerNorm backward + residual gradient add + (optionally) the dropout mask of the
//                    branch below it; per-block partial sums for the LayerNorm weight / bias gradients,
//                    reduced in fixed order
// Dropout masks are a counter-based hash of (seed, site, element): backward regenerates them, nothing is stored.
// Parameter vector per layer, in nn.TransformerEncoderLayer's registration order:
//   in_proj_weight [3d][d], in_proj_bias [3d], out_proj.weight [d][d], out_proj.bias [d],
//   linear1.weight [ff][d], linear1.bias [ff], linear2.weight [d][ff], linear2.bias [d],
//   norm1.weight [d], norm1.bias [d], norm2.weight [d], norm2.bias [d]
#pragma once
#include <hip/hip_runtime.h>

#include "linear.h"

namespace igi {

constexpr int TOK_D = 32, TOK_DH = 16, TOK_MAX_S = 8;
constexpr float TOK_LN_EPS = 1e-5f;

struct TokenPlan {
  int B, S, d, H, ff, L;
  long long R;           // token rows
  float p;               // dropout probability (0 when not training)
  long long per_layer;   // parameters per layer
  long long o_inw, o_inb, o_ow, o_ob, o_w1, o_b1, o_w2, o_b2, o_n1w, o_n1b, o_n2w, o_n2b;
  // saved activations per layer (float offsets inside a layer's slab)
  long long a_x, a_st1, a_xn1, a_qkv, a_ctx, a_x1, a_st2, a_xn2, a_z, a_h, a_layer;
  // backward scratch (float offsets after the L slabs)
  long long s_g0, s_g1, s_rA, s_rB, s_wide0, s_wide1, s_lnpart, s_lin;
  size_t lin_bytes, total_bytes;
  int ln_blocks;
  long long s_part; int bwd_samples, bwd_grid;   // fused backward (k_token_bwd)
};
constexpr int TB_ROWS = 64;                     // token rows per workgroup of the fused backward

static inline long long ru4ll(long long x) { return (x + 3) & ~3LL; }

static int make_token_plan(const igi_token_cfg* c, TokenPlan* p) {
  if (!c || c->batch < 1 || c->seq < 1 || c->layers < 1 || c->ff < 4 || c->dropout < 0.f || c->dropout >= 1.f)
    return IGI_E_BADARG;
  if (c->d_model != TOK_D || c->nhead * TOK_DH != TOK_D || c->seq > TOK_MAX_S || (c->ff & 3) || c->layers > 8)
    return IGI_E_UNSUPPORTED;
  p->B = c->batch; p->S = c->seq; p->d = c->d_model; p->H = c->nhead; p->ff = c->ff; p->L = c->layers;
  p->R = (long long)c->batch * c->seq;
  if (p->R > (1LL << 26)) return IGI_E_UNSUPPORTED;
  p->p = c->training ? c->dropout : 0.f;
  const int d = p->d, ff = p->ff;
  long long o = 0;
  p->o_inw = o; o += 3LL * d * d;
  p->o_inb = o; o += 3 * d;
  p->o_ow = o; o += (long long)d * d;
  p->o_ob = o; o += d;
  p->o_w1 = o; o += (long long)ff * d;
  p->o_b1 = o; o += ff;
  p->o_w2 = o; o += (long long)d * ff;
  p->o_b2 = o; o += d;
  p->o_n1w = o; o += d;
  p->o_n1b = o; o += d;
  p->o_n2w = o; o += d;
  p->o_n2b = o; o += d;
  p->per_layer = o;
  const long long R = p->R;
  long long a = 0;
  auto take = [&](long long n) { long long r = a; a += ru4ll(n); return r; };
  p->a_x = take(R * d); p->a_st1 = take(R * 2); p->a_xn1 = take(R * d); p->a_qkv = take(R * 3 * d);
  p->a_ctx = take(R * d); p->a_x1 = take(R * d); p->a_st2 = take(R * 2); p->a_xn2 = take(R * d);
  p->a_z = take(R * ff); p->a_h = take(R * ff);
  p->a_layer = a;
  long long s = a * p->L;
  auto stake = [&](long long n) { long long r = s; s += ru4ll(n); return r; };
  p->ln_blocks = (int)((R + 63) / 64 < 256 ? (R + 63) / 64 : 256);
  p->s_g0 = stake(R * d); p->s_g1 = stake(R * d); p->s_rA = stake(R * d); p->s_rB = stake(R * d);
  const int wide = 3 * d > ff ? 3 * d : ff;
  p->s_wide0 = stake(R * wide); p->s_wide1 = stake(R * wide);
  p->s_lnpart = stake((long long)p->ln_blocks * 2 * d * 2 * p->L);   // one per layer norm: the sums are deferred
  p->bwd_samples = (int)(TB_ROWS / p->S);           // fused backward: samples per workgroup (see k_token_bwd)
  if ((long long)p->bwd_samples * 256 > p->B) p->bwd_samples = (int)(p->B / 256 > 1 ? p->B / 256 : 1);
  p->bwd_grid = (int)((p->B + p->bwd_samples - 1) / p->bwd_samples);
  // one 100 KB gradient record per workgroup: beyond 1024 of them (>= 21 K samples of three tokens) the backward runs
  // launch by launch on split-row slabs instead, and no record space is reserved
  if (p->bwd_grid > 1024) p->bwd_grid = 0;
  p->s_part = stake((long long)p->bwd_grid * p->per_layer * p->L);   // one gradient record per workgroup
  p->s_lin = stake(0);
  size_t lb = linear_workspace_bytes(R, d, 3 * d);
  const size_t c1 = linear_workspace_bytes(R, d, d), c2 = linear_workspace_bytes(R, d, ff),
               c3 = linear_workspace_bytes(R, ff, d);
  if (c1 > lb) lb = c1;
  if (c2 > lb) lb = c2;
  if (c3 > lb) lb = c3;
  p->lin_bytes = lb;
  p->total_bytes = sizeof(float) * (size_t)s + lb * 4 * (size_t)p->L + 64;   // one Linear workspace per call (deferred sums)
  return 0;
}

// ---- dropout: keep iff hash(seed, site, index) >= p * 2^32; kept values are scaled by 1/(1-p)
__device__ __forceinline__ unsigned int tok_hash(unsigned long long seed, unsigned int site, unsigned int idx) {
  unsigned int x = idx * 0x9E3779B1u ^ (unsigned int)seed ^ (site * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  x += (unsigned int)(seed >> 32);
  x ^= x >> 15; x *= 0x2C1B3C6Du;
  x ^= x >> 12; x *= 0x297A2D39u;
  x ^= x >> 15;
  return x;
}
struct Drop {
  unsigned long long seed;
  unsigned int site, thresh;  // thresh = 0: no dropout
  float scale;
};
static inline Drop make_drop(float p, unsigned long long seed, unsigned int site) {
  Drop d;
  d.seed = seed; d.site = site;
  d.thresh = p > 0.f ? (unsigned int)((double)p * 4294967296.0) : 0u;
  d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}
__device__ __forceinline__ Drop make_drop_dev(float p, unsigned long long seed, unsigned int site) {   // = make_drop on the device
  Drop d;
  d.seed = seed; d.site = site;
  d.thresh = p > 0.f ? (unsigned int)((double)p * 4294967296.0) : 0u;
  d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}
__device__ __forceinline__ float drop_apply(const Drop& d, unsigned int idx, float v) {
  if (d.thresh == 0u) return v;
  return tok_hash(d.seed, d.site, idx) >= d.thresh ? v * d.scale : 0.f;
}

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16-lane DPP row, in every lane
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}
__device__ __forceinline__ float half32_sum(float v) {  // sum over 32 consecutive lanes
  v = row16_sum(v);
  return v + __shfl_xor(v, 16, 64);
}

// x_out = x_prev + drop(delta)  (delta may be null: x_out = x_prev), then xn = LN(x_out) (gamma may be
// null: residual only).  Half wave per row.
__global__ __launch_bounds__(256) void k_resid_ln_fwd(const float* __restrict__ xprev, const float* __restrict__ delta,
                                                      Drop dr, float* __restrict__ xout,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ xn, float* __restrict__ stats, long long R) {
  const int f = threadIdx.x & 31;
  const long long hw = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const long long nhw = ((long long)gridDim.x * blockDim.x) >> 5;
  for (long long r = hw; r < R; r += nhw) {
    const long long i = r * TOK_D + f;
    float v = xprev[i];
    if (delta) v += drop_apply(dr, (unsigned int)i, delta[i]);
    if (xout) xout[i] = v;
    if (gamma) {
      const float mean = half32_sum(v) * (1.0f / TOK_D);
      const float c = v - mean;
      const float var = half32_sum(c * c) * (1.0f / TOK_D);
      const float rstd = 1.0f / sqrtf(var + TOK_LN_EPS);
      xn[i] = c * rstd * gamma[f] + beta[f];
      if (f == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    }
  }
}

// LayerNorm backward + residual:  dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dxn * gamma.
// Optionally also writes dmask = drop_mask(dx) for the branch feeding the residual BELOW this norm.
// Per-block partial sums of (dxn * xhat, dxn) per feature go to part[block][2][32].
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dxn, const float* __restrict__ x,
                                                const float* __restrict__ stats, const float* __restrict__ gamma,
                                                const float* __restrict__ dres, float* __restrict__ dx, Drop dr,
                                                float* __restrict__ dmask, float* __restrict__ part, long long R) {
  __shared__ float red[2][8][TOK_D];
  const int f = threadIdx.x & 31, hwl = threadIdx.x >> 5;
  const long long hw = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const long long nhw = ((long long)gridDim.x * blockDim.x) >> 5;
  const float gm = gamma[f];
  float sg = 0.f, sb = 0.f;
  for (long long r = hw; r < R; r += nhw) {
    const long long i = r * TOK_D + f;
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    const float xhat = (x[i] - mean) * rstd;
    const float d = dxn[i];
    sg += d * xhat;
    sb += d;
    const float g = d * gm;
    const float m1 = half32_sum(g) * (1.0f / TOK_D);
    const float m2 = half32_sum(g * xhat) * (1.0f / TOK_D);
    const float v = dres[i] + rstd * (g - m1 - xhat * m2);
    dx[i] = v;
    if (dmask) dmask[i] = drop_apply(dr, (unsigned int)i, v);
  }
  red[0][hwl][f] = sg;
  red[1][hwl][f] = sb;
  __syncthreads();
  if (threadIdx.x < 2 * TOK_D) {
    const int w = threadIdx.x / TOK_D, ff = threadIdx.x % TOK_D;
    float a = 0.f;
    for (int q = 0; q < 8; ++q) a += red[w][q][ff];
    part[((long long)blockIdx.x * 2 + w) * TOK_D + ff] = a;
  }
}

// masked copy: dst = drop_mask(src)
__global__ __launch_bounds__(256) void k_drop_copy(const float* __restrict__ src, float* __restrict__ dst, Drop dr,
                                                   long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = drop_apply(dr, (unsigned int)i, src[i]);
}

// attention core.  16 lanes per (sample, head); lane = head dimension.  qkv row = [q(32) | k(32) | v(32)].
template <int S>
__global__ __launch_bounds__(256) void k_attn_fwd(const float* __restrict__ qkv, float* __restrict__ ctx, Drop dr,
                                                  long long pairs, int H) {
  const int dl = threadIdx.x & 15;
  const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
  const float scale = 0.25f;  // 1 / sqrt(16)
  for (long long g = g0; g < pairs; g += ng) {
    const long long b = g / H;
    const int h = (int)(g - b * H);
    float q[S], k[S], v[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float* row = qkv + (b * S + s) * (3 * TOK_D) + h * TOK_DH + dl;
      q[s] = row[0]; k[s] = row[TOK_D]; v[s] = row[2 * TOK_D];
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
      float sc[S], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < S; ++j) { sc[j] = row16_sum(q[i] * k[j]) * scale; mx = fmaxf(mx, sc[j]); }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < S; ++j) { sc[j] = __expf(sc[j] - mx); den += sc[j]; }
      const float inv = 1.0f / den;
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const float pij = drop_apply(dr, (unsigned int)((g * S + i) * S + j), sc[j] * inv);
        o += pij * v[j];
      }
      ctx[(b * S + i) * TOK_D + h * TOK_DH + dl] = o;
    }
  }
}

template <int S>
__global__ __launch_bounds__(256) void k_attn_bwd(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                  float* __restrict__ dqkv, Drop dr, long long pairs, int H) {
  const int dl = threadIdx.x & 15;
  const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
  const float scale = 0.25f;
  for (long long g = g0; g < pairs; g += ng) {
    const long long b = g / H;
    const int h = (int)(g - b * H);
    float q[S], k[S], v[S], dc[S], dq[S], dk[S], dv[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float* row = qkv + (b * S + s) * (3 * TOK_D) + h * TOK_DH + dl;
      q[s] = row[0]; k[s] = row[TOK_D]; v[s] = row[2 * TOK_D];
      dc[s] = dctx[(b * S + s) * TOK_D + h * TOK_DH + dl];
      dq[s] = 0.f; dk[s] = 0.f; dv[s] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
      float pr[S], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < S; ++j) { pr[j] = row16_sum(q[i] * k[j]) * scale; mx = fmaxf(mx, pr[j]); }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < S; ++j) { pr[j] = __expf(pr[j] - mx); den += pr[j]; }
      const float inv = 1.0f / den;
      float dp[S], dot = 0.f;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        pr[j] *= inv;
        // the mask factor m_ij in {0, 1/(1-p)}: P_drop = P * m
        const float m = drop_apply(dr, (unsigned int)((g * S + i) * S + j), 1.0f);
        dv[j] += pr[j] * m * dc[i];
        dp[j] = row16_sum(dc[i] * v[j]) * m;     // d(loss)/dP_ij
        dot += dp[j] * pr[j];
      }
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const float ds = pr[j] * (dp[j] - dot) * scale;  // d(loss)/d(score_ij) incl. the 1/sqrt(dh)
        dq[i] += ds * k[j];
        dk[j] += ds * q[i];
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float* row = dqkv + (b * S + s) * (3 * TOK_D) + h * TOK_DH + dl;
      row[0] = dq[s]; row[TOK_D] = dk[s]; row[2 * TOK_D] = dv[s];
    }
  }
}

// h = drop(gelu(z)), exact erf form (nn.GELU default / activation="gelu")
__global__ __launch_bounds__(256) void k_gelu_fwd(const float* __restrict__ z, float* __restrict__ h, Drop dr,
                                                  long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x = z[i];
    h[i] = drop_apply(dr, (unsigned int)i, 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)));
  }
}
__global__ __launch_bounds__(256) void k_gelu_bwd(const float* __restrict__ dh, const float* __restrict__ z,
                                                  float* __restrict__ dz, Drop dr, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x = z[i];
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    dz[i] = drop_apply(dr, (unsigned int)i, dh[i]) * (cdf + x * pdf);
  }
}

static inline int tok_blocks(long long n, int per_block, int cap = 2048) {
  long long b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

template <typename F>
static int attn_dispatch(int S, F&& f) {
  switch (S) {
    case 1: f(std::integral_constant<int, 1>()); return 0;
    case 2: f(std::integral_constant<int, 2>()); return 0;
    case 3: f(std::integral_constant<int, 3>()); return 0;
    case 4: f(std::integral_constant<int, 4>()); return 0;
    case 5: f(std::integral_constant<int, 5>()); return 0;
    case 6: f(std::integral_constant<int, 6>()); return 0;
    case 7: f(std::integral_constant<int, 7>()); return 0;
    case 8: f(std::integral_constant<int, 8>()); return 0;
  }
  return IGI_E_UNSUPPORTED;
}

static inline float* tok_ws(void* ws) {
  return reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15);
}

// dropout sites of layer l
enum { SITE_ATTN = 0, SITE_SA = 1, SITE_FF_ACT = 2, SITE_FF = 3 };

// ---------------------------------------------------------------------------------------------------------------------
// The whole forward as ONE launch (round 5).  As 8 launches per layer + 1 (four GEMMs, two residual + LayerNorm kernels,
// attention, GELU) the stack was 17 launches of 4 - 8 us for ~0.15 GFLOP.  Here a workgroup of four waves carries up to
// 128 token rows (32 samples; 16 when a sample has more than four tokens) through every layer with the residual stream,
// the LayerNorm outputs, q / k / v and the feed-forward activations in LDS; a layer's four weight matrices are brought into
// LDS once per workgroup and layer; every activation the backward pass reads (TokenPlan::a_*) is written out once.
// Arithmetic = the separate kernels', operation by operation: the Linears are the same fmaf chains on
// v_mfma_f32_32x32x2_f32 in the LDS-DMA kernel's k order (K = 32 and 128 are whole k-tiles), LayerNorm is the same
// half-wave-per-row code, attention the same 16-lanes-per-(sample, head) code, the same erf GELU, the same dropout hash on
// the same element indices -- outputs and saved activations are bit-identical (tests/test_gpu_token_encoder.py).
// ---------------------------------------------------------------------------------------------------------------------
// Waves per workgroup of the two fused kernels.  A workgroup's time is its chain of ~20 phases per layer, each as wide as the
// workgroup: at 8192 x 2 tokens the forward ran 59.5 / 43.5 / 37.6 us with 256 / 512 / 1024 threads, the backward 95 / 64.6 /
// 69.4 (2048 x 3: 39 / 32.9 / 28.6 and 57.5 / 44.9 / 60.7: at 1024 threads the backward's attention pass has 128 registers).
#ifndef TF_THREADS_N
#define TF_THREADS_N 1024
#endif
#ifndef TB_THREADS_N
#define TB_THREADS_N 512
#endif
constexpr int TF_THREADS = TF_THREADS_N, TF_NW = TF_THREADS / 64, TF_HW = TF_THREADS / 32;   // forward
constexpr int TB_THREADS = TB_THREADS_N, TB_NW = TB_THREADS / 64, TB_HW = TB_THREADS / 32;   // backward
constexpr int TF_ROWS = 128;                 // token rows per workgroup (LDS images are this tall)
constexpr int TF_LDX = TOK_D + 4;            // 36: row pitch of the 32-wide images (16-byte reads of 16 rows: 16 bank groups)
constexpr int TF_LDZ = 128 + 4;              // 132: q|k|v (96 used) and the feed-forward activations
constexpr int TF_FF = 128;
constexpr int TF_W_FLOATS = 3 * TOK_D * TF_LDX + TOK_D * TF_LDX + TF_FF * TF_LDX + TOK_D * TF_LDZ;
constexpr int TF_LDS_FLOATS = 2 * TF_ROWS * TF_LDX + TF_ROWS * TF_LDZ + TF_W_FLOATS;

struct TokFwdArgs {
  const float* x; const float* params; float* y; float* W;   // W: the (aligned) workspace base
  long long R, per_layer, a_layer;
  long long o_inw, o_inb, o_ow, o_ob, o_w1, o_b1, o_w2, o_b2, o_n1w, o_n1b, o_n2w, o_n2b;
  long long a_x, a_st1, a_xn1, a_qkv, a_ctx, a_x1, a_st2, a_xn2, a_z, a_h;
  int S, H, L, rows_per_wg;
  float p; unsigned long long seed;
};

template <int S>
__global__ __launch_bounds__(TF_THREADS) void k_token_fwd(const TokFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bx = smem;                               // residual stream [rows][36]
  float* bn = bx + TF_ROWS * TF_LDX;              // LayerNorm output / attention context / branch outputs [rows][36]
  float* bz = bn + TF_ROWS * TF_LDX;              // q|k|v, then z / h [rows][132]
  float* win = bz + TF_ROWS * TF_LDZ;             // in_proj [96][36]
  float* wout = win + 3 * TOK_D * TF_LDX;         // out_proj [32][36]
  float* w1 = wout + TOK_D * TF_LDX;              // linear1 [128][36]
  float* w2 = w1 + TF_FF * TF_LDX;                // linear2 [32][132]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const long long r0 = (long long)blockIdx.x * a.rows_per_wg;
  const int nrows = (int)min((long long)a.rows_per_wg, a.R - r0);
  const int mtiles = (nrows + 31) >> 5;
  const int f = tid & 31, hw = tid >> 5;          // LayerNorm: half wave hw (of 8) owns rows hw, hw + 8, ...

  // C[rows][N] = A[rows][K] . Wimg[N][K]^T + bias, into LDS (pitch ldc) and, if G, to global (pitch N); tiles round-robin
  auto linear = [&](const float* A, int lda, int K, const float* Wimg, int ldw, int N, const float* bias, float* C, int ldc,
                    float* G) {
    const int ntiles = N >> 5;
    for (int t = wave; t < mtiles * ntiles; t += TF_NW) {
      const int mt = t / ntiles, nt = t - mt * ntiles;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* ap = A + (32 * mt + l31) * lda + 4 * h;
      const float* bp = Wimg + (32 * nt + l31) * ldw + 4 * h;
      for (int c = 0; c < K; c += 8) {            // the LDS-DMA kernel's order: k = 8 c + 4 h + j at step j
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + c);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
      }
      const int n = 32 * nt + l31;
      const float b = bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[r] + b;
        C[row * ldc + n] = v;
        if (G && row < nrows) G[(r0 + row) * N + n] = v;
      }
    }
  };
  // xout = xprev (+ drop(delta)) ; xn = LN(xout): k_resid_ln_fwd's arithmetic on LDS rows
  auto resid_ln = [&](const float* xprev_g, bool have_delta, const Drop& dr, float* xout_g, const float* gamma, const float* beta,
                      float* xn_g, float* stats_g) {
    for (int row = hw; row < nrows; row += TF_HW) {
      const long long i = (r0 + row) * TOK_D + f;
      float v = xprev_g ? xprev_g[i] : bx[row * TF_LDX + f];
      if (have_delta) v += drop_apply(dr, (unsigned int)i, bn[row * TF_LDX + f]);
      bx[row * TF_LDX + f] = v;
      if (xout_g) xout_g[i] = v;
      if (gamma) {
        const float mean = half32_sum(v) * (1.0f / TOK_D);
        const float c = v - mean;
        const float var = half32_sum(c * c) * (1.0f / TOK_D);
        const float rstd = 1.0f / sqrtf(var + TOK_LN_EPS);
        const float o = c * rstd * gamma[f] + beta[f];
        bn[row * TF_LDX + f] = o;
        xn_g[i] = o;
        if (f == 0) { stats_g[2 * (r0 + row)] = mean; stats_g[2 * (r0 + row) + 1] = rstd; }
      }
    }
  };

  Drop pending = make_drop_dev(0.f, a.seed, 0);
  for (int l = 0; l < a.L; ++l) {
    const float* P = a.params + (long long)l * a.per_layer;
    float* A = a.W + (long long)l * a.a_layer;
    // ---- this layer's weights -> LDS (coalesced 16-byte loads); the previous layer's readers are behind its last barrier
    for (int u = tid; u < 3 * TOK_D * 8; u += TF_THREADS)
      *reinterpret_cast<float4*>(win + (u >> 3) * TF_LDX + 4 * (u & 7)) = *reinterpret_cast<const float4*>(P + a.o_inw + 4 * u);
    for (int u = tid; u < TOK_D * 8; u += TF_THREADS)
      *reinterpret_cast<float4*>(wout + (u >> 3) * TF_LDX + 4 * (u & 7)) = *reinterpret_cast<const float4*>(P + a.o_ow + 4 * u);
    for (int u = tid; u < TF_FF * 8; u += TF_THREADS)
      *reinterpret_cast<float4*>(w1 + (u >> 3) * TF_LDX + 4 * (u & 7)) = *reinterpret_cast<const float4*>(P + a.o_w1 + 4 * u);
    for (int u = tid; u < TOK_D * 32; u += TF_THREADS)
      *reinterpret_cast<float4*>(w2 + (u >> 5) * TF_LDZ + 4 * (u & 31)) = *reinterpret_cast<const float4*>(P + a.o_w2 + 4 * u);
    // ---- x_l = x_{l-1} + drop(ff branch of l - 1) (l == 0: the input), xn1 = LN1(x_l)
    resid_ln(l == 0 ? a.x : nullptr, l > 0, pending, A + a.a_x, P + a.o_n1w, P + a.o_n1b, A + a.a_xn1, A + a.a_st1);
    __syncthreads();
    linear(bn, TF_LDX, TOK_D, win, TF_LDX, 3 * TOK_D, P + a.o_inb, bz, TF_LDZ, A + a.a_qkv);
    __syncthreads();
    // ---- attention: 16 lanes per (sample, head), k_attn_fwd's arithmetic on the LDS rows; context -> bn
    {
      const Drop da = make_drop_dev(a.p, a.seed, 4 * l + SITE_ATTN);
      const int dl = tid & 15;
      const int nsamp = nrows / S;
      const float scale = 0.25f;
      for (int g = tid >> 4; g < nsamp * a.H; g += TF_THREADS / 16) {
        const int bl = g / a.H, hh = g - bl * a.H;
        const long long gg = (r0 / S + bl) * a.H + hh;       // the global (sample, head) index: dropout element indices
        float q[S], k[S], v[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
          const float* row = bz + (bl * S + s2) * TF_LDZ + hh * TOK_DH + dl;
          q[s2] = row[0]; k[s2] = row[TOK_D]; v[s2] = row[2 * TOK_D];
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
          float sc[S], mx = -INFINITY;
#pragma unroll
          for (int j = 0; j < S; ++j) { sc[j] = row16_sum(q[i] * k[j]) * scale; mx = fmaxf(mx, sc[j]); }
          float den = 0.f;
#pragma unroll
          for (int j = 0; j < S; ++j) { sc[j] = __expf(sc[j] - mx); den += sc[j]; }
          const float inv = 1.0f / den;
          float o = 0.f;
#pragma unroll
          for (int j = 0; j < S; ++j) {
            const float pij = drop_apply(da, (unsigned int)((gg * S + i) * S + j), sc[j] * inv);
            o += pij * v[j];
          }
          bn[(bl * S + i) * TF_LDX + hh * TOK_DH + dl] = o;
          A[a.a_ctx + (r0 + bl * S + i) * TOK_D + hh * TOK_DH + dl] = o;
        }
      }
    }
    __syncthreads();
    linear(bn, TF_LDX, TOK_D, wout, TF_LDX, TOK_D, P + a.o_ob, bz, TF_LDZ, nullptr);    // sa branch -> bz[:, 0..31]
    __syncthreads();
    // ---- x1 = x + drop(sa branch), xn2 = LN2(x1)   (the branch is read from bz here)
    {
      const Drop ds = make_drop_dev(a.p, a.seed, 4 * l + SITE_SA);
      for (int row = hw; row < nrows; row += TF_HW) {
        const long long i = (r0 + row) * TOK_D + f;
        float v = bx[row * TF_LDX + f] + drop_apply(ds, (unsigned int)i, bz[row * TF_LDZ + f]);
        bx[row * TF_LDX + f] = v;
        A[a.a_x1 + i] = v;
        const float mean = half32_sum(v) * (1.0f / TOK_D);
        const float c = v - mean;
        const float var = half32_sum(c * c) * (1.0f / TOK_D);
        const float rstd = 1.0f / sqrtf(var + TOK_LN_EPS);
        const float o = c * rstd * P[a.o_n2w + f] + P[a.o_n2b + f];
        bn[row * TF_LDX + f] = o;
        A[a.a_xn2 + i] = o;
        if (f == 0) { A[a.a_st2 + 2 * (r0 + row)] = mean; A[a.a_st2 + 2 * (r0 + row) + 1] = rstd; }
      }
    }
    __syncthreads();
    linear(bn, TF_LDX, TOK_D, w1, TF_LDX, TF_FF, P + a.o_b1, bz, TF_LDZ, A + a.a_z);
    __syncthreads();
    // ---- h = drop(gelu(z)), in place
    {
      const Drop dg = make_drop_dev(a.p, a.seed, 4 * l + SITE_FF_ACT);
      for (int e = tid; e < nrows * TF_FF; e += TF_THREADS) {
        const int row = e >> 7, c = e & 127;
        const float xz = bz[row * TF_LDZ + c];
        const long long i = (r0 + row) * TF_FF + c;
        const float hv = drop_apply(dg, (unsigned int)i, 0.5f * xz * (1.0f + erff(xz * 0.70710678118654752f)));
        bz[row * TF_LDZ + c] = hv;
        A[a.a_h + i] = hv;
      }
    }
    __syncthreads();
    linear(bz, TF_LDZ, TF_FF, w2, TF_LDZ, TOK_D, P + a.o_b2, bn, TF_LDX, nullptr);      // ff branch -> bn
    __syncthreads();
    pending = make_drop_dev(a.p, a.seed, 4 * l + SITE_FF);
  }
  // ---- y = x1_{L-1} + drop(ff branch)
  for (int row = hw; row < nrows; row += TF_HW) {
    const long long i = (r0 + row) * TOK_D + f;
    a.y[i] = bx[row * TF_LDX + f] + drop_apply(pending, (unsigned int)i, bn[row * TF_LDX + f]);
  }
}

static inline bool token_fused_enabled() {
  const char* e = getenv("IGI_TOKEN_FUSED");   // read per call (one call per forward pass): the parity test switches it
  return !e || atoi(e) != 0;
}

static int token_forward(const igi_token_cfg* c, const float* x, const float* params, float* y, void* workspace,
                         size_t workspace_bytes, unsigned long long seed, hipStream_t s) {
  TokenPlan p;
  int rc = make_token_plan(c, &p);
  if (rc) return rc;
  if (!x || !params || !y || !workspace) return IGI_E_BADARG;
  if (workspace_bytes < p.total_bytes) return IGI_E_WORKSPACE;
  float* W = tok_ws(workspace);
  // (the alignment terms are linear_forward's conditions for the LDS-DMA kernel, whose k order the fused kernel reproduces)
  if (token_fused_enabled() && p.ff == TF_FF && p.d == TOK_D && p.R >= 4 && aligned16(params) && !bf16_mode() &&
      ((p.per_layer | p.o_inw | p.o_ow | p.o_w1 | p.o_w2 | p.a_layer | p.a_xn1 | p.a_ctx | p.a_xn2 | p.a_h) & 3) == 0) {
    TokFwdArgs a;
    a.x = x; a.params = params; a.y = y; a.W = W;
    a.R = p.R; a.per_layer = p.per_layer; a.a_layer = p.a_layer;
    a.o_inw = p.o_inw; a.o_inb = p.o_inb; a.o_ow = p.o_ow; a.o_ob = p.o_ob; a.o_w1 = p.o_w1; a.o_b1 = p.o_b1;
    a.o_w2 = p.o_w2; a.o_b2 = p.o_b2; a.o_n1w = p.o_n1w; a.o_n1b = p.o_n1b; a.o_n2w = p.o_n2w; a.o_n2b = p.o_n2b;
    a.a_x = p.a_x; a.a_st1 = p.a_st1; a.a_xn1 = p.a_xn1; a.a_qkv = p.a_qkv; a.a_ctx = p.a_ctx; a.a_x1 = p.a_x1;
    a.a_st2 = p.a_st2; a.a_xn2 = p.a_xn2; a.a_z = p.a_z; a.a_h = p.a_h;
    a.S = p.S; a.H = p.H; a.L = p.L; a.p = p.p; a.seed = seed;
    // samples per workgroup: up to 128 token rows, but no fewer than one workgroup per CU while the batch allows it (2048
    // samples x 3 tokens as 64 workgroups of 96 rows ran 78 us; a workgroup's time is its serial chain of phases)
    int samples = (int)(TF_ROWS / p.S);
    if ((long long)samples * 256 > p.B) samples = (int)(p.B / 256 > 1 ? p.B / 256 : 1);
    a.rows_per_wg = samples * p.S;
    const int grid = (int)((p.B + samples - 1) / samples);
    rc = attn_dispatch(p.S, [&](auto sc) {
      constexpr int SS = decltype(sc)::value;
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_token_fwd<SS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(float) * TF_LDS_FLOATS));
        attr = true;
      }
      hipLaunchKernelGGL((k_token_fwd<SS>), dim3(grid), dim3(TF_THREADS), sizeof(float) * TF_LDS_FLOATS, s, a);
    });
    if (rc) return rc;
    return (int)hipGetLastError();
  }
  const long long R = p.R;
  const int d = p.d, ff = p.ff;
  const int rb = tok_blocks(R, 8);  // 8 half-wave rows per 256-thread block
  // layer 0 input + LN1
  const float* xin = x;
  const float* branch = nullptr;  // pending residual branch (ff output of the previous layer)
  Drop pending = make_drop(0.f, seed, 0);
  for (int l = 0; l < p.L; ++l) {
    const float* P = params + (long long)l * p.per_layer;
    float* A = W + (long long)l * p.a_layer;
    // x_l = x_{l-1} + drop(ff branch of l-1)  (l == 0: copy of the input), xn1 = LN1(x_l)
    hipLaunchKernelGGL(k_resid_ln_fwd, dim3(rb), dim3(256), 0, s, xin, branch, pending, A + p.a_x, P + p.o_n1w,
                       P + p.o_n1b, A + p.a_xn1, A + p.a_st1, R);
    if ((rc = linear_forward(A + p.a_xn1, d, P + p.o_inw, P + p.o_inb, A + p.a_qkv, 3 * d, R, d, 3 * d, LIN_NONE, s)))
      return rc;
    const long long pairs = (long long)p.B * p.H;
    const Drop da = make_drop(p.p, seed, 4 * l + SITE_ATTN);
    rc = attn_dispatch(p.S, [&](auto sc) {
      constexpr int SS = decltype(sc)::value;
      hipLaunchKernelGGL((k_attn_fwd<SS>), dim3(tok_blocks(pairs, 16)), dim3(256), 0, s, A + p.a_qkv, A + p.a_ctx, da,
                         pairs, p.H);
    });
    if (rc) return rc;
    float* t0 = W + p.s_g0;  // sa branch output (not needed by backward)
    if ((rc = linear_forward(A + p.a_ctx, d, P + p.o_ow, P + p.o_ob, t0, d, R, d, d, LIN_NONE, s))) return rc;
    hipLaunchKernelGGL(k_resid_ln_fwd, dim3(rb), dim3(256), 0, s, A + p.a_x, t0, make_drop(p.p, seed, 4 * l + SITE_SA),
                       A + p.a_x1, P + p.o_n2w, P + p.o_n2b, A + p.a_xn2, A + p.a_st2, R);
    if ((rc = linear_forward(A + p.a_xn2, d, P + p.o_w1, P + p.o_b1, A + p.a_z, ff, R, d, ff, LIN_NONE, s))) return rc;
    hipLaunchKernelGGL(k_gelu_fwd, dim3(tok_blocks(R * ff, 256)), dim3(256), 0, s, A + p.a_z, A + p.a_h,
                       make_drop(p.p, seed, 4 * l + SITE_FF_ACT), R * ff);
    float* t1 = W + p.s_g1;
    if ((rc = linear_forward(A + p.a_h, ff, P + p.o_w2, P + p.o_b2, t1, d, R, ff, d, LIN_NONE, s))) return rc;
    xin = A + p.a_x1;
    branch = t1;
    pending = make_drop(p.p, seed, 4 * l + SITE_FF);
  }
  // y = x1_{L-1} + drop(ff branch)
  hipLaunchKernelGGL(k_resid_ln_fwd, dim3(rb), dim3(256), 0, s, xin, branch, pending, y, (const float*)nullptr,
                     (const float*)nullptr, (float*)nullptr, (float*)nullptr, R);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole backward as ONE launch + one sum (round 5).  As separate launches a backward pass was eight Linear levels
// (weight + data gradient each), four LayerNorm, two attention, two GELU kernels, a masked copy and the deferred sums:
// ~20 launches of 5 - 15 us for ~0.45 GFLOP (124 us at 2048 x 3 tokens, ~200 us at 8192 x 2).  Here a workgroup keeps the
// gradient of the residual stream of its <= 64 token rows in LDS and walks the layers top down: per layer the four
// weight matrices are staged TRANSPOSED (the data gradients then are the forward's k-contiguous MFMA loops), saved
// activations are staged once each, every weight gradient is a 32 x 32 MFMA tile over the workgroup's rows (operands
// read along the row axis of the LDS arrays), bias / LayerNorm parameter gradients are column sums of the passes that
// produce their operands.  A workgroup writes ONE record of all parameter gradients; k_slab_reduce sums the records
// in fixed order (bitwise reproducible, no atomics).  Same formulas as the kernels above (k_ln_bwd, k_attn_bwd,
// k_gelu_bwd, the dropout hash on the same element numbers); sums associate differently (per-workgroup records instead
// of split-row slabs), so results agree with the launch-per-operation path to rounding, not bitwise.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TB_LDW = 3 * TOK_D + 4;        // 100: row pitch of the transposed in_proj image
constexpr int TB_RED = 3 * TB_THREADS;
constexpr int TB_LDS_FLOATS = 2 * TB_ROWS * TF_LDX + 2 * TB_ROWS * TF_LDZ + TF_FF * TF_LDX + TOK_D * TF_LDZ + TOK_D * TF_LDX +
                              TOK_D * TB_LDW + TB_RED;

struct TokBwdArgs {
  const float* dy; const float* params; float* dx; float* W; float* part;
  long long R, per_layer, a_layer, P;
  long long o_inw, o_inb, o_ow, o_ob, o_w1, o_b1, o_w2, o_b2, o_n1w, o_n1b, o_n2w, o_n2b;
  long long a_x, a_st1, a_xn1, a_qkv, a_ctx, a_x1, a_st2, a_xn2, a_z, a_h;
  int S, H, L, rows_per_wg;
  float p; unsigned long long seed;
};

template <int S>
__global__ __launch_bounds__(TB_THREADS) void k_token_bwd(const TokBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bg = smem;                               // gradient w.r.t. the residual stream [rows][36]
  float* bd = bg + TB_ROWS * TF_LDX;              // 32-wide operand of the current product [rows][36]
  float* bw = bd + TB_ROWS * TF_LDX;              // wide gradient (dh / dz, dctx) [rows][132]
  float* ba = bw + TB_ROWS * TF_LDZ;              // staged activations, dqkv [rows][132]
  float* w2t = ba + TB_ROWS * TF_LDZ;             // linear2^T [128][36]
  float* w1t = w2t + TF_FF * TF_LDX;              // linear1^T [32][132]
  float* wot = w1t + TOK_D * TF_LDZ;              // out_proj^T [32][36]
  float* wit = wot + TOK_D * TF_LDX;              // in_proj^T [32][100]
  float* red = wit + TOK_D * TB_LDW;              // column-sum partials
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const long long r0 = (long long)blockIdx.x * a.rows_per_wg;
  const int nrows = (int)min((long long)a.rows_per_wg, a.R - r0);
  const int mtiles = (nrows + 31) >> 5;
  const int f = tid & 31, hw = tid >> 5;
  float* G0 = a.part + (long long)blockIdx.x * a.P;

  // C[rows][N] = A[rows][K] . Wimg[N][K]^T (the forward's loop, no bias)
  auto dgrad = [&](const float* A, int lda, int K, const float* Wimg, int ldw, int N, float* C, int ldc) {
    const int ntiles = N >> 5;
    for (int t = wave; t < mtiles * ntiles; t += TB_NW) {
      const int mt = t / ntiles, nt = t - mt * ntiles;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* ap = A + (32 * mt + l31) * lda + 4 * h;
      const float* bp = Wimg + (32 * nt + l31) * ldw + 4 * h;
      for (int c = 0; c < K; c += 8) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + c);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) C[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h) * ldc + 32 * nt + l31] = acc[r];
    }
  };
  // dW[M][N] = sum_r Y[r][m] X[r][n] over the workgroup's rows (rows past nrows hold zeros) -> this workgroup's record
  auto wgrad = [&](const float* Y, int ldy, int M, const float* X, int ldx, int N, float* dst) {
    const int ntiles = N >> 5;
    const int total = (M >> 5) * ntiles;
    for (int t = wave; t < total; t += TB_NW) {
      const int mt = t / ntiles, nt = t - mt * ntiles;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* yp = Y + h * ldy + 32 * mt + l31;
      const float* xp = X + h * ldx + 32 * nt + l31;
      for (int rb = 0; rb < 32 * mtiles; rb += 16) {   // eight k-steps' operands in flight in front of their MFMAs
        float ya[8], xa[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { ya[q] = yp[(rb + 2 * q) * ldy]; xa[q] = xp[(rb + 2 * q) * ldx]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[q], xa[q], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h) * N + 32 * nt + l31] = acc[r];
    }
  };
  // rows of a saved activation [R][width] -> LDS (16-byte pieces)
  auto stage = [&](const float* g, int width, float* dst, int ld) {
    const int w4 = width >> 2;
    for (int e = tid; e < nrows * w4; e += TB_THREADS) {
      const int row = e / w4, c = e - row * w4;
      *reinterpret_cast<float4*>(dst + row * ld + 4 * c) = *reinterpret_cast<const float4*>(g + (r0 + row) * width + 4 * c);
    }
  };
  // sum of the per-half-wave partials of a 32-wide column sum -> dst[0..31]  (red + 256 * slot)
  auto put32 = [&](int slot, float v) { red[TB_THREADS * slot + hw * 32 + f] = v; };
  auto sum32 = [&](int slot, float* dst) {
    if (tid < 32) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < TB_HW; ++q) t += red[TB_THREADS * slot + q * 32 + tid];
      dst[tid] = t;
    }
  };

  for (int e = tid; e < TB_LDS_FLOATS; e += TB_THREADS) smem[e] = 0.f;   // rows past nrows stay zero for the whole kernel
  __syncthreads();
  for (int row = hw; row < nrows; row += TB_HW) bg[row * TF_LDX + f] = a.dy[(r0 + row) * TOK_D + f];

  for (int l = a.L - 1; l >= 0; --l) {
    const float* P = a.params + (long long)l * a.per_layer;
    const float* A = a.W + (long long)l * a.a_layer;
    float* G = G0 + (long long)l * a.per_layer;
    // ---- the layer's weights, transposed: image[n = input feature][k = output feature]
    for (int u = tid; u < TOK_D * TF_FF; u += TB_THREADS) {
      w2t[(u & 127) * TF_LDX + (u >> 7)] = P[a.o_w2 + u];            // linear2.weight [32][128]
      w1t[(u & 31) * TF_LDZ + (u >> 5)] = P[a.o_w1 + u];             // linear1.weight [128][32]
    }
    for (int u = tid; u < TOK_D * TOK_D; u += TB_THREADS) wot[(u & 31) * TF_LDX + (u >> 5)] = P[a.o_ow + u];
    for (int u = tid; u < 3 * TOK_D * TOK_D; u += TB_THREADS) wit[(u & 31) * TB_LDW + (u >> 5)] = P[a.o_inw + u];
    // ---- 1. dff = mask_ff(dres); db2
    {
      const Drop dr = make_drop_dev(a.p, a.seed, 4 * l + SITE_FF);
      float sb = 0.f;
      for (int row = hw; row < nrows; row += TB_HW) {
        const float v = drop_apply(dr, (unsigned int)((r0 + row) * TOK_D + f), bg[row * TF_LDX + f]);
        bd[row * TF_LDX + f] = v;
        sb += v;
      }
      put32(0, sb);
    }
    stage(A + a.a_h, TF_FF, ba, TF_LDZ);
    __syncthreads();
    sum32(0, G + a.o_b2);
    // ---- 2. dW2 = dff^T h; dh = dff W2
    wgrad(bd, TF_LDX, TOK_D, ba, TF_LDZ, TF_FF, G + a.o_w2);
    dgrad(bd, TF_LDX, TOK_D, w2t, TF_LDX, TF_FF, bw, TF_LDZ);
    __syncthreads();
    stage(A + a.a_z, TF_FF, ba, TF_LDZ);
    __syncthreads();
    // ---- 3. dz = mask_act(dh) gelu'(z); db1
    {
      const Drop dr = make_drop_dev(a.p, a.seed, 4 * l + SITE_FF_ACT);
      const int c = tid & 127;
      float sb = 0.f;
      for (int row = tid >> 7; row < nrows; row += TB_THREADS / 128) {
        const float x = ba[row * TF_LDZ + c];
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
        const float v = drop_apply(dr, (unsigned int)((r0 + row) * TF_FF + c), bw[row * TF_LDZ + c]) * (cdf + x * pdf);
        bw[row * TF_LDZ + c] = v;
        sb += v;
      }
      red[tid] = sb;
    }
    __syncthreads();
    if (tid < TF_FF) {
      float t = red[tid];
#pragma unroll
      for (int q = 1; q < TB_THREADS / 128; ++q) t += red[128 * q + tid];
      G[a.o_b1 + tid] = t;
    }
    stage(A + a.a_xn2, TOK_D, ba, TF_LDZ);
    __syncthreads();
    // ---- 4. dW1 = dz^T xn2; dxn2 = dz W1
    wgrad(bw, TF_LDZ, TF_FF, ba, TF_LDZ, TOK_D, G + a.o_w1);
    dgrad(bw, TF_LDZ, TF_FF, w1t, TF_LDZ, TOK_D, bd, TF_LDX);
    __syncthreads();
    // ---- 5. LayerNorm 2 backward + residual: dx1; dsa = mask_sa(dx1); d(norm2), d(out_proj.bias)
    {
      const Drop dr = make_drop_dev(a.p, a.seed, 4 * l + SITE_SA);
      const float gm = P[a.o_n2w + f];
      float sg = 0.f, sb = 0.f, so = 0.f;
      for (int row = hw; row < nrows; row += TB_HW) {
        const long long i = (r0 + row) * TOK_D + f;
        const float mean = A[a.a_st2 + 2 * (r0 + row)], rstd = A[a.a_st2 + 2 * (r0 + row) + 1];
        const float xhat = (A[a.a_x1 + i] - mean) * rstd;
        const float d = bd[row * TF_LDX + f];
        sg += d * xhat;
        sb += d;
        const float g = d * gm;
        const float m1 = half32_sum(g) * (1.0f / TOK_D);
        const float m2 = half32_sum(g * xhat) * (1.0f / TOK_D);
        const float v = bg[row * TF_LDX + f] + rstd * (g - m1 - xhat * m2);
        bg[row * TF_LDX + f] = v;
        const float ds = drop_apply(dr, (unsigned int)i, v);
        bd[row * TF_LDX + f] = ds;
        so += ds;
      }
      put32(0, sg); put32(1, sb); put32(2, so);
    }
    stage(A + a.a_ctx, TOK_D, ba, TF_LDZ);
    __syncthreads();
    sum32(0, G + a.o_n2w); sum32(1, G + a.o_n2b); sum32(2, G + a.o_ob);
    // ---- 6. dWo = dsa^T ctx; dctx = dsa Wo
    wgrad(bd, TF_LDX, TOK_D, ba, TF_LDZ, TOK_D, G + a.o_ow);
    dgrad(bd, TF_LDX, TOK_D, wot, TF_LDX, TOK_D, bw, TF_LDZ);
    __syncthreads();
    // ---- 7. attention backward (k_attn_bwd's arithmetic): dqkv -> ba; d(in_proj.bias)
    {
      const Drop dr = make_drop_dev(a.p, a.seed, 4 * l + SITE_ATTN);
      const int dl = tid & 15, slot = tid >> 4;          // slot parity = head (H == 2)
      const int nsamp = nrows / S;
      const float scale = 0.25f;
      float sq = 0.f, sk = 0.f, sv = 0.f;
      for (int g = slot; g < nsamp * a.H; g += TB_THREADS / 16) {
        const int bl = g / a.H, hh = g - bl * a.H;
        const long long gg = (r0 / S + bl) * a.H + hh;
        float q[S], k[S], v[S], dc[S], dq[S], dk[S], dv[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
          const float* row = A + a.a_qkv + (r0 + bl * S + s2) * (3 * TOK_D) + hh * TOK_DH + dl;
          q[s2] = row[0]; k[s2] = row[TOK_D]; v[s2] = row[2 * TOK_D];
          dc[s2] = bw[(bl * S + s2) * TF_LDZ + hh * TOK_DH + dl];
          dq[s2] = 0.f; dk[s2] = 0.f; dv[s2] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
          float pr[S], mx = -INFINITY;
#pragma unroll
          for (int j = 0; j < S; ++j) { pr[j] = row16_sum(q[i] * k[j]) * scale; mx = fmaxf(mx, pr[j]); }
          float den = 0.f;
#pragma unroll
          for (int j = 0; j < S; ++j) { pr[j] = __expf(pr[j] - mx); den += pr[j]; }
          const float inv = 1.0f / den;
          float dp[S], dot = 0.f;
#pragma unroll
          for (int j = 0; j < S; ++j) {
            pr[j] *= inv;
            const float m = drop_apply(dr, (unsigned int)((gg * S + i) * S + j), 1.0f);
            dv[j] += pr[j] * m * dc[i];
            dp[j] = row16_sum(dc[i] * v[j]) * m;
            dot += dp[j] * pr[j];
          }
#pragma unroll
          for (int j = 0; j < S; ++j) {
            const float ds = pr[j] * (dp[j] - dot) * scale;
            dq[i] += ds * k[j];
            dk[j] += ds * q[i];
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) {
          float* row = ba + (bl * S + s2) * TF_LDZ + hh * TOK_DH + dl;
          row[0] = dq[s2]; row[TOK_D] = dk[s2]; row[2 * TOK_D] = dv[s2];
          sq += dq[s2]; sk += dk[s2]; sv += dv[s2];
        }
      }
      red[slot * 16 + dl] = sq; red[TB_THREADS + slot * 16 + dl] = sk; red[2 * TB_THREADS + slot * 16 + dl] = sv;
    }
    __syncthreads();
    if (tid < 3 * TOK_D) {
      const int which = tid >> 5, c = tid & 31, hh = c >> 4, dl = c & 15;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < TB_THREADS / 32; ++q) t += red[TB_THREADS * which + (2 * q + hh) * 16 + dl];   // the slots of head hh, in order
      G[a.o_inb + tid] = t;
    }
    stage(A + a.a_xn1, TOK_D, bw, TF_LDZ);
    __syncthreads();
    // ---- 8. dWin = dqkv^T xn1; dxn1 = dqkv Win
    wgrad(ba, TF_LDZ, 3 * TOK_D, bw, TF_LDZ, TOK_D, G + a.o_inw);
    dgrad(ba, TF_LDZ, 3 * TOK_D, wit, TB_LDW, TOK_D, bd, TF_LDX);
    __syncthreads();
    // ---- 9. LayerNorm 1 backward + residual: the gradient of this layer's input; d(norm1)
    {
      const float gm = P[a.o_n1w + f];
      float sg = 0.f, sb = 0.f;
      for (int row = hw; row < nrows; row += TB_HW) {
        const long long i = (r0 + row) * TOK_D + f;
        const float mean = A[a.a_st1 + 2 * (r0 + row)], rstd = A[a.a_st1 + 2 * (r0 + row) + 1];
        const float xhat = (A[a.a_x + i] - mean) * rstd;
        const float d = bd[row * TF_LDX + f];
        sg += d * xhat;
        sb += d;
        const float g = d * gm;
        const float m1 = half32_sum(g) * (1.0f / TOK_D);
        const float m2 = half32_sum(g * xhat) * (1.0f / TOK_D);
        bg[row * TF_LDX + f] += rstd * (g - m1 - xhat * m2);
      }
      put32(0, sg); put32(1, sb);
    }
    __syncthreads();
    sum32(0, G + a.o_n1w); sum32(1, G + a.o_n1b);
    // (the next layer's first pass reads bg rows this thread wrote and writes red slot 0 after its weights loop; the
    //  sums above read red: one more barrier keeps them apart)
    __syncthreads();
  }
  for (int row = hw; row < nrows; row += TB_HW) a.dx[(r0 + row) * TOK_D + f] = bg[row * TF_LDX + f];
}

constexpr int TB_FUSED_MAX_S = 4;   // tokens per sample the one-launch backward takes (tests/test_host_api.py guards its registers)

static inline bool token_bwd_fused_enabled() {
  const char* e = getenv("IGI_TOKEN_FUSED_BWD");   // read per call (one call per backward pass): A/B and the parity test
  return !e || atoi(e) != 0;
}

static int token_backward(const igi_token_cfg* c, const float* dy, const float* params, float* dx, float* grads,
                          void* workspace, size_t workspace_bytes, unsigned long long seed, hipStream_t s) {
  TokenPlan p;
  int rc = make_token_plan(c, &p);
  if (rc) return rc;
  if (!dy || !params || !dx || !grads || !workspace) return IGI_E_BADARG;
  if (workspace_bytes < p.total_bytes) return IGI_E_WORKSPACE;
  float* W = tok_ws(workspace);
  // (S <= TB_FUSED_MAX_S: with five or more tokens per sample the attention pass of k_token_bwd holds more than 256 registers
  //  -- 88 to 940 bytes of scratch per lane in the S = 5 .. 8 instantiations -- so those run the launch-per-operation
  //  backward below, which does not spill; the instantiations are not built)
  if (token_bwd_fused_enabled() && p.S <= TB_FUSED_MAX_S && p.bwd_grid > 0 && p.ff == TF_FF && p.d == TOK_D && p.H == 2 && !bf16_mode() &&
      ((p.a_layer | p.a_xn1 | p.a_ctx | p.a_xn2 | p.a_h | p.a_z) & 3) == 0) {
    TokBwdArgs a;
    a.dy = dy; a.params = params; a.dx = dx; a.W = W; a.part = W + p.s_part;
    a.R = p.R; a.per_layer = p.per_layer; a.a_layer = p.a_layer; a.P = p.per_layer * p.L;
    a.o_inw = p.o_inw; a.o_inb = p.o_inb; a.o_ow = p.o_ow; a.o_ob = p.o_ob; a.o_w1 = p.o_w1; a.o_b1 = p.o_b1;
    a.o_w2 = p.o_w2; a.o_b2 = p.o_b2; a.o_n1w = p.o_n1w; a.o_n1b = p.o_n1b; a.o_n2w = p.o_n2w; a.o_n2b = p.o_n2b;
    a.a_x = p.a_x; a.a_st1 = p.a_st1; a.a_xn1 = p.a_xn1; a.a_qkv = p.a_qkv; a.a_ctx = p.a_ctx; a.a_x1 = p.a_x1;
    a.a_st2 = p.a_st2; a.a_xn2 = p.a_xn2; a.a_z = p.a_z; a.a_h = p.a_h;
    a.S = p.S; a.H = p.H; a.L = p.L; a.p = p.p; a.seed = seed;
    a.rows_per_wg = p.bwd_samples * p.S;
    rc = attn_dispatch(p.S, [&](auto sc) {
      constexpr int SS = decltype(sc)::value;
      if constexpr (SS <= TB_FUSED_MAX_S) {
        static bool attr = false;
        if (!attr) {
          (void)hipFuncSetAttribute((const void*)k_token_bwd<SS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(float) * TB_LDS_FLOATS));
          attr = true;
        }
        hipLaunchKernelGGL((k_token_bwd<SS>), dim3(p.bwd_grid), dim3(TB_THREADS), sizeof(float) * TB_LDS_FLOATS, s, a);
      }
    });
    if (rc) return rc;
    SegTable t;
    t.n = 1;
    t.wide = 1;
    Segment& sg = t.s[0];
    sg.dst = 0; sg.src = a.part; sg.stride = a.P; sg.count = (int)a.P; sg.cols = (int)a.P; sg.src_ld = 0; sg.nparts = p.bwd_grid;
    hipLaunchKernelGGL(k_slab_reduce, dim3(SLAB_GX, 1), dim3(RED_THREADS), 0, s, t, grads);
    return (int)hipGetLastError();
  }
  const long long R = p.R;
  const int d = p.d, ff = p.ff;
  float* g0 = W + p.s_g0;
  float* g1 = W + p.s_g1;
  float* rA = W + p.s_rA;   // residual-stream gradient entering a layer from above
  float* rB = W + p.s_rB;   // residual-stream gradient between the two blocks of a layer
  float* w0 = W + p.s_wide0;
  float* w1 = W + p.s_wide1;
  // the sums of every split-row partial of this pass (eight Linears x (weight, bias), four layer norms) are queued and
  // run as ONE launch at the end: each producer therefore keeps its own partial buffer
  SplitSumTable sums;
  int n_lin = 0, n_ln = 0;
  auto lin_ws_next = [&]() { return (void*)(reinterpret_cast<char*>(W + p.s_lin) + (size_t)(n_lin++) * p.lin_bytes); };
  auto lnpart_next = [&]() { return W + p.s_lnpart + (long long)(n_ln++) * p.ln_blocks * 2 * d; };
  const float* dres = dy;   // gradient w.r.t. the current residual stream
  const float* dbr;         // gradient w.r.t. the branch output feeding it (after the dropout mask)
  // top: branch gradient = drop_mask(dy) with the last layer's ff site
  if (p.p > 0.f) {
    hipLaunchKernelGGL(k_drop_copy, dim3(tok_blocks(R * d, 256)), dim3(256), 0, s, dy, g1,
                       make_drop(p.p, seed, 4 * (p.L - 1) + SITE_FF), R * d);
    dbr = g1;
  } else {
    dbr = dy;
  }
  for (int l = p.L - 1; l >= 0; --l) {
    const float* P = params + (long long)l * p.per_layer;
    float* G = grads + (long long)l * p.per_layer;
    float* A = W + (long long)l * p.a_layer;
    // ---- ff block: f = linear2(h)
    if ((rc = linear_backward(A + p.a_h, ff, P + p.o_w2, nullptr, 0, dbr, d, w0, ff, G + p.o_w2, G + p.o_b2, R, ff, d,
                              LIN_NONE, lin_ws_next(), p.lin_bytes, s, &sums)))
      return rc;
    hipLaunchKernelGGL(k_gelu_bwd, dim3(tok_blocks(R * ff, 256)), dim3(256), 0, s, w0, A + p.a_z, w1,
                       make_drop(p.p, seed, 4 * l + SITE_FF_ACT), R * ff);
    if ((rc = linear_backward(A + p.a_xn2, d, P + p.o_w1, nullptr, 0, w1, ff, g0, d, G + p.o_w1, G + p.o_b1, R, d, ff,
                              LIN_NONE, lin_ws_next(), p.lin_bytes, s, &sums)))
      return rc;
    // ---- LN2 backward + residual: dx1 and its masked copy for the sa branch (g1)
    float* dx1 = rB;
    float* lnpart = lnpart_next();
    hipLaunchKernelGGL(k_ln_bwd, dim3(p.ln_blocks), dim3(256), 0, s, g0, A + p.a_x1, A + p.a_st2, P + p.o_n2w, dres,
                       dx1, make_drop(p.p, seed, 4 * l + SITE_SA), p.p > 0.f ? g1 : (float*)nullptr, lnpart, R);
    // norm2.weight and norm2.bias are adjacent
    if (!split_sum_defer(sums, G + p.o_n2w, lnpart, 2LL * d, 2LL * d, p.ln_blocks))
      split_sum(G + p.o_n2w, lnpart, 2LL * d, p.ln_blocks, 2LL * d, s);
    const float* da = p.p > 0.f ? g1 : dx1;
    // ---- sa block: a = out_proj(ctx)
    if ((rc = linear_backward(A + p.a_ctx, d, P + p.o_ow, nullptr, 0, da, d, g0, d, G + p.o_ow, G + p.o_ob, R, d, d,
                              LIN_NONE, lin_ws_next(), p.lin_bytes, s, &sums)))
      return rc;
    const long long pairs = (long long)p.B * p.H;
    const Drop datt = make_drop(p.p, seed, 4 * l + SITE_ATTN);
    rc = attn_dispatch(p.S, [&](auto sc) {
      constexpr int SS = decltype(sc)::value;
      hipLaunchKernelGGL((k_attn_bwd<SS>), dim3(tok_blocks(pairs, 16)), dim3(256), 0, s, A + p.a_qkv, g0, w1, datt,
                         pairs, p.H);
    });
    if (rc) return rc;
    if ((rc = linear_backward(A + p.a_xn1, d, P + p.o_inw, nullptr, 0, w1, 3 * d, g0, d, G + p.o_inw, G + p.o_inb, R, d,
                              3 * d, LIN_NONE, lin_ws_next(), p.lin_bytes, s, &sums)))
      return rc;
    // ---- LN1 backward + residual: gradient of this layer's input; masked copy for the ff branch of l-1
    float* dxl = (l == 0) ? dx : rA;
    const bool mask_below = (l > 0) && p.p > 0.f;
    lnpart = lnpart_next();
    hipLaunchKernelGGL(k_ln_bwd, dim3(p.ln_blocks), dim3(256), 0, s, g0, A + p.a_x, A + p.a_st1, P + p.o_n1w, dx1, dxl,
                       make_drop(p.p, seed, 4 * (l - 1) + SITE_FF), mask_below ? g1 : (float*)nullptr, lnpart, R);
    // norm1.weight and norm1.bias are adjacent
    if (!split_sum_defer(sums, G + p.o_n1w, lnpart, 2LL * d, 2LL * d, p.ln_blocks))
      split_sum(G + p.o_n1w, lnpart, 2LL * d, p.ln_blocks, 2LL * d, s);
    dres = dxl;
    dbr = mask_below ? g1 : dxl;
  }
  split_sum_flush(sums, s);
  return (int)hipGetLastError();
}

}  // namespace igi
