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
};

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

static int token_forward(const igi_token_cfg* c, const float* x, const float* params, float* y, void* workspace,
                         size_t workspace_bytes, unsigned long long seed, hipStream_t s) {
  TokenPlan p;
  int rc = make_token_plan(c, &p);
  if (rc) return rc;
  if (!x || !params || !y || !workspace) return IGI_E_BADARG;
  if (workspace_bytes < p.total_bytes) return IGI_E_WORKSPACE;
  float* W = tok_ws(workspace);
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

static int token_backward(const igi_token_cfg* c, const float* dy, const float* params, float* dx, float* grads,
                          void* workspace, size_t workspace_bytes, unsigned long long seed, hipStream_t s) {
  TokenPlan p;
  int rc = make_token_plan(c, &p);
  if (rc) return rc;
  if (!dy || !params || !dx || !grads || !workspace) return IGI_E_BADARG;
  if (workspace_bytes < p.total_bytes) return IGI_E_WORKSPACE;
  float* W = tok_ws(workspace);
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
