// Depth / segmentation image encoder of the student: DepthOnlyFCBackbone54x96
// (algo/models/transformer/tact.py:81-113), forward and backward:
//   (B,1,54,96) -Conv(1->32,k5)-> (32,50,92) -MaxPool2-> (32,25,46) -ELU- Conv(32->64,k3)-> (64,23,44) -ELU-
//   Flatten (c,y,x) -Linear(64768->128)-ELU-Linear(128->latent)
//
// MI355X design
//   * conv1 + max-pool + ELU is ONE kernel (k_depth_conv1_fwd): a workgroup stages its image in LDS
//     (20 KB) and runs the 25-tap convolution on the MFMA pipe: a 32x32x2 tile = 8 pooled pixels x their
//     4 pool positions (rows) x 32 channels (columns), 16 MFMA steps cover the 25 taps (+7 zero taps);
//     the A operand is gathered straight from the LDS image, the B operand (the 800 weights) lives in
//     registers.  The tile-row order puts the four pool positions of a pooled pixel into four consecutive
//     accumulator registers of ONE lane, so pooling (and its arg-max, 1 byte) needs no cross-lane traffic.
//     The 50x92x32 pre-pool map (589 KB / image) is never written.
//   * conv2 is an implicit GEMM on gemm_dma.h (channels-last activations, im2col in the loader) with
//     bias+ELU / ELU-gradient epilogues, exactly like the tactile convolutions (tactile.h).
//   * the 64768 -> 128 Linear is a split-K GEMM over a weight copy whose columns are permuted from torch's
//     flatten order (c, y, x) to the channels-last order (y, x, c) of the activations; its gradient is
//     permuted back.
//   * conv1's weight gradient (the input needs none) routes each pooled gradient to its arg-max position
//     and accumulates 25 taps per (pooled pixel, channel) on the VALU against the LDS-resident image.
// Parameter vector in state_dict order: image_compression.{0,3,6,8}.{weight,bias}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/igi_ppo.h"
#include "gemm_dma.h"
#include "linear.h"
#include "tactile.h"

namespace igi {

constexpr int DP_H = 54, DP_W = 96, DP_PH = 25, DP_PW = 46, DP_H2 = 23, DP_W2 = 44;
constexpr int DP_C1 = 32, DP_C2 = 64, DP_FC = 128;
constexpr int DP_P1 = DP_PH * DP_PW;          // 1150 pooled pixels
constexpr int DP_P2 = DP_H2 * DP_W2;          // 1012 conv2 outputs
constexpr int DP_FLAT = DP_P2 * DP_C2;        // 64768
constexpr int DP_TILES = (DP_P1 + 7) / 8;     // conv1 MFMA tiles per image

struct DepthPlan {
  int B, L;
  long long o_c1w, o_c1b, o_c2w, o_c2b, o_f1w, o_f1b, o_f2w, o_f2b, P;
  size_t w_zero, w_a1, w_idx1, w_w2r, w_w2d, w_a2, w_wfp, w_fcslab, w_h3, w_dh3, w_dz3, w_dz2, w_g1, w_c2slab,
      w_c2red, w_c1part, w_c1red, w_dwfp, w_lin, w_total;
  size_t lin_bytes;
  int sk_fc, kchunk_fc, sk_c2, c1_blocks;
};

static int make_depth_plan(const igi_depth_cfg* c, DepthPlan* p) {
  if (!c || c->batch < 1 || c->latent_dim < 4 || (c->latent_dim & 3)) return IGI_E_BADARG;
  if (c->batch % 32 || c->batch > 32768) return IGI_E_UNSUPPORTED;  // k-ranges of the weight gradients are whole k-tiles
  p->B = c->batch; p->L = c->latent_dim;
  long long o = 0;
  p->o_c1w = o; o += DP_C1 * 25;
  p->o_c1b = o; o += DP_C1;
  p->o_c2w = o; o += DP_C2 * DP_C1 * 9;
  p->o_c2b = o; o += DP_C2;
  p->o_f1w = o; o += (long long)DP_FC * DP_FLAT;
  p->o_f1b = o; o += DP_FC;
  p->o_f2w = o; o += (long long)p->L * DP_FC;
  p->o_f2b = o; o += p->L;
  p->P = o;
  // split-K of the big Linear's forward: whole 32-row k-tiles per split, no empty split
  int sk = dma_choose_splitk(p->B, DP_FC, DP_FLAT, 1);
  int kchunk = ((DP_FLAT + sk - 1) / sk + DMA_BK - 1) / DMA_BK * DMA_BK;
  sk = (DP_FLAT + kchunk - 1) / kchunk;
  p->sk_fc = sk; p->kchunk_fc = kchunk;
  p->sk_c2 = dma_choose_splitk(288, DP_C2, p->B * DP_P2, 1);
  p->c1_blocks = p->B < 512 ? p->B : 512;
  size_t w = 0;
  auto take = [&](size_t bytes) { size_t at = w; w += (bytes + 255) & ~(size_t)255; return at; };
  const size_t B = (size_t)p->B;
  p->w_zero = take(256);
  p->w_a1 = take(sizeof(float) * B * DP_P1 * DP_C1);
  p->w_idx1 = take(B * DP_P1 * DP_C1);
  p->w_w2r = take(sizeof(float) * DP_C2 * 288);
  p->w_w2d = take(sizeof(float) * DP_C1 * 576);
  p->w_a2 = take(sizeof(float) * B * DP_FLAT);
  p->w_wfp = take(sizeof(float) * (size_t)DP_FC * DP_FLAT);
  p->w_fcslab = take(sizeof(float) * (size_t)p->sk_fc * B * DP_FC);
  p->w_h3 = take(sizeof(float) * B * DP_FC);
  p->w_dh3 = take(sizeof(float) * B * DP_FC);
  p->w_dz3 = take(sizeof(float) * B * DP_FC);
  p->w_dz2 = take(sizeof(float) * B * DP_FLAT);
  p->w_g1 = take(sizeof(float) * B * DP_P1 * DP_C1);
  p->w_c2slab = take(sizeof(float) * (size_t)p->sk_c2 * (288 * DP_C2 + DP_C2));
  p->w_c2red = take(sizeof(float) * 288 * DP_C2);
  p->w_c1part = take(sizeof(float) * (size_t)p->c1_blocks * DP_C1 * 26);
  p->w_c1red = take(sizeof(float) * DP_C1 * 26);
  p->w_dwfp = take(sizeof(float) * (size_t)DP_FC * DP_FLAT);
  p->lin_bytes = linear_workspace_bytes(p->B, DP_FC, p->L);
  p->w_lin = take(p->lin_bytes + 64);
  p->w_total = w;
  return 0;
}

// ---- conv1 (5x5, 1 -> 32) + bias + 2x2 max-pool + ELU: one workgroup per image, MFMA 32x32x2
__global__ __launch_bounds__(256) void k_depth_conv1_fwd(const float* __restrict__ x, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, float* __restrict__ a1,
                                                         unsigned char* __restrict__ idx1, int B) {
  __shared__ float img[DP_H * DP_W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  // B operand: B[k = 2s + h][n = l31] = w1[n][tap k] (taps >= 25 are zero)
  float bw[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int tap = 2 * s + h;
    bw[s] = tap < 25 ? w1[l31 * 25 + tap] : 0.f;
  }
  const float bias = b1[l31];
  // tap -> offset inside the image, for this lane's 16 k-steps
  int toff[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int tap = (2 * s + h) < 25 ? 2 * s + h : 0;
    toff[s] = (tap / 5) * DP_W + tap % 5;
  }
  const int j = l31 >> 2, pos = l31 & 3;  // this lane's A row: pooled pixel j of the tile, pool position pos
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    const float* src = x + (long long)b * DP_H * DP_W;
    for (int e = threadIdx.x * 4; e < DP_H * DP_W; e += 256 * 4)
      *reinterpret_cast<float4*>(img + e) = *reinterpret_cast<const float4*>(src + e);
    __syncthreads();
    for (int t = wave; t < DP_TILES; t += 4) {
      int pp = 8 * t + j;
      if (pp > DP_P1 - 1) pp = DP_P1 - 1;
      const int py = pp / DP_PW, px = pp - py * DP_PW;
      const int base = (2 * py + (pos >> 1)) * DP_W + 2 * px + (pos & 1);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(img[base + toff[s]], bw[s], acc, 0, 0, 0);
      // acc[4q + i]: pooled pixel 2q + h of the tile, pool position i, channel l31
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float m = acc[4 * q];
        int am = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
          if (acc[4 * q + i] > m) { m = acc[4 * q + i]; am = i; }
        const int opp = 8 * t + 2 * q + h;
        if (opp < DP_P1) {
          const long long o = ((long long)b * DP_P1 + opp) * DP_C1 + l31;
          a1[o] = elu1(m + bias);
          idx1[o] = (unsigned char)am;
        }
      }
    }
  }
}

// conv1 weight / bias gradient: g1 = d(loss)/d(pooled pre-ELU map) (channels-last), routed to the arg-max
// position.  32-lane groups (lane = channel) walk the pooled pixels; per-block partials [32][26].
__global__ __launch_bounds__(256) void k_depth_conv1_wgrad(const float* __restrict__ x, const float* __restrict__ g1,
                                                           const unsigned char* __restrict__ idx1,
                                                           float* __restrict__ part, int B) {
  __shared__ float img[DP_H * DP_W];
  __shared__ float red[8][DP_C1 * 26];
  const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float acc[25], accb = 0.f;
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = 0.f;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    const float* src = x + (long long)b * DP_H * DP_W;
    for (int e = threadIdx.x * 4; e < DP_H * DP_W; e += 256 * 4)
      *reinterpret_cast<float4*>(img + e) = *reinterpret_cast<const float4*>(src + e);
    __syncthreads();
    for (int pp = grp; pp < DP_P1; pp += 8) {
      const long long o = ((long long)b * DP_P1 + pp) * DP_C1 + c;
      const float g = g1[o];
      const int pos = idx1[o];
      const int py = pp / DP_PW, px = pp - py * DP_PW;
      const float* ip = img + (2 * py + (pos >> 1)) * DP_W + 2 * px + (pos & 1);
      accb += g;
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) acc[ky * 5 + kx] += g * ip[ky * DP_W + kx];
    }
  }
#pragma unroll
  for (int t = 0; t < 25; ++t) red[grp][c * 26 + t] = acc[t];
  red[grp][c * 26 + 25] = accb;
  __syncthreads();
  for (int e = threadIdx.x; e < DP_C1 * 26; e += 256) {
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += red[q][e];
    part[(long long)blockIdx.x * (DP_C1 * 26) + e] = s;
  }
}

// partial-summed [32][26] -> conv1 weight [32][25] and bias [32]
__global__ __launch_bounds__(256) void k_depth_conv1_gout(const float* __restrict__ red, float* __restrict__ gw,
                                                          float* __restrict__ gb) {
  for (int e = threadIdx.x; e < DP_C1 * 26; e += 256) {
    const int c = e / 26, t = e - c * 26;
    if (t < 25) gw[c * 25 + t] = red[e];
    else gb[c] = red[e];
  }
}

// fc1 weight: torch column (c * 1012 + p)  <->  channels-last column (p * 64 + c)
__global__ __launch_bounds__(256) void k_depth_perm_fc(const float* __restrict__ src, float* __restrict__ dst,
                                                       int to_channels_last) {
  const long long total = (long long)DP_FC * DP_FLAT;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long n = e / DP_FLAT;
    const int k = (int)(e - n * DP_FLAT);  // index in the DESTINATION layout
    int ks;
    if (to_channels_last) { const int pp = k / DP_C2, c = k - pp * DP_C2; ks = c * DP_P2 + pp; }
    else { const int c = k / DP_P2, pp = k - c * DP_P2; ks = pp * DP_C2 + c; }
    dst[e] = src[n * DP_FLAT + ks];
  }
}

// h3 = elu(sum of the split-K slabs + bias)
__global__ __launch_bounds__(256) void k_depth_fc_finish(const float* __restrict__ slab, int sk, long long stride,
                                                         const float* __restrict__ bias, float* __restrict__ h3,
                                                         long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = slab[i];
    for (int q = 1; q < sk; ++q) s += slab[q * stride + i];
    h3[i] = elu1(s + bias[i % DP_FC]);
  }
}

__global__ __launch_bounds__(256) void k_depth_elu_grad(const float* __restrict__ dy, const float* __restrict__ y,
                                                        float* __restrict__ dz, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dz[i] = dy[i] * elu1_grad_from_out(y[i]);
}

static int depth_forward(const igi_depth_cfg* c, const float* x, const float* params, float* y, void* ws,
                         size_t ws_bytes, hipStream_t s) {
  DepthPlan p;
  int rc = make_depth_plan(c, &p);
  if (rc) return rc;
  if (!x || !params || !y || !ws) return IGI_E_BADARG;
  if (ws_bytes < p.w_total) return IGI_E_WORKSPACE;
  float* zero = twsp<float>(ws, p.w_zero);
  float* a1 = twsp<float>(ws, p.w_a1);
  float* a2 = twsp<float>(ws, p.w_a2);
  float* w2r = twsp<float>(ws, p.w_w2r);
  float* w2d = twsp<float>(ws, p.w_w2d);
  float* wfp = twsp<float>(ws, p.w_wfp);
  float* slab = twsp<float>(ws, p.w_fcslab);
  float* h3 = twsp<float>(ws, p.w_h3);
  IGI_HIP_TRY(hipMemsetAsync(zero, 0, 256, s));
  {
    ConvWJobs jw;
    jw.j[0] = ConvWJob{params + p.o_c2w, w2r, w2d, DP_C2, DP_C1, 3, 3, DP_C1};
    jw.j[1] = jw.j[0]; jw.j[2] = jw.j[0];
    hipLaunchKernelGGL(k_tactile_pack_w, dim3(72, 1), dim3(256), 0, s, jw);
  }
  hipLaunchKernelGGL(k_depth_perm_fc, dim3(2048), dim3(256), 0, s, params + p.o_f1w, wfp, 1);
  hipLaunchKernelGGL(k_depth_conv1_fwd, dim3(p.B < 2048 ? p.B : 2048), dim3(256), 0, s, x, params + p.o_c1w,
                     params + p.o_c1b, a1, twsp<unsigned char>(ws, p.w_idx1), p.B);
  {  // conv2 + bias + ELU: (B,25,46,32) -> (B,23,44,64)
    GemmArgs g;
    g.A = a1; g.gather = 1; g.conv = conv_desc(zero, DP_H2, DP_W2, DP_PH, DP_PW, DP_C1, 1, 0, 3, 3);
    g.B = w2r; g.ldb = 288;
    g.M = p.B * DP_P2; g.N = DP_C2; g.K = 288; g.lda = 288;
    g.C = a2; g.ldc = DP_C2; g.bias = params + p.o_c2b; g.epilogue = EPI_BIAS_ELU;
    IGI_HIP_TRY(gemm(g, true, true, s));
  }
  {  // Linear(64768 -> 128), split over k
    GemmArgs g;
    g.A = a2; g.lda = DP_FLAT;
    g.B = wfp; g.ldb = DP_FLAT;
    g.M = p.B; g.N = DP_FC; g.K = DP_FLAT;
    g.C = slab; g.ldc = DP_FC;
    g.splitk = p.sk_fc; g.kchunk = p.kchunk_fc; g.sCsplit = (long long)p.B * DP_FC;
    IGI_HIP_TRY(gemm(g, true, true, s));
    const long long n = (long long)p.B * DP_FC;
    hipLaunchKernelGGL(k_depth_fc_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slab, p.sk_fc, n,
                       params + p.o_f1b, h3, n);
  }
  return linear_forward(h3, DP_FC, params + p.o_f2w, params + p.o_f2b, y, p.L, p.B, DP_FC, p.L, LIN_NONE, s);
}

static int depth_backward(const igi_depth_cfg* c, const float* x, const float* dy, const float* params, float* grads,
                          void* ws, size_t ws_bytes, hipStream_t s) {
  DepthPlan p;
  int rc = make_depth_plan(c, &p);
  if (rc) return rc;
  if (!x || !dy || !params || !grads || !ws) return IGI_E_BADARG;
  if (ws_bytes < p.w_total) return IGI_E_WORKSPACE;
  float* zero = twsp<float>(ws, p.w_zero);
  float* a1 = twsp<float>(ws, p.w_a1);
  float* a2 = twsp<float>(ws, p.w_a2);
  float* w2d = twsp<float>(ws, p.w_w2d);
  float* wfp = twsp<float>(ws, p.w_wfp);
  float* h3 = twsp<float>(ws, p.w_h3);
  float* dh3 = twsp<float>(ws, p.w_dh3);
  float* dz3 = twsp<float>(ws, p.w_dz3);
  float* dz2 = twsp<float>(ws, p.w_dz2);
  float* g1 = twsp<float>(ws, p.w_g1);
  float* c2slab = twsp<float>(ws, p.w_c2slab);
  float* c2red = twsp<float>(ws, p.w_c2red);
  float* c1part = twsp<float>(ws, p.w_c1part);
  float* dwfp = twsp<float>(ws, p.w_dwfp);
  // Linear(128 -> latent)
  if ((rc = linear_backward(h3, DP_FC, params + p.o_f2w, nullptr, 0, dy, p.L, dh3, DP_FC, grads + p.o_f2w,
                            grads + p.o_f2b, p.B, DP_FC, p.L, LIN_NONE, twsp<void>(ws, p.w_lin), p.lin_bytes + 64, s)))
    return rc;
  const long long n3 = (long long)p.B * DP_FC;
  hipLaunchKernelGGL(k_depth_elu_grad, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, dh3, h3, dz3, n3);
  {  // fc1 weight gradient (channels-last column order) + bias: dW[128][64768] = dz3^T a2
    GemmArgs g;
    g.A = dz3; g.lda = DP_FC;
    g.B = a2; g.ldb = DP_FLAT;
    g.M = DP_FC; g.N = DP_FLAT; g.K = p.B;
    g.C = dwfp; g.ldc = DP_FLAT; g.Cbias = grads + p.o_f1b;
    IGI_HIP_TRY(gemm(g, false, false, s));
    hipLaunchKernelGGL(k_depth_perm_fc, dim3(2048), dim3(256), 0, s, dwfp, grads + p.o_f1w, 0);
  }
  {  // d(conv2 pre-activation) = (dz3 . Wp) * elu'(a2)
    GemmArgs g;
    g.A = dz3; g.lda = DP_FC;
    g.B = wfp; g.ldb = DP_FLAT;
    g.M = p.B; g.N = DP_FLAT; g.K = DP_FC;
    g.C = dz2; g.ldc = DP_FLAT; g.aux = a2; g.ldaux = DP_FLAT; g.epilogue = EPI_ELUGRAD;
    IGI_HIP_TRY(gemm(g, true, false, s));
  }
  {  // conv2 weight gradient, transposed (taps on M), split over the output pixels
    GemmArgs g;
    g.A = a1; g.gather = 3; g.conv = conv_desc(zero, DP_H2, DP_W2, DP_PH, DP_PW, DP_C1, 1, 0, 3, 3); g.lda = 288;
    g.B = dz2; g.ldb = DP_C2;
    g.M = 288; g.N = DP_C2; g.K = p.B * DP_P2;
    g.C = c2slab; g.ldc = DP_C2; g.Cbias = c2slab + (long long)p.sk_c2 * 288 * DP_C2; g.bias_from_b = 1;
    g.splitk = p.sk_c2; g.sCsplit = 288LL * DP_C2; g.sCbiasSplit = DP_C2;
    IGI_HIP_TRY(gemm(g, false, false, s));
    split_sum(c2red, c2slab, 288LL * DP_C2, p.sk_c2, 288LL * DP_C2, s);
    split_sum(grads + p.o_c2b, c2slab + (long long)p.sk_c2 * 288 * DP_C2, DP_C2, p.sk_c2, DP_C2, s);
    ConvWJobs ju;
    ju.j[0] = ConvWJob{c2red, grads + p.o_c2w, nullptr, DP_C2, DP_C1, 3, 3, DP_C1};
    ju.j[1] = ju.j[0]; ju.j[2] = ju.j[0];
    hipLaunchKernelGGL(k_tactile_unpack_gw, dim3(72, 1), dim3(256), 0, s, ju);
  }
  {  // conv2 data gradient -> g1 = d(pooled pre-ELU map) = (dz2 (*) flipped W2) * elu'(a1)
    GemmArgs g;
    g.A = dz2; g.gather = 1; g.conv = conv_desc(zero, DP_PH, DP_PW, DP_H2, DP_W2, DP_C2, 1, 2, 3, 3); g.lda = 576;
    g.B = w2d; g.ldb = 576;
    g.M = p.B * DP_P1; g.N = DP_C1; g.K = 576;
    g.C = g1; g.ldc = DP_C1; g.aux = a1; g.ldaux = DP_C1; g.epilogue = EPI_ELUGRAD;
    IGI_HIP_TRY(gemm(g, true, true, s));
  }
  hipLaunchKernelGGL(k_depth_conv1_wgrad, dim3(p.c1_blocks), dim3(256), 0, s, x, g1,
                     twsp<unsigned char>(ws, p.w_idx1), c1part, p.B);
  float* c1red = twsp<float>(ws, p.w_c1red);
  split_sum(c1red, c1part, (long long)DP_C1 * 26, p.c1_blocks, (long long)DP_C1 * 26, s);
  hipLaunchKernelGGL(k_depth_conv1_gout, dim3(1), dim3(256), 0, s, c1red, grads + p.o_c1w, grads + p.o_c1b);
  return (int)hipGetLastError();
}

}  // namespace igi
