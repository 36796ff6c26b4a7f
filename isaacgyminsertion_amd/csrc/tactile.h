// AllSight tactile encoder (algo/models/transformer/tactile_cnn.py:62-79): Conv(3->32,k8,s2)+ReLU,
// Conv(32->64,k4)+ReLU, Conv(64->64,k3)+ReLU, SpatialSoftArgmax(normalize=True), Linear(128->latent);
// forward and backward.
//
// MI355X design: activations are kept channels-last so every convolution (forward, data gradient,
// weight gradient) is an implicit GEMM on the exact-fp32 MFMA kernel of gemm_dma.h -- the im2col
// matrix is never materialised, the LDS-DMA loader gathers 16-byte channel groups of the input
// pixels straight into the LDS ring (out-of-image taps of the data-gradient read a zero page):
//   forward  a_out[m][co] = relu(b[co] + sum_{ky,kx,c} a_in[pix(m,ky,kx)][c] * Wr[co][ky][kx][c])
//   dgrad    dz_in[m][c]  = relu'(a_in) * sum_{ky,kx,co} dz_out[pix'(m,ky,kx)][co] * Wd[c][ky][kx][co]
//            (full correlation: pad = K-1, taps flipped in the repacked weight Wd)
//   wgrad    dWr[co][(ky,kx,c)] = sum_m dz_out[m][co] * a_in[pix(m,ky,kx)][c]   (split over m)
// conv1's FORWARD reads the caller's NCHW images as they are (ConvDesc::planar): its taps are ordered (c, ky, kx) --
// torch's own weight layout, no repacking -- so a 32-float k-tile is four kernel rows of one colour plane and
// K = 3 * 64 = 192: six k-tiles instead of the eight of a channels-last copy padded to four channels (488 -> 380 us at
// 8192 images).  Its WEIGHT GRADIENT keeps that padded channels-last copy (made by the forward, one kernel row = 128
// contiguous bytes): on the planar tensor the same product fetches 32-byte runs and is bound by the fetches
// (553 -> 858 us, measured), and a 192-row tap tile saves no MFMA over the 256-row one.  Spatial soft-argmax reproduces the reference's coordinate grid quirk
// (tactile_cnn.py:32-58): for flat position k of the row-major h*w map,
// x-weight = linspace(-1,1,w)[k / h], y-weight = linspace(-1,1,h)[k % h]; output [x0,y0,x1,y1,...].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/igi_ppo.h"
#include "gemm_dma.h"
#include "teacher.h"

namespace igi {

constexpr int TC_C1 = 32, TC_C2 = 64, TC_C3 = 64;

struct TactilePlan {
  int B, H, W, L;
  int H1, W1, H2, W2, H3, W3;
  long long M1, M2, M3;
  // parameter offsets (torch order: cnn.0.{w,b}, cnn.2.{w,b}, cnn.4.{w,b}, cnn.7.{w,b})
  long long o_w1, o_b1, o_w2, o_b2, o_w3, o_b3, o_wf, o_bf, P;
  // workspace (bytes)
  size_t w_zero, w_xin, w_w2r, w_w3r, w_w2d, w_w3d, w_a1, w_a2, w_a3, w_ssa_part, w_sstat, w_feat, w_dfeat, w_dz3,
      w_dz2, w_dz1, w_slab, w_gr, w_total;
  int ssa_fused;   // conv3's tiles emit the soft-argmax partials (k_softargmax_combine finishes them)
  int sk1, sk2, sk3, skf;
  long long s_w1, s_b1, s_w2, s_b2, s_w3, s_b3, s_wf, s_bf, slab_floats;  // slab offsets (floats)
  long long g_w1r, g_w2r, g_w3r;                                          // reduced repacked grads (floats)
};

static int make_tactile_plan(const igi_tactile_cfg* c, TactilePlan* p) {
  memset(p, 0, sizeof(*p));
  if (!c || c->batch < 1 || c->height < 16 || c->width < 16 || c->latent_dim < 4 || (c->latent_dim & 3))
    return IGI_E_BADARG;
  if (c->batch % 32) return IGI_E_UNSUPPORTED;  // weight-gradient k-ranges are whole 32-row k-tiles
  p->B = c->batch; p->H = c->height; p->W = c->width; p->L = c->latent_dim;
  p->H1 = (p->H - 8) / 2 + 1; p->W1 = (p->W - 8) / 2 + 1;
  p->H2 = p->H1 - 3; p->W2 = p->W1 - 3;
  p->H3 = p->H2 - 2; p->W3 = p->W2 - 2;
  if (p->H3 < 2 || p->W3 < 2) return IGI_E_BADARG;
  p->M1 = (long long)p->B * p->H1 * p->W1;
  p->M2 = (long long)p->B * p->H2 * p->W2;
  p->M3 = (long long)p->B * p->H3 * p->W3;
  if (p->M1 * 32 >= (1LL << 31) || (long long)p->B * p->H * p->W * 4 >= (1LL << 31)) return IGI_E_UNSUPPORTED;
  long long o = 0;
  p->o_w1 = o; o += TC_C1 * 3 * 64;
  p->o_b1 = o; o += TC_C1;
  p->o_w2 = o; o += TC_C2 * TC_C1 * 16;
  p->o_b2 = o; o += TC_C2;
  p->o_w3 = o; o += TC_C3 * TC_C2 * 9;
  p->o_b3 = o; o += TC_C3;
  p->o_wf = o; o += (long long)p->L * 128;
  p->o_bf = o; o += p->L;
  p->P = o;
  size_t w = 0;
  auto take = [&](size_t bytes) { size_t at = w; w += (size_t)ru64((long long)bytes); return at; };
  p->w_zero = take(256);
  p->w_xin = take(sizeof(float) * (size_t)p->B * p->H * p->W * 4);
  p->w_w2r = take(sizeof(float) * TC_C2 * 512);
  p->w_w3r = take(sizeof(float) * TC_C3 * 576);
  p->w_w2d = take(sizeof(float) * TC_C1 * 1024);
  p->w_w3d = take(sizeof(float) * TC_C2 * 576);
  p->w_a1 = take(sizeof(float) * p->M1 * TC_C1);
  p->w_a2 = take(sizeof(float) * p->M2 * TC_C2);
  p->w_a3 = take(sizeof(float) * p->M3 * TC_C3);
  p->ssa_fused = conv_ssa_fusable(p->M3, TC_C3, p->H3 * p->W3) ? 1 : 0;
  p->w_ssa_part = p->ssa_fused ? take(sizeof(float) * (size_t)(p->M3 / 32) * TC_C3 * 4) : 0;
  p->w_sstat = take(sizeof(float) * (size_t)p->B * 64 * 2);
  p->w_feat = take(sizeof(float) * (size_t)p->B * 128);
  p->w_dfeat = take(sizeof(float) * (size_t)p->B * 128);
  p->w_dz3 = take(sizeof(float) * p->M3 * TC_C3);
  p->w_dz2 = take(sizeof(float) * p->M2 * TC_C2);
  p->w_dz1 = take(sizeof(float) * p->M1 * TC_C1);
  // weight gradients are computed transposed (taps on the 128-row M side, output channels on N):
  // with 32-64 output channels the natural orientation would leave half or more of every 128-row tile empty
  // Two resident workgroups per CU as soon as each still reduces over >= 32 k-tiles (dma_choose_splitk asks for 128:
  // right for the big square products it was tuned on, but these tall-and-thin ones -- a few tap tiles by 32 / 64 output
  // channels -- then ran one workgroup per CU, conv1's on HALF the CUs, at 2048 images: conv1 260 -> 121 us, conv2
  // 303 -> 280, conv3 269 -> 239 at the configs[3] share; at 8192 images conv1 533 -> 453, the others already had two)
  auto two_per_cu = [](int sk, long long K, int tap_tiles) {
    const int s2 = 512 / tap_tiles;
    return (s2 > sk && K / s2 >= 32LL * DMA_BK) ? s2 : sk;
  };
  p->sk1 = two_per_cu(dma_choose_splitk(256, TC_C1, (int)p->M1, 1), p->M1, 1);
  {
    static int tall = -1;
    if (tall < 0) { const char* e = getenv("IGI_CONV_TALL"); tall = e ? atoi(e) : 3; }
    p->sk2 = two_per_cu(dma_choose_splitk(512, TC_C2, (int)p->M2, 1, tall > 2 ? 256 : DMA_BM), p->M2, tall > 2 ? 2 : 4);
  }
  {
    // (576 taps: three 192-tap tiles when gemm() will take them -- whole groups of 32 images, see conv_pw_ok -- else five
    // 128-tap tiles)
    static int tall = -1;
    if (tall < 0) { const char* e = getenv("IGI_CONV_TALL"); tall = e ? atoi(e) : 3; }
    const bool t192 = tall > 2 && conv_bm192_on() && (p->B % 32) == 0;
    p->sk3 = two_per_cu(dma_choose_splitk(576, TC_C3, (int)p->M3, 1, t192 ? 192 : DMA_BM), p->M3, t192 ? 3 : 5);
  }
  p->skf = dma_choose_splitk(p->L, 128, p->B, 1);
  if (const char* e = getenv("IGI_TAC_SK")) {   // "s1,s2,s3" (0 = keep): split factors of the three weight gradients, for A/B runs
    int v[3] = {0, 0, 0}, i = 0;
    for (const char* q = e; *q && i < 3; ++i) { v[i] = atoi(q); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
    auto cap = [](int sk, long long rows) { const long long m = rows / 128 > 1 ? rows / 128 : 1; return (int)(sk > m ? m : sk); };
    if (v[0] > 0) p->sk1 = cap(v[0], p->M1);
    if (v[1] > 0) p->sk2 = cap(v[1], p->M2);
    if (v[2] > 0) p->sk3 = cap(v[2], p->M3);
  }
  long long s = 0;
  p->s_w1 = s; s += (long long)p->sk1 * TC_C1 * 256;
  p->s_b1 = s; s += (long long)p->sk1 * TC_C1;
  p->s_w2 = s; s += (long long)p->sk2 * TC_C2 * 512;
  p->s_b2 = s; s += (long long)p->sk2 * TC_C2;
  p->s_w3 = s; s += (long long)p->sk3 * TC_C3 * 576;
  p->s_b3 = s; s += (long long)p->sk3 * TC_C3;
  p->s_wf = s; s += (long long)p->skf * p->L * 128;
  p->s_bf = s; s += (long long)p->skf * p->L;
  p->slab_floats = s;
  p->w_slab = take(sizeof(float) * (size_t)s);
  long long gr = 0;
  p->g_w1r = gr; gr += TC_C1 * 256;
  p->g_w2r = gr; gr += TC_C2 * 512;
  p->g_w3r = gr; gr += TC_C3 * 576;
  p->w_gr = take(sizeof(float) * (size_t)gr);
  p->w_total = w;
  return 0;
}

template <typename T>
static inline T* twsp(void* ws, size_t off) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off);
}

// (B,3,H,W) -> (B,H,W,4) with a zero fourth channel, for conv1's weight gradient; also clears the zero page
__global__ __launch_bounds__(256) void k_tactile_pack_input(const float* __restrict__ x, int B, int H, int W,
                                                            float* __restrict__ xin, float* __restrict__ zero) {
  if (blockIdx.x == 0 && threadIdx.x < 64) zero[threadIdx.x] = 0.f;
  const long long HW = (long long)H * W, total = (long long)B * HW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / HW, pix = e - b * HW;
    const float* src = x + b * 3 * HW + pix;
    float4 v = make_float4(src[0], src[HW], src[2 * HW], 0.f);
    *reinterpret_cast<float4*>(xin + e * 4) = v;
  }
}

// torch (co,ci,kh,kw) -> forward layout Wr[co][ky][kx][cpad]; and, for the data gradient, the flipped
// Wd[ci][ky'][kx'][co] = W[co][ci][KH-1-ky'][KW-1-kx'] (dgrad != nullptr).
// (one launch for all layers: blockIdx.y = job; as a launch per layer these 5 us kernels were launch latency)
struct ConvWJob { const float* src; float* dst; float* dst2; int CO, CI, KH, KW, CP; };
struct ConvWJobs { ConvWJob j[3]; };
__global__ __launch_bounds__(256) void k_tactile_pack_w(const ConvWJobs jobs) {
  const ConvWJob& jb = jobs.j[blockIdx.y];
  const float* __restrict__ w = jb.src;
  float* __restrict__ wr = jb.dst;
  float* __restrict__ wd = jb.dst2;
  const int CO = jb.CO, CI = jb.CI, KH = jb.KH, KW = jb.KW, CP = jb.CP;
  const int total = CO * KH * KW * CP;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int c = e % CP;
    const int kx = (e / CP) % KW;
    const int ky = (e / (CP * KW)) % KH;
    const int co = e / (CP * KW * KH);
    const float v = (c < CI) ? w[((co * CI + c) * KH + ky) * KW + kx] : 0.f;
    wr[e] = v;
    if (wd && c < CI) wd[((c * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)) * CO + co] = v;
  }
}

// reduced gradient gWr[(ky,kx,cpad)][co] (taps on the GEMM's M side) -> torch layout (co,ci,kh,kw)
__global__ __launch_bounds__(256) void k_tactile_unpack_gw(const ConvWJobs jobs) {
  const ConvWJob& jb = jobs.j[blockIdx.y];
  const float* __restrict__ gwr = jb.src;
  float* __restrict__ gw = jb.dst;
  const int CO = jb.CO, CI = jb.CI, KH = jb.KH, KW = jb.KW, CP = jb.CP;
  const int total = CO * CI * KH * KW;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int kx = e % KW;
    const int ky = (e / KW) % KH;
    const int c = (e / (KW * KH)) % CI;
    const int co = e / (KW * KH * CI);
    gw[e] = gwr[(((ky * KW + kx) * CP) + c) * CO + co];  // reduced gradient is [tap][co]
  }
}

__device__ __forceinline__ float linspace_pm1(int i, int steps) {
  // torch.linspace(-1, 1, steps)[i] in fp32 (symmetric evaluation, as ATen does)
  const float step = 2.0f / (float)(steps - 1);
  return (i < steps / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(steps - i - 1));
}

// one wave per image, lane = channel (64): softmax over the H3*W3 positions of channel `lane`.
// Loads are issued eight positions at a time (independent), the sums stay strictly in position order;
// the (k / h, k % h) grid coordinates advance incrementally instead of two divisions per position.
__global__ __launch_bounds__(64) void k_softargmax_fwd(const float* __restrict__ a3, int P, int h, int w,
                                                       float* __restrict__ feat, float* __restrict__ sstat) {
  const int b = blockIdx.x, c = threadIdx.x;
  const float* src = a3 + (long long)b * P * 64 + c;
  float m = -INFINITY;
  int k = 0;
  for (; k + 8 <= P; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
  }
  for (; k < P; ++k) m = fmaxf(m, src[(long long)k * 64]);
  float s = 0.f, sx = 0.f, sy = 0.f;
  int q = 0, r = 0;  // q = k / h, r = k % h
  for (k = 0; k + 8 <= P; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float e = expf(v[u] - m);
      s += e;
      sx += e * linspace_pm1(q, w);
      sy += e * linspace_pm1(r, h);
      if (++r == h) { r = 0; ++q; }
    }
  }
  for (; k < P; ++k) {
    const float e = expf(src[(long long)k * 64] - m);
    s += e;
    sx += e * linspace_pm1(q, w);
    sy += e * linspace_pm1(r, h);
    if (++r == h) { r = 0; ++q; }
  }
  feat[(long long)b * 128 + 2 * c] = sx / s;
  feat[(long long)b * 128 + 2 * c + 1] = sy / s;
  sstat[((long long)b * 64 + c) * 2] = m;
  sstat[((long long)b * 64 + c) * 2 + 1] = s;
}

// Merges the per-32-row-group partials the conv3 tiles emitted (gemm_dma.h: ssa_tile_partials) into the image's
// soft-argmax: thread = (image, channel); groups in position order; (max_g, s_g, sx_g, sy_g) -> m = max_g max_g,
// s = sum_g s_g exp(max_g - m) (likewise sx, sy), feature = (sx / s, sy / s), sstat = (m, s) for the backward pass.
__global__ __launch_bounds__(256) void k_softargmax_combine(const float* __restrict__ part, int B, int groups,
                                                            float* __restrict__ feat, float* __restrict__ sstat) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * 64) return;
  const int b = (int)(t >> 6), c = (int)(t & 63);
  const float4* p = reinterpret_cast<const float4*>(part) + ((long long)b * groups) * 64 + c;
  float m = -INFINITY;
  for (int g0 = 0; g0 < groups; g0 += 6) {   // six groups per round trip (192 positions = 6 groups, 576 = 18)
    float4 v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = p[(long long)min(g0 + u, groups - 1) * 64];
#pragma unroll
    for (int u = 0; u < 6; ++u) m = fmaxf(m, v[u].x);
  }
  float s = 0.f, sx = 0.f, sy = 0.f;
  for (int g0 = 0; g0 < groups; g0 += 6) {
    float4 v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = p[(long long)min(g0 + u, groups - 1) * 64];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      if (g0 + u < groups) {
        const float w = expf(v[u].x - m);
        s += v[u].y * w;
        sx += v[u].z * w;
        sy += v[u].w * w;
      }
    }
  }
  feat[(long long)b * 128 + 2 * c] = sx / s;
  feat[(long long)b * 128 + 2 * c + 1] = sy / s;
  sstat[((long long)b * 64 + c) * 2] = m;
  sstat[((long long)b * 64 + c) * 2 + 1] = s;
}

// d(pre-ReLU conv3 output) = relu'(a3) * softmax_k * (gx*(xw_k - fx) + gy*(yw_k - fy))
__global__ __launch_bounds__(64) void k_softargmax_bwd(const float* __restrict__ a3, const float* __restrict__ feat,
                                                       const float* __restrict__ sstat,
                                                       const float* __restrict__ dfeat, int P, int h, int w,
                                                       float* __restrict__ dz3) {
  const int b = blockIdx.x, c = threadIdx.x;
  const float* src = a3 + (long long)b * P * 64 + c;
  float* dst = dz3 + (long long)b * P * 64 + c;
  const float m = sstat[((long long)b * 64 + c) * 2], s = sstat[((long long)b * 64 + c) * 2 + 1];
  const float fx = feat[(long long)b * 128 + 2 * c], fy = feat[(long long)b * 128 + 2 * c + 1];
  const float gx = dfeat[(long long)b * 128 + 2 * c], gy = dfeat[(long long)b * 128 + 2 * c + 1];
  int q = 0, r = 0;  // q = k / h, r = k % h
  int k = 0;
  for (; k + 8 <= P; k += 8) {   // eight positions per round trip (one load per trip made the map a chain of P latencies)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float sm = expf(v[u] - m) / s;
      const float d = sm * (gx * (linspace_pm1(q, w) - fx) + gy * (linspace_pm1(r, h) - fy));
      dst[(long long)(k + u) * 64] = (v[u] > 0.f) ? d : 0.f;
      if (++r == h) { r = 0; ++q; }
    }
  }
  for (; k < P; ++k) {
    const float v = src[(long long)k * 64];
    const float sm = expf(v - m) / s;
    const float d = sm * (gx * (linspace_pm1(q, w) - fx) + gy * (linspace_pm1(r, h) - fy));
    dst[(long long)k * 64] = (v > 0.f) ? d : 0.f;
    if (++r == h) { r = 0; ++q; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Standalone SpatialSoftArgmax (tactile_cnn.py:7-58) on an NCHW tensor, any channel count: one wave per (image,
// channel) row of P = h*w contiguous values, lanes stride the positions.  Same coordinate quirk as above (flat index
// k -> x-weight grid_w[k / h], y-weight grid_h[k % h]); normalize = 0 uses the integer grids arange(w) / arange(h).
// out[row] = (E[x], E[y]) (interleaved per channel, as the reference's cat + view gives); stat[row] = (max, sum exp).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ssa_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ssa_coord(int i, int steps, int normalize) {
  return normalize ? linspace_pm1(i, steps) : (float)i;
}
__global__ __launch_bounds__(256) void k_ssa_fwd(const float* __restrict__ x, long long rows, int h, int w,
                                                 int normalize, float* __restrict__ out, float* __restrict__ stat) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int P = h * w;
  const float* src = x + row * P;
  float m = -INFINITY;
  for (int k = lane; k < P; k += 64) m = fmaxf(m, src[k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f, sx = 0.f, sy = 0.f;
  for (int k = lane; k < P; k += 64) {
    const float e = expf(src[k] - m);
    const int q = k / h, r = k - q * h;
    s += e;
    sx += e * ssa_coord(q, w, normalize);
    sy += e * ssa_coord(r, h, normalize);
  }
  s = ssa_wave_sum(s); sx = ssa_wave_sum(sx); sy = ssa_wave_sum(sy);
  if (lane == 0) {
    out[2 * row] = sx / s;
    out[2 * row + 1] = sy / s;
    stat[2 * row] = m;
    stat[2 * row + 1] = s;
  }
}
// dx[k] = softmax_k * (gx (xw_k - E[x]) + gy (yw_k - E[y]))
__global__ __launch_bounds__(256) void k_ssa_bwd(const float* __restrict__ x, const float* __restrict__ out,
                                                 const float* __restrict__ stat, const float* __restrict__ dout,
                                                 long long rows, int h, int w, int normalize, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int P = h * w;
  const float m = stat[2 * row], s = stat[2 * row + 1];
  const float fx = out[2 * row], fy = out[2 * row + 1], gx = dout[2 * row], gy = dout[2 * row + 1];
  for (int k = lane; k < P; k += 64) {
    const int q = k / h, r = k - q * h;
    const float sm = expf(x[row * P + k] - m) / s;
    dx[row * P + k] = sm * (gx * (ssa_coord(q, w, normalize) - fx) + gy * (ssa_coord(r, h, normalize) - fy));
  }
}
static int spatial_softargmax_forward(const float* x, long long rows, int h, int w, int normalize, float* out,
                                      float* stat, hipStream_t s) {
  if (!x || !out || !stat || rows < 0 || h < 1 || w < 1 || (long long)h * w > (1 << 24)) return IGI_E_BADARG;
  if (normalize && (h < 2 || w < 2)) return IGI_E_BADARG;      // linspace(-1, 1, 1) has no step
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_ssa_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, h, w, normalize, out, stat);
  return (int)hipGetLastError();
}
static int spatial_softargmax_backward(const float* x, const float* out, const float* stat, const float* dout,
                                       long long rows, int h, int w, int normalize, float* dx, hipStream_t s) {
  if (!x || !out || !stat || !dout || !dx || rows < 0 || h < 1 || w < 1) return IGI_E_BADARG;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_ssa_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, out, stat, dout, rows, h, w,
                     normalize, dx);
  return (int)hipGetLastError();
}

static ConvDesc conv_desc(const float* zero, int OH, int OW, int IH, int IW, int C, int stride, int pad, int KH,
                          int KW, int planar = 0) {
  ConvDesc d;
  d.planar = planar;
  d.zero = zero; d.OW = OW; d.OHW = OH * OW; d.IH = IH; d.IW = IW; d.C = C; d.stride = stride; d.pad = pad;
  d.KW = KW; d.KH = KH; d.ntaps = KH * KW * C;
  d.dOW = make_fastdiv(OW); d.dOHW = make_fastdiv(OH * OW); d.dC = make_fastdiv(C); d.dKW = make_fastdiv(KW);
  d.dTPP = make_fastdiv(C >= 32 ? C / 32 : 1);
  return d;
}

static int tactile_forward(const igi_tactile_cfg* c, const float* x, const float* params, float* y, void* ws,
                           size_t ws_bytes, hipStream_t s) {
  TactilePlan p;
  int rc = make_tactile_plan(c, &p);
  if (rc) return rc;
  if (!x || !params || !y || !ws) return IGI_E_BADARG;
  if (ws_bytes < p.w_total) return IGI_E_WORKSPACE;
  float* zero = twsp<float>(ws, p.w_zero);
  float* xin = twsp<float>(ws, p.w_xin);
  float *w2r = twsp<float>(ws, p.w_w2r), *w3r = twsp<float>(ws, p.w_w3r);
  float *w2d = twsp<float>(ws, p.w_w2d), *w3d = twsp<float>(ws, p.w_w3d);
  float *a1 = twsp<float>(ws, p.w_a1), *a2 = twsp<float>(ws, p.w_a2), *a3 = twsp<float>(ws, p.w_a3);
  {
    long long tot = (long long)p.B * p.H * p.W;
    int nb = (int)((tot + 255) / 256);   // one pixel per thread: a capped grid made each thread a chain of dependent round trips
    if (nb > (1 << 20)) nb = 1 << 20;
    hipLaunchKernelGGL(k_tactile_pack_input, dim3(nb), dim3(256), 0, s, x, p.B, p.H, p.W, xin, zero);
  }
  {
    ConvWJobs jw;
    jw.j[0] = ConvWJob{params + p.o_w2, w2r, w2d, TC_C2, TC_C1, 4, 4, TC_C1};
    jw.j[1] = ConvWJob{params + p.o_w3, w3r, w3d, TC_C3, TC_C2, 3, 3, TC_C2};
    jw.j[2] = jw.j[1];
    hipLaunchKernelGGL(k_tactile_pack_w, dim3(144, 2), dim3(256), 0, s, jw);
  }
  {  // conv1: (B,3,H,W) -> (B,H1,W1,32), straight from the caller's tensor
    GemmArgs g;
    g.A = x; g.gather = 1; g.conv = conv_desc(zero, p.H1, p.W1, p.H, p.W, 3, 2, 0, 8, 8, 1);
    g.B = params + p.o_w1; g.ldb = 192;   // torch's (co, c, ky, kx) IS the planar tap order
    g.M = (int)p.M1; g.N = TC_C1; g.K = 192; g.lda = 192;
    g.C = a1; g.ldc = TC_C1; g.bias = params + p.o_b1; g.epilogue = EPI_BIAS_RELU;
    IGI_HIP_TRY(gemm(g, true, true, s));
  }
  {  // conv2
    GemmArgs g;
    g.A = a1; g.gather = 1; g.conv = conv_desc(zero, p.H2, p.W2, p.H1, p.W1, TC_C1, 1, 0, 4, 4);
    g.B = w2r; g.ldb = 512;
    g.M = (int)p.M2; g.N = TC_C2; g.K = 512; g.lda = 512;
    g.C = a2; g.ldc = TC_C2; g.bias = params + p.o_b2; g.epilogue = EPI_BIAS_RELU;
    IGI_HIP_TRY(gemm(g, true, true, s));
  }
  {  // conv3
    GemmArgs g;
    g.A = a2; g.gather = 1; g.conv = conv_desc(zero, p.H3, p.W3, p.H2, p.W2, TC_C2, 1, 0, 3, 3);
    g.B = w3r; g.ldb = 576;
    g.M = (int)p.M3; g.N = TC_C3; g.K = 576; g.lda = 576;
    g.C = a3; g.ldc = TC_C3; g.bias = params + p.o_b3; g.epilogue = EPI_BIAS_RELU;
    if (p.ssa_fused) {   // the tiles emit the soft-argmax partials of their 32-row groups (tactile_cnn.py:46-58 fused in)
      g.ssa_part = twsp<float>(ws, p.w_ssa_part); g.ssa_P = p.H3 * p.W3; g.ssa_h = p.H3; g.ssa_w = p.W3;
    }
    IGI_HIP_TRY(gemm(g, true, true, s));
  }
  if (p.ssa_fused) {
    const int groups = p.H3 * p.W3 / 32;
    ProfScope ps(PC_SOFTARGMAX_FWD, s, 0.0, 16.0 * (double)(p.M3 / 32) * TC_C3 * 2);   // two passes over the partials
    IGI_LAUNCH(k_softargmax_combine, dim3((p.B * 64 + 255) / 256), dim3(256), 0, s, twsp<float>(ws, p.w_ssa_part), p.B,
               groups, twsp<float>(ws, p.w_feat), twsp<float>(ws, p.w_sstat));
  } else {
    ProfScope ps(PC_SOFTARGMAX_FWD, s, 0.0, 2.0 * 4.0 * (double)p.M3 * TC_C3);   // two passes over a3
    IGI_LAUNCH(k_softargmax_fwd, dim3(p.B), dim3(64), 0, s, a3, p.H3 * p.W3, p.H3, p.W3,
               twsp<float>(ws, p.w_feat), twsp<float>(ws, p.w_sstat));
  }
  {  // Linear(128 -> latent)
    GemmArgs g;
    g.A = twsp<float>(ws, p.w_feat); g.lda = 128;
    g.B = params + p.o_wf; g.ldb = 128;
    g.M = p.B; g.N = p.L; g.K = 128;
    g.C = y; g.ldc = p.L; g.bias = params + p.o_bf; g.epilogue = EPI_BIAS;
    IGI_HIP_TRY(gemm(g, true, true, s));
  }
  return (int)hipGetLastError();
}

static int tactile_backward(const igi_tactile_cfg* c, const float* dy, const float* params, float* grads,
                            void* ws, size_t ws_bytes, hipStream_t s) {
  TactilePlan p;
  int rc = make_tactile_plan(c, &p);
  if (rc) return rc;
  if (!dy || !params || !grads || !ws) return IGI_E_BADARG;
  if (ws_bytes < p.w_total) return IGI_E_WORKSPACE;
  float* zero = twsp<float>(ws, p.w_zero);
  float* xin = twsp<float>(ws, p.w_xin);
  float *w2d = twsp<float>(ws, p.w_w2d), *w3d = twsp<float>(ws, p.w_w3d);
  float *a1 = twsp<float>(ws, p.w_a1), *a2 = twsp<float>(ws, p.w_a2), *a3 = twsp<float>(ws, p.w_a3);
  float *dz1 = twsp<float>(ws, p.w_dz1), *dz2 = twsp<float>(ws, p.w_dz2), *dz3 = twsp<float>(ws, p.w_dz3);
  float* feat = twsp<float>(ws, p.w_feat);
  float* dfeat = twsp<float>(ws, p.w_dfeat);
  float* slab = twsp<float>(ws, p.w_slab);
  float* gr = twsp<float>(ws, p.w_gr);
  {  // d(feat) = dy . Wfc
    GemmArgs g;
    g.A = dy; g.lda = p.L;
    g.B = params + p.o_wf; g.ldb = 128;
    g.M = p.B; g.N = 128; g.K = p.L;
    g.C = dfeat; g.ldc = 128;
    IGI_HIP_TRY(gemm(g, true, false, s));
  }
  {  // dWfc = dy^T feat, dbfc
    GemmArgs g;
    g.A = dy; g.lda = p.L;
    g.B = feat; g.ldb = 128;
    g.M = p.L; g.N = 128; g.K = p.B;
    g.C = slab + p.s_wf; g.ldc = 128; g.Cbias = slab + p.s_bf;
    g.splitk = p.skf; g.sCsplit = (long long)p.L * 128; g.sCbiasSplit = p.L;
    IGI_HIP_TRY(gemm(g, false, false, s));
  }
  {
    ProfScope ps(PC_SOFTARGMAX_BWD, s, 0.0, 2.0 * 4.0 * (double)p.M3 * TC_C3);   // a3 read, dz3 written
    IGI_LAUNCH(k_softargmax_bwd, dim3(p.B), dim3(64), 0, s, a3, feat, twsp<float>(ws, p.w_sstat), dfeat,
               p.H3 * p.W3, p.H3, p.W3, dz3);
  }
  {  // conv3 weight gradient
    GemmArgs g;
    g.A = a2; g.gather = 3; g.conv = conv_desc(zero, p.H3, p.W3, p.H2, p.W2, TC_C2, 1, 0, 3, 3); g.lda = 576;
    g.B = dz3; g.ldb = TC_C3;
    g.M = 576; g.N = TC_C3; g.K = (int)p.M3;
    g.C = slab + p.s_w3; g.ldc = TC_C3; g.Cbias = slab + p.s_b3; g.bias_from_b = 1;
    g.splitk = p.sk3; g.sCsplit = (long long)TC_C3 * 576; g.sCbiasSplit = TC_C3;
    IGI_HIP_TRY(gemm(g, false, false, s));
  }
  {  // conv3 data gradient -> dz2 = relu'(a2) * (dz3 (*) flipped W3)
    GemmArgs g;
    g.A = dz3; g.gather = 1; g.conv = conv_desc(zero, p.H2, p.W2, p.H3, p.W3, TC_C3, 1, 2, 3, 3); g.lda = 576;
    g.B = w3d; g.ldb = 576;
    g.M = (int)p.M2; g.N = TC_C2; g.K = 576;
    g.C = dz2; g.ldc = TC_C2; g.aux = a2; g.ldaux = TC_C2; g.epilogue = EPI_RELUGRAD;
    g.flop_credit = (double)p.M3 / (double)p.M2;   // algorithmic MACs = the forward's: the padded correlation's zero taps carry none
    IGI_HIP_TRY(gemm(g, true, true, s));
  }
  {  // conv2 weight gradient
    GemmArgs g;
    g.A = a1; g.gather = 3; g.conv = conv_desc(zero, p.H2, p.W2, p.H1, p.W1, TC_C1, 1, 0, 4, 4); g.lda = 512;
    g.B = dz2; g.ldb = TC_C2;
    g.M = 512; g.N = TC_C2; g.K = (int)p.M2;
    g.C = slab + p.s_w2; g.ldc = TC_C2; g.Cbias = slab + p.s_b2; g.bias_from_b = 1;
    g.splitk = p.sk2; g.sCsplit = (long long)TC_C2 * 512; g.sCbiasSplit = TC_C2;
    IGI_HIP_TRY(gemm(g, false, false, s));
  }
  {  // conv2 data gradient -> dz1
    GemmArgs g;
    g.A = dz2; g.gather = 1; g.conv = conv_desc(zero, p.H1, p.W1, p.H2, p.W2, TC_C2, 1, 3, 4, 4); g.lda = 1024;
    g.B = w2d; g.ldb = 1024;
    g.M = (int)p.M1; g.N = TC_C1; g.K = 1024;
    g.C = dz1; g.ldc = TC_C1; g.aux = a1; g.ldaux = TC_C1; g.epilogue = EPI_RELUGRAD;
    g.flop_credit = (double)p.M2 / (double)p.M1;
    IGI_HIP_TRY(gemm(g, true, true, s));
  }
  {  // conv1 weight gradient (the input needs no gradient)
    GemmArgs g;
    g.A = xin; g.gather = 3; g.conv = conv_desc(zero, p.H1, p.W1, p.H, p.W, 4, 2, 0, 8, 8); g.lda = 256;
    g.B = dz1; g.ldb = TC_C1;
    g.M = 256; g.N = TC_C1; g.K = (int)p.M1;
    g.C = slab + p.s_w1; g.ldc = TC_C1; g.Cbias = slab + p.s_b1; g.bias_from_b = 1;
    g.splitk = p.sk1; g.sCsplit = (long long)TC_C1 * 256; g.sCbiasSplit = TC_C1;
    IGI_HIP_TRY(gemm(g, false, false, s));
  }
  // fixed-order split-k reduction: biases and the fc layer land in `grads`, conv weights in the
  // repacked scratch and are then un-permuted to torch's (co,ci,kh,kw)
  SegTable t;
  t.n = 0;
  t.wide = 1;
  auto add = [&](float* dst_base, long long dst, const float* src, long long stride, int count, int nparts) {
    Segment& sg = t.s[t.n++];
    sg.dst = dst + (dst_base - grads);  // relative to `grads`
    sg.src = src; sg.stride = stride; sg.count = count; sg.cols = count; sg.src_ld = 0; sg.nparts = nparts;
  };
  add(gr, p.g_w1r, slab + p.s_w1, (long long)TC_C1 * 256, TC_C1 * 256, p.sk1);
  add(gr, p.g_w2r, slab + p.s_w2, (long long)TC_C2 * 512, TC_C2 * 512, p.sk2);
  add(gr, p.g_w3r, slab + p.s_w3, (long long)TC_C3 * 576, TC_C3 * 576, p.sk3);
  add(grads, p.o_b1, slab + p.s_b1, TC_C1, TC_C1, p.sk1);
  add(grads, p.o_b2, slab + p.s_b2, TC_C2, TC_C2, p.sk2);
  add(grads, p.o_b3, slab + p.s_b3, TC_C3, TC_C3, p.sk3);
  add(grads, p.o_wf, slab + p.s_wf, (long long)p.L * 128, p.L * 128, p.skf);
  add(grads, p.o_bf, slab + p.s_bf, p.L, p.L, p.skf);
  {
    // hundreds of partials per element at bench scale (the convolutions' reductions run over millions of rows): 64
    // blocks per segment left the sum on a quarter of the chip (125 us at 8192 images); IGI_TAC_RED_GX blocks per segment
    static int gx = -1;
    if (gx < 0) { const char* e = getenv("IGI_TAC_RED_GX"); gx = e ? atoi(e) : 256; if (gx < 1) gx = 1; }
    hipLaunchKernelGGL(k_slab_reduce, dim3(gx, t.n), dim3(RED_THREADS), 0, s, t, grads);
  }
  {
    ConvWJobs ju;
    ju.j[0] = ConvWJob{gr + p.g_w1r, grads + p.o_w1, nullptr, TC_C1, 3, 8, 8, 4};
    ju.j[1] = ConvWJob{gr + p.g_w2r, grads + p.o_w2, nullptr, TC_C2, TC_C1, 4, 4, TC_C1};
    ju.j[2] = ConvWJob{gr + p.g_w3r, grads + p.o_w3, nullptr, TC_C3, TC_C2, 3, 3, TC_C2};
    hipLaunchKernelGGL(k_tactile_unpack_gw, dim3(144, 3), dim3(256), 0, s, ju);
  }
  return (int)hipGetLastError();
}

}  // namespace igi
