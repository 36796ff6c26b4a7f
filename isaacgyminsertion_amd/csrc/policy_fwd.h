// The rollout-side policy step as ONE persistent kernel per 32 environments (frozen_ppo.py:343-366 `model_act` +
// :655-665 of `play_steps`; models_split.py:120-164 `act` = env_mlp -> [obs | latent] -> actor / critic trunks -> heads
// -> Normal sample), for the reference's network shape (priv 64 -> 256 -> 128 -> 8, [obs 15 | latent 8] -> 512 -> 256 -> 128,
// <= 7 actions).  Other shapes keep the layer-by-layer launches (teacher.h, teacher_policy_step).
//
// Why: at 4096 environments the layer-by-layer step is seven dependent launches of 64 - 256 workgroups, 80 us for 3.4
// GFLOP (25 us of matrix pipe) -- every launch pays its fill, first DMA round trip, epilogue and drain with nothing to
// overlap them, and the activations go out to HBM and come back between launches.  Round 5 tried the 32-row chain kernel of
// the student's MLPs (register-staged weights, two barriers per 64-wide chunk): 103 us, a chain of global -> register ->
// LDS round trips.
//
// Shape of this kernel:
//  * a workgroup = 32 environments x ONE net (blockIdx & 1: actor | critic); both kinds run the small env_mlp for their
//    rows (22 % of a workgroup's work, twice) so that the 0.7 MB of trunk weights a workgroup streams are one net's, and
//    256 workgroups fill the chip at 4096 environments;
//  * every activation of the 32 rows stays in LDS as k-contiguous XOR-swizzled [32][32] images (the operand layout of
//    gemm_dma.h / env_mlp.h): 32 + 64 KB in two regions that the layers alternate between, nothing goes to HBM;
//  * weights are streamed by LDS-DMA in [32 outputs][16 k] chunks (2 KB) through WAVE-PRIVATE three-deep rings: wave w owns
//    output-column tile(s) w (+ 8) of the layer, requests chunk i + 2, waits (counted vmcnt) for chunk i + 1, reads its
//    fragments, and issues the eight MFMAs of chunk i -- no workgroup barrier inside a layer, one between layers;
//  * v_mfma_f32_32x32x2_f32, k order = the GEMM kernels' (k-tiles in order, pairs k, k + 4 inside every group of eight),
//    bias + tanh exactly as their epilogue: the trunk outputs are bit-identical to the per-layer launches; the 8-wide latent
//    layer and the heads are VALU dot products in another summation order (fp32 rounding apart);
//  * the heads, the Normal sample from the caller's noise, neglogp, the value de-normalisation and the seven arena writes
//    (k_heads_act_store's arithmetic) close the kernel: 2 launches per policy step (k_policy_stage + this).
#pragma once
#include "dma_util.h"
#include "gemm_f32.h"

namespace igi {

struct PolicyFwdArgs {
  const float* priv; int ldp;          // [rows][64] normalised privileged input (k_policy_stage)
  const float* xcat; int ldx;          // [rows][32]: normalised obs in columns 0 .. obs - 1, zeros behind obs + 8
  int rows, obs, act;
  const float *eW1, *eb1, *eW2, *eb2, *eW3, *eb3;      // env_mlp: [256][64], [128][256], [8][128]
  const float* w1p;                                    // [2][512][32] zero-padded first trunk layer (k_policy_stage refreshes it)
  const float *tb1, *tW2, *tb2, *tW3, *tb3;            // actor trunk; critic = + ac_block floats
  long long ac_block;
  const float *Wmu, *bmu, *Wv, *bv, *logstd;
  // sampling + arena (the arguments of k_heads_act_store)
  const float* noise; const double* rms_value; float eps;
  float *actions_t, *nlp_t, *values_t, *mus_t, *sigmas_t, *actions_clamped, *values_out;
};

constexpr int PF_ROWS = 32;
constexpr int PF_THREADS = 512;
constexpr int PF_IMG = 32 * 32;                 // one [32 rows][32 k] activation image (floats)
constexpr int PF_P = 8 * PF_IMG;                // region P: layer-1 output (8 images), xcat, trunk-2 output
constexpr int PF_Q = 16 * PF_IMG;               // region Q: input (2), env layer 2 (4), trunk-1 output (16), trunk-3 output (4)
constexpr int PF_CH = 32 * 16;                  // one weight chunk [32 outputs][16 k]
constexpr int PF_NS = 3;                        // ring depth per wave
constexpr int PF_RING = 8 * PF_NS * PF_CH;
constexpr int PF_SMALL = 8 * 128 /* W3 */ + 8 * 128 /* head rows: mu 0.., value 7 */ + 2 * 32 * 8 /* partial dots */ + 32 * 8 /* latent | mu */;
constexpr int PF_LDS_FLOATS = PF_P + PF_Q + PF_RING + PF_SMALL;
static_assert(PF_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
constexpr float PF_LOG_SQRT_2PI = 0.918938533204672741780329736406f;

static inline bool policy_fwd_shape_ok(int obs, int priv, int act, int npl, const int* pu, int nl, const int* u) {
  return priv == 64 && npl == 3 && pu[0] == 256 && pu[1] == 128 && pu[2] == 8 && nl == 3 && u[0] == 512 && u[1] == 256 &&
         u[2] == 128 && obs >= 1 && obs + 8 <= 32 && act >= 1 && act <= 7;
}

// IGI_POLICY_FUSED=0: the layer-by-layer policy step (A/B)
static inline bool policy_fwd_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_POLICY_FUSED"); on = e ? atoi(e) != 0 : 1; }
  return on != 0;
}

// float index of accumulator element r of a lane in a k-contiguous swizzled [32][32] image: see env_img_bases (env_mlp.h)
__device__ __forceinline__ void pf_img_bases(int l31, int h, int (&ib)[4]) {
  const int sk[4] = {0, 1, 4, 5};
#pragma unroll
  for (int k = 0; k < 4; ++k) ib[k] = (4 * h * 8 + ((l31 >> 2) ^ (sk[k] + 2 * h))) * 4 + (l31 & 3);
}

template <int I, int N, class F>
__device__ __forceinline__ void pf_for(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pf_for<I + 1, N>(f);
  }
}

// s_waitcnt vmcnt(N) with N a compile-time constant
template <int N>
__device__ __forceinline__ void pf_wait_vm() {
  static_assert(N >= 0 && N <= 12, "vmcnt immediates of the ring depths in use");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  static_assert((N & 1) == 0, "two requests per chunk");
}

struct PfWave {
  float* ring;                 // this wave's PF_NS chunks
  int l31, h, lane, wave;
  int ib[4];
  int aoff;                    // float offset of this lane's row in an activation image: l31 * 32
  int asw;                     // (l31 >> 1) & 7
  int boff, bsw;               // chunk: l31 * 16, (l31 >> 2) & 3
};

// One Linear + Tanh layer of the block: out[32][N] = tanh(in[32][K] . W[N][K]^T + b) for THIS wave's output-column tiles
// n = wave + NW j (j < NT); waves whose first tile lies beyond N / 32 skip the layer.  `ain`: K / 32 input images;
// `aout`: N / 32 output images (another LDS region); W rows ldw floats apart.  The caller puts a workgroup barrier behind.
// NW: the waves that compute (waves 0 .. NW - 1 own tiles wave + NW j); the others return at once.
// NS: depth of the wave's ring (NS - 1 chunks in flight ahead of the one being multiplied).
template <int K, int N, int NT, int NW = 8, int NS = PF_NS>
__device__ __forceinline__ void pf_layer(const PfWave& w, const float* __restrict__ ain, const float* __restrict__ W, int ldw,
                                         const float (&bias)[NT], float* __restrict__ aout) {
  constexpr int KC = K / 16, NI = KC * NT;       // items = (k-chunk c, tile j), c-major
  static_assert(K % 32 == 0 && N % 32 == 0, "whole images");
  static_assert(NT * NW * 32 >= N, "every output tile has an owner");
  if (w.wave >= NW || w.wave * 32 >= N) return;
  // the lane's two 16-byte pieces of a chunk request: DMA instruction i covers rows 16 i .. 16 i + 15, four pieces per row;
  // piece slot s of row r holds k-piece s ^ ((r >> 2) & 3) (so that the fragments' ds_read_b128 are conflict-free)
  unsigned goff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 16 * i + (w.lane >> 2), p = (w.lane & 3) ^ ((r >> 2) & 3);
    goff[i] = 4u * ((unsigned)r * (unsigned)ldw + 4u * (unsigned)p);
  }
  auto issue = [&](auto item_c) {
    constexpr int I = decltype(item_c)::value;
    constexpr int c = I / NT, j = I % NT;
    float* dst = w.ring + (I % NS) * PF_CH;
    const char* base = reinterpret_cast<const char*>(uniform_ptr(W + (long long)(32 * (w.wave + NW * j)) * ldw + 16 * c));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      asm volatile("" : "+v"(goff[i]));
      dma16(reinterpret_cast<const float*>(base + goff[i]), dst + 256 * i);
    }
  };
  struct Frag { f32x4 a[2], b[2]; };
  auto read = [&](auto item_c, Frag& f) {
    constexpr int I = decltype(item_c)::value;
    constexpr int c = I / NT;
    const float* ch = w.ring + (I % NS) * PF_CH;
    const float* img = ain + (c >> 1) * PF_IMG;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int g8 = 2 * (c & 1) + g;
      f.a[g] = *reinterpret_cast<const f32x4*>(img + w.aoff + (((2 * g8 + w.h) ^ w.asw) << 2));
      f.b[g] = *reinterpret_cast<const f32x4*>(ch + w.boff + (((2 * g + w.h) ^ w.bsw) << 2));
    }
  };
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  Frag fr[2];
  constexpr int D = NS - 1;                      // chunks in flight
  constexpr int PRO = D < NI ? D : NI;
  auto prologue = [&](auto item_c) { issue(item_c); };
  pf_for<0, PRO>(prologue);
  pf_wait_vm<2 * (PRO - 1)>();                   // item 0 has landed once only the younger requests are outstanding
  read(std::integral_constant<int, 0>{}, fr[0]);
  auto step = [&](auto item_c) {
    constexpr int I = decltype(item_c)::value;
    if constexpr (I + D < NI) issue(std::integral_constant<int, (I + D < NI ? I + D : 0)>{});
    if constexpr (I + 1 < NI) {
      // item I + 1 has landed once only the requests of the items behind it (two instructions each) are outstanding
      constexpr int last = (I + D < NI - 1) ? I + D : NI - 1;
      pf_wait_vm<2 * (last - (I + 1))>();
      read(std::integral_constant<int, (I + 1 < NI ? I + 1 : 0)>{}, fr[(I + 1) & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    const Frag& f = fr[I & 1];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[I % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[g][q], f.b[g][q], acc[I % NT], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // (fully unrolled: ring slots, image addresses and the vmcnt immediates are compile-time constants)
  pf_for<0, NI>(step);
  // bias + tanh into the next layer's operand images
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float* img = aout + (w.wave + NW * j) * PF_IMG;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      img[w.ib[(r >> 1) & 3] + ((r & 3) + 8 * (r >> 2)) * 32] = fast_tanh(acc[j][r] + bias[j]);
  }
}

// dot products of the block's 32 rows (images at `img`, 4 of them = 128 k) with the eight 128-wide rows at `wrows`:
// thread (row = tid & 31, out = (tid >> 5) & 7, half = tid >> 8) sums its 64 k; the halves meet in `part`.
__device__ __forceinline__ void pf_dot8(const float* __restrict__ img, const float* __restrict__ wrows, float* __restrict__ part, int tid) {
  const int row = tid & 31, out = (tid >> 5) & 7, half = tid >> 8;
  const int sw = (row >> 1) & 7;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {           // the half's two 32-k images
    const float* im = img + (2 * half + t) * PF_IMG + row * 32;
    const float* wr = wrows + out * 128 + (2 * half + t) * 32;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(im + ((p ^ sw) << 2));
      const f32x4 v = *reinterpret_cast<const f32x4*>(wr + 4 * p);
#pragma unroll
      for (int q = 0; q < 4; ++q) s = fmaf(x[q], v[q], s);
    }
  }
  part[(half * 32 + row) * 8 + out] = s;
}

__global__ __launch_bounds__(PF_THREADS) void k_policy_fwd(const PolicyFwdArgs a) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  float* P = smem;
  float* Q = P + PF_P;
  float* ring0 = Q + PF_Q;
  float* w3s = ring0 + PF_RING;          // [8][128] last env layer
  float* hws = w3s + 8 * 128;            // [8][128] head rows: mu 0 .. act - 1, value at row 7
  float* part = hws + 8 * 128;           // [2][32][8]
  float* lat = part + 2 * 32 * 8;        // [32][8]: latent, later mu
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = (int)blockIdx.x & 1, blk = (int)blockIdx.x >> 1;     // net (0 actor, 1 critic), row block
  const int m0 = blk * PF_ROWS;
  PfWave w;
  w.ring = ring0 + wave * (PF_NS * PF_CH);
  w.lane = lane; w.wave = wave; w.l31 = lane & 31; w.h = lane >> 5;
  pf_img_bases(w.l31, w.h, w.ib);
  w.aoff = w.l31 * 32; w.asw = (w.l31 >> 1) & 7;
  w.boff = w.l31 * 16; w.bsw = (w.l31 >> 2) & 3;
  const float* tb1 = a.tb1 + z * a.ac_block;
  const float* tW2 = a.tW2 + z * a.ac_block;
  const float* tb2 = a.tb2 + z * a.ac_block;
  const float* tW3 = a.tW3 + z * a.ac_block;
  const float* tb3 = a.tb3 + z * a.ac_block;
  const float* w1p = a.w1p + (long long)z * 512 * 32;

  // ---- everything a later phase would wait a round trip for, requested now
  const int c31 = 32 * wave + w.l31;
  float be1[1] = {a.eb1[c31]};
  float be2[1] = {a.eb2[min(c31, 127)]};
  float bt1[2] = {tb1[c31], tb1[256 + c31]};
  float bt2[1] = {tb2[c31]};
  float bt3[1] = {tb3[min(c31, 127)]};
  const int xrow = tid >> 4, xc = 2 * (tid & 15);                    // this thread's two xcat elements
  const float2 xv = *reinterpret_cast<const float2*>(a.xcat + (long long)min(m0 + xrow, a.rows - 1) * a.ldx + xc);
  float w3v[2], hwv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + PF_THREADS * i;                               // 0 .. 1023
    w3v[i] = a.eW3[e];
    const int r = e >> 7, k = e & 127;
    hwv[i] = r < a.act ? a.Wmu[r * 128 + k] : (r == 7 ? a.Wv[k] : 0.f);
  }
  {  // the input block: two [32][32] images, four 1 KB pieces each, one per wave
    const int k = wave >> 2, i = wave & 3;
    const int m = 8 * i + (lane >> 3);
    const int row = min(m0 + m, a.rows - 1);
    dma16(a.priv + (long long)row * a.ldp + k * 32 + 4 * ((lane & 7) ^ ((m >> 1) & 7)), Q + k * PF_IMG + 256 * i);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) { w3s[tid + PF_THREADS * i] = w3v[i]; hws[tid + PF_THREADS * i] = hwv[i]; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- env_mlp: 64 -> 256 (Q -> P), 256 -> 128 (P -> Q), 128 -> 8 (Q -> lat)
  pf_layer<64, 256, 1>(w, Q, a.eW1, 64, be1, P);
  __syncthreads();
  pf_layer<256, 128, 1>(w, P, a.eW2, 256, be2, Q);
  __syncthreads();
  pf_dot8(Q, w3s, part, tid);
  __syncthreads();
  if (tid < 256) {
    const int row = tid & 31, out = tid >> 5;
    lat[row * 8 + out] = fast_tanh((part[row * 8 + out] + part[(32 + row) * 8 + out]) + a.eb3[out]);
  }
  __syncthreads();
  // ---- xcat = [obs_n | latent | 0] as ONE image in P
  {
    const int sw = (xrow >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = xc + q;
      const float v = (c >= a.obs && c < a.obs + 8) ? lat[xrow * 8 + (c - a.obs)] : (q ? xv.y : xv.x);
      P[xrow * 32 + (((c >> 2) ^ sw) << 2) + (c & 3)] = v;
    }
  }
  __syncthreads();
  // ---- trunk of net z: 32 -> 512 (P -> Q), 512 -> 256 (Q -> P), 256 -> 128 (P -> Q)
  pf_layer<32, 512, 2>(w, P, w1p, 32, bt1, Q);
  __syncthreads();
  pf_layer<512, 256, 1>(w, Q, tW2, 512, bt2, P);
  __syncthreads();
  pf_layer<256, 128, 1>(w, P, tW3, 256, bt3, Q);
  __syncthreads();
  // ---- heads: rows 0 .. act - 1 of hws = mu, row 7 = value
  pf_dot8(Q, hws, part, tid);
  __syncthreads();
  if (z == 0) {
    // thread (row, action q): k_heads_act_store's per-action arithmetic; the neglogp terms meet in action order
    const int row = tid & 31, q = tid >> 5, grow = m0 + row;
    if (tid < 256 && q < a.act && grow < a.rows) {
      const float m = (part[row * 8 + q] + part[(32 + row) * 8 + q]) + a.bmu[q];
      const float sig = expf(m * 0.f + a.logstd[q]);
      const long long ia = (long long)grow * a.act + q;
      const float av = m + sig * a.noise[ia];
      const float x = av - m;
      lat[row * 8 + q] = ((x * x) / (2.0f * (sig * sig)) + logf(sig)) + PF_LOG_SQRT_2PI;
      a.actions_t[ia] = av;
      a.mus_t[ia] = m;
      a.sigmas_t[ia] = sig;
      a.actions_clamped[ia] = fminf(fmaxf(av, -1.0f), 1.0f);
    }
    __syncthreads();
    if (tid < 32 && m0 + tid < a.rows) {
      float nlp = 0.f;
      for (int k = 0; k < a.act; ++k) nlp += lat[tid * 8 + k];
      a.nlp_t[m0 + tid] = nlp;
    }
  } else if (tid < 32 && m0 + tid < a.rows) {
    const int row = tid, grow = m0 + row;
    float v = (part[row * 8 + 7] + part[(32 + row) * 8 + 7]) + a.bv[0];
    if (a.rms_value) {
      const float vm = (float)a.rms_value[0], vd = sqrtf((float)a.rms_value[1] + a.eps);
      v = vd * fminf(fmaxf(v, -5.0f), 5.0f) + vm;
    }
    a.values_t[grow] = v;
    a.values_out[grow] = v;
  }
}

static hipError_t policy_forward(const PolicyFwdArgs& a, hipStream_t s) {
  if (a.rows < 1 || !aligned16(a.priv) || !aligned16(a.xcat) || (a.ldp & 3) || a.ldp < 64 || a.ldx != 32 || !aligned16(a.eW1) ||
      !aligned16(a.eW2) || !aligned16(a.w1p) || !aligned16(a.tW2) || !aligned16(a.tW3) || (a.ac_block & 3))
    return hipErrorInvalidValue;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)k_policy_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * PF_LDS_FLOATS));
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int blocks = (a.rows + PF_ROWS - 1) / PF_ROWS;
  const double macs = 64.0 * 256 + 256.0 * 128 + 128.0 * 8 + 2.0 * ((a.obs + 8) * 512.0 + 512.0 * 256 + 256.0 * 128) + 128.0 * (a.act + 1);
  ProfScope ps(PC_POLICY_FWD, s, 2.0 * macs * a.rows, 4.0 * a.rows * (64.0 + 32 + 4 * a.act + 3) + 4.0 * 404501);
  IGI_LAUNCH(k_policy_fwd, dim3(2 * blocks), dim3(PF_THREADS), sizeof(float) * PF_LDS_FLOATS, s, a);
  return hipGetLastError();
}

}  // namespace igi
