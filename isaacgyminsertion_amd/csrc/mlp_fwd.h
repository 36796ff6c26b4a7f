// A chain of Linear (+ fused activation) layers, FORWARD, as one launch (the student's small MLPs: lin encoder
// 15 -> 64 -> 32, point-cloud compress 512 -> 64 -> 32, decoder output stack 96 -> 32 -> 256 -> 128 -> 64 -> 32 + the action
// head 32 -> 6, tact.py:137-212, 337-339, 367-369, 407-410).  As one launch per Linear these layers were 10 of the student
// step's launches at ~7 us each for ~1 us of arithmetic (profiles/r04_student_c4_kernel_stats.csv).
//
// One workgroup (8 waves: one 32-column output tile each) carries 32 rows through every layer: the hidden activations never leave LDS (two ping-pong
// [32][256] images) except for the copy each layer writes out once for the backward pass (igi_mlp_backward reads y[l]);
// the weights arrive through LDS in 64-wide k-chunks; the first layer's input is streamed the same way, so its width
// is not limited by LDS.  Outputs <= 256 per layer.
//
// Arithmetic = the per-layer launches', bit for bit: every output element is ONE fmaf chain on v_mfma_f32_32x32x2_f32
// -- k-ordered in pairs (0, 4), (1, 5), (2, 6), (3, 7) inside every group of eight where igi_linear_forward would have
// run the LDS-DMA kernel (dma_eligible: K % 32 == 0, aligned operands, >= 4 rows), in plain pairs (0, 1), (2, 3), ...
// where it would have run the generic kernel (gemm_f32.h) -- followed by the same bias + activation expression
// (tests/test_gpu_linear.py compares the two paths with torch.equal).
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_dma.h"
#include "gemm_f32.h"
#include "linear.h"

namespace igi {

constexpr int MF_MAX_LAYERS = 8;
constexpr int MF_ROWS = 32;
constexpr int MF_WAVES = 8;
constexpr int MF_THREADS = MF_WAVES * 64;
constexpr int MF_MAXW = 256;                 // widest layer OUTPUT: one 32-column tile per wave
constexpr int MF_KC = 64;                    // k-chunk staged per step
constexpr int MF_MAX_STEPS = 40;             // k-chunks of the whole chain
constexpr int MF_ALD = MF_MAXW + 4;          // row pitch of an activation image: 16-byte reads of 16 rows hit 16 bank groups
constexpr int MF_WLD = MF_KC + 4;            // row pitch of the weight / input chunk images
constexpr int MF_LDS_FLOATS = 2 * MF_ROWS * MF_ALD + MF_MAXW * MF_WLD + MF_ROWS * MF_WLD;

struct MlpFwdArgs {
  const float* x; int ldx; long long rows; int n, nsteps;
  int dims[MF_MAX_LAYERS + 1], acts[MF_MAX_LAYERS], dma_order[MF_MAX_LAYERS], vec_w[MF_MAX_LAYERS], vec_y[MF_MAX_LAYERS];
  int vec_x;
  const float* W[MF_MAX_LAYERS];
  const float* b[MF_MAX_LAYERS];
  float* y[MF_MAX_LAYERS];
  int ldy[MF_MAX_LAYERS];
  unsigned char step_layer[MF_MAX_STEPS];    // k-chunk s belongs to layer step_layer[s], starts at k = 64 * step_chunk[s]
  unsigned char step_chunk[MF_MAX_STEPS];
};

__global__ __launch_bounds__(MF_THREADS) void k_mlp_fwd(const MlpFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act0 = smem;
  float* act1 = act0 + MF_ROWS * MF_ALD;
  float* wch = act1 + MF_ROWS * MF_ALD;       // [<= 256 outputs][chunk]
  float* xch = wch + MF_MAXW * MF_WLD;        // [32 rows][chunk] of the first layer's input
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const long long r0 = (long long)blockIdx.x * MF_ROWS;
  const int nrows = (int)min((long long)MF_ROWS, a.rows - r0);
  const int lq = tid & 15, lr = tid >> 4;     // loader: float4 column of the chunk, row (+ 32 i)

  // One k-chunk of one layer = weights W[n][k0 .. k0 + kc) for every output n (+ the input rows for layer 0), fetched
  // into registers a step AHEAD (the loads fly under the previous chunk's MFMAs) and parked in LDS between two barriers.
  // (two steps ahead, two register sets: one chunk's 32 MFMAs per wave are ~1 us, a load from L2 with every workgroup
  // pulling the same weights is 1.5 - 2.5 us)
  float4 wrs[2][8], xrs[2];
  auto fetch = [&](int s, float4 (&wr)[8], float4& xr) {
    const int l = a.step_layer[s], K = a.dims[l], N = a.dims[l + 1];
    const int k0 = MF_KC * a.step_chunk[s], kc = min(MF_KC, K - k0);
    const float* __restrict__ W = a.W[l];
    const int col = 4 * lq;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = lr + 32 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && col < kc) {
        const float* p = W + (long long)n * K + k0 + col;
        if (a.vec_w[l] && col + 3 < kc) v = *reinterpret_cast<const float4*>(p);
        else {
          v.x = p[0];
          if (col + 1 < kc) v.y = p[1];
          if (col + 2 < kc) v.z = p[2];
          if (col + 3 < kc) v.w = p[3];
        }
      }
      wr[i] = v;
    }
    xr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l == 0 && lr < nrows && col < kc) {
      const float* p = a.x + (r0 + lr) * a.ldx + k0 + col;
      if (a.vec_x && col + 3 < kc) xr = *reinterpret_cast<const float4*>(p);
      else {
        xr.x = p[0];
        if (col + 1 < kc) xr.y = p[1];
        if (col + 2 < kc) xr.z = p[2];
        if (col + 3 < kc) xr.w = p[3];
      }
    }
  };
  fetch(0, wrs[0], xrs[0]);
  if (a.nsteps > 1) fetch(1, wrs[1], xrs[1]);
  f32x16 acc;
  auto step = [&](int s, float4 (&wr)[8], float4& xr) {
    const int l = a.step_layer[s], K = a.dims[l], N = a.dims[l + 1];
    const int k0 = MF_KC * a.step_chunk[s], kc = min(MF_KC, K - k0);
    const int kcp = (kc + 7) & ~7;              // zero-filled up to whole groups of eight (adding 0 * 0 is exact)
    const int ntiles = (N + 31) >> 5;
    const float* src = (l & 1) ? act0 : act1;   // layer l reads image (l - 1) & 1 and writes image l & 1
    float* dst = (l & 1) ? act1 : act0;
    __syncthreads();                            // the previous chunk's reads of wch / xch are done
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (lr + 32 * i < 32 * ntiles) *reinterpret_cast<float4*>(wch + (lr + 32 * i) * MF_WLD + 4 * lq) = wr[i];
    if (l == 0) *reinterpret_cast<float4*>(xch + lr * MF_WLD + 4 * lq) = xr;
    __syncthreads();
    if (s + 2 < a.nsteps) fetch(s + 2, wr, xr);      // into the set that has just been parked
    if (k0 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    if (wave < ntiles) {
      const float* ap = (l == 0) ? xch + l31 * MF_WLD : src + l31 * MF_ALD + k0;
      const float* bp = wch + (32 * wave + l31) * MF_WLD;
      if (a.dma_order[l]) {
        // LDS-DMA kernel's order: lane half h feeds k = 8 c + 4 h + j to step j of group c
        for (int c = 0; c < kcp; c += 8) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(ap + c + 4 * h);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + c + 4 * h);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
        }
      } else {
        // generic kernel's order: step s multiplies k = 2 s + h
        for (int c = 0; c < kcp; c += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[c + h], bp[c + h], acc, 0, 0, 0);
      }
    }
    if (k0 + kc < K) return;
    // ---- the layer is complete: bias + activation into its image (accumulator layout: column l31, rows
    //      (r & 3) + 8 (r >> 2) + 4 h), then the copy the backward pass reads, whole rows, 16 bytes per lane where possible
    if (wave < ntiles) {
      const int act = a.acts[l];
      const int n = 32 * wave + l31;
      const float bv = (a.b[l] && n < N) ? a.b[l][n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (act == LIN_TANH) v = fast_tanh(v + bv);
        else if (act == LIN_RELU) v = fmaxf(v + bv, 0.f);
        else if (a.b[l]) v = v + bv;
        dst[((r & 3) + 8 * (r >> 2) + 4 * h) * MF_ALD + n] = v;
      }
    }
    __syncthreads();
    float* __restrict__ y = a.y[l];
    const int ldy = a.ldy[l];
    if (a.vec_y[l]) {
      const int c4 = N >> 2;
      for (int q = lq; q < c4; q += 16)
        if (lr < nrows)
          *reinterpret_cast<float4*>(y + (r0 + lr) * ldy + 4 * q) = *reinterpret_cast<const float4*>(dst + lr * MF_ALD + 4 * q);
    } else {
      for (int c = lq; c < N; c += 16)
        if (lr < nrows) y[(r0 + lr) * ldy + c] = dst[lr * MF_ALD + c];
    }
  };
  for (int s = 0; s < a.nsteps; s += 2) {
    step(s, wrs[0], xrs[0]);
    if (s + 1 < a.nsteps) step(s + 1, wrs[1], xrs[1]);
  }
}

// ys[l]: [rows][dims[l + 1]] with row pitch ldy[l] (NULL pitches: dense).  Returns IGI_E_UNSUPPORTED for chains this
// kernel does not take (a layer wider than 256 outputs, the bf16-input mode): the caller runs the layers one by one.
static int mlp_forward(const float* x, int ldx, long long rows, int n, const int32_t* dims, const int32_t* acts,
                       const float* const* weight, const float* const* bias, float* const* ys, const int32_t* ldys,
                       hipStream_t s) {
  if (!x || !dims || !acts || !weight || !bias || !ys || n < 1 || n > MF_MAX_LAYERS || rows < 0 || rows > (1LL << 30))
    return IGI_E_BADARG;
  if (rows == 0) return 0;
  if (bf16_mode()) return IGI_E_UNSUPPORTED;
  MlpFwdArgs a;
  a.x = x; a.ldx = ldx; a.rows = rows; a.n = n;
  if (ldx < dims[0]) return IGI_E_BADARG;
  a.vec_x = aligned16(x) && (ldx & 3) == 0;
  for (int l = 0; l <= n; ++l) {
    if (dims[l] < 1) return IGI_E_BADARG;
    a.dims[l] = dims[l];
  }
  for (int l = 0; l < n; ++l) {
    const int K = dims[l], N = dims[l + 1];
    if (N > MF_MAXW) return IGI_E_UNSUPPORTED;
    if (acts[l] < 0 || acts[l] > 2 || !weight[l] || !ys[l] || (acts[l] != LIN_NONE && !bias[l])) return IGI_E_BADARG;
    a.acts[l] = acts[l];
    a.W[l] = weight[l]; a.b[l] = bias[l]; a.y[l] = ys[l];
    a.ldy[l] = ldys ? ldys[l] : N;
    if (a.ldy[l] < N) return IGI_E_BADARG;
    a.vec_w[l] = aligned16(weight[l]) && (K & 3) == 0;
    a.vec_y[l] = aligned16(ys[l]) && (a.ldy[l] & 3) == 0 && (N & 3) == 0;
    // which chain order igi_linear_forward's launch of this layer would have used: the very predicate gemm() applies
    GemmArgs g;
    g.A = l == 0 ? x : ys[l - 1]; g.lda = l == 0 ? ldx : a.ldy[l - 1];
    g.B = weight[l]; g.ldb = K;
    g.bias = bias[l];
    g.M = (int)rows; g.N = N; g.K = K;
    g.C = ys[l]; g.ldc = a.ldy[l];
    a.dma_order[l] = dma_eligible(g, true, true) ? 1 : 0;
  }
  a.nsteps = 0;
  for (int l = 0; l < n; ++l)
    for (int c = 0; c * MF_KC < dims[l]; ++c) {
      if (a.nsteps >= MF_MAX_STEPS || c > 255) return IGI_E_UNSUPPORTED;
      a.step_layer[a.nsteps] = (unsigned char)l;
      a.step_chunk[a.nsteps] = (unsigned char)c;
      ++a.nsteps;
    }
  static bool attr = false;
  if (!attr) {
    IGI_HIP_TRY(hipFuncSetAttribute((const void*)k_mlp_fwd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(sizeof(float) * MF_LDS_FLOATS)));
    attr = true;
  }
  double fl = 0, by = 4.0 * rows * dims[0];
  for (int l = 0; l < n; ++l) { fl += 2.0 * rows * dims[l] * (double)dims[l + 1]; by += 4.0 * (rows + dims[l]) * (double)dims[l + 1]; }
  ProfScope ps(PC_MLP_FWD, s, fl, by);
  IGI_LAUNCH(k_mlp_fwd, dim3((unsigned)((rows + MF_ROWS - 1) / MF_ROWS)), dim3(MF_THREADS), sizeof(float) * MF_LDS_FLOATS, s, a);
  return (int)hipGetLastError();
}

}  // namespace igi
