// Segmented point-cloud encoder (algo/models/transformer/pointnets.py:12-42):
//   y[b][c] = max_n ( W2 . gelu(W1 . x[b][n] + b1) + b2 )[c],  W1 (64,3), W2 (256,64), erf-GELU;
// forward (+ argmax) and backward.
//
// The reference materialises (B,N,64) and (B,N,256) activations in HBM (512 KB per object per
// sample at N = 400).  Here one persistent workgroup streams samples: each wave keeps its W2 fragments
// in registers, the 64-wide hidden rows of 32 points are produced on the VALU straight into a
// double-buffered LDS MFMA-operand tile, the 32x256 second-layer block comes from exact-fp32 MFMA
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
constexpr int PN_BWD_BLOCKS = 256;   // backward: one workgroup per CU (85 KB of LDS each), one 67 KB partial per workgroup
constexpr int PN_LD = PN_OUT + 1;
constexpr int PN_WLD = PN_H + 4;      // row pitch of the forward's W2 staging rows in LDS
#ifndef PN_STAGE_W2
#define PN_STAGE_W2 1               // 0: the lanes' W2 rows straight from global memory (A/B builds)
#endif

// erf-GELU.  Phi(x) = 0.5 * erfc(-x / sqrt 2) with erfc(z) = 2^(z * Q(z)) on [0, 4] (Q: degree 9, fitted to
// -log2(erfc(z)) / z with the error weighted by erfc(z) * z; erfc(4) = 1.5e-8 is below half an ulp of 1, so |z| is
// clamped there):  Phi = 1 - e/2 for x >= 0 and e/2 for x < 0 -- no cancellation on the negative side.  In fp32 the
// absolute error of Phi is 7.3e-8 and of x * Phi 6.1e-7 over |x| <= 9, the same as the reference's own
// 0.5 * x * (1 + erff(x / sqrt 2)) evaluated in fp32 (6.8e-7: the rounding of 1 + erf) -- measured against fp64 in
// tools/probes/gelu_fit.py, which also regenerates the coefficients.  One range, no branch, one v_exp_f32: 21 vector
// instructions per value against ~45 for erff + the formula, and the polynomial runs two values per instruction
// (v_pk_fma_f32) in the forward kernel, whose hidden-layer production is VALU time the fp32 MFMA cannot overlap.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pn_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f32x2 pn_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <typename T>
__device__ __forceinline__ T gelu_q(T z) {   // -log2(e) folded in: erfc(z) = exp2(z * gelu_q(z))
  T p = (T)(-1.909314714e-06f);
  p = pn_fma(p, z, (T)(3.273919096e-05f));
  p = pn_fma(p, z, (T)(-2.497497791e-04f));
  p = pn_fma(p, z, (T)(1.086700535e-03f));
  p = pn_fma(p, z, (T)(-2.619071391e-03f));
  p = pn_fma(p, z, (T)(3.822097281e-04f));
  p = pn_fma(p, z, (T)(2.757399108e-02f));
  p = pn_fma(p, z, (T)(-1.482662003e-01f));
  p = pn_fma(p, z, (T)(-9.184483787e-01f));
  p = pn_fma(p, z, (T)(-1.627907028e+00f));
  return p;
}
// Beyond the fitted range the exponent keeps falling with the UNCLAMPED argument, zu * Q(4): the tail 2^(-6.49 zu) is
// continuous at z = 4 and underflows to zero, so x * Phi(x) -> -0 for very negative x as in the reference (with the
// clamped argument Phi stayed at erfc(4) / 2 = 7.7e-9 and x * Phi grew linearly: -7.7e-5 at x = -1e4, found by
// tests/test_gpu_pointnet.py::test_pointnet_gelu_sweep_against_fp64).  Same instruction count; bit-identical for |x| <= 5.65.
__device__ __forceinline__ float gelu_cdf(float x) {
  const float zu = fabsf(x) * 0.70710678118654752440f;
  const float z = fminf(zu, 4.0f);
  const float s = 0.5f * __builtin_amdgcn_exp2f(zu * gelu_q(z));
  return x >= 0.f ? 1.0f - s : s;
}
__device__ __forceinline__ f32x2 gelu_cdf(f32x2 x) {
  const f32x2 zu = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const f32x2 z = __builtin_elementwise_min(zu, (f32x2)(4.0f));
  const f32x2 a = zu * gelu_q(z);
  f32x2 s;
  s.x = __builtin_amdgcn_exp2f(a.x); s.y = __builtin_amdgcn_exp2f(a.y);
  s = s * 0.5f;
  const f32x2 t = 1.0f - s;
  f32x2 r;
  r.x = x.x >= 0.f ? t.x : s.x; r.y = x.y >= 0.f ? t.y : s.y;
  return r;
}
__device__ __forceinline__ float gelu_erf(float x) { return x * gelu_cdf(x); }
__device__ __forceinline__ float gelu_erf_grad(float x) {   // Phi(x) + x * phi(x)
  return gelu_cdf(x) + x * (__builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f) * 0.39894228040143267794f);
}

// Several objects of one cloud tensor in ONE launch (round 6): the plug and the socket encoders of the student
// (tact.py:542-555: obs_pcl[:, :400] and obs_pcl[:, 400:800], two PointNets with their own weights) used to be two forward
// and two backward launches per optimizer step, each with its ~23 us fill (W2 fragments into registers, first tiles, drain).
// A workgroup serves ONE object (wave-uniform index into this table, in the kernarg segment): its parameters, the float
// offset of the object's first point inside a cloud row and its point count; outputs go straight into the concatenated
// (B, objects * 256) encoding (and its arg-max), which is what compress_pcl_enc reads -- no concatenation launch either.
constexpr int PN_MAX_OBJ = 4;
struct PnObjs {
  const float* params[PN_MAX_OBJ];
  int x_off[PN_MAX_OBJ], N[PN_MAX_OBJ];
  int nobj, bpo;                      // objects; workgroups per object
};

// Forward.  One workgroup streams clouds; wave w owns output columns 64w..64w+63 (two 32-column MFMA blocks).
//  * the wave's W2 fragments (64 values per lane) and the thread's first-layer rows (8 hidden units: 32 values) live in
//    registers for the whole kernel -- the k-loop reads only the hidden tile from LDS (one ds_read per two MFMAs);
//  * the hidden tile is double-buffered: the rows of tile t+1 are produced before the MFMAs of tile t are issued and one
//    barrier per tile orders both buffers;
//  * the running maximum is kept per accumulator slot (value + tile number: compare + two selects per element) and the
//    slots are merged once per cloud -- the first maximum in point order, as before.
__device__ __forceinline__ void pn_produce(float px, float py, float pz, const float4* __restrict__ w1 /* + kq*8 */,
                                           float* __restrict__ hs /* + kq*8*32 + m_h */) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    // hidden units j, j+1 as one packed pair: (W1[k][0], W1[k+1][0]), (..[1]), (..[2]), (b1[k], b1[k+1]) -- two
    // broadcast LDS reads
    const float4 wa = w1[j], wb = w1[j + 1];
    const f32x2 wx = {wa.x, wa.y}, wy = {wa.z, wa.w}, wz = {wb.x, wb.y}, bb = {wb.z, wb.w};
    const f32x2 pre = pn_fma(wz, (f32x2)(pz), pn_fma(wy, (f32x2)(py), pn_fma(wx, (f32x2)(px), bb)));
    const f32x2 g = pre * gelu_cdf(pre);
    hs[j * 32] = g.x;
    hs[(j + 1) * 32] = g.y;
  }
}

// the workgroup's tile stream: tile 0..ntiles-1 of cloud blockIdx.x, then of cloud blockIdx.x + gridDim.x, ...  The
// coordinates of a tile are loaded TWO tiles before its MFMAs (one before its hidden rows are produced): issued right
// before the production they would stall it for a full memory latency on every tile.
struct PnStream {
  const float* base;   // cloud of the next tile to fetch
  int cloud, tile, stride;   // stride: workgroups that share this object's clouds
};
__device__ __forceinline__ void pn_fetch(PnStream& st, int B, int N, int ntiles, int m_h, long long cloud_step,
                                         float& px, float& py, float& pz) {
  const int n = st.tile * 32 + m_h;
  px = 0.f; py = 0.f; pz = 0.f;
  if (st.cloud < B && n < N) { px = st.base[n * 3]; py = st.base[n * 3 + 1]; pz = st.base[n * 3 + 2]; }
  if (++st.tile == ntiles) { st.tile = 0; st.cloud += st.stride; st.base += cloud_step; }
}

// COLMAX (EXPERIMENT, IGI_PN_COLMAX=1, forward timing only -- VERDICT round 5 item 4(ii)): the running maximum per LANE
// and accumulator (a v_max3 tree over the lane's 16 rows of a tile, one compare, two selects: 22 instead of 96 vector
// instructions per tile) with only the TILE of the maximum recorded; the arg-max written is tile * 32 (the row inside the tile
// would have to be resolved afterwards), so the backward must not be run on it.  It measures the most that variant could
// gain in the forward before its resolve pass is paid for.
template <bool COLMAX>
__global__ __launch_bounds__(256, 2) void k_pointnet_fwd(const float* __restrict__ x, long long xpitch, int B,
                                                         const PnObjs o, float* __restrict__ y, long long ypitch,
                                                         int* __restrict__ argmax) {
  __shared__ __attribute__((aligned(16))) float Hs[2][PN_H * 32];   // [buffer][64 k][32 m]
  __shared__ float4 W1s[PN_H];
  // which object, which of its workgroups (wave-uniform: the table is read with scalar loads)
  const int obj = __builtin_amdgcn_readfirstlane((int)blockIdx.x / o.bpo), bid = (int)blockIdx.x - obj * o.bpo, nbl = o.bpo;
  const float* __restrict__ params = o.params[obj];
  const int N = o.N[obj];
  x += o.x_off[obj];
  y += obj * PN_OUT;
  if (argmax) argmax += obj * PN_OUT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int c0 = wave * 64 + l31, c1 = c0 + 32;
  float wb0[PN_H / 2], wb1[PN_H / 2];   // B operands of the 32 k-steps: W2[c][2*k2 + h]
  if (PN_STAGE_W2) {
    // the wave's 64 W2 rows through LDS, 32 rows per pass: eight coalesced 16-byte loads per lane and pass (1 KB per
    // instruction), then each lane reads its row back.  Read straight from global memory a lane's row is 256 bytes from
    // its neighbour's: every load instruction touched 64 cache lines for 16 useful bytes each, 4096 line requests per
    // wave through the texture path -- most of the kernel's ~25 us fixed cost at 4 clouds per workgroup.
    // (round 6: the staging region is the wave's own and a wave's LDS operations execute in order, so the passes need a
    //  wave-level fence only -- with a workgroup barrier between pass 1's loads and its LDS writes the compiler parked the
    //  eight loaded values in scratch, 144 bytes per lane: tests/test_host_api.py now asks for zero)
    __shared__ __attribute__((aligned(16))) float Wst[4][32 * PN_WLD];
    float* ws = Wst[wave];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const float4* src = reinterpret_cast<const float4*>(params + PN_OW2 + (wave * 64 + 32 * pass) * PN_H);
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[j * 64 + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = j * 64 + lane;
        *reinterpret_cast<float4*>(ws + (q >> 4) * PN_WLD + 4 * (q & 15)) = v[j];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float4* row = reinterpret_cast<const float4*>(ws + l31 * PN_WLD);
#pragma unroll
      for (int j = 0; j < PN_H / 4; ++j) {
        const float4 r = row[j];
        if (pass == 0) { wb0[2 * j] = h ? r.y : r.x; wb0[2 * j + 1] = h ? r.w : r.z; }
        else { wb1[2 * j] = h ? r.y : r.x; wb1[2 * j + 1] = h ? r.w : r.z; }
      }
      // (the row reads above are in the wave's LDS queue ahead of the next pass's writes)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  } else {
    // a lane's two W2 rows as 16-byte loads (the even or the odd elements are kept)
    const float4* r0 = reinterpret_cast<const float4*>(params + PN_OW2 + c0 * PN_H);
    const float4* r1 = reinterpret_cast<const float4*>(params + PN_OW2 + c1 * PN_H);
#pragma unroll
    for (int j = 0; j < PN_H / 4; ++j) {
      const float4 v0 = r0[j], v1 = r1[j];
      wb0[2 * j] = h ? v0.y : v0.x; wb0[2 * j + 1] = h ? v0.w : v0.z;
      wb1[2 * j] = h ? v1.y : v1.x; wb1[2 * j + 1] = h ? v1.w : v1.z;
    }
  }
  if (tid < PN_H) {   // pair layout (see pn_produce): entry 2q = (x_k, x_k+1, y_k, y_k+1), 2q + 1 = (z_k, z_k+1, b_k, b_k+1), k = 2q
    const int k = tid & ~1;
    const float* wk = params + PN_OW1 + k * 3;
    W1s[tid] = (tid & 1) ? make_float4(wk[2], wk[5], params[PN_OB1 + k], params[PN_OB1 + k + 1])
                         : make_float4(wk[0], wk[3], wk[1], wk[4]);
  }
  const int m_h = tid & 31, kq = tid >> 5;  // hidden-tile production: point m_h, hidden units kq*8..+7
  const float b2_0 = params[PN_OB2 + c0], b2_1 = params[PN_OB2 + c1];
  const int ntiles = (N + 31) / 32;         // <= 256: the tile number of a slot's maximum is an 8-bit field (launcher)
  const bool ragged = (N & 31) != 0;
  const int hoff = kq * 8 * 32 + m_h;
  const float4* w1 = W1s + kq * 8;
  int buf = 0;
  const long long cloud_step = (long long)nbl * xpitch;   // xpitch: floats between clouds (>= 3 N: a slice of a wider cloud tensor)
  PnStream st = {x + (long long)bid * xpitch, bid, 0, nbl};
  float px, py, pz;
  __syncthreads();
  pn_fetch(st, B, N, ntiles, m_h, cloud_step, px, py, pz);
  pn_produce(px, py, pz, w1, &Hs[0][hoff]);
  pn_fetch(st, B, N, ntiles, m_h, cloud_step, px, py, pz);   // tile 1: produced during tile 0's iteration
  __syncthreads();
  for (int b = bid; b < B; b += nbl) {
    float bv0[16], bv1[16];
    unsigned bt0[4] = {0u, 0u, 0u, 0u}, bt1[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 16; ++r) { bv0[r] = -INFINITY; bv1[r] = -INFINITY; }
    float cm0 = -INFINITY, cm1 = -INFINITY;   // COLMAX: the lane's running maximum per accumulator and its tile
    int ct0 = 0, ct1 = 0;
    for (int rt = 0; rt < ntiles; ++rt) {
      // the whole A fragment of this tile first (32 reads in flight; read next to their MFMAs they are reloaded into
      // one register pair and every fourth MFMA waits a full LDS latency), the production below hides the latency
      float a[PN_H / 2];
      {
        const float* hb = &Hs[buf][h * 32 + l31];
#pragma unroll
        for (int k2 = 0; k2 < PN_H / 2; ++k2) a[k2] = hb[k2 * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      // hidden rows of the stream's next tile (zeros past the last cloud: never read), then the fetch for the one after
      pn_produce(px, py, pz, w1, &Hs[buf ^ 1][hoff]);
      pn_fetch(st, B, N, ntiles, m_h, cloud_step, px, py, pz);
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k2 = 0; k2 < PN_H / 2; ++k2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2], wb0[k2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2], wb1[k2], acc1, 0, 0, 0);
      }
      if (ragged && rt == ntiles - 1) {   // rows past the cloud never win: -inf > x is false
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool in = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < N;
          acc0[r] = in ? acc0[r] : -INFINITY;
          acc1[r] = in ? acc1[r] : -INFINITY;
        }
      }
      if constexpr (COLMAX) {
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        auto tree = [](const f32x16& a) {
          float m = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]);
#pragma unroll
          for (int r = 3; r + 1 < 16; r += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, a[r]), a[r + 1]);
          return __builtin_fmaxf(m, a[15]);
        };
        const float m0 = tree(acc0), m1 = tree(acc1);
        const bool u0 = m0 > cm0, u1 = m1 > cm1;     // strict: the first tile keeps a tie
        cm0 = u0 ? m0 : cm0; ct0 = u0 ? rt : ct0;
        cm1 = u1 ? m1 : cm1; ct1 = u1 ? rt : ct1;
      } else {
        // strict '>' keeps the first tile of a slot's maximum.  Three instructions per element -- compare, select the
        // value, select the tile number into its byte (SDWA: the other three bytes of the word are preserved); the
        // compiler's form (compare, bit-field insert, two selects) takes four and keeps 32 compare masks in SGPR pairs,
        // which it then spills.  VALU
        // time is not hidden here: the fp32 MFMA runs on the SIMD's FMA lanes (MFMA-busy + VALU-active cycles add up
        // to the kernel's duration in the SQ counters).
        const unsigned rtv = (unsigned)rt;
        // the accumulators come out of the matrix pipe: 18 wait states between the last MFMA and a vector read
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#define PN_UPD3(acc, bv, btw, BYTE)                                                                                  \
  asm volatile("v_cmp_gt_f32 vcc, %2, %0\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t"                                       \
               "v_cndmask_b32_sdwa %1, %1, %3, vcc dst_sel:" BYTE " dst_unused:UNUSED_PRESERVE src0_sel:" BYTE       \
               " src1_sel:BYTE_0"                                                                                    \
               : "+v"(bv), "+v"(btw)                                                                                 \
               : "v"(acc), "v"(rtv)                                                                                  \
               : "vcc")
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          PN_UPD3(acc0[4 * q + 0], bv0[4 * q + 0], bt0[q], "BYTE_0"); PN_UPD3(acc1[4 * q + 0], bv1[4 * q + 0], bt1[q], "BYTE_0");
          PN_UPD3(acc0[4 * q + 1], bv0[4 * q + 1], bt0[q], "BYTE_1"); PN_UPD3(acc1[4 * q + 1], bv1[4 * q + 1], bt1[q], "BYTE_1");
          PN_UPD3(acc0[4 * q + 2], bv0[4 * q + 2], bt0[q], "BYTE_2"); PN_UPD3(acc1[4 * q + 2], bv1[4 * q + 2], bt1[q], "BYTE_2");
          PN_UPD3(acc0[4 * q + 3], bv0[4 * q + 3], bt0[q], "BYTE_3"); PN_UPD3(acc1[4 * q + 3], bv1[4 * q + 3], bt1[q], "BYTE_3");
        }
#undef PN_UPD3
      }
      __syncthreads();
      buf ^= 1;
    }
    // merge the slots (rows ascend with r inside a tile): the first maximum in point order
    float best0 = -INFINITY, best1 = -INFINITY;
    int bi0 = 0, bi1 = 0;
    if constexpr (COLMAX) { best0 = cm0; best1 = cm1; bi0 = ct0 * 32; bi1 = ct1 * 32; }
    else
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int i0 = (int)((bt0[r >> 2] >> (8 * (r & 3))) & 0xFFu) * 32 + ro;
      const int i1 = (int)((bt1[r >> 2] >> (8 * (r & 3))) & 0xFFu) * 32 + ro;
      if (bv0[r] > best0 || (bv0[r] == best0 && i0 < bi0)) { best0 = bv0[r]; bi0 = i0; }
      if (bv1[r] > best1 || (bv1[r] == best1 && i1 < bi1)) { best1 = bv1[r]; bi1 = i1; }
    }
    // the two lane halves hold interleaved row groups of the same column
    const float o0 = __shfl_xor(best0, 32, 64), o1 = __shfl_xor(best1, 32, 64);
    const int oi0 = __shfl_xor(bi0, 32, 64), oi1 = __shfl_xor(bi1, 32, 64);
    if (o0 > best0 || (o0 == best0 && oi0 < bi0)) { best0 = o0; bi0 = oi0; }
    if (o1 > best1 || (o1 == best1 && oi1 < bi1)) { best1 = o1; bi1 = oi1; }
    if (h == 0) {
      y[(long long)b * ypitch + c0] = best0 + b2_0;
      y[(long long)b * ypitch + c1] = best1 + b2_1;
      if (argmax) {
        argmax[(long long)b * ypitch + c0] = bi0;
        argmax[(long long)b * ypitch + c1] = bi1;
      }
    }
  }
}

// thread c = output column.  Per cloud: recompute the hidden row of its argmax point (two hidden units per packed
// instruction, the forward's rounding), accumulate dW2[c][:] in REGISTERS (64 per thread: the LDS read-modify-write per
// element it replaces was a third of the loop) and db2, form t[k] = dy*W2[c][k]*gelu'(pre) with W2 read from an LDS
// copy ([k][c], staged once: the per-element global load it replaces sat in the dependency chain) and reduce
// t[k] * (x, y, z, 1) over the 256 columns into dW1 / db1 (16 k at a time through LDS).
#ifndef PNB_Q
#define PNB_Q 4    // threads per output column of the backward kernel (1, 2 or 4)
#endif
constexpr int PNB_THREADS = 256 * PNB_Q, PNB_ROUNDS = 4 / PNB_Q;
__global__ __launch_bounds__(PNB_THREADS) void k_pointnet_bwd(const float* __restrict__ x, long long xpitch, int B,
                                                      const PnObjs o,
                                                      const float* __restrict__ dy, long long dypitch, const int* __restrict__ argmax,
                                                      long long ipitch, float* __restrict__ partial) {
  const int obj = __builtin_amdgcn_readfirstlane((int)blockIdx.x / o.bpo), bid = (int)blockIdx.x - obj * o.bpo, nbl = o.bpo;
  const float* __restrict__ params = o.params[obj];
  x += o.x_off[obj];
  dy += obj * PN_OUT;
  argmax += obj * PN_OUT;
  // 256 PNB_Q threads: thread (c = tid % 256, hf = tid / 256) owns output column c for the hidden units of chunks hf,
  // hf + PNB_Q, ... (16 each) -- a cloud is one dependent chain per thread (recompute the hidden row of the column's arg-max point), so
  // the workgroup is made PNB_Q times as wide instead of each thread walking all 64 units; every
  // sum keeps its order (per (c, k) over the clouds, per chunk over the columns): results are bit-identical.
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* W2t = sm;                           // [64 k][256 c]; at the end (with Ts) the [64 k][257] transposition buffer of dW2
  float* Ts = W2t + PN_H * PN_OUT;           // [PNB_Q][16 k][256 c]
  float* Xs = Ts + PNB_Q * 16 * PN_OUT + PN_H;       // [256 c][4]  (x, y, z, 1) of the argmax point (PN_H floats of slack: 64 x 257 fits below)
  float4* W1p = reinterpret_cast<float4*>(Xs + PN_OUT * 4);   // pair layout of pn_produce: 64 entries
  const int tid = threadIdx.x, c = tid & 255, hf = tid >> 8;
  {
    const float4* w2 = reinterpret_cast<const float4*>(params + PN_OW2 + c * PN_H);
#pragma unroll
    for (int q = 0; q < PN_H / 4 / PNB_Q; ++q) {
      const int k4 = (PN_H / 4 / PNB_Q) * hf + q;
      const float4 v = w2[k4];
      W2t[(4 * k4 + 0) * PN_OUT + c] = v.x; W2t[(4 * k4 + 1) * PN_OUT + c] = v.y;
      W2t[(4 * k4 + 2) * PN_OUT + c] = v.z; W2t[(4 * k4 + 3) * PN_OUT + c] = v.w;
    }
    if (tid < PN_H) {
      const int k = tid & ~1;
      const float* wk = params + PN_OW1 + k * 3;
      W1p[tid] = (tid & 1) ? make_float4(wk[2], wk[5], params[PN_OB1 + k], params[PN_OB1 + k + 1])
                           : make_float4(wk[0], wk[3], wk[1], wk[4]);
    }
  }
  __syncthreads();
  float db2 = 0.f;
  f32x2 dw2[8 * PNB_ROUNDS];
#pragma unroll
  for (int q = 0; q < 8 * PNB_ROUNDS; ++q) dw2[q] = (f32x2)(0.f);
  // reduction role inside a half: k_local = (tid / 16) % 16, part = tid % 16; part 0 keeps dW1 / db1 for k = chunk*16 + k_local
  const int kl = (tid >> 4) & 15, part = tid & 15;
  float* Th = Ts + hf * 16 * PN_OUT;
  float acc1[PNB_ROUNDS][4];
#pragma unroll
  for (int q = 0; q < PNB_ROUNDS; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[q][j] = 0.f;

  for (int b = bid; b < B; b += nbl) {
    const float g = dy[(long long)b * dypitch + c];
    const int n = argmax[(long long)b * ipitch + c];
    const float* xp = x + (long long)b * xpitch + (long long)n * PN_IN;
    const float px = xp[0], py = xp[1], pz = xp[2];
    db2 += g;
    if (hf == 0) *reinterpret_cast<float4*>(Xs + c * 4) = make_float4(px, py, pz, 1.0f);
#pragma unroll
    for (int c2 = 0; c2 < PNB_ROUNDS; ++c2) {
      const int chunk = PNB_Q * c2 + hf;
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        const int k = chunk * 16 + j;
        const float4 wa = W1p[k], wb = W1p[k + 1];
        const f32x2 wx = {wa.x, wa.y}, wy = {wa.z, wa.w}, wz = {wb.x, wb.y}, bb = {wb.z, wb.w};
        // the forward's rounding of the pre-activation (pn_produce), one Phi for the value and the derivative
        const f32x2 pre = pn_fma(wz, (f32x2)(pz), pn_fma(wy, (f32x2)(py), pn_fma(wx, (f32x2)(px), bb)));
        const f32x2 cdf = gelu_cdf(pre);
        dw2[c2 * 8 + (j >> 1)] = pn_fma((f32x2)(g), pre * cdf, dw2[c2 * 8 + (j >> 1)]);
        const f32x2 a = pre * pre * -0.72134752044448170368f;
        f32x2 e;
        e.x = __builtin_amdgcn_exp2f(a.x); e.y = __builtin_amdgcn_exp2f(a.y);
        const f32x2 w2 = {W2t[k * PN_OUT + c], W2t[(k + 1) * PN_OUT + c]};
        const f32x2 t = ((f32x2)(g) * w2) * pn_fma(pre, e * 0.39894228040143267794f, cdf);
        Th[j * PN_OUT + c] = t.x;
        Th[(j + 1) * PN_OUT + c] = t.y;
      }
      __syncthreads();
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = part + 16 * q;
        const float t = Th[kl * PN_OUT + cc];
        const float4 xv = *reinterpret_cast<const float4*>(Xs + cc * 4);
        s0 += t * xv.x; s1 += t * xv.y; s2 += t * xv.z; s3 += t * xv.w;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {  // the 16 `part` lanes of one k are adjacent lanes
        s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64); s3 += __shfl_xor(s3, o, 64);
      }
      acc1[c2][0] += s0; acc1[c2][1] += s1; acc1[c2][2] += s2; acc1[c2][3] += s3;
      __syncthreads();
    }
  }
  // per-workgroup partial in parameter layout
  float* out = partial + (long long)blockIdx.x * PN_P;     // [object][workgroup of the object][PN_P]
  if (part == 0) {
#pragma unroll
    for (int c2 = 0; c2 < PNB_ROUNDS; ++c2) {
      const int k = (PNB_Q * c2 + hf) * 16 + kl;
      out[PN_OW1 + k * 3 + 0] = acc1[c2][0];
      out[PN_OW1 + k * 3 + 1] = acc1[c2][1];
      out[PN_OW1 + k * 3 + 2] = acc1[c2][2];
      out[PN_OB1 + k] = acc1[c2][3];
    }
  }
  if (hf == 0) out[PN_OB2 + c] = db2;
  // parameter layout W2[c][k]: through LDS ([k][c], rows padded to 257) so that consecutive lanes write consecutive k (a
  // lane writing its own row c put 64 lanes on 64 different 256-byte rows per instruction)
  float* dW2s = sm;
#pragma unroll
  for (int c2 = 0; c2 < PNB_ROUNDS; ++c2)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = (PNB_Q * c2 + hf) * 16 + 2 * q;
      dW2s[k * PN_LD + c] = dw2[c2 * 8 + q].x;
      dW2s[(k + 1) * PN_LD + c] = dw2[c2 * 8 + q].y;
    }
  __syncthreads();
  for (int e = tid; e < PN_OUT * PN_H; e += PNB_THREADS) out[PN_OW2 + e] = dW2s[(e & (PN_H - 1)) * PN_LD + (e >> 6)];
}

static inline int pn_blocks(int64_t B) { return (int)(B < PN_BLOCKS ? B : PN_BLOCKS); }
// IGI_PN_COLMAX=1: the column-level running maximum (forward timing experiment: its arg-max is the TILE only, see the kernel)
static inline bool pn_colmax_experiment() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_PN_COLMAX"); on = e ? atoi(e) != 0 : 0; }
  return on != 0;
}
// workgroups per object: the chip's slots (512 forward: two per CU; 256 backward: one per CU) shared by the objects
static inline int pn_blocks_multi(int64_t B, int nobj, int total) {
  int64_t per = total / (nobj > 0 ? nobj : 1);
  if (per < 1) per = 1;
  return (int)(B < per ? B : per);
}
static size_t pointnet_workspace_bytes(int64_t B) { return sizeof(float) * (size_t)PN_P * pn_blocks(B); }
static size_t pointnet_workspace_bytes_multi(int64_t B, int nobj) {
  return sizeof(float) * (size_t)PN_P * (size_t)nobj * pn_blocks_multi(B, nobj, PN_BWD_BLOCKS);
}

static int pn_objs(PnObjs& o, int nobj, const float* const* params, const int32_t* x_off, const int32_t* N, int64_t x_pitch) {
  if (nobj < 1 || nobj > PN_MAX_OBJ || !params || !N) return IGI_E_BADARG;
  o.nobj = nobj;
  for (int i = 0; i < PN_MAX_OBJ; ++i) {
    const int k = i < nobj ? i : 0;
    if (!params[k] || N[k] < 1) return IGI_E_BADARG;
    if (N[k] > 8192) return IGI_E_UNSUPPORTED;   // 8-bit tile numbers (the reference's clouds: 400 points per object)
    o.params[i] = params[k]; o.N[i] = N[k]; o.x_off[i] = x_off ? x_off[k] : 0;
    if (o.x_off[i] < 0 || (int64_t)o.x_off[i] + (int64_t)N[k] * PN_IN > x_pitch) return IGI_E_BADARG;
  }
  return 0;
}

// x_pitch: floats between consecutive clouds (0 = dense, 3 N) -- a column slice of a wider (batch, points, 3) tensor runs in
// place (tact.py:542-566 slices the plug / socket / goal clouds out of one tensor); dy_pitch likewise for a slice of the
// gradient of the concatenated encodings
static int pointnet_forward(const float* x, int64_t x_pitch, int64_t B, int N, const float* params, float* y, int* argmax,
                            hipStream_t s) {
  if (!x || !params || !y || B < 1 || N < 1 || B > (1 << 30)) return IGI_E_BADARG;
  if (x_pitch == 0) x_pitch = (int64_t)N * PN_IN;
  if (x_pitch < (int64_t)N * PN_IN) return IGI_E_BADARG;
  PnObjs o;
  const int32_t n32 = N;
  int rc = pn_objs(o, 1, &params, nullptr, &n32, x_pitch);
  if (rc) return rc;
  o.bpo = pn_blocks(B);
  {
    // algorithmic: 2 * (3*64 + 64*256) flop per point; 12 B/point in, 256 values + 256 indices per cloud out
    ProfScope ps(PC_POINTNET_FWD, s, 2.0 * (PN_IN * PN_H + PN_H * PN_OUT) * (double)B * N,
                 12.0 * (double)B * N + 8.0 * PN_OUT * (double)B);
    if (pn_colmax_experiment()) IGI_LAUNCH(k_pointnet_fwd<true>, dim3(o.bpo), dim3(256), 0, s, x, (long long)x_pitch, (int)B, o, y, (long long)PN_OUT, argmax);
    else IGI_LAUNCH(k_pointnet_fwd<false>, dim3(o.bpo), dim3(256), 0, s, x, (long long)x_pitch, (int)B, o, y, (long long)PN_OUT, argmax);
  }
  return (int)hipGetLastError();
}

// nobj objects of one cloud tensor in one launch: object i = points x_off[i] / 3 .. of every cloud (x_off in floats), N[i]
// points, parameters params[i]; y and argmax are (B, nobj * 256): object i's encoding in columns 256 i ..
static int pointnet_forward_multi(int nobj, const float* x, int64_t x_pitch, int64_t B, const int32_t* x_off, const int32_t* N,
                                  const float* const* params, float* y, int* argmax, hipStream_t s) {
  if (!x || !y || B < 1 || B > (1 << 30) || x_pitch < 1) return IGI_E_BADARG;
  PnObjs o;
  int rc = pn_objs(o, nobj, params, x_off, N, x_pitch);
  if (rc) return rc;
  o.bpo = pn_blocks_multi(B, nobj, PN_BLOCKS);
  double pts = 0;
  for (int i = 0; i < nobj; ++i) pts += (double)B * N[i];
  {
    ProfScope ps(PC_POINTNET_FWD, s, 2.0 * (PN_IN * PN_H + PN_H * PN_OUT) * pts, 12.0 * pts + 8.0 * PN_OUT * (double)B * nobj);
    if (pn_colmax_experiment()) IGI_LAUNCH(k_pointnet_fwd<true>, dim3(o.bpo * nobj), dim3(256), 0, s, x, (long long)x_pitch, (int)B, o, y,
                                           (long long)PN_OUT * nobj, argmax);
    else IGI_LAUNCH(k_pointnet_fwd<false>, dim3(o.bpo * nobj), dim3(256), 0, s, x, (long long)x_pitch, (int)B, o, y,
                    (long long)PN_OUT * nobj, argmax);
  }
  return (int)hipGetLastError();
}

static int pn_backward_launch(const float* x, int64_t x_pitch, int64_t B, const PnObjs& o, const float* dy, int64_t dy_pitch,
                              const int* argmax, int64_t i_pitch, float* grads, void* ws, hipStream_t s) {
  if (pn_colmax_experiment() && !getenv("IGI_PN_COLMAX_TIMING")) return IGI_E_UNSUPPORTED;   // its arg-max is the tile only: timing runs say so explicitly
  const int nb = o.bpo;
  const size_t shm = sizeof(float) * (PN_H * PN_OUT + PNB_Q * 16 * PN_OUT + PN_H + PN_OUT * 4 + 4 * PN_H);
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
    ProfScope ps(PC_POINTNET_BWD, s, (double)B * o.nobj * PN_OUT * (4.0 * PN_IN * PN_H + 4.0 * PN_H),
                 (double)B * o.nobj * (8.0 * PN_OUT + 12.0 * PN_OUT) + 4.0 * PN_P * nb * o.nobj);
    IGI_LAUNCH(k_pointnet_bwd, dim3(nb * o.nobj), dim3(PNB_THREADS), shm, s, x, (long long)x_pitch, (int)B, o, dy,
               (long long)dy_pitch, argmax, (long long)i_pitch, partial);
  }
  SegTable t;
  t.n = o.nobj;
  t.wide = 1;
  for (int i = 0; i < o.nobj; ++i) {   // grads: [object][PN_P]
    Segment& sg = t.s[i];
    sg.dst = (long long)i * PN_P; sg.src = partial + (size_t)i * nb * PN_P; sg.stride = PN_P; sg.count = PN_P; sg.cols = PN_P;
    sg.src_ld = 0; sg.nparts = nb;
  }
  hipLaunchKernelGGL(k_slab_reduce, dim3(SLAB_GX, o.nobj), dim3(RED_THREADS), 0, s, t, grads);
  return (int)hipGetLastError();
}

static int pointnet_backward(const float* x, int64_t x_pitch, int64_t B, int N, const float* params, const float* dy,
                             int64_t dy_pitch, const int* argmax, float* grads, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!x || !params || !dy || !argmax || !grads || !ws || B < 1 || N < 1) return IGI_E_BADARG;
  if (x_pitch == 0) x_pitch = (int64_t)N * PN_IN;
  if (dy_pitch == 0) dy_pitch = PN_OUT;
  if (x_pitch < (int64_t)N * PN_IN || dy_pitch < PN_OUT) return IGI_E_BADARG;
  if (ws_bytes < pointnet_workspace_bytes(B)) return IGI_E_WORKSPACE;
  PnObjs o;
  const int32_t n32 = N;
  int rc = pn_objs(o, 1, &params, nullptr, &n32, x_pitch);
  if (rc) return rc;
  o.bpo = (int)(B < PN_BWD_BLOCKS ? B : PN_BWD_BLOCKS);
  return pn_backward_launch(x, x_pitch, B, o, dy, dy_pitch, argmax, PN_OUT, grads, ws, s);
}

// grads: [nobj][PN_P]; dy (row pitch dy_pitch >= nobj * 256) and argmax (B, nobj * 256) as the forward wrote them
static int pointnet_backward_multi(int nobj, const float* x, int64_t x_pitch, int64_t B, const int32_t* x_off, const int32_t* N,
                                   const float* const* params, const float* dy, int64_t dy_pitch, const int* argmax,
                                   float* grads, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!x || !dy || !argmax || !grads || !ws || B < 1 || x_pitch < 1) return IGI_E_BADARG;
  PnObjs o;
  int rc = pn_objs(o, nobj, params, x_off, N, x_pitch);
  if (rc) return rc;
  if (dy_pitch == 0) dy_pitch = (int64_t)PN_OUT * nobj;
  if (dy_pitch < (int64_t)PN_OUT * nobj) return IGI_E_BADARG;
  if (ws_bytes < pointnet_workspace_bytes_multi(B, nobj)) return IGI_E_WORKSPACE;
  o.bpo = pn_blocks_multi(B, nobj, PN_BWD_BLOCKS);
  return pn_backward_launch(x, x_pitch, B, o, dy, dy_pitch, argmax, (int64_t)PN_OUT * nobj, grads, ws, s);
}

}  // namespace igi
