// Exact-fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), the contraction behind every
// Linear forward / dgrad / wgrad of the teacher and student networks.
//
//   C[m][n] (+)= sum_k A(m,k) * B(n,k)          (optionally batched and split along k)
//
// Operand element addressing is a template parameter, so the three products of a Linear layer
// need no transposed copies:
//   forward  Y  = act(X W^T + b) : A = X  [m][k] (k-contiguous), B = W [n][k] (k-contiguous)
//   dgrad    dX = dZ W           : A = dZ [m][k] (k-contiguous), B(n,k) = W[k][n] (n-contiguous)
//   wgrad    dW = dZ^T X         : A(m,k) = dZ[k][m], B(n,k) = X[k][n] (both reduction-major)
//
// Work decomposition (CDNA4): 256-thread workgroup = 4 waves (one per SIMD); each wave owns
// TM x TN accumulator tiles of 32x32 (16 fp32 regs per lane each).  Both operand tiles live in LDS
// reduction-major ([BK][rows+4]) so a wave reads its MFMA operands with conflict-free
// ds_read_b32 (lane&31 -> consecutive floats, lane>>5 -> k parity).  Global->LDS staging is
// register double-buffered: the loads of k-tile t+1 are in flight while the MFMAs of tile t
// run; one barrier per k-tile.  f32 MFMA issues at the f32 VALU rate (64 cyc / 32x32x2 per
// SIMD), so one wave per SIMD with >=2 independent accumulators already saturates the pipe and
// LDS/global bandwidth needs are low (8 B/clk/CU for a 128x128 tile).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prof.h"

namespace igi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Epilogue { EPI_STORE = 0, EPI_BIAS_TANH = 1, EPI_TANHGRAD = 2, EPI_BIAS = 3, EPI_BIAS_RELU = 4,
                EPI_RELUGRAD = 5, EPI_BIAS_ELU = 6, EPI_ELUGRAD = 7, EPI_COUNT = 8 };
// nn.ELU(alpha = 1): x > 0 ? x : expm1(x); its derivative from the OUTPUT a: a > 0 ? 1 : a + 1
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float elu1_grad_from_out(float a) { return a > 0.f ? 1.0f : a + 1.0f; }

// im2col addressing for the implicit-GEMM convolutions (channels-last activations): GEMM row
// m = (image b, output y, output x); tap (ky, kx) reads input pixel (oy*stride+ky-pad, ox*stride+kx-pad),
// out-of-image taps read a zero page.  Used by the LDS-DMA kernel only (gemm_dma.h).
// n / d for 0 <= n < 2^31 as one multiply-high and one shift (d fixed per launch): the im2col
// loaders decompose a row / tap index per lane per 16-byte fetch, and a hardware-less 32-bit
// division (~35 VALU instructions) there made the address math, not the MFMA pipe, the bottleneck.
struct FastDiv {
  unsigned int mul = 0, shr = 0, d = 1;
};
static inline FastDiv make_fastdiv(unsigned int d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) return f;
  unsigned int l = 0;
  while ((1u << l) < d) ++l;
  const unsigned int p = 31 + l;
  f.mul = (unsigned int)(((1ULL << p) + d - 1) / d);
  f.shr = p - 32;
  return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  return f.d <= 1 ? n : (int)(__umulhi((unsigned int)n, f.mul) >> f.shr);
}
// n / d with the magic number when the launcher provided the one for THIS d, the plain division otherwise
__device__ __forceinline__ int fdiv_checked(int n, int d, const FastDiv& f) {
  if (d == 1) return n;
  if (f.d == (unsigned int)d) return (int)(__umulhi((unsigned int)n, f.mul) >> f.shr);
  return n / d;
}

struct ConvDesc {
  const float* zero = nullptr;  // >= 16 bytes of zeros
  int OW = 0, OHW = 0;          // output grid per image
  int IH = 0, IW = 0, C = 0;    // input image, C floats per pixel (4, or a multiple of 32)
  int stride = 1, pad = 0, KW = 0;
  int ntaps = 0;                // KH*KW*C (columns of the im2col matrix)
  int fast32 = 0;               // pad == 0 and the image tensor spans < 4 GB: no bounds tests, 32-bit byte offsets (set by the launcher)
  int pmajor = 0;               // data gradient on position-major tiles (gemm_dma.h, GATHER == 4): set by the launcher
  int KH = 0;                   // kernel height (pmajor path)
  int nb32 = 0;                 // weight gradient over position-major rows (GATHER == 5): images / 32
  int planar = 0;               // forward only (GATHER == 1): the input is [image][C planes][IH][IW] (the caller's NCHW
                                // tensor, C = 3), KH = KW = 8 and the taps are ordered (c, ky, kx) = torch's weight layout:
                                // a 32-float k-tile is four kernel rows of one plane, K = 64 * C, no padding channel
  FastDiv dNB32;
  FastDiv dOW, dOHW, dC, dKW, dTPP;  // divisors OW, OHW, C, KW, C/32
};

struct GemmArgs {
  const float* A = nullptr;
  const float* B = nullptr;
  float* C = nullptr;
  const float* bias = nullptr;  // [N] (EPI_BIAS_TANH / EPI_BIAS)
  const float* aux = nullptr;   // [M][ldaux] saved activation (EPI_TANHGRAD)
  float* Cbias = nullptr;       // wgrad only: per-split column sums of A over k -> bias gradient [M]
  int M = 0, N = 0, K = 0;
  int lda = 0, ldb = 0, ldc = 0, ldaux = 0;
  long long sA = 0, sB = 0, sC = 0, sBias = 0, sAux = 0, sCbias = 0;  // batch strides (elements)
  int nbatch = 1, splitk = 1, kchunk = 0;                              // kchunk: k-range per split
  long long sCsplit = 0, sCbiasSplit = 0;                              // split strides (elements)
  int epilogue = EPI_STORE, accumulate = 0;
  int vecA = 0, vecB = 0;  // 16-byte global loads allowed for the operand (alignment checked on host)
  // row head fused into the epilogue of a bias+tanh layer whose output tile spans the whole row (N <= 128):
  // head_out[m][q] = tanh(sum_c act[m][c] * head_W[q][c] + head_b[q]), q < head_n <= 8 (gemm_dma.h, HEAD kernel)
  const float* head_W = nullptr;
  const float* head_b = nullptr;
  float* head_out = nullptr;
  int head_ld = 0, head_n = 0;
  // row dots fused into a tanh'-epilogue data-gradient tile (gemm_dma.h, multi kernel): besides C the tile emits
  //   rowdot_out[(z * n_tiles + nt) * M + m][q] = sum_{n in tile} C[z][m][n] * rowdot_W[q][z * rowdot_kz + n],  q < 8
  // i.e. its share of the product of the fresh dZ with eight more weight columns (the latent columns of the first trunk
  // layer): the consumer adds the n_tiles * nbatch partials per row instead of re-reading dZ from HBM.
  const float* rowdot_W = nullptr;
  float* rowdot_out = nullptr;
  int rowdot_ld = 0, rowdot_kz = 0;
  // spatial soft-argmax partials from the tiles of the last tactile convolution (gemm_dma.h, tall 256 x 64 forward tile
  // with the bias + ReLU epilogue): for every 32-row group g of the output (rows 32 g .. 32 g + 31 belong to ONE image when
  // H_out * W_out is a multiple of 32) and channel c, ssa_part[g][c] = (max, sum e, sum e * xw, sum e * yw) with
  // e = exp(a - max) over the group's positions; k_softargmax_combine merges an image's groups (tactile.h)
  float* ssa_part = nullptr;
  int ssa_P = 0, ssa_h = 0, ssa_w = 0;
  // weight gradient of the layer BELOW fused into a tanh'-epilogue data-gradient tile (gemm_dma.h, multi kernel, together
  // with rowdot_*): the fresh dZ tile (rows m, columns n of THIS product = outputs of the layer below) is multiplied,
  // transposed, with the layer below's 32-wide input rows lw_X[m][0..31] -- lw_out[part][batch][n][j] += sum_m C[m][n] *
  // lw_X[m][j], and the bias gradient sum_m C[m][n] -- accumulated in registers over the lw_chain consecutive row tiles a
  // workgroup computes; C itself is then NOT written (nobody else reads it).  part = row-tile group.
  const float* lw_X = nullptr;
  float* lw_out = nullptr;
  float* lw_bias = nullptr;
  int lw_ldx = 0, lw_chain = 0;
  int lw_xw = 0;   // real input width (<= 31): column 31 of the padded rows carries the ONE that yields the bias gradient
  long long lw_sPart = 0, lw_sNet = 0, lw_bsPart = 0, lw_bsNet = 0;
  int wide_epi = 0;        // LDS-DMA kernel: LDS-staged 16-byte epilogue stores allowed (set by its launcher)
  int gather = 0;          // 0 none | 1 A = im2col gather (k-contiguous) | 2 B = im2col (reduction-major)
                           // | 3 A = im2col (reduction-major): conv weight gradient with taps on the M side
  int bias_from_b = 0;     // Cbias = column sums of B over k (instead of A), indexed by n
  double flop_credit = 1.0;  // profiler only: algorithmic / executed flops (zero-padded operands, e.g. K 23 run as 32)
  // LDS-DMA kernels: divisors of the workgroup's tile decomposition (n_tiles, m_tiles, splitk), set by the launchers
  // (dma_set_divs).  A division by a run-time value is expanded through the vector unit's float reciprocal (~15
  // instructions, two of them quarter-rate) -- five of them per workgroup, paid in full beside fp32 MFMAs; with the
  // magic numbers the decomposition is five scalar multiplies.  d == 0: not set, the kernel divides.
  FastDiv dNT{0, 0, 0}, dMT{0, 0, 0}, dSK{0, 0, 0};
  ConvDesc conv;
};

// tanh(x) = 1 - 2/(exp(2x)+1): v_exp_f32 + v_rcp_f32; abs error <= ~1.5e-7 (fp32 tolerance of the
// path is 1e-4 relative on gradients), 6 instructions instead of ~40 for ocml tanhf.
// Written as the instructions it should be: exp(2x) = exp2(x * 2 log2(e)) with the constant 2 * fl(log2 e) -- the same
// bits as (x + x) * fl(log2 e), a scaling by two is exact -- and 1 - 2r as one fma (2r is exact, one rounding either
// way).  Two instructions fewer per element than "1.0f - 2.0f * rcp(expf(2.0f * x) + 1.0f)" compiles to under
// -ffp-contract=off, for the same result bit by bit; beside exact-fp32 MFMAs every vector instruction is paid in full.
// No clamp of the argument: v_exp_f32 saturates to +inf / 0, v_rcp_f32(inf) = 0 and rcp(0 + 1) = 1, so |x| beyond the
// range where exp(2x) is finite gives exactly +1 / -1 -- the values the clamped version (|x| <= 15) produced.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * __builtin_bit_cast(float, 0x4038aa3bu));
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}


// Epilogue of one 32x32 accumulator tile (C/D layout of v_mfma_f32_32x32x2: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)).  EPI is compile-time so the element loop is branch-free;
// auxiliary loads are issued together (clamped addresses) before any store.
template <int EPI, bool ACCUM>
__device__ __forceinline__ void epilogue_tile(const f32x16& acc, float* __restrict__ C, int ldc,
                                              const float* __restrict__ bias,
                                              const float* __restrict__ aux, int ldaux, int rbase,
                                              int col, int M, int N) {
  const bool colok = col < N;
  const int colc = colok ? col : N - 1;
  float bv = 0.f;
  if (EPI == EPI_BIAS_TANH || EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_ELU) bv = bias[colc];
  float t[16], prev[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = rbase + (r & 3) + 8 * (r >> 2);
    const int rowc = row < M ? row : M - 1;
    if (EPI == EPI_TANHGRAD || EPI == EPI_RELUGRAD || EPI == EPI_ELUGRAD) t[r] = aux[rowc * ldaux + colc];
    if (ACCUM) prev[r] = C[rowc * ldc + colc];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = rbase + (r & 3) + 8 * (r >> 2);
    float v = acc[r];
    if (ACCUM) v += prev[r];
    if (EPI == EPI_BIAS_TANH) v = fast_tanh(v + bv);
    else if (EPI == EPI_BIAS) v = v + bv;
    else if (EPI == EPI_TANHGRAD) v = v * (1.0f - t[r] * t[r]);
    else if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bv, 0.f);
    else if (EPI == EPI_RELUGRAD) v = (t[r] > 0.f) ? v : 0.f;
    else if (EPI == EPI_BIAS_ELU) v = elu1(v + bv);
    else if (EPI == EPI_ELUGRAD) v = v * elu1_grad_from_out(t[r]);
    if (colok && row < M) C[row * ldc + col] = v;
  }
}

#define IGI_EPILOGUE_DISPATCH(CALL)                                                    \
  do {                                                                                 \
    if (g.accumulate) {                                                                \
      if (g.epilogue == EPI_TANHGRAD) { CALL(EPI_TANHGRAD, true); }                    \
      else if (g.epilogue == EPI_BIAS_TANH) { CALL(EPI_BIAS_TANH, true); }             \
      else if (g.epilogue == EPI_BIAS) { CALL(EPI_BIAS, true); }                       \
      else if (g.epilogue == EPI_BIAS_RELU) { CALL(EPI_BIAS_RELU, true); }             \
      else if (g.epilogue == EPI_RELUGRAD) { CALL(EPI_RELUGRAD, true); }               \
      else if (g.epilogue == EPI_BIAS_ELU) { CALL(EPI_BIAS_ELU, true); }               \
      else if (g.epilogue == EPI_ELUGRAD) { CALL(EPI_ELUGRAD, true); }                 \
      else { CALL(EPI_STORE, true); }                                                  \
    } else {                                                                           \
      if (g.epilogue == EPI_TANHGRAD) { CALL(EPI_TANHGRAD, false); }                   \
      else if (g.epilogue == EPI_BIAS_TANH) { CALL(EPI_BIAS_TANH, false); }            \
      else if (g.epilogue == EPI_BIAS) { CALL(EPI_BIAS, false); }                      \
      else if (g.epilogue == EPI_BIAS_RELU) { CALL(EPI_BIAS_RELU, false); }            \
      else if (g.epilogue == EPI_RELUGRAD) { CALL(EPI_RELUGRAD, false); }              \
      else if (g.epilogue == EPI_BIAS_ELU) { CALL(EPI_BIAS_ELU, false); }              \
      else if (g.epilogue == EPI_ELUGRAD) { CALL(EPI_ELUGRAD, false); }                \
      else { CALL(EPI_STORE, false); }                                                 \
    }                                                                                  \
  } while (0)

constexpr int GEMM_BK = 16;
constexpr int GEMM_THREADS = 256;

// ---- global -> register tile fetch ---------------------------------------------------------
// k-contiguous source: unit u -> (row = u/4, k4 = u%4), 4 consecutive k per unit.
// reduction-major source: unit u -> (k = u/(ROWS/4), r4 = u%(ROWS/4)), 4 consecutive rows per unit.
template <int ROWS, bool KC>
struct TileRegs {
  static constexpr int UNITS = ROWS * GEMM_BK / 4;
  static constexpr int NU = (UNITS + GEMM_THREADS - 1) / GEMM_THREADS;
  float4 v[NU];

  __device__ __forceinline__ void fetch(const float* __restrict__ src, int ld, int r0, int rmax,
                                        int k0, int kend, int vec, int tid) {
#pragma unroll
    for (int p = 0; p < NU; ++p) {
      const int u = tid + p * GEMM_THREADS;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (UNITS % GEMM_THREADS == 0 || u < UNITS) {
        if (KC) {
          const int r = r0 + (u >> 2);
          const int k = k0 + ((u & 3) << 2);
          if (r < rmax) {
            const float* p0 = src + (long long)r * ld + k;
            if (vec && k + 3 < kend) {
              x = *reinterpret_cast<const float4*>(p0);
            } else {
              if (k + 0 < kend) x.x = p0[0];
              if (k + 1 < kend) x.y = p0[1];
              if (k + 2 < kend) x.z = p0[2];
              if (k + 3 < kend) x.w = p0[3];
            }
          }
        } else {
          constexpr int R4 = ROWS / 4;
          const int k = k0 + u / R4;
          const int r = r0 + ((u % R4) << 2);
          if (k < kend) {
            const float* p0 = src + (long long)k * ld + r;
            if (vec && r + 3 < rmax) {
              x = *reinterpret_cast<const float4*>(p0);
            } else {
              if (r + 0 < rmax) x.x = p0[0];
              if (r + 1 < rmax) x.y = p0[1];
              if (r + 2 < rmax) x.z = p0[2];
              if (r + 3 < rmax) x.w = p0[3];
            }
          }
        }
      }
      v[p] = x;
    }
  }

  // LDS image: S[k][row], leading dimension LD = ROWS + 4
  __device__ __forceinline__ void stash(float* __restrict__ S, int tid) const {
    constexpr int LD = ROWS + 4;
#pragma unroll
    for (int p = 0; p < NU; ++p) {
      const int u = tid + p * GEMM_THREADS;
      if (UNITS % GEMM_THREADS == 0 || u < UNITS) {
        if (KC) {
          const int r = u >> 2;
          const int k = (u & 3) << 2;
          S[(k + 0) * LD + r] = v[p].x;
          S[(k + 1) * LD + r] = v[p].y;
          S[(k + 2) * LD + r] = v[p].z;
          S[(k + 3) * LD + r] = v[p].w;
        } else {
          constexpr int R4 = ROWS / 4;
          const int k = u / R4;
          const int r = (u % R4) << 2;
          *reinterpret_cast<float4*>(&S[k * LD + r]) = v[p];
        }
      }
    }
  }
};

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_kernel(const GemmArgs g) {
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;    // 32x32 MFMA tiles per wave
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
  constexpr int LDA_S = BM + 4, LDB_S = BN + 4;
  constexpr int BK = GEMM_BK;

  __shared__ __attribute__((aligned(16))) float lds[2 * BK * LDA_S + 2 * BK * LDB_S];
  float* As = lds;
  float* Bs = lds + 2 * BK * LDA_S;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const int batch = blockIdx.z / g.splitk;
  const int split = blockIdx.z % g.splitk;

  const float* A = g.A + batch * g.sA;
  const float* B = g.B + batch * g.sB;
  int k_begin = 0, k_end = g.K;
  if (g.splitk > 1) {
    k_begin = split * g.kchunk;
    k_end = min(g.K, k_begin + g.kchunk);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bsum = 0.f;  // bias-gradient column sum (wgrad), one A column per thread
  const bool do_bsum = (g.Cbias != nullptr) && (blockIdx.x == 0) && (tid < BM);

  TileRegs<BM, A_KC> ra;
  TileRegs<BN, B_KC> rb;
  const int nk = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;

  if (nk > 0) {
    ra.fetch(A, g.lda, m0, g.M, k_begin, k_end, g.vecA, tid);
    rb.fetch(B, g.ldb, n0, g.N, k_begin, k_end, g.vecB, tid);
    ra.stash(As, tid);
    rb.stash(Bs, tid);
  }
  __syncthreads();

  const int khalf = lane >> 5;
  const int l31 = lane & 31;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      const int k0 = k_begin + (kt + 1) * BK;
      ra.fetch(A, g.lda, m0, g.M, k0, k_end, g.vecA, tid);
      rb.fetch(B, g.ldb, n0, g.N, k0, k_end, g.vecB, tid);
    }
    const float* as = As + cur * BK * LDA_S + wm * WTM + l31;
    const float* bs = Bs + cur * BK * LDB_S + wn * WTN + l31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[(kk + khalf) * LDA_S + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[(kk + khalf) * LDB_S + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_bsum) {
      const float* ac = As + cur * BK * LDA_S + tid;
#pragma unroll
      for (int k = 0; k < BK; ++k) bsum += ac[k * LDA_S];
    }
    if (kt + 1 < nk) {
      ra.stash(As + (cur ^ 1) * BK * LDA_S, tid);
      rb.stash(Bs + (cur ^ 1) * BK * LDB_S, tid);
    }
    __syncthreads();
  }

  // ---- epilogue
  float* C = g.C + batch * g.sC + split * g.sCsplit;
  const float* bias = g.bias ? g.bias + batch * g.sBias : nullptr;
  const float* aux = g.aux ? g.aux + batch * g.sAux : nullptr;
#define IGI_EPI_CALL(E, ACC)                                                                   \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      epilogue_tile<E, ACC>(acc[i][j], C, g.ldc, bias, aux, g.ldaux,                           \
                            m0 + wm * WTM + i * 32 + 4 * khalf, n0 + wn * WTN + j * 32 + l31, g.M, g.N)
  IGI_EPILOGUE_DISPATCH(IGI_EPI_CALL);
#undef IGI_EPI_CALL
  if (do_bsum && (m0 + tid) < g.M) {
    g.Cbias[batch * g.sCbias + split * g.sCbiasSplit + m0 + tid] = bsum;
  }
}

template <int BM, int BN, int WGM, int WGN>
static hipError_t launch_cfg(const GemmArgs& g, bool akc, bool bkc, hipStream_t s) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.nbatch * g.splitk);
  dim3 block(GEMM_THREADS);
  if (akc && bkc)
    IGI_LAUNCH((gemm_f32_kernel<BM, BN, WGM, WGN, true, true>), grid, block, 0, s, g);
  else if (akc && !bkc)
    IGI_LAUNCH((gemm_f32_kernel<BM, BN, WGM, WGN, true, false>), grid, block, 0, s, g);
  else if (!akc && !bkc)
    IGI_LAUNCH((gemm_f32_kernel<BM, BN, WGM, WGN, false, false>), grid, block, 0, s, g);
  else
    IGI_LAUNCH((gemm_f32_kernel<BM, BN, WGM, WGN, false, true>), grid, block, 0, s, g);
  return hipGetLastError();
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tile shape by output extent: narrow outputs (latent / head widths) get narrow tiles so the
// MFMA work wasted on padding stays small.
static inline void gemm_tile_for(int M, int N, int* bm, int* bn) {
  if (N <= 32) { *bm = 128; *bn = 32; }
  else if (M <= 32) { *bm = 32; *bn = 128; }
  else if (N <= 64) { *bm = 128; *bn = 64; }
  else { *bm = 128; *bn = 128; }
}

static hipError_t launch_gemm(GemmArgs g, bool akc, bool bkc, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return hipSuccess;
  // the epilogue indexes C / aux with 32-bit offsets
  if ((long long)g.M * g.ldc >= (1LL << 31) || (long long)g.M * (g.ldaux + 1) >= (1LL << 31))
    return hipErrorInvalidValue;
  if (g.splitk < 1) g.splitk = 1;
  if (g.splitk > 1 && g.kchunk <= 0) {
    int c = (g.K + g.splitk - 1) / g.splitk;
    g.kchunk = (c + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
  }
  // 16-byte loads need the base, the leading dimension and every batch/split offset aligned
  g.vecA = aligned16(g.A) && (g.lda % 4 == 0) && (g.sA % 4 == 0);
  g.vecB = aligned16(g.B) && (g.ldb % 4 == 0) && (g.sB % 4 == 0);
  if (g.splitk > 1 && (g.kchunk % 4 != 0)) {  // a k-contiguous operand would start mid-vector
    if (akc) g.vecA = 0;
    if (bkc) g.vecB = 0;
  }
  int bm, bn;
  gemm_tile_for(g.M, g.N, &bm, &bn);
  // algorithmic work of this launch: 2*M*N*K flops per batch; bytes = operands once + output once
  const double fl = 2.0 * g.M * g.N * (double)g.K * g.nbatch * g.flop_credit;
  const double by = 4.0 * g.nbatch * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * g.splitk);
  ProfScope ps(PC_GEMM_GENERIC, s, fl, by);
  if (bm == 128 && bn == 32) return launch_cfg<128, 32, 4, 1>(g, akc, bkc, s);
  if (bm == 32 && bn == 128) return launch_cfg<32, 128, 1, 4>(g, akc, bkc, s);
  if (bm == 128 && bn == 64) return launch_cfg<128, 64, 4, 1>(g, akc, bkc, s);
  return launch_cfg<128, 128, 2, 2>(g, akc, bkc, s);
}

}  // namespace igi
