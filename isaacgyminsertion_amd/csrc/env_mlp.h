// env_mlp forward as ONE launch: priv -> Linear+Tanh -> Linear+Tanh -> Linear+Tanh (the <= 8-wide latent), i.e.
// models_split.py:27-38 / :185-199 for a 64-row block per workgroup with every activation of the block kept in LDS.
// Built for the reference's shape (priv_info 64 -> 256 -> 128 -> 8, train yaml priv_mlp.units); other shapes run
// layer by layer through the GEMM launchers.
//
// Why a kernel of its own: as separate launches the three layers are two short GEMM launches (12 + 20 us for 1.6 GFLOP
// at minibatch 16384; the matrix pipes need 10 us), each of which pays a DMA round trip, a handful of k-tiles, an
// epilogue and a drain.  Here the 64 x 256 first-layer output never leaves the CU before it is consumed (it is still
// written out once, for the backward pass).
//
// Shape of the kernel -- what the phase stamps of tools/probes/env_fwd_probe.hip said about four earlier shapes:
//  * 16384 rows are 512 row blocks of 32 x 4 column quarters of 32: two 32 x 32 MFMA tiles per SIMD and no more, so
//    the run time is one workgroup's own serial chain (two workgroups per CU start together and stay in lockstep:
//    their stalls coincide, and staggering them only moved the end);
//  * 8 waves sharing the weights through an LDS-DMA ring (the GEMM kernels' structure): 28 us -- a 16 KB weight chunk
//    is 16 MFMAs per wave, and the block-wide barrier per chunk serialised fragment reads, wave A's MFMAs, wave B's
//    MFMAs and the wait for the next chunk;
//  * weights straight into registers: 31 us, bound by the texture path (32 cache lines per 16-byte-per-lane load);
//  * wave-private rings, one 32 x 32 tile per wave, two workgroups per CU: 24-26 us -- no barrier per chunk any
//    more, but a wave's 16 dependent MFMAs per chunk leave the pipe idle while it issues DMA, waits and reads fragments.
// So: ONE wave per SIMD that owns a column quarter for BOTH 32-row halves of a 64-row block.  Its two accumulators
// alternate on the matrix pipe (no dependent-issue stall), the weight fragments are read once for both, chunk c + 1's
// fragments are requested before chunk c's MFMAs (register double buffer), the private DMA ring runs three chunks
// ahead, and the tanh epilogue of a finished tile is issued element by element between the MFMAs of the following
// chunks (16 or 8 elements per 32 MFMAs; each costs its ~45 issue cycles there as well -- exact-fp32 MFMAs and vector
// instructions do not overlap on a SIMD -- but there is no separate epilogue phase in which the waves wait for each
// other).  Waves meet at three points where data crosses them.
//
// Arithmetic is the LDS-DMA GEMM's (gemm_dma.h): the same k-contiguous XOR-swizzled operand images, the same MFMA
// order (pairs k, k+4 inside every group of eight, k-tiles in order), the same fast_tanh, and the head's fmaf chain --
// outputs are bit-identical to the per-layer launches (tests/test_gpu_teacher.py checks it).
#pragma once
#include "gemm_dma.h"

#ifndef ENV_TS
#define ENV_TS(i)   // tools/probes/env_fwd_probe.hip stamps the phases of one workgroup through this hook
#endif

namespace igi {

struct EnvFwdArgs {
  const float* priv; int ldp;           // [M][K1] normalised privileged input
  const float* W1; const float* b1;     // [N1][K1], [N1]
  const float* W2; const float* b2;     // [N2][N1] (row pitch ldw2), [N2]
  const float* W3; const float* b3;     // [N3][N2], [N3]
  float* e1; int lde1;                  // [ceil(M / 64) * 64][N1]: whole row blocks are stored
  float* e2; int lde2;                  // [M][N2]
  float* out; int ldo;                  // [M][N3] (strided: the latent columns of xcat)
  int M, K1, N1, N2, N3;
  int ldw2;
};

constexpr int ENV_BM = 64;
constexpr int ENV_WAVES = 4;
constexpr int ENV_THREADS = ENV_WAVES * 64;
constexpr int ENV_IMG = 32 * DMA_BK;           // one [32][32] operand image (floats): a k-tile of one 32-row half
constexpr int ENV_NS = 4;                      // wave-private weight ring: chunk c in registers, c + 1 .. c + 3 in LDS / in flight
constexpr int ENV_EPLD = 32 + 4;               // row pitch of a wave's [64][32] slice of the second layer's output
constexpr int ENV_K1 = 64, ENV_N1 = 256, ENV_N2 = 128;
constexpr int ENV_KT1 = ENV_K1 / DMA_BK, ENV_KT2 = ENV_N1 / DMA_BK, ENV_NT1 = ENV_N1 / 128;
constexpr int ENV_NCH1 = ENV_NT1 * ENV_KT1, ENV_NCH = ENV_NCH1 + ENV_KT2;   // 4 + 8 weight chunks of 128 columns x 32 k
static_assert(ENV_NT1 == 2 && ENV_KT1 == 2 && ENV_KT2 == 8, "the chunk schedule below is written out for this shape");
static_assert(ENV_BM * ENV_EPLD <= ENV_NS * ENV_IMG, "the output slice reuses the wave's drained ring");
constexpr int ENV_LDS_FLOATS = 2 * ENV_KT1 * ENV_IMG + 2 * ENV_KT2 * ENV_IMG + ENV_WAVES * ENV_NS * ENV_IMG + 8 * ENV_N2;

static inline bool env_fwd_supported(int K1, int N1, int N2, int N3) {
  return K1 == ENV_K1 && N1 == ENV_N1 && N2 == ENV_N2 && N3 >= 1 && N3 <= 8;
}

// Where accumulator element r of a lane (row (r & 3) + 8 (r >> 2) + 4 h, column l31) goes in a k-contiguous swizzled
// [32][32] image: float index ib[(r >> 1) & 3] + 32 * ((r & 3) + 8 (r >> 2)).  The swizzle term ((row >> 1) & 7) takes
// only the four values {0, 1, 4, 5} + 2 h over a lane's sixteen rows, so four lane bases and immediate offsets do.
__device__ __forceinline__ void env_img_bases(int l31, int h, int (&ib)[4]) {
  const int sk[4] = {0, 1, 4, 5};
#pragma unroll
  for (int k = 0; k < 4; ++k) ib[k] = (4 * h * 8 + ((l31 >> 2) ^ (sk[k] + 2 * h))) * 4 + (l31 & 3);
}

struct EnvFrags { f32x4 a[2][4], b[4]; };    // one chunk's operands: A of both row halves, B (shared)

// fragments of a swizzled [32][32] image: lane (row l31, h) feeds k = 8 g + 4 h .. + 3 to MFMA group g
__device__ __forceinline__ void env_frag(const float* img, int l31, int h, f32x4 (&f)[4]) {
  const int sw = (l31 >> 1) & 7;
#pragma unroll
  for (int g8 = 0; g8 < 4; ++g8) f[g8] = *reinterpret_cast<const f32x4*>(img + (l31 * 8 + ((2 * g8 + h) ^ sw)) * 4);
}

struct EnvWave {
  const EnvFwdArgs& a;
  float *P, *E1, *ring;
  int l31, h, wn, lane;
  int ib[4];
  unsigned off1[4], off2[4];   // byte offsets of this lane's four 16-byte pieces of a weight chunk (layer 1 / layer 2)
  f32x16 t0[2], t1[2], z[2];      // layer-1 slice 0, slice 1, layer 2; [row half]
  EnvFrags F[2];
  float bias1[2];

  __device__ __forceinline__ EnvWave(const EnvFwdArgs& a_) : a(a_) {}

  // chunk C of this wave: rows 32 wn .. of a 128-row weight slice, 32 k -- four DMA instructions of eight whole
  // 128-byte rows each, landing as the same swizzled k-contiguous image the GEMM kernels use
  template <int C>
  __device__ __forceinline__ void issue() {
    float* st = ring + (C % ENV_NS) * ENV_IMG;
    constexpr bool first = C < ENV_NCH1;
    // wave-uniform base (scalar unit) + 32-bit lane offset: the `global_load_lds_dwordx4 v_off, s[base]` form, no
    // vector instruction per request (the empty asm keeps the offset's zero-extension next to the add, see DmaPtrs)
    const char* base = reinterpret_cast<const char*>(first ? a.W1 + (long long)(C / ENV_KT1) * 128 * ENV_K1 + (C % ENV_KT1) * DMA_BK
                                                           : a.W2 + (C - ENV_NCH1) * DMA_BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned& o = first ? off1[i] : off2[i];
      asm volatile("" : "+v"(o));
      dma16(reinterpret_cast<const float*>(base + o), st + 256 * i);
    }
  }
  template <int C>
  __device__ __forceinline__ const float* a_img(int half) const {   // A operand of chunk C: input block or layer-1 images
    return C < ENV_NCH1 ? P + ((C % ENV_KT1) * 2 + half) * ENV_IMG : E1 + ((C - ENV_NCH1) * 2 + half) * ENV_IMG;
  }
  // layer-2 chunks 4 and 8 read images whose last elements were written (by other waves) during the chunk before
  static constexpr bool a_late(int C) { return C == ENV_NCH1 || C == ENV_NCH1 + 4; }

  template <int C>
  __device__ __forceinline__ void chunk() {
    // request chunk C + 3 (its stage held chunk C - 1, whose fragments were read a chunk ago), then wait until only
    // the requests younger than chunk C + 1 are outstanding: C + 1 has landed, its fragments are requested now and
    // arrive under this chunk's MFMAs
    if (C + 3 < ENV_NCH) issue<(C + 3 < ENV_NCH ? C + 3 : 0)>();
    ENV_TS(2 * C);
    if (C + 1 < ENV_NCH) {
      if (C + 3 < ENV_NCH) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (C + 2 < ENV_NCH) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      EnvFrags& nx = F[(C + 1) & 1];
      env_frag(ring + ((C + 1) % ENV_NS) * ENV_IMG, l31, h, nx.b);
      if (!a_late(C + 1)) {
        env_frag(a_img<(C + 1 < ENV_NCH ? C + 1 : 0)>(0), l31, h, nx.a[0]);
        env_frag(a_img<(C + 1 < ENV_NCH ? C + 1 : 0)>(1), l31, h, nx.a[1]);
      }
    }
    EnvFrags& cu = F[C & 1];
    if (a_late(C)) {
      env_frag(a_img<C>(0), l31, h, cu.a[0]);
      env_frag(a_img<C>(1), l31, h, cu.a[1]);
    }
    if (C == ENV_NCH - 1) {
      // the last chunk carries the global copy of layer 1 (the backward pass needs it) -- not an earlier one: vmcnt
      // retires in order, so a store (acknowledged after ~3 k cycles) in front of a weight request holds that
      // request's wait for as long.  One 16-byte unit per lane and image, eight lanes = one 128-byte row segment.
      const int tid = wn * 64 + lane, srow = tid >> 3, sunit = tid & 7;
      const float* e1s = E1 + (srow * 8 + (sunit ^ ((srow >> 1) & 7))) * 4;
      const int m0 = blockIdx.x * ENV_BM;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float* e1p = a.e1 + (long long)(m0 + 32 * half + srow) * a.lde1 + 4 * sunit;
#pragma unroll
        for (int j = 0; j < ENV_KT2; ++j)
          *reinterpret_cast<f32x4*>(e1p + j * DMA_BK) = *reinterpret_cast<const f32x4*>(e1s + (j * 2 + half) * ENV_IMG);
      }
    }
    // which accumulators this chunk feeds, and which finished tile's epilogue rides between its MFMAs
    f32x16* acc = C < ENV_KT1 ? t0 : C < ENV_NCH1 ? t1 : z;
    constexpr bool fresh = C == 0 || C == ENV_KT1 || C == ENV_NCH1;
    constexpr int epi_n = (C >= ENV_KT1 && C < ENV_NCH1) ? 16 : (C >= ENV_NCH1 && C < ENV_NCH1 + 4) ? 8 : 0;   // elements (of 32)
    constexpr int epi_0 = (C >= ENV_KT1 && C < ENV_NCH1) ? 16 * (C - ENV_KT1) : (C >= ENV_NCH1 && C < ENV_NCH1 + 4) ? 8 * (C - ENV_NCH1) : 0;
    const f32x16* prev = C < ENV_NCH1 ? t0 : t1;
    const int pslice = C < ENV_NCH1 ? 0 : 1;
    if (fresh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    }
    // (exact-fp32 MFMAs and plain vector instructions do not overlap on a SIMD: an element costs its ~45 issue cycles
    // wherever it is placed -- pinning it behind the first accumulator's MFMA, where a bf16 kernel would hide it,
    // measured slower than the scheduler's own placement -- so the interleave only spares the waves a separate phase)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cu.a[0][i >> 2][i & 3], cu.b[i >> 2][i & 3], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cu.a[1][i >> 2][i & 3], cu.b[i >> 2][i & 3], acc[1], 0, 0, 0);
      if (epi_n > 0 && i % (epi_n > 0 ? 16 / epi_n : 1) == 0) {
        const int e = epi_0 + i / (epi_n > 0 ? 16 / epi_n : 1);   // 0 .. 31: half e >> 4, accumulator element e & 15
        const int half = e >> 4, r = e & 15;
        float* img = E1 + ((pslice * 4 + wn) * 2 + half) * ENV_IMG;
        img[ib[(r >> 1) & 3] + ((r & 3) + 8 * (r >> 2)) * DMA_BK] = fast_tanh(prev[half][r] + bias1[pslice]);
      }
    }
    ENV_TS(2 * C + 1);
    if (C == ENV_NCH1 - 1 || C == ENV_NCH1 + 3) {     // a layer-1 slice is complete in LDS: the next chunk reads all of it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
};

__global__ __launch_bounds__(ENV_THREADS) void k_env_fwd(const EnvFwdArgs a) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;         // wave tile: both 32-row halves x columns 32 wn .. of a 128-wide slice
  const int m0 = blockIdx.x * ENV_BM;
  EnvWave w(a);
  w.P = smem;                                        // [k-tile][half] images of the input block
  w.E1 = w.P + 2 * ENV_KT1 * ENV_IMG;                // [k-tile][half] images of layer 1's output = layer 2's A operand
  float* ring0 = w.E1 + 2 * ENV_KT2 * ENV_IMG;
  w.ring = ring0 + wn * (ENV_NS * ENV_IMG);          // THIS wave's weight chunks: nobody else touches them
  float* wsh = ring0 + ENV_WAVES * ENV_NS * ENV_IMG; // head weights [N3][128]
  w.l31 = l31; w.h = h; w.wn = wn; w.lane = lane;
  env_img_bases(l31, h, w.ib);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = 8 * i + (lane >> 3), k4 = (lane & 7) ^ ((m >> 1) & 7);
    w.off1[i] = 4u * ((unsigned)(wn * 32 + m) * ENV_K1 + 4u * k4);
    w.off2[i] = 4u * ((unsigned)(wn * 32 + m) * (unsigned)a.ldw2 + 4u * k4);
  }

  ENV_TS(40);
  // ordinary loads first: they are then older than everything the counted waits reason about
  const float bias2 = a.b2[wn * 32 + l31];
  w.bias1[0] = a.b1[wn * 32 + l31];
  w.bias1[1] = a.b1[128 + wn * 32 + l31];
  const float b3a = a.b3[min(2 * wn, a.N3 - 1)], b3b = a.b3[min(2 * wn + 1, a.N3 - 1)];
  float w3v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w3v[i] = a.W3[min(tid + ENV_THREADS * i, a.N3 * ENV_N2 - 1)];
  // the input block: 16 pieces of eight 128-byte rows, four per wave
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pid = wn + 4 * q, k = pid >> 3, half = (pid >> 2) & 1, i = pid & 3;
    const int m = 8 * i + (lane >> 3);
    const int row = min(m0 + 32 * half + m, a.M - 1);
    dma16(a.priv + (long long)row * a.ldp + k * DMA_BK + 4 * ((lane & 7) ^ ((m >> 1) & 7)), w.P + (k * 2 + half) * ENV_IMG + 256 * i);
  }
  w.issue<0>();
  w.issue<1>();
  w.issue<2>();
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (tid + ENV_THREADS * i < a.N3 * ENV_N2) wsh[tid + ENV_THREADS * i] = w3v[i];
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the input block and chunk 0
  __syncthreads();
  env_frag(w.ring, l31, h, w.F[0].b);
  env_frag(w.P, l31, h, w.F[0].a[0]);
  env_frag(w.P + ENV_IMG, l31, h, w.F[0].a[1]);

  w.chunk<0>(); w.chunk<1>(); w.chunk<2>(); w.chunk<3>();
  w.chunk<4>(); w.chunk<5>(); w.chunk<6>(); w.chunk<7>();
  w.chunk<8>(); w.chunk<9>(); w.chunk<10>(); w.chunk<11>();

  // ---- layer 2 complete: bias + tanh into this wave's [64][32] slice (over its drained ring), then the head and the global copy
  float* ep = w.ring;
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * half + (r & 3) + 8 * (r >> 2) + 4 * h;
      ep[row * ENV_EPLD + l31] = fast_tanh(w.z[half][r] + bias2);
    }
  ENV_TS(24);
  __syncthreads();
  ENV_TS(25);
  const int srow = tid >> 3, sunit = tid & 7;
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 32 * half + srow;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ring0 + j * (ENV_NS * ENV_IMG) + row * ENV_EPLD + 4 * sunit);
      if (m0 + row < a.M) *reinterpret_cast<f32x4*>(a.e2 + (long long)(m0 + row) * a.lde2 + j * 32 + 4 * sunit) = v;
    }
  ENV_TS(26);
  // the head: lane = row of the block, wave = a pair of outputs -- its weights are wave-uniform (one LDS broadcast
  // read per 16 bytes), the row is read once for both
  const float* wq0 = wsh + min(2 * wn, a.N3 - 1) * ENV_N2;
  const float* wq1 = wsh + min(2 * wn + 1, a.N3 - 1) * ENV_N2;
  float hacc0 = 0.f, hacc1 = 0.f;
  // same accumulation order as the MFMA k-loop: pairs (k, k+4) inside every group of eight
#pragma unroll 4
  for (int c8 = 0; c8 < ENV_N2; c8 += 8) {
    const float* x = ring0 + (c8 >> 5) * (ENV_NS * ENV_IMG) + lane * ENV_EPLD + (c8 & 31);
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(x), x1 = *reinterpret_cast<const f32x4*>(x + 4);
    const f32x4 u0 = *reinterpret_cast<const f32x4*>(wq0 + c8), u1 = *reinterpret_cast<const f32x4*>(wq0 + c8 + 4);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(wq1 + c8), v1 = *reinterpret_cast<const f32x4*>(wq1 + c8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hacc0 = fmaf(x0[j], u0[j], hacc0);
      hacc1 = fmaf(x0[j], v0[j], hacc1);
      hacc0 = fmaf(x1[j], u1[j], hacc0);
      hacc1 = fmaf(x1[j], v1[j], hacc1);
    }
  }
  ENV_TS(27);
  if (m0 + lane < a.M) {
    if (2 * wn < a.N3) a.out[(long long)(m0 + lane) * a.ldo + 2 * wn] = fast_tanh(hacc0 + b3a);
    if (2 * wn + 1 < a.N3) a.out[(long long)(m0 + lane) * a.ldo + 2 * wn + 1] = fast_tanh(hacc1 + b3b);
  }
  ENV_TS(41);
}

// -> hipErrorInvalidValue when the shapes are not the ones this kernel is built for (the caller runs the layers
// as separate launches)
static hipError_t env_mlp_forward(const EnvFwdArgs& a, hipStream_t s) {
  if (!env_fwd_supported(a.K1, a.N1, a.N2, a.N3) || a.M < 1 || !aligned16(a.priv) || !aligned16(a.W1) || !aligned16(a.W2) ||
      !aligned16(a.e1) || !aligned16(a.e2) || (a.ldp & 3) || (a.lde1 & 3) || (a.lde2 & 3) || a.ldp < a.K1 ||
      a.lde1 < a.N1 || a.lde2 < a.N2 || (a.ldw2 & 3) || a.ldw2 < a.N1)
    return hipErrorInvalidValue;
  const size_t shm = sizeof(float) * ENV_LDS_FLOATS;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)k_env_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ENV_LDS_FLOATS));
    if (e != hipSuccess) return e;
    attr = true;
  }
  const double fl = 2.0 * a.M * ((double)a.K1 * a.N1 + (double)a.N1 * a.N2 + (double)a.N2 * a.N3);
  const double by = 4.0 * ((double)a.M * (a.K1 + a.N1 + a.N2 + a.N3) + (double)a.K1 * a.N1 + (double)a.N1 * a.N2);
  ProfScope ps(PC_ENV_FWD, s, fl, by);
  IGI_LAUNCH(k_env_fwd, dim3((a.M + ENV_BM - 1) / ENV_BM), dim3(ENV_THREADS), shm, s, a);
  return hipGetLastError();
}

}  // namespace igi
