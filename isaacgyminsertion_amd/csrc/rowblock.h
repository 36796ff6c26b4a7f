// One backward LEVEL of a Linear + Tanh stack as a persistent row-block kernel (autograd of models_split.py:27-38,
// called from frozen_ppo.py:583-585): for the layer  Z = X W^T + b,  X = tanh(previous pre-activation),
//
//     dX = (dZ W) * (1 - X^2)          data gradient into the layer below, times tanh'
//     dW = dZ^T X ,  db = sum_rows dZ  this layer's weight / bias gradient
//
// Both products consume the same dZ rows and the same X rows.  As two kinds of tiles in one grid (gemm_dma.h,
// gemm_dma_wgrad_multi_kernel) every dZ block and every X block crosses the chip twice, the weight gradient is cut
// into 64 split-K slabs per net, and each of the ~770 short-lived workgroups pays its own fill and drain.
//
// Here ONE resident workgroup per CU owns a 64-column slice c of the layer's input and a contiguous range of rows,
// and walks the range in 64-row blocks:
//
//   per block   dZ[64][128] and X[64][slice] arrive ONCE, by LDS-DMA, in a two-stage ring;
//               waves 0-3  dX[64][slice] = dZ . W[:, slice]   (W fragments live in registers for the whole kernel: 64
//                          values per lane, brought in through LDS-DMA once), tanh' from the X block already in LDS; the
//                          epilogue of block b - 1 is issued between the MFMA groups of block b;
//               waves 4-7  dW[:, slice] += dZ^T . X           accumulators stay in registers across ALL blocks (+ the
//                          bias gradient from the very dZ values they feed to the matrix pipe); they also issue every
//                          LDS-DMA request, so the other wave of each SIMD starts its MFMAs right behind the barrier;
//   per kernel  one dW partial per workgroup: ranges (32 / 64) partials per element instead of 64 split-K slabs.
//
// Each SIMD hosts one wave of either role (waves w and w + 4 share a SIMD), 64 MFMAs per wave and block, no exchange
// between waves: ONE barrier per block (publish the block that has landed / retire the stage that is refilled),
// exactly the GEMM k-loop's protocol.  The dZ image is k-contiguous and XOR-swizzled over the row's 32 16-byte units
// (unit u of row r holds columns 4 (u ^ (r & 15)) ..+3, swizzle on the DMA's per-lane SOURCE address) so that the same
// image serves the data gradient's ds_read_b128 (lane = row, four k) and the weight gradient's ds_read_b32
// (lane = column, k = row), both conflict-free.  The four slices of a row range sit on one XCD: dZ is fetched from HBM
// once and three more times from that XCD's L2.
//
// MODE 2 (LOWX): the finished data-gradient element -- lane = column, register = row pair: the A-operand layout of
// v_mfma_f32_32x32x2_f32 -- is multiplied in place with the input rows of the layer BELOW (right operand from L2): that
// layer's weight / bias gradient leaves with this kernel's partials and dX is never written.
//
// MODE 3 (LATZ, round 6; = MODE 2 +): dZ is not read but FORMED at the head of every block from the 8-wide latent gradient
// (the backward of env_mlp's last layer, k_latent_bwd's arithmetic): the kernel stages the layer's output activations where
// it staged dZ, the weight-gradient waves accumulate the rank-8 weight gradient from the untouched image on the matrix pipe
// (v_mfma_f32_16x16x4_f32), a barrier, all waves rewrite the image in place, a barrier.  One launch and the dZ round trip
// less per optimizer step; +4.5 us in this kernel against the 9 us launch it replaces (profiles/r06_latz_ab.log).
//
// What it bought (tools/probes/rb_level_probe.hip): the block loop runs at 0.80 of the fp32
// matrix peak -- the GEMM k-loop's steady-state rate -- + ~6 us per launch; half the HBM bytes of the tile levels; 3 us
// per level in the update.  The levels were bound by the MFMA rate the chip sustains, not by bytes; the gains came from
// what the resident structure then allowed (MODE 2, fewer partials, one launch less).
//
// Arithmetic: the data gradient's MFMA chain is the LDS-DMA GEMM's (k-tiles in order, pairs k, k + 4 inside every group
// of eight) -- its output is bit-identical to gemm_dma_body<128, true, false, ..., TANHGRAD_ONLY>; the weight gradient
// sums its rows in another (fixed) grouping than the split-K slabs did.
#pragma once
#include <type_traits>

#include "dma_util.h"
#include "gemm_f32.h"

namespace igi {

constexpr int RB_KO = 128;                          // width of dZ = this layer's output width
constexpr int RB_S = 64;                            // input-column slice of a workgroup
constexpr int RB_BR = 64;                           // rows per block
constexpr int RB_Z_FLOATS = RB_BR * RB_KO;          // 8192
constexpr int RB_X_FLOATS = RB_BR * RB_S;           // 4096
constexpr int RB_STAGE = RB_Z_FLOATS + RB_X_FLOATS;
constexpr int RB_EPLD = 32 + 4;                     // row pitch of a data-gradient wave's [32][32] parking slice
constexpr int RB_PARK = 4 * 32 * RB_EPLD;
constexpr int RB_WPLD = RB_S + 4;                   // row pitch of a weight-gradient wave's final [32][64] parking slice
constexpr int RB_DLT = 64 + 4;                        // row pitch of the transposed dl image (16-byte reads of 16 rows: conflict-free)
constexpr int RB_LATZ = 64 * 8 + 8 * 128 + 4 * 256 + 16 * RB_DLT;   // LATZ: dl of the block, the latent layer's weights, the closing exchange, dl transposed
constexpr int RB_LDS_FLOATS = 2 * RB_STAGE + RB_PARK;
constexpr int RB_LDS_FLOATS_LATZ = RB_LDS_FLOATS + RB_LATZ;
constexpr int RB_THREADS = 512;
static_assert(4 * 32 * RB_WPLD <= RB_STAGE, "the final weight-gradient tiles park in stage 0");
static_assert(2 * (32 * 66 + 32) <= RB_PARK, "the LOWX exchange uses the parking slices");

struct RbLevelArgs {
  const float* dZ = nullptr; int ldz = 0; long long sZ = 0;        // [nets][rows][128]
  const float* W = nullptr; int ldw = 0; long long sW = 0;         // [nets][128][IN]   (torch layout [out][in])
  const float* X = nullptr; int ldx = 0; long long sX = 0;         // [nets][rows][IN]  tanh outputs of the layer below
  float* dX = nullptr; int lddx = 0; long long sdX = 0;            // [nets][rows][IN]
  float* dWp = nullptr; int ldwp = 0; long long sWpart = 0, sWnet = 0;   // partials [ranges][nets][128][ldwp]
  float* dBp = nullptr; long long sBpart = 0, sBnet = 0;                 // partials [ranges][nets][128]
  int rows = 0, IN = 0, nets = 1;
  int nslices = 0, ranges = 0, nblocks = 0;     // IN / 64; row ranges per net (= partial count); rows / 64
  int variant = 0;                              // A/B switches of tools/probes/rb_level_probe.hip (IGI_RB_VARIANT)
  // LOWX (nets == 1): the weight / bias gradient of the layer BELOW from the finished data-gradient tiles -- lx_X = that
  // layer's input rows [rows][64]; lx_W partials [ranges][IN][lx_ldw], lx_B partials [ranges][IN] -- and dX is NOT written
  const float* lx_X = nullptr; int lx_ld = 0;
  float* lx_W = nullptr; int lx_ldw = 0; long long lx_sPart = 0;
  float* lx_B = nullptr; long long lx_bsPart = 0;
  // LATZ (round 6, MODE 3 = LOWX + LATZ, nets == 1; teacher.h latz_fuse_ref): dZ is not READ but formed in the block's prologue from the
  // 8-wide latent gradient -- `dZ` then points at the layer's OUTPUT activations Y = tanh(.) [rows][128] (staged exactly
  // as dZ would be) and the staged image is rewritten in place:  dl[row][k] = (sum_t lz_parts[t][row][k]) * (1 - lat^2),
  // dZ[row][c] = (sum_k dl[row][k] * lz_W3[k][c]) * (1 - Y[row][c]^2)  -- k_latent_bwd's arithmetic, expression by
  // expression.  The rank-8 weight / bias gradient of the latent layer, dW3[k][c] = sum_rows dl[row][k] * Y[row][c], leaves
  // as one record [8 * 128 | 8] per row range (slice s owns columns 32 s ..; slice 0 the bias).
  const float* lz_parts = nullptr; int lz_tiles = 0; long long lz_tstride = 0;   // [tiles][rows][8], tile stride in floats
  const float* lz_lat = nullptr; int lz_ldlat = 0;                               // latent values (tanh outputs), [rows][>= 8]
  const float* lz_W3 = nullptr;                                                  // [8][128]
  float* lz_rec = nullptr; long long lz_srec = 0;                                // records [ranges][lz_srec], lz_srec >= 8 * 128 + 8
};

// How many row ranges (= weight-gradient partials per net) for a level: one workgroup per CU when the rows allow it.
static inline int rb_level_ranges(int rows, int IN, int nets) {
  const int nslices = IN / RB_S, nblocks = rows / RB_BR;
  int r = 256 / (nets * nslices);
  if (r > nblocks) r = nblocks;
  return r < 1 ? 1 : r;
}

// IGI_RB_LEVEL=0: the tile kernels for every level (A/B)
static inline bool rb_level_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_RB_LEVEL"); on = e ? atoi(e) : 1; }
  return on != 0;
}

// Shapes this kernel is built for (the caller runs the tile kernels otherwise).
static inline bool rb_level_shape_ok(long long rows, int KO, int IN, int nets) {
  const bool on = rb_level_enabled();
  return on && KO == RB_KO && IN >= RB_S && (IN % RB_S) == 0 && IN <= 1024 && rows >= 4 * RB_BR && (rows % RB_BR) == 0 &&
         rows < (1 << 24) && nets >= 1 && nets <= 2 && !bf16_mode();
}

// NETS: one instantiation per level (trunk: actor + critic, env_mlp: one net), so that a kernel trace tells them apart.
// MODE 0: data gradient parked in LDS and stored 16 bytes per lane; 1: 4-byte stores from the accumulators (probe);
// 2: the data gradient is not stored -- element by element it is the A operand of a second product, the weight gradient of
// the layer below, dW_below[c][j] += dX[row][c] * X_below[row][j] (j < 64: two more accumulator tiles per wave) -- the
// accumulator layout IS the operand layout (lane = column, register = row pair), so nothing goes through LDS.
template <int MODE, int NETS>
__global__ __launch_bounds__(RB_THREADS) void k_rb_level(const RbLevelArgs a) {
  constexpr bool DIRECT = MODE == 1, LOWX = MODE == 2 || MODE == 3, LATZ = MODE == 3;
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // ---- which (net, row range, column slice): the slices of one range are neighbours on one XCD (blocks b and b + 8
  //      share an XCD), so the range's dZ blocks are served from that XCD's L2 after the first fetch
  const int total_ranges = NETS * a.ranges;
  int slice, gr;
  {
    const int bid = blockIdx.x;
    if ((total_ranges & 7) == 0) {
      const int x = bid & 7, p = bid >> 3;
      const int q = p / a.nslices;
      slice = p - q * a.nslices;
      gr = x * (total_ranges >> 3) + q;
    } else {
      gr = bid / a.nslices;
      slice = bid - gr * a.nslices;
    }
  }
  slice = __builtin_amdgcn_readfirstlane(slice);
  gr = __builtin_amdgcn_readfirstlane(gr);
  const int net = NETS == 1 ? 0 : __builtin_amdgcn_readfirstlane(gr / a.ranges), range = gr - net * a.ranges;
  const int b0 = __builtin_amdgcn_readfirstlane((int)((long long)range * a.nblocks / a.ranges));
  const int b1 = __builtin_amdgcn_readfirstlane((int)((long long)(range + 1) * a.nblocks / a.ranges));
  const int nb = b1 - b0;
  const int c0 = slice * RB_S;
  const float* dZ = a.dZ + net * a.sZ + (long long)b0 * RB_BR * a.ldz;
  const float* X = a.X + net * a.sX + (long long)b0 * RB_BR * a.ldx + c0;
  float* dX = a.dX + net * a.sdX + (long long)b0 * RB_BR * a.lddx + c0;
  const float* W = a.W + net * a.sW + c0;

  // ---- roles.  Waves 0-3: data gradient of tile (row half rt, column half ct) of the block's [64][64] output.
  //      Waves 4-7: weight gradient rows 32 ot ..+31 (of the 128 outputs) x the slice's 64 columns, AND every LDS-DMA
  //      request: right behind a barrier the data-gradient wave of a SIMD starts its MFMAs while the other wave
  //      requests the next block (with all eight waves requesting, the matrix pipes idled for the ~0.2 us that takes).
  //      Each role runs its own loop (its own register allocation); both execute one s_barrier per block.
  const bool dgrad_role = wave < 4;
  const int w4 = wave & 3;

  // ---- LATZ: pieces both roles run at the head of every block (between the barrier that publishes the staged block and
  //      the block's MFMAs): the rank-8 weight gradient from the untouched activations, a barrier, the in-place rewrite
  //      of the image into dZ, a barrier.
  float* dlb = smem + 2 * RB_STAGE + RB_PARK;          // [64][8]
  float* w3s = dlb + 64 * 8;                           // [8][128]
  float* lzx = w3s + 8 * 128;                          // [4 waves][8 k][32 columns] closing exchange
  float* dlt = lzx + 4 * 256;                          // [16][RB_DLT]: dl transposed (rows 8 .. 15 stay zero): the A operand below
  // dW3[k][32 slice + c] += sum_rows dl[row][k] * Y[row][c] on the matrix pipe, by the weight-gradient waves: wave w4 takes
  // rows 16 w4 ..+15 of the block as four v_mfma_f32_16x16x4_f32 steps per 16-column tile -- lane (n = lane & 15,
  // q = lane >> 4) feeds A[k-index n][row 4 q + t] (ONE 16-byte read of the transposed dl image serves the four steps) and
  // B[row 4 q + t][column n] (a 4-byte read of the untouched activations, 64 distinct banks) to step t; register r of lane
  // (n, q) holds dW3[k = 4 q + r][column n] (lanes q >= 2 multiply the zero rows).  As vector code (32 rows x one column per
  // thread, all eight waves) this phase cost ~0.8 us per block: 64 LDS reads and ~45 address instructions per 8 rows.
  typedef float f32x4m __attribute__((ext_vector_type(4)));
  f32x4m acc3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  auto latz_head = [&](const float* zs_c, auto wg_c) __attribute__((always_inline)) {
    constexpr bool WG = decltype(wg_c)::value;          // called by a weight-gradient wave
    float* zs = const_cast<float*>(zs_c);
    if constexpr (WG) {
      const int n = lane & 15, q = lane >> 4;
      const f32x4 av = *reinterpret_cast<const f32x4*>(dlt + n * RB_DLT + 16 * w4 + 4 * q);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rr = 4 * q + t;                        // (row & 15) of this lane's row of step t
        const float* yr = zs + (16 * w4 + rr) * RB_KO + (n & 3);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const float bv = yr[4 * ((8 * slice + 4 * jt + (n >> 2)) ^ rr)];
          acc3[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, acc3[jt], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      // a thread owns ONE column group (its 8 x 4 latent weights are read once per block) and rows rg + 16 i: unit
      // cg ^ rg of each (row & 15 == rg), a whole row per half-wave
      const int cg = tid & 31, rg = tid >> 5;
      f32x4 wv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) wv[k] = *reinterpret_cast<const f32x4*>(w3s + k * 128 + 4 * cg);
      float* zp0 = zs + rg * RB_KO + 4 * (cg ^ rg);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rg + 16 * i;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(dlb + row * 8), d1 = *reinterpret_cast<const f32x4*>(dlb + row * 8 + 4);
        f32x4* zp = reinterpret_cast<f32x4*>(zp0 + 16 * i * RB_KO);
        const f32x4 y = *zp;
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dk = k < 4 ? d0[k & 3] : d1[k & 3];
#pragma unroll
          for (int j = 0; j < 4; ++j) sacc[j] = fmaf(dk, wv[k][j], sacc[j]);
        }
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = sacc[j] * (1.0f - y[j] * y[j]);
        *zp = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  if (dgrad_role) {
    const int rt = w4 >> 1, ct = w4 & 1;
    __builtin_amdgcn_s_barrier();     // block 0 and W are in LDS
    asm volatile("" ::: "memory");
    float wf[64];                      // W[k = 8 c + 4 h + j][c0 + 32 ct + l31] at index 4 c + j
    int zq[16];                        // float offset of k-group 2 c + h of this lane's row in the dZ image
    {
      const float* wp = smem + RB_STAGE + 4 * h * RB_S + 32 * ct + l31;
#pragma unroll
      for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[4 * c + j] = wp[(8 * c + j) * RB_S];
      const int m = 32 * rt + l31;
#pragma unroll
      for (int c = 0; c < 16; ++c) zq[c] = m * RB_KO + 4 * ((2 * c + h) ^ (m & 15));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();     // W is in registers: stage 1 may be refilled
    asm volatile("" ::: "memory");
    float* park = smem + 2 * RB_STAGE + w4 * (32 * RB_EPLD);
    const int prow = lane >> 3, pc4 = lane & 7;
    unsigned lo = DIRECT ? 4u * ((unsigned)(4 * h) * (unsigned)a.lddx + (unsigned)l31)
                         : 4u * ((unsigned)prow * (unsigned)a.lddx + 4u * (unsigned)pc4);
    // The epilogue of block b - 1 (tanh' from the X values read while that block was resident, then the stores) is
    // issued BETWEEN the MFMA groups of block b: its vector instructions cost their issue slots wherever they stand
    // (exact-fp32 MFMAs and VALU do not overlap on a SIMD), but its LDS round trips and the store issue are no phase of
    // their own in which this SIMD's other wave waits at the barrier.
    f32x16 acc_prev;
    float xv_prev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_prev[r] = 0.f; xv_prev[r] = 0.f; }
    f32x16 lxacc[2];
    float lxb = 0.f, pb0[16], pb1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { lxacc[0][r] = 0.f; lxacc[1][r] = 0.f; pb0[r] = 0.f; pb1[r] = 0.f; }
    // B operands of element r of block bp: X_below[row (r & 3) + 8 (r >> 2) + 4 h of the wave's row half][l31 (+ 32)],
    // 128 contiguous bytes per half-wave straight from L2 (the layer's whole input is a few MB), requested two elements ahead
    // (address = wave-uniform row pointer, scalar unit, + ONE 32-bit lane offset: per-lane 64-bit row pointers cost 32 registers)
    const float* lxbase = LOWX ? a.lx_X + ((long long)b0 * RB_BR + 32 * rt) * a.lx_ld : nullptr;
    unsigned lxo = 4u * ((unsigned)(4 * h) * (unsigned)a.lx_ld + (unsigned)l31);
    auto lx_load = [&](int bp, int r) {
      const char* q = reinterpret_cast<const char*>(uniform_ptr(lxbase + ((long long)bp * RB_BR + (r & 3) + 8 * (r >> 2)) * a.lx_ld));
      asm volatile("" : "+v"(lxo));
      pb0[r] = *reinterpret_cast<const float*>(q + lxo);
      pb1[r] = *reinterpret_cast<const float*>(q + lxo + 128);
    };
    float vprev[16];                   // LOWX: the previous block's finished data gradient (kept instead of acc_prev + xv_prev)
#pragma unroll
    for (int r = 0; r < 16; ++r) vprev[r] = 0.f;
    auto epi_elem = [&](int bp, int r) {     // accumulator layout: column l31, rows (r & 3) + 8 (r >> 2) + 4 h
      const float v = LOWX ? vprev[r] : acc_prev[r] * (1.0f - xv_prev[r] * xv_prev[r]);
      if (LOWX) {
        lxb += v;
        lxacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, pb0[r], lxacc[0], 0, 0, 0);
        lxacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, pb1[r], lxacc[1], 0, 0, 0);
      } else if (DIRECT) {
        char* ob = reinterpret_cast<char*>(dX + ((long long)bp * RB_BR + 32 * rt) * a.lddx + 32 * ct);
        asm volatile("" : "+v"(lo));
        *reinterpret_cast<float*>(ob + (size_t)((r & 3) + 8 * (r >> 2)) * a.lddx * 4 + lo) = v;
      } else {
        park[((r & 3) + 8 * (r >> 2) + 4 * h) * RB_EPLD + l31] = v;
      }
    };
    f32x4 pv;
    auto epi_read = [&](int it) { pv = *reinterpret_cast<const f32x4*>(park + (8 * it + prow) * RB_EPLD + 4 * pc4); };
    auto epi_store = [&](int bp, int it) {
      char* ob = reinterpret_cast<char*>(dX + ((long long)bp * RB_BR + 32 * rt) * a.lddx + 32 * ct);
      asm volatile("" : "+v"(lo));
      *reinterpret_cast<f32x4*>(ob + (size_t)it * 8 * a.lddx * 4 + lo) = pv;
    };
    auto block = [&](int b, auto stg) {
      constexpr int S = decltype(stg)::value;
      if (b > 0) {                     // (block 0 was published by the prologue's barriers)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      const float* zs = smem + S * RB_STAGE;
      const float* xs = zs + RB_Z_FLOATS;
      if constexpr (LATZ) latz_head(zs, std::false_type{});
      // this block's tanh' operands, for the epilogue that rides in the NEXT block (read now: the stage is refilled then)
      const float* xc = xs + (32 * rt + 4 * h) * RB_S + 32 * ct + l31;
      float xv[16];
      if (!LOWX) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xv[r] = xc[((r & 3) + 8 * (r >> 2)) * RB_S];
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      f32x4 fa[2];
      // (LOWX: the sixteen fragment offsets are recomputed, two vector instructions each, instead of held in registers)
      const int zrow = (32 * rt + l31) * RB_KO, zsw = (32 * rt + l31) & 15;
      auto zoffs = [&](int c) { return LOWX ? zrow + 4 * ((2 * c + h) ^ zsw) : zq[c]; };
      fa[0] = *reinterpret_cast<const f32x4*>(zs + zoffs(0));
      const bool epi = b > 0 && !(a.variant & 8);
      if (LOWX && epi) { lx_load(b - 1, 0); lx_load(b - 1, 1); }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        // the NEXT group's fragment is requested before this group's MFMAs and nothing may cross (left alone, the
        // scheduler sinks every LDS read to just in front of its use: read, lgkmcnt(0), two MFMAs, read, ...)
        if (c < 15) fa[(c + 1) & 1] = *reinterpret_cast<const f32x4*>(zs + zoffs(c + 1));
        if (epi && MODE == 0 && c >= 9 && c < 13) epi_read(c - 9);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][j], wf[4 * c + j], acc, 0, 0, 0);
        if (epi) {
          if (LOWX) { if (c + 2 < 16) lx_load(b - 1, c + 2); epi_elem(b - 1, c); }
          else if (DIRECT) epi_elem(b - 1, c);
          else if (c < 8) { epi_elem(b - 1, 2 * c); epi_elem(b - 1, 2 * c + 1); }
          else if (c >= 9 && c < 13) epi_store(b - 1, c - 9);    // the read was requested in front of this group's MFMAs
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LOWX) {
        // (register budget: the finished values are formed here, behind the block's MFMAs -- 16 LDS reads and 48 vector
        // instructions that do not ride between MFMA groups -- instead of keeping accumulators AND tanh' operands alive
        // through the next block beside the second product's 32 accumulator registers)
#pragma unroll
        for (int r = 0; r < 16; ++r) xv[r] = xc[((r & 3) + 8 * (r >> 2)) * RB_S];
#pragma unroll
        for (int r = 0; r < 16; ++r) vprev[r] = acc[r] * (1.0f - xv[r] * xv[r]);
      } else {
        acc_prev = acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) xv_prev[r] = xv[r];
      }
    };
    int b = 0;
    for (; b + 2 <= nb; b += 2) {
      block(b, std::integral_constant<int, 0>{});
      block(b + 1, std::integral_constant<int, 1>{});
    }
    if (b < nb) block(b, std::integral_constant<int, 0>{});
    __syncthreads();                 // (the weight-gradient waves park their tiles in stage 0 behind this)
    if constexpr (LATZ) {            // the latent layer's gradient record of this row range: the four row groups meet here
      float* rec = a.lz_rec + (long long)range * a.lz_srec;
      float s3 = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) s3 += lzx[g * 256 + tid];     // [k][32 columns] of the four weight-gradient waves, in order
      rec[(tid >> 5) * 128 + 32 * slice + (tid & 31)] = s3;
      if (slice == 0 && tid < 8) {
        float sdb = 0.f;
        for (int j = 0; j < 32; ++j) sdb += dlb[tid + 8 * j];
        rec[8 * 128 + tid] = sdb;
      }
    }
    if (!(a.variant & 8)) {          // the last block's epilogue, beside the weight-gradient waves' final stores
      if (LOWX) { lx_load(nb - 1, 0); lx_load(nb - 1, 1); }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (LOWX && r + 2 < 16) lx_load(nb - 1, r + 2);
        epi_elem(nb - 1, r);
      }
      if (MODE == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) { epi_read(it); epi_store(nb - 1, it); }
      }
    }
    if (LOWX) {
      // the two row halves (rt = 0, 1) of a column half meet: rt = 1 parks [32 c][64 j] + its bias sums, rt = 0 adds and
      // stores the workgroup's partial record (accumulator layout: lane = input column j, register = output c)
      constexpr int XLD = 64 + 2;
      float* ex = smem + 2 * RB_STAGE + ct * (32 * XLD + 32);
      const float bsum2 = lxb + __shfl_xor(lxb, 32, 64);
      if (rt == 1) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) ex[((r & 3) + 8 * (r >> 2) + 4 * h) * XLD + 32 * n + l31] = lxacc[n][r];
        if (h == 0) ex[32 * XLD + l31] = bsum2;
      }
      __syncthreads();
      if (rt == 0) {
        float* wout = a.lx_W + range * a.lx_sPart + (long long)(c0 + 32 * ct) * a.lx_ldw;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
            wout[(long long)c * a.lx_ldw + 32 * n + l31] = lxacc[n][r] + ex[c * XLD + 32 * n + l31];
          }
        if (h == 0) a.lx_B[range * a.lx_bsPart + c0 + 32 * ct + l31] = bsum2 + ex[32 * XLD + l31];
      }
    }
    return;
  }

  // ---- weight-gradient waves
  // LDS-DMA requests of one block: dZ = 32 pieces of two 512-byte rows (piece i = w4 + 4 q), X = 16 pieces of four
  // 256-byte row segments; twelve per wave.  Wave-uniform block address (scalar unit) + a lane offset fixed for the
  // kernel (the swizzle term of dZ depends on the parity of q: two offsets).
  const int zr0 = 2 * w4 + (lane >> 5);                    // row of piece w4 (+ 8 q)
  unsigned zoff[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    zoff[par] = 4u * ((unsigned)zr0 * (unsigned)a.ldz + 4u * (unsigned)((lane & 31) ^ ((zr0 + 8 * par) & 15)));
  unsigned xoff = 4u * ((unsigned)(4 * w4 + (lane >> 4)) * (unsigned)a.ldx + 4u * (unsigned)(lane & 15));   // (+ 16 q rows)
  auto issue = [&](int b, auto stg) {
    float* st = smem + decltype(stg)::value * RB_STAGE;
    const char* zb = reinterpret_cast<const char*>(uniform_ptr(dZ + (long long)b * RB_BR * a.ldz));
    const char* xb = reinterpret_cast<const char*>(uniform_ptr(X + (long long)b * RB_BR * a.ldx));
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      asm volatile("" : "+v"(zoff[q & 1]));   // keeps "scalar base + zext(lane offset)" together (see DmaPtrs)
      dma16(reinterpret_cast<const float*>(zb + (size_t)q * 8 * a.ldz * 4 + zoff[q & 1]), st + 256 * (w4 + 4 * q));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      asm volatile("" : "+v"(xoff));
      dma16(reinterpret_cast<const float*>(xb + (size_t)q * 16 * a.ldx * 4 + xoff), st + RB_Z_FLOATS + 256 * (w4 + 4 * q));
    }
  };
  issue(0, std::integral_constant<int, 0>{});
  // the slice of W, [128 k][64 columns], rides to the still idle stage 1 the same way (32 pieces of four 256-byte rows):
  // as 64 strided loads per lane straight into the data-gradient waves' registers it cost 2.5 us at the head of every
  // workgroup (tools/probes/rb_level_probe.hip)
  {
    const unsigned woff = 4u * ((unsigned)(4 * w4 + (lane >> 4)) * (unsigned)a.ldw + 4u * (unsigned)(lane & 15));
    const char* wb = reinterpret_cast<const char*>(uniform_ptr(W));
#pragma unroll
    for (int q = 0; q < 8; ++q)
      dma16(reinterpret_cast<const float*>(wb + (size_t)q * 16 * a.ldw * 4 + woff), smem + RB_STAGE + 256 * (w4 + 4 * q));
  }
  // LATZ: this thread's two (row, k) pairs of the block's latent gradient: pair p = wt + 256 i = row * 8 + k
  const int wt = tid - 256;
  float lzp[2][8], lzl[2] = {0.f, 0.f};
  float dbacc = 0.f;                  // this thread's share of the latent layer's bias gradient (k = wt & 7)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 8; ++t) lzp[i][t] = 0.f;
  auto lz_fetch = [&](int b) __attribute__((always_inline)) {        // the row-dot partials and latent values of block b of this range
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pr = wt + 256 * i;
      const long long grow = (long long)(b0 + b) * RB_BR + (pr >> 3);
      const float* pp = a.lz_parts + grow * 8 + (pr & 7);
#pragma unroll
      for (int t = 0; t < 8; ++t) lzp[i][t] = t < a.lz_tiles ? pp[(long long)t * a.lz_tstride] : 0.f;
      lzl[i] = a.lz_lat[grow * a.lz_ldlat + (pr & 7)];
    }
  };
  auto lz_publish = [&]() __attribute__((always_inline)) {           // k_latent_bwd's expressions: partials added in tile order, times tanh'
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float acc_ = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < a.lz_tiles) acc_ += lzp[i][t];
      const float dl = acc_ * (1.0f - lzl[i] * lzl[i]);
      dlb[wt + 256 * i] = dl;
      dlt[((wt + 256 * i) & 7) * RB_DLT + ((wt + 256 * i) >> 3)] = dl;
      dbacc += dl;
    }
  };
  if constexpr (LATZ) {
    for (int e = wt; e < 8 * 128; e += 256) w3s[e] = a.lz_W3[e];
    for (int e = wt; e < 8 * RB_DLT; e += 256) dlt[8 * RB_DLT + e] = 0.f;
    lz_fetch(0);
    lz_publish();
  }
  const int ot = w4;
  f32x16 accw[2];
  float bsum = 0.f;
  const bool do_bias = (slice == 0) && (a.dBp != nullptr);
#pragma unroll
  for (int r = 0; r < 16; ++r) { accw[0][r] = 0.f; accw[1][r] = 0.f; }
  int za[2][4];                      // dZ image offsets of (column 32 ot + l31, row 4 h + j) for chunk c even / odd
  {
    const int o = 32 * ot + l31;
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 8 * par + 4 * h + j;           // (row & 15) of chunk c with (c & 1) == par
        za[par][j] = (4 * h + j) * RB_KO + 4 * ((o >> 2) ^ r) + (o & 3);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (LATZ: this wave's dl / W3 writes)
  __builtin_amdgcn_s_barrier();       // block 0 and W are in LDS for everyone
  __builtin_amdgcn_s_barrier();       // the data-gradient waves have copied W out of stage 1
  asm volatile("" ::: "memory");
  {
    auto block = [&](int b, auto stg, auto bias_c) __attribute__((always_inline)) {
      constexpr int S = decltype(stg)::value;
      constexpr bool BIAS = decltype(bias_c)::value;    // a compile-time copy of do_bias: the compiler if-converts the run-time test
      if (b > 0) {
        if constexpr (LATZ) lz_publish();                   // dl of block b (fetched a block ago): its readers of block b - 1 are done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // block b (requested a block ago) has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // ... for everyone, and everyone is past the other stage
        asm volatile("" ::: "memory");
      }
      const float* zs = smem + S * RB_STAGE;
      if constexpr (LATZ) {
        latz_head(zs, std::true_type{});
        if (b + 1 < nb) lz_fetch(b + 1);
      }
      const float* xb_ = zs + RB_Z_FLOATS + 4 * h * RB_S + l31;
      float fa[2][4], fb0[2][4], fb1[2][4];
      auto load = [&](int c, float (&A)[4], float (&B0)[4], float (&B1)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          A[j] = zs[8 * c * RB_KO + za[c & 1][j]];
          B0[j] = xb_[(8 * c + j) * RB_S];
          B1[j] = xb_[(8 * c + j) * RB_S + 32];
        }
      };
      load(0, fa[0], fb0[0], fb1[0]);
      if (b + 1 < nb) issue(b + 1, std::integral_constant<int, 1 - S>{});
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < 7) load(c + 1, fa[(c + 1) & 1], fb0[(c + 1) & 1], fb1[(c + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);   // (see the data-gradient loop)
        if constexpr (BIAS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bsum += fa[c & 1][j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][j], fb0[c & 1][j], accw[0], 0, 0, 0);
          accw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][j], fb1[c & 1][j], accw[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto run = [&](auto bias_c) {
      int b = 0;
      for (; b + 2 <= nb; b += 2) {
        block(b, std::integral_constant<int, 0>{}, bias_c);
        block(b + 1, std::integral_constant<int, 1>{}, bias_c);
      }
      if (b < nb) block(b, std::integral_constant<int, 0>{}, bias_c);
    };
    if (do_bias) run(std::true_type{}); else run(std::false_type{});
  }

  if constexpr (LATZ) {   // this half's share of the latent layer's gradients, for the data-gradient waves behind the barrier
    if (lane < 32) {                   // lanes q < 2 hold k = 4 q + r
      const int n = lane & 15, q = lane >> 4;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) lzx[w4 * 256 + (4 * q + r) * 32 + 16 * jt + n] = acc3[jt][r];
    }
    dlb[wt] = dbacc;
  }
  // ---- the workgroup's weight-gradient partial: accumulators -> LDS (stage 0 is idle) -> 16-byte stores
  __syncthreads();
  if (!((a.variant & 4) && accw[0][0] != 12345.f)) {
    float* wp = smem + ot * (32 * RB_WPLD);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        wp[((r & 3) + 8 * (r >> 2) + 4 * h) * RB_WPLD + 32 * n + l31] = accw[n][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* out = a.dWp + range * a.sWpart + net * a.sWnet + (long long)(32 * ot) * a.ldwp + c0;
    const int prow = lane >> 4, pc4 = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(wp + (4 * it + prow) * RB_WPLD + 4 * pc4);
      *reinterpret_cast<f32x4*>(out + (long long)(4 * it + prow) * a.ldwp + 4 * pc4) = v;
    }
    if (do_bias) {
      // lane (l31, h) summed rows 4 h + j (mod 8) of column 32 ot + l31: add the two halves in fixed order
      const float other = __shfl_xor(bsum, 32, 64);
      if (h == 0) a.dBp[range * a.sBpart + net * a.sBnet + 32 * ot + l31] = bsum + other;
    }
  }
  if (LOWX) __syncthreads();          // (the data-gradient waves' exchange of their second product)
}

static hipError_t rb_level_backward(RbLevelArgs a, hipStream_t s, int prof_class) {
  if (!rb_level_shape_ok(a.rows, RB_KO, a.IN, a.nets)) return hipErrorNotSupported;
  if (!aligned16(a.dZ) || !aligned16(a.X) || (!a.lx_W && (!a.dX || !aligned16(a.dX))) || !aligned16(a.dWp) || (a.ldz & 3) || (a.ldx & 3) ||
      (a.lddx & 3) || (a.ldwp & 3) || (a.sZ & 3) || (a.sX & 3) || (a.sdX & 3) || (a.sWpart & 3) || (a.sWnet & 3) ||
      a.ldz < RB_KO || a.ldx < a.IN || a.lddx < a.IN || a.ldwp < a.IN || a.ldw < a.IN ||
      !aligned16(a.W) || (a.ldw & 3) || (a.sW & 3) ||
      (long long)RB_BR * a.ldz * 4 >= (1LL << 31) || (long long)RB_BR * a.ldx * 4 >= (1LL << 31))
    return hipErrorNotSupported;
  a.nslices = a.IN / RB_S;
  a.nblocks = a.rows / RB_BR;
  static int variant = -1;
  if (variant < 0) { const char* e = getenv("IGI_RB_VARIANT"); variant = e ? atoi(e) : 0; }
  a.variant = variant;
  if (a.ranges < 1 || a.ranges > a.nblocks) return hipErrorInvalidValue;
  const bool lowx = a.lx_W != nullptr;
  if (lowx && (a.nets != 1 || !a.lx_X || !a.lx_B || a.lx_ld < 64 || a.lx_ldw < 64)) return hipErrorInvalidValue;
  const bool latz = a.lz_parts != nullptr;
  if (latz && (!lowx || !a.lz_lat || !a.lz_W3 || !a.lz_rec || a.lz_tiles < 1 || a.lz_tiles > 8 || a.lz_srec < 8 * 128 + 8 ||
               a.lz_ldlat < 8 || a.ldz != RB_KO))
    return hipErrorInvalidValue;
  static bool attr = false;
  if (!attr) {
    const void* ks[5] = {(const void*)k_rb_level<0, 1>, (const void*)k_rb_level<0, 2>, (const void*)k_rb_level<1, 1>,
                         (const void*)k_rb_level<1, 2>, (const void*)k_rb_level<2, 1>};
    for (const void* k : ks) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * RB_LDS_FLOATS));
      if (e != hipSuccess) return e;
    }
    hipError_t e = hipFuncSetAttribute((const void*)k_rb_level<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(float) * RB_LDS_FLOATS_LATZ));
    if (e != hipSuccess) return e;
    attr = true;
  }
  const double fl = 4.0 * a.nets * (double)a.rows * RB_KO * a.IN + (lowx ? 2.0 * (double)a.rows * a.IN * 64 : 0.0) +
                    (latz ? 4.0 * (double)a.rows * RB_KO * 8 : 0.0);   // (the rank-8 products counted once, not per column slice)
  const double by = 4.0 * a.nets * ((double)a.rows * (RB_KO + 2.0 * a.IN) + (double)RB_KO * a.IN * (1 + a.ranges));
  ProfScope ps(prof_class, s, fl, by);
  const dim3 grid(a.nets * a.ranges * a.nslices);
  const size_t shm = sizeof(float) * RB_LDS_FLOATS;
  if (latz) IGI_LAUNCH((k_rb_level<3, 1>), grid, dim3(RB_THREADS), sizeof(float) * RB_LDS_FLOATS_LATZ, s, a);
  else if (lowx) IGI_LAUNCH((k_rb_level<2, 1>), grid, dim3(RB_THREADS), shm, s, a);
  else if (a.variant & 32) {
    if (a.nets == 1) IGI_LAUNCH((k_rb_level<1, 1>), grid, dim3(RB_THREADS), shm, s, a);
    else IGI_LAUNCH((k_rb_level<1, 2>), grid, dim3(RB_THREADS), shm, s, a);
  } else {
    if (a.nets == 1) IGI_LAUNCH((k_rb_level<0, 1>), grid, dim3(RB_THREADS), shm, s, a);
    else IGI_LAUNCH((k_rb_level<0, 2>), grid, dim3(RB_THREADS), shm, s, a);
  }
  return hipGetLastError();
}

}  // namespace igi
