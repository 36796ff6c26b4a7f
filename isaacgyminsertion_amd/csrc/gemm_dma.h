// Pipelined exact-fp32 MFMA GEMM for the large, aligned contractions (the bulk of the FLOPs):
// same contract as gemm_f32.h (C[m][n] = sum_k A(m,k) B(n,k), batched / split-k, fused epilogues)
// but the operand tiles travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no
// ds_write pass) through a 3-deep ring, so two k-tiles (64-96 KB per CU) are always in flight
// while the MFMAs of the current one run: the register-staged kernel exposed the HBM/L2 latency
// once per k-tile (measured: MFMA pipe 32-57 % busy, waves parked in s_waitcnt).
//
//   workgroup  : 512 threads = 8 waves, two per SIMD (while one wave computes addresses / waits on
//                LDS the other's MFMAs keep the pipe busy), tile 128 x BN (BN = 64 | 128 | 256),
//                k-tile 32; waves 2x4 (BN >= 128: 64x64 | 64x32 per wave) or 4x2 (BN = 64: 32x32)
//   LDS ring   : 3 stages x (A 16 KB + B 16|32 KB) = 96 | 144 KB  -> one workgroup per CU
//   k-contiguous operand  : LDS image [row][8 units of 16 B], unit p of row m holds k-group
//                p ^ ((m>>1)&7) (XOR swizzle applied on the per-lane SOURCE address; the LDS-DMA
//                destination is lane-linear) -> ds_read_b128 of 4 consecutive k, conflict-free
//   reduction-major operand: LDS image [32 k][rows] copied verbatim -> ds_read_b32, conflict-free
//   MFMA k order: for k-chunk c (8 wide) and lane half h, step j uses k = 8c + 4h + j for BOTH
//                operands (any bijection is valid as long as A and B agree)
//   schedule   : per k-tile  s_waitcnt vmcnt(<loads of one tile>) ; s_barrier ; issue tile t+2 ;
//                64|128 MFMAs.  The single barrier both publishes tile t and frees stage (t-1)%3.
// Requirements (checked by the launcher, otherwise gemm_f32.h runs): M,N >= 4, every split's k-range
// a multiple of 32, 16-byte aligned bases / leading dimensions, N % 4 == 0 for row-major operands.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gemm_f32.h"
#include "dma_util.h"
#include "rowblock.h"

namespace igi {

constexpr int DMA_BK = 32;
constexpr int DMA_BM = 128;
constexpr int DMA_NS = 3;

// Issue this wave's share of one operand tile (ROWS x 32 floats) into `stage`.
constexpr int DMA_WAVES = 8;
constexpr int DMA_THREADS = DMA_WAVES * 64;

template <int ROWS, bool KC>
__device__ __forceinline__ void dma_tile(const float* __restrict__ src, int ld, int r0, int rmax,
                                         int k0, float* stage, int wave, int lane) {
  constexpr int NINSTR = ROWS * DMA_BK * 4 / 1024;  // 1 KiB per wave-instruction
  static_assert(NINSTR % DMA_WAVES == 0 || NINSTR < DMA_WAVES, "tile must split evenly over the waves");
  constexpr int NQ = NINSTR >= DMA_WAVES ? NINSTR / DMA_WAVES : 1;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int i = wave + DMA_WAVES * q;
    if (NINSTR < DMA_WAVES && i >= NINSTR) break;  // a 32-row tile is four instructions: waves 0-3 fetch it
    const float* g;
    if (KC) {
      const int m = 8 * i + (lane >> 3);
      const int p = lane & 7;
      const int k4 = p ^ ((m >> 1) & 7);
      const int r = min(r0 + m, rmax - 1);
      g = src + (long long)r * ld + k0 + 4 * k4;
    } else {
      const int f = 256 * i + 4 * lane;
      const int k = f / ROWS;
      const int m = f % ROWS;
      const int r = min(r0 + m, rmax - 4);
      g = src + (long long)(k0 + k) * ld + r;
    }
    dma16(g, stage + 256 * i);
  }
}

// The same tile with the per-lane source offsets computed ONCE per output tile (they only advance along k).  The
// address of a DMA instruction is (wave-uniform base + k advance, on the scalar unit) + (32-bit lane offset): the
// `global_load_lds_dwordx4 v_off, s[base]` form, i.e. no vector instruction per k-tile -- beside exact-fp32 MFMAs
// every vector instruction is paid in full (issue-side counters), and the 64-bit per-lane pointer
// version spent six to eight of them per k-tile.
template <int ROWS, bool KC>
struct DmaPtrs {
  static constexpr int NINSTR = ROWS * DMA_BK * 4 / 1024;
  static constexpr int NQ = NINSTR >= DMA_WAVES ? NINSTR / DMA_WAVES : 1;
  const float* base;   // wave-uniform: first row / column of the tile
  unsigned off[NQ];    // bytes from base (a tile spans <= 256 rows: < 4 GB for any row pitch the launchers accept)
  long long kstep;     // elements per unit of k
};
template <int ROWS, bool KC>
__device__ __forceinline__ void dma_ptrs_init(DmaPtrs<ROWS, KC>& d, const float* __restrict__ src, int ld, int r0,
                                              int rmax, int wave, int lane) {
  d.kstep = KC ? 1 : ld;
  d.base = uniform_ptr(KC ? src + (long long)r0 * ld : src + r0);
#pragma unroll
  for (int q = 0; q < DmaPtrs<ROWS, KC>::NQ; ++q) {
    const int i = wave + DMA_WAVES * q;
    if (KC) {
      const int m = 8 * i + (lane >> 3);
      const int k4 = (lane & 7) ^ ((m >> 1) & 7);
      const int r = min(m, rmax - 1 - r0);
      d.off[q] = 4u * ((unsigned)r * (unsigned)ld + 4u * (unsigned)k4);
    } else {
      const int f = 256 * i + 4 * lane;
      const int k = f / ROWS;
      const int m = f % ROWS;
      const int r = min(m, rmax - 4 - r0);
      d.off[q] = 4u * ((unsigned)k * (unsigned)ld + (unsigned)r);
    }
  }
}
template <int ROWS, bool KC>
__device__ __forceinline__ void dma_ptrs_issue(DmaPtrs<ROWS, KC>& d, int k0, float* stage, int wave) {
  const char* b = reinterpret_cast<const char*>(d.base + (long long)k0 * d.kstep);
#pragma unroll
  for (int q = 0; q < DmaPtrs<ROWS, KC>::NQ; ++q) {
    const int i = wave + DMA_WAVES * q;
    if (DmaPtrs<ROWS, KC>::NINSTR < DMA_WAVES && i >= DmaPtrs<ROWS, KC>::NINSTR) break;
    // the empty asm keeps the zero-extension of the offset next to the add: hoisted out of the k-loop (it is
    // loop-invariant) the instruction selector no longer sees "scalar base + zext(32-bit lane offset)" and falls
    // back to a 64-bit vector add per instruction
    asm volatile("" : "+v"(d.off[q]));
    dma16(reinterpret_cast<const float*>(b + d.off[q]), stage + 256 * i);
  }
}

// im2col gather, k-contiguous operand (conv forward / dgrad): row m = (b, oy, ox); k-tile kt is one
// kernel row of 8 four-channel pixels (C == 4, KW == 8) or one 32-channel slice of tap kt / (C/32).
// The rows a lane fetches do not change along k, so their (image, y, x) decomposition is done once per
// output tile (GatherRows); per k-tile only the wave-uniform tap offset is added.
template <int ROWS>
struct GatherRows {
  static constexpr int NQ = ROWS * DMA_BK * 4 / 1024 / DMA_WAVES;
  int base[NQ];   // element offset of input pixel (oy*stride - pad, ox*stride - pad) of the row's image
  int iy0[NQ], ix0[NQ];
  unsigned off[NQ];   // fast32: byte offset of the lane's 16 bytes inside tap (0, 0): 4 * (base + 4 * k4)
};

template <int ROWS>
__device__ __forceinline__ void gather_rows_init(GatherRows<ROWS>& gr, const ConvDesc& cd, int r0, int rmax,
                                                 int wave, int lane) {
#pragma unroll
  for (int q = 0; q < GatherRows<ROWS>::NQ; ++q) {
    const int i = wave + DMA_WAVES * q;
    const int m = 8 * i + (lane >> 3);
    const int r = min(r0 + m, rmax - 1);
    const int b = fdiv(r, cd.dOHW);
    const int rem = r - b * cd.OHW;
    const int oy = fdiv(rem, cd.dOW);
    const int ox = rem - oy * cd.OW;
    gr.iy0[q] = oy * cd.stride - cd.pad;
    gr.ix0[q] = ox * cd.stride - cd.pad;
    if (cd.planar) {   // the lane's 16 bytes of a k-tile: kernel row k4 / 2 (of the tile's four), pixels 4 * (k4 % 2) ..+3
      const int k4 = (lane & 7) ^ ((m >> 1) & 7);
      gr.base[q] = (b * cd.C * cd.IH + gr.iy0[q]) * cd.IW + gr.ix0[q];
      gr.off[q] = 4u * (unsigned)(gr.base[q] + (k4 >> 1) * cd.IW + 4 * (k4 & 1));
      continue;
    }
    gr.base[q] = ((b * cd.IH + gr.iy0[q]) * cd.IW + gr.ix0[q]) * cd.C;
    gr.off[q] = 4u * (unsigned)(gr.base[q] + 4 * ((lane & 7) ^ ((m >> 1) & 7)));
  }
}

template <int ROWS>
__device__ __forceinline__ void dma_tile_gather_kc(const float* __restrict__ src, const ConvDesc& cd,
                                                   GatherRows<ROWS>& gr, int kt, float* stage, int wave,
                                                   int lane) {
  // wave-uniform part of the tap
  int ky, kx0, coff0;
  if (cd.planar) {   // k-tile kt = plane kt / 2, kernel rows 4 * (kt % 2) ..+3; always the test-free path (dma_eligible)
    const char* b = reinterpret_cast<const char*>(src + ((kt >> 1) * cd.IH + 4 * (kt & 1)) * cd.IW);
#pragma unroll
    for (int q = 0; q < GatherRows<ROWS>::NQ; ++q) {
      asm volatile("" : "+v"(gr.off[q]));
      dma16(reinterpret_cast<const float*>(b + gr.off[q]), stage + 256 * (wave + DMA_WAVES * q));
    }
    return;
  }
  if (cd.C == 4) { ky = kt; kx0 = 0; coff0 = 0; }
  else {
    const int pix = fdiv(kt, cd.dTPP);
    coff0 = (kt - pix * (cd.C >> 5)) * 32;
    ky = fdiv(pix, cd.dKW);
    kx0 = pix - ky * cd.KW;
  }
  if (cd.fast32) {
    // unpadded convolution (every forward layer): each tap of each output position is inside the image, so a request
    // is (tap address, scalar) + (the lane's byte offset, fixed for the tile) -- one vector instruction less than
    // nothing: none.  The tested path below costs ~16 per request, and beside fp32 MFMAs each is paid in full.
    const char* b = reinterpret_cast<const char*>(src + ((ky * cd.IW + kx0) * cd.C + coff0));
#pragma unroll
    for (int q = 0; q < GatherRows<ROWS>::NQ; ++q) {
      asm volatile("" : "+v"(gr.off[q]));   // keeps the zero-extension next to the add (see DmaPtrs)
      dma16(reinterpret_cast<const float*>(b + gr.off[q]), stage + 256 * (wave + DMA_WAVES * q));
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < GatherRows<ROWS>::NQ; ++q) {
    const int i = wave + DMA_WAVES * q;
    const int m = 8 * i + (lane >> 3);
    const int k4 = (lane & 7) ^ ((m >> 1) & 7);
    const int kx = (cd.C == 4) ? k4 : kx0;
    const int coff = (cd.C == 4) ? 0 : coff0 + 4 * k4;
    const int iy = gr.iy0[q] + ky, ix = gr.ix0[q] + kx;
    const bool inb = (iy >= 0) && (iy < cd.IH) && (ix >= 0) && (ix < cd.IW);
    const float* g = inb ? src + (gr.base[q] + (ky * cd.IW + kx) * cd.C + coff) : cd.zero;
    dma16(g, stage + 256 * i);
  }
}

// im2col gather, reduction-major operand (conv wgrad): LDS image [32 rows m][ROWS taps]; element
// (tap n = (ky,kx,c), row m) = input pixel of output position m at tap (ky,kx), channel c.  The taps
// a lane fetches are fixed along k (GatherTaps, once per tile); the row index advances by 32 per k-tile.
template <int ROWS>
struct GatherTaps {
  static constexpr int NQ = ROWS * DMA_BK * 4 / 1024 / DMA_WAVES;
  int ky[NQ], kx[NQ], toff[NQ];  // toff = (ky*IW + kx)*C + c
};

template <int ROWS>
__device__ __forceinline__ void gather_taps_init(GatherTaps<ROWS>& gt, const ConvDesc& cd, int n0, int wave,
                                                 int lane) {
#pragma unroll
  for (int q = 0; q < GatherTaps<ROWS>::NQ; ++q) {
    const int i = wave + DMA_WAVES * q;
    const int f = 256 * i + 4 * lane;
    const int n = min(n0 + (f % ROWS), cd.ntaps - 4);
    const int pix = fdiv(n, cd.dC);
    const int c = n - pix * cd.C;
    gt.ky[q] = fdiv(pix, cd.dKW);
    gt.kx[q] = pix - gt.ky[q] * cd.KW;
    gt.toff[q] = (gt.ky[q] * cd.IW + gt.kx[q]) * cd.C + c;
  }
}

template <int ROWS>
__device__ __forceinline__ void dma_tile_gather_rm(const float* __restrict__ src, const ConvDesc& cd,
                                                   const GatherTaps<ROWS>& gt, int k0, float* stage, int wave,
                                                   int lane) {
#pragma unroll
  for (int q = 0; q < GatherTaps<ROWS>::NQ; ++q) {
    const int i = wave + DMA_WAVES * q;
    const int f = 256 * i + 4 * lane;
    const int r = k0 + f / ROWS;
    const int b = fdiv(r, cd.dOHW);
    const int rem = r - b * cd.OHW;
    const int oy = fdiv(rem, cd.dOW);
    const int ox = rem - oy * cd.OW;
    const int iy0 = oy * cd.stride - cd.pad, ix0 = ox * cd.stride - cd.pad;
    const int iy = iy0 + gt.ky[q], ix = ix0 + gt.kx[q];
    const bool inb = (iy >= 0) && (iy < cd.IH) && (ix >= 0) && (ix < cd.IW);
    const float* g = inb ? src + (((b * cd.IH + iy0) * cd.IW + ix0) * cd.C + gt.toff[q]) : cd.zero;
    dma16(g, stage + 256 * i);
  }
}

// torch.linspace(-1, 1, steps)[i] in fp32 (symmetric evaluation, as ATen does)
__device__ __forceinline__ float ssa_linspace(int i, int steps) {
  const float step = 2.0f / (float)(steps - 1);
  return (i < steps / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(steps - i - 1));
}

// Soft-argmax partials of one finished 256 x 64 tile whose activated values sit in the waves' LDS slices
// ([wave = wm * WGN + wn][WTM rows][WTN + 4]): thread (c = tid % 64, rg = tid / 64) owns channel c of rows 32 rg .. +31
// (one image: P % 32 == 0 and the tile starts on a multiple of 256) and writes (max, sum e, sum e xw, sum e yw) with the
// reference's coordinate quirk (flat position k of the h x w map: x-weight linspace(w)[k / h], y-weight linspace(h)[k % h];
// tactile_cnn.py:32-58).  Positions are visited in order, eight LDS reads in flight.
template <int WTM, int WTN, int WGN>
__device__ __forceinline__ void ssa_tile_partials(const GemmArgs& g, const float* smem, int m0, int n0, int tid) {
  constexpr int EPLD = WTN + 4;
  const int c = tid & 63, rg = tid >> 6;
  const int wn = c / WTN, cl = c - wn * WTN;
  const int r0 = 32 * rg;
  const int wm = r0 / WTM, rl = r0 - wm * WTM;
  const float* col = smem + (wm * WGN + wn) * (WTM * EPLD) + rl * EPLD + cl;
  // two passes over the group's 32 LDS values (eight in flight) instead of 32 live registers: LDS reads are not vector
  // instructions, and the kernel keeps two workgroups per CU
  float mx = col[0];
#pragma unroll 8
  for (int i = 1; i < 32; ++i) mx = fmaxf(mx, col[i * EPLD]);
  const int k0 = (m0 + r0) % g.ssa_P;   // first position of the group inside its image
  int q = k0 / g.ssa_h, r = k0 - q * g.ssa_h;
  // 100 M exponentials per 8192 images run beside the convolution's MFMAs here, every vector instruction paid in full:
  // v_exp_f32 on (v - max) * log2(e) (arguments in [-max, 0]; ~3e-7 relative, the backward recomputes the softmax from
  // the (max, sum) this produces), linspace steps hoisted, the x weight refreshed only when the position's k / h changes
  const int hw = g.ssa_w, hh = g.ssa_h;
  const float stepw = 2.0f / (float)(hw - 1), steph = 2.0f / (float)(hh - 1);
  auto lin = [](int i, int steps, float step) {
    return (i < steps / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(steps - i - 1));
  };
  float xw = lin(q, hw, stepw);
  float s = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {
    const float e = __builtin_amdgcn_exp2f((col[i * EPLD] - mx) * 1.44269504088896340736f);
    s += e;
    sx += e * xw;
    sy += e * lin(r, hh, steph);
    if (++r == hh) { r = 0; ++q; xw = lin(q, hw, stepw); }
  }
  if (n0 + c < g.N)
    *reinterpret_cast<float4*>(g.ssa_part + ((long long)((m0 + r0) >> 5) * g.N + n0 + c) * 4) = make_float4(mx, s, sx, sy);
}

// Position-major data-gradient tile (GATHER == 4, see gemm_dma_body): tile mt = (image block blk, input position pos).
struct PmTile {
  int blk, pos, iy0, ix0;      // top-left tap of the position in the (dz) input image: (oy * stride - pad, ox * stride - pad)
  unsigned mask, mrem;         // in-image taps (bit ky * KW + kx); the not-yet-requested ones
  int tap, slice, tpp, nk;     // next k-tile to request: (tap, 32-channel slice); slices per tap; k-tiles of this tile
  unsigned off[8];             // per-lane byte offsets of this lane's rows inside a tap: image stride * row + 16 * k-group
};

template <int ROWS>
__device__ __forceinline__ void pm_tile_init(PmTile& t, const ConvDesc& cd, int mt, int wave, int lane) {
  static_assert(ROWS * DMA_BK * 4 / 1024 / DMA_WAVES <= 8, "off[] slots");
  t.blk = __builtin_amdgcn_readfirstlane(fdiv(mt, cd.dOHW));
  // every field of the walk below is wave-uniform: say so (readfirstlane), or the per-k-tile address of pm_issue --
  // a 64-bit multiply-add -- is computed on the vector unit (14 instructions, four of them quarter-rate integer
  // multiplies, per k-tile; beside fp32 MFMAs each is paid in full: 13 % vector-active cycles in the SQ counters)
  t.pos = __builtin_amdgcn_readfirstlane(mt - t.blk * cd.OHW);
  const int oy = __builtin_amdgcn_readfirstlane(fdiv(t.pos, cd.dOW));
  const int ox = __builtin_amdgcn_readfirstlane(t.pos - oy * cd.OW);
  t.iy0 = __builtin_amdgcn_readfirstlane(oy * cd.stride - cd.pad);
  t.ix0 = __builtin_amdgcn_readfirstlane(ox * cd.stride - cd.pad);
  unsigned m = 0;
  const int ntap = cd.KH * cd.KW;
  for (int tp = 0; tp < ntap; ++tp) {   // scalar loop (<= 16 taps)
    const int ky = fdiv(tp, cd.dKW), kx = tp - ky * cd.KW;
    const bool ok = (unsigned)(t.iy0 + ky) < (unsigned)cd.IH && (unsigned)(t.ix0 + kx) < (unsigned)cd.IW;
    m |= ok ? (1u << tp) : 0u;
  }
  t.mask = t.mrem = __builtin_amdgcn_readfirstlane(m);
  t.tpp = cd.C >> 5;
  t.nk = __builtin_popcount(t.mask) * t.tpp;
  t.tap = t.mask ? __builtin_ctz(t.mask) : 0;
  t.slice = 0;
  const unsigned img = (unsigned)(cd.IH * cd.IW * cd.C);   // floats per image (ROWS images < 4 GB: checked by the launcher)
#pragma unroll
  for (int q = 0; q < ROWS * DMA_BK * 4 / 1024 / DMA_WAVES; ++q) {
    const int i = wave + DMA_WAVES * q;
    const int mrow = 8 * i + (lane >> 3);
    t.off[q] = 4u * ((unsigned)mrow * img + 4u * (unsigned)((lane & 7) ^ ((mrow >> 1) & 7)));
  }
}

// requests the A tile of the next in-image (tap, slice) into `stage` and returns that k-tile's first k index in the
// repacked weight (k = (tap * C + c)): the B operand tile is then fetched from there
template <int ROWS>
__device__ __forceinline__ int pm_issue(PmTile& t, const float* __restrict__ src, const ConvDesc& cd, float* stage,
                                        int wave) {
  const int ky = fdiv(t.tap, cd.dKW), kx = t.tap - ky * cd.KW;
  const long long pix = ((long long)t.blk * ROWS * cd.IH + (t.iy0 + ky)) * cd.IW + (t.ix0 + kx);
  const char* b = reinterpret_cast<const char*>(src + pix * cd.C + t.slice * 32);
  const int kb = (t.tap * t.tpp + t.slice) * DMA_BK;
#pragma unroll
  for (int q = 0; q < ROWS * DMA_BK * 4 / 1024 / DMA_WAVES; ++q) {
    asm volatile("" : "+v"(t.off[q]));   // keeps the zero-extension next to the add (see DmaPtrs)
    dma16(reinterpret_cast<const float*>(b + t.off[q]), stage + 256 * (wave + DMA_WAVES * q));
  }
  if (++t.slice == t.tpp) {   // scalar walk over the set bits of the mask
    t.slice = 0;
    t.mrem &= t.mrem - 1;
    t.tap = t.mrem ? __builtin_ctz(t.mrem) : 0;
  }
  return kb;
}

// Weight gradient of an unpadded convolution with the reduction walking the rows POSITION-MAJOR (GATHER == 5): k-tile kt
// = 32 consecutive images at ONE output position (kt = j * OHW + pos: image-group major, so that consecutive k-tiles
// slide the window over the same 32 images and re-read their overlap from L2; position-major over the whole batch made
// the 8 x 8 x 4 conv1 product HBM-bound: 3.2 GB of window fetches per launch, 643 -> 750 us).  The im2col operand's request is then
// (scalar address of that position's window in image 32 j) + (lane offset fixed for the whole kernel: the lane's image
// inside the group and its tap) -- the row-major walk of GATHER == 3 decomposes every request's row into (image, y, x)
// on the vector unit (~25 instructions per request, each paid in full beside exact-fp32 MFMAs).  dZ's rows of a k-tile
// are one image apart.  Same products, summed in another (fixed) order.
template <int ROWS>
struct PwTaps {
  static constexpr int NQ = ROWS * DMA_BK * 4 / 1024 / DMA_WAVES;
  unsigned off[NQ];
};
template <int ROWS>
__device__ __forceinline__ void pw_init(PwTaps<ROWS>& t, const ConvDesc& cd, int n0, int wave, int lane) {
  const unsigned img = (unsigned)(cd.IH * cd.IW * cd.C);
#pragma unroll
  for (int q = 0; q < PwTaps<ROWS>::NQ; ++q) {
    const int i = wave + DMA_WAVES * q;
    const int f = 256 * i + 4 * lane;
    const int krow = f / ROWS;
    const int n = min(n0 + (f % ROWS), cd.ntaps - 4);
    const int pix = fdiv(n, cd.dC);
    const int c = n - pix * cd.C;
    const int ky = fdiv(pix, cd.dKW), kx = pix - ky * cd.KW;
    t.off[q] = 4u * ((unsigned)krow * img + (unsigned)((ky * cd.IW + kx) * cd.C + c));
  }
}
template <int ROWS>
__device__ __forceinline__ void pw_issue(PwTaps<ROWS>& t, const float* __restrict__ src, const ConvDesc& cd, int pos,
                                         int j, float* stage, int wave) {
  const int oy = fdiv(pos, cd.dOW), ox = pos - oy * cd.OW;
  const long long pix = ((long long)(32 * j) * cd.IH + oy * cd.stride) * cd.IW + ox * cd.stride;
  const char* b = reinterpret_cast<const char*>(src + pix * cd.C);
#pragma unroll
  for (int q = 0; q < PwTaps<ROWS>::NQ; ++q) {
    asm volatile("" : "+v"(t.off[q]));
    dma16(reinterpret_cast<const float*>(b + t.off[q]), stage + 256 * (wave + DMA_WAVES * q));
  }
}

// Wide epilogue: a lane of the MFMA C/D layout owns one column and 16 scattered rows, so storing
// straight from the accumulators issues 4-byte stores that touch two 128-B lines per wave-instruction
// (measured: a 1-k-tile launch writing 67 MB took 39 us = 1.7 TB/s, store-issue bound).  Instead each
// wave parks its raw WTM x WTN tile in its own slice of the (now idle) LDS ring and reads it back
// row-major, 16 bytes per lane: bias / saved activations are fetched as float4 and every global store
// wave-instruction covers whole 128..256-B row segments (4x fewer, 4x wider stores).
template <int EPI>
__device__ __forceinline__ float4 epi_apply4(float4 v, float4 b, float4 t) {
  if (EPI == EPI_BIAS_TANH) {
    v.x = fast_tanh(v.x + b.x); v.y = fast_tanh(v.y + b.y); v.z = fast_tanh(v.z + b.z); v.w = fast_tanh(v.w + b.w);
  } else if (EPI == EPI_BIAS) {
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  } else if (EPI == EPI_BIAS_RELU) {
    v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
  } else if (EPI == EPI_TANHGRAD) {
    v.x *= (1.0f - t.x * t.x); v.y *= (1.0f - t.y * t.y); v.z *= (1.0f - t.z * t.z); v.w *= (1.0f - t.w * t.w);
  } else if (EPI == EPI_RELUGRAD) {
    v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f; v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
  } else if (EPI == EPI_BIAS_ELU) {
    v.x = elu1(v.x + b.x); v.y = elu1(v.y + b.y); v.z = elu1(v.z + b.z); v.w = elu1(v.w + b.w);
  } else if (EPI == EPI_ELUGRAD) {
    v.x *= elu1_grad_from_out(t.x); v.y *= elu1_grad_from_out(t.y); v.z *= elu1_grad_from_out(t.z);
    v.w *= elu1_grad_from_out(t.w);
  }
  return v;
}

template <int EPI, int WTM, int WTN, bool KEEP = false, bool STORE = true>
__device__ __forceinline__ void epilogue_rows(float* __restrict__ ep, float* __restrict__ C, int ldc,
                                              const float* __restrict__ bias, const float* __restrict__ aux,
                                              int ldaux, int row0, int col0, int M, int N, int lane) {
  constexpr int EPLD = WTN + 4;
  constexpr int C4 = WTN / 4;        // float4 per tile row
  constexpr int RPI = 64 / C4;       // rows per wave-instruction
  const int c4 = lane % C4, rl = lane / C4;
  const int col = col0 + 4 * c4;
  if (col >= N) return;              // N % 4 == 0 on this path: a float4 is all in or all out
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI == EPI_BIAS_TANH || EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_ELU)
    b = *reinterpret_cast<const float4*>(bias + col);
  // four row groups at a time: all LDS / aux loads are issued (row index clamped) before the first
  // store, so the loop is not a chain of dependent load -> store round trips
  static_assert((WTM / RPI) % 4 == 0, "row groups come in fours");
  if (row0 + WTM <= M) {
    // whole tile rows (wave-uniform test): no clamps, no predicates, and every address is a wave-uniform base (scalar
    // unit) plus ONE 32-bit lane offset computed here -- the general path below spends ~6 vector instructions per
    // 16-byte access on 64-bit row arithmetic, and beside fp32 MFMAs each of them is paid in full
    char* cb = reinterpret_cast<char*>(C + (long long)row0 * ldc + col0);
    const char* ab = reinterpret_cast<const char*>(aux + (long long)row0 * ldaux + col0);
    unsigned lo = 4u * ((unsigned)rl * (unsigned)ldc + 4u * c4), la = 4u * ((unsigned)rl * (unsigned)ldaux + 4u * c4);
    for (int it0 = 0; it0 < WTM / RPI; it0 += 4) {
      float4 v[4], t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const float4*>(ep + ((it0 + u) * RPI + rl) * EPLD + 4 * c4);
        t[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == EPI_TANHGRAD || EPI == EPI_RELUGRAD || EPI == EPI_ELUGRAD) {
          asm volatile("" : "+v"(la));
          t[u] = *reinterpret_cast<const float4*>(ab + (long long)(it0 + u) * RPI * ldaux * 4 + la);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 o = epi_apply4<EPI>(v[u], b, t[u]);
        asm volatile("" : "+v"(lo));
        if (STORE) *reinterpret_cast<float4*>(cb + (long long)(it0 + u) * RPI * ldc * 4 + lo) = o;
        if (KEEP) *reinterpret_cast<float4*>(ep + ((it0 + u) * RPI + rl) * EPLD + 4 * c4) = o;
      }
    }
    return;
  }
  for (int it0 = 0; it0 < WTM / RPI; it0 += 4) {
    float4 v[4], t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = (it0 + u) * RPI + rl;
      const int rowc = min(row0 + r, M - 1);
      v[u] = *reinterpret_cast<const float4*>(ep + r * EPLD + 4 * c4);
      t[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (EPI == EPI_TANHGRAD || EPI == EPI_RELUGRAD || EPI == EPI_ELUGRAD)
        t[u] = *reinterpret_cast<const float4*>(aux + rowc * ldaux + col);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = row0 + (it0 + u) * RPI + rl;
      const float4 o = epi_apply4<EPI>(v[u], b, t[u]);
      if (STORE && row < M) *reinterpret_cast<float4*>(C + row * ldc + col) = o;
      if (KEEP) *reinterpret_cast<float4*>(ep + ((it0 + u) * RPI + rl) * EPLD + 4 * c4) = o;  // activated tile stays in LDS
    }
  }
}

// STORE_ONLY drops every fused epilogue but the plain store (weight-gradient launches): the grouped kernel
// instantiates the body four times, and with all eight epilogues in each copy the compiler spilled the
// by-value problem table to scratch (1.2 KB/lane, 37 -> 90 us).
// HEAD: bias+tanh layer whose single n-tile holds whole rows, followed by a <= 8-wide tanh head computed from
// the LDS-staged activated tile (the 128 -> 8 latent layer of env_mlp: one launch less per step).
// BF16IN (opt-in, IGI_GEMM_BF16=1; SURVEY section 7 "offer bf16-input fp32-accumulate as an opt-in mode"): the
// fp32 tiles land in LDS exactly as before, each lane converts the eight k it feeds to one
// v_mfma_f32_32x32x16_bf16 (round to nearest even, v_cvt_pk_bf16_f32) -- 16x the matrix rate, ~3 significant
// digits per product, fp32 accumulation.  NOT the arithmetic the headline number is measured in.
// Hook: a caller-supplied epilogue that consumes the raw accumulators instead of any of the built-in ones (the last
// trunk layer's forward with the policy / value heads, the PPO loss and the head backward behind it: teacher.h,
// TrunkLossHook).  hook->prefetch(...) runs right behind the first tile's DMA requests (loads issued there land under the
// k-loop), hook->epilogue(...) after the last k-tile (every DMA of this wave has landed; other waves may still read the ring).
struct NoHook {};
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// x = hi + mid + lo exactly (see dma_util.h, x3_mode): three round-to-nearest-even conversions and two exact differences
__device__ __forceinline__ void split3_bf16(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)x[j];
    const float r = x[j] - (float)h;
    const __bf16 m = (__bf16)r;
    const float q = r - (float)m;
    hi[j] = h; mid[j] = m; lo[j] = (__bf16)q;
  }
}
template <bool STORE_ONLY, bool HEAD, bool BF16IN, bool TANHGRAD_ONLY, class Hook>
struct STORE_ONLY_OK { static constexpr bool value = !HEAD && !BF16IN && !TANHGRAD_ONLY && std::is_same<Hook, NoHook>::value; };
// LOWW (with TANHGRAD_ONLY + row dots): the tile also feeds the weight gradient of the layer below (GemmArgs::lw_*) into
// *lw_acc, which the caller keeps across the consecutive row tiles of a workgroup, and does not write its C tile.
template <int BN, bool A_KC, bool B_KC, int GATHER, int NS, int BM = DMA_BM, bool STORE_ONLY = false, bool HEAD = false,
          int BF16IN = 0, bool TANHGRAD_ONLY = false, class Hook = NoHook, bool LOWW = false>
__device__ __forceinline__ void gemm_dma_body(const GemmArgs& g, int n_tiles, int m_tiles, int bid, Hook* hook = nullptr,
                                              f32x16* lw_acc = nullptr) {
  // KG == 2 (the 192-row tile: three 32-row MFMA tiles do not split over eight waves): the waves form two groups of four
  // that share the SAME 2 x 2 arrangement of 96 x 32 wave tiles and split every k-tile's four 8-k groups between them;
  // the two partial accumulators meet once, in the epilogue.  24 MFMAs per wave and barrier instead of 16.
  constexpr int KG = (BM == 192) ? 2 : 1;
  constexpr int NW = DMA_WAVES / KG;
  constexpr int WGM = KG == 2 ? 2 : ((BN == 32) ? 8 : ((BN == 64) ? 4 : 2)), WGN = NW / WGM;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  static_assert(KG == 1 || (BN == 64 && STORE_ONLY_OK<STORE_ONLY, HEAD, (BF16IN != 0), TANHGRAD_ONLY, Hook>::value),
                "the 192-row tile is built for the plain-store weight-gradient products");
  static_assert(BN != 32 || NS == 2, "waves issue unequal DMA counts on a 32-wide tile: no counted vmcnt waits");
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  constexpr int A_FLOATS = BM * DMA_BK, B_FLOATS = BN * DMA_BK;
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  constexpr int LPT = (BN == 32) ? 4 : (A_FLOATS + B_FLOATS) * 4 / 1024 / DMA_WAVES;  // DMA instructions per wave per k-tile
  extern __shared__ __attribute__((aligned(1024))) float smem[];

  // the wave index is made scalar: LDS-DMA destinations (M0) and tile offsets are then computed on the scalar unit
  // instead of v_readfirstlane round trips in front of every DMA instruction
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave / NW, wl = wave % NW;
  const int wm = wl / WGN, wn = wl % WGN;
  const int l31 = lane & 31, h = lane >> 5;

  // the tile index is wave-uniform: say so, and everything derived from it (tile origin, k range, operand bases --
  // including the integer divisions) stays on the scalar unit
  // (integer division by a runtime value is expanded through the VECTOR unit's float reciprocal, so its results live
  // in vector registers even when every input is scalar: pull them back)
  bid = __builtin_amdgcn_readfirstlane(bid);
  const int bq = __builtin_amdgcn_readfirstlane(fdiv_checked(bid, n_tiles, g.dNT));      // bid / n_tiles
  const int nt = bid - bq * n_tiles;
  const int z = __builtin_amdgcn_readfirstlane(fdiv_checked(bq, m_tiles, g.dMT));        // bid / (n_tiles * m_tiles)
  const int mt = bq - z * m_tiles;
  const int batch = __builtin_amdgcn_readfirstlane(fdiv_checked(z, g.splitk, g.dSK)), split = z - batch * g.splitk;
  const int n0 = nt * BN, m0 = mt * BM;

  const float* A = g.A + batch * g.sA;
  const float* B = g.B + batch * g.sB;
  int k_begin = 0, k_end = g.K;
  if (g.splitk > 1) {
    k_begin = split * g.kchunk;
    k_end = min(g.K, k_begin + g.kchunk);
  }
  int nk = (k_end - k_begin) / DMA_BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;
  const bool do_bsum = (g.Cbias != nullptr) && (g.bias_from_b ? (mt == 0 && tid < BN) : (nt == 0 && tid < BM));

  // GATHER == 4: the data gradient of a padded (full-correlation) convolution on POSITION-MAJOR tiles.  A 256-row tile
  // is ONE input position of 256 consecutive images, so every row of the tile has the same set of in-image taps:
  // the out-of-image taps (conv2: 31 % of the 16, conv3: 26 % of the 9 on the 32 x 64 maps) are not executed at all
  // -- the row-major tiles of GATHER == 1 feed them from a zero page and pay a bounds test per request --, a request
  // is (scalar tap address) + (a lane offset fixed for the tile), and consecutive tiles walk the positions of one
  // image block, so neighbouring tiles re-read each other's patches from L2.  Output row j of the tile is image
  // blk * 256 + j at that position: the epilogue's row stride is one image.
  PmTile pm;
  if constexpr (GATHER == 4) pm_tile_init<BM>(pm, g.conv, mt, wave, lane);
  PwTaps<BM> pw;
  if constexpr (GATHER == 5) pw_init<BM>(pw, g.conv, m0, wave, lane);
  GatherRows<BM> grows;
  GatherTaps<BM> gtaps_a;
  GatherTaps<BN> gtaps_b;
  // GATHER == 6 = GATHER == 1 + the fused soft-argmax partials in the epilogue: its own instantiation, so the other
  // im2col forward products do not carry that epilogue (its first version held 32 values per lane: 172 VGPRs, one
  // workgroup per CU, for BOTH users of the shared instantiation -- tests/test_host_api.py guards the register count)
  constexpr bool G1 = (GATHER == 1 || GATHER == 6);
  if (G1) gather_rows_init<BM>(grows, g.conv, m0, g.M, wave, lane);
  if (GATHER == 3) gather_taps_init<BM>(gtaps_a, g.conv, m0, wave, lane);
  if (GATHER == 2) gather_taps_init<BN>(gtaps_b, g.conv, n0, wave, lane);
  DmaPtrs<BM, A_KC> pa;
  DmaPtrs<BN, B_KC> pb;
  if (!G1 && GATHER != 3 && GATHER != 4 && GATHER != 5) dma_ptrs_init<BM, A_KC>(pa, A, g.lda, m0, g.M, wave, lane);
  if (GATHER == 5) dma_ptrs_init<BN, B_KC>(pb, B, g.conv.OHW * g.ldb, n0, g.N, wave, lane);   // dZ rows of a k-tile: one image apart
  else if (GATHER != 2) dma_ptrs_init<BN, B_KC>(pb, B, g.ldb, n0, g.N, wave, lane);
  auto issue = [&](int t, auto stg) {   // stg: compile-time ring stage of k-tile t (= t % NS)
    float* st = smem + decltype(stg)::value * STAGE;
    if constexpr (GATHER == 4) {   // the next in-image tap (k-tiles are requested strictly in order: stateful walk)
      const int kb = pm_issue<BM>(pm, A, g.conv, st, wave);
      dma_ptrs_issue<BN, B_KC>(pb, kb, st + A_FLOATS, wave);
      return;
    }
    const int k0 = k_begin + t * DMA_BK;
    if constexpr (GATHER == 5) {
      const int kt = k0 / DMA_BK;
      const int j = fdiv(kt, g.conv.dOHW), pos = kt - j * g.conv.OHW;   // image group major: consecutive k-tiles slide the window over the SAME 32 images (L2 reuse)
      pw_issue<BM>(pw, A, g.conv, pos, j, st, wave);
      const char* bb = reinterpret_cast<const char*>(pb.base + ((long long)(32 * j) * g.conv.OHW + pos) * g.ldb);
#pragma unroll
      for (int q = 0; q < DmaPtrs<BN, B_KC>::NQ; ++q) {
        const int i = wave + DMA_WAVES * q;
        if (DmaPtrs<BN, B_KC>::NINSTR < DMA_WAVES && i >= DmaPtrs<BN, B_KC>::NINSTR) break;
        asm volatile("" : "+v"(pb.off[q]));
        dma16(reinterpret_cast<const float*>(bb + pb.off[q]), st + A_FLOATS + 256 * i);
      }
      return;
    }
    if (G1) dma_tile_gather_kc<BM>(A, g.conv, grows, k0 / DMA_BK, st, wave, lane);
    else if (GATHER == 3) dma_tile_gather_rm<BM>(A, g.conv, gtaps_a, k0, st, wave, lane);
    else dma_ptrs_issue<BM, A_KC>(pa, k0, st, wave);
    if (GATHER == 2) dma_tile_gather_rm<BN>(B, g.conv, gtaps_b, k0, st + A_FLOATS, wave, lane);
    else dma_ptrs_issue<BN, B_KC>(pb, k0, st + A_FLOATS, wave);
  };

  if constexpr (GATHER == 4) nk = pm.nk;   // in-image taps x channel slices of this tile's position
  // prologue: NS-1 tiles in flight
  if (nk > 0) issue(0, std::integral_constant<int, 0>{});
  if (NS > 2 && nk > 1) issue(1, std::integral_constant<int, 1>{});
  if constexpr (!std::is_same<Hook, NoHook>::value) hook->prefetch(g, m0, batch, tid);

  // One k-tile with its ring stage S a compile-time constant (the loop below is unrolled by NS): every LDS address of
  // the tile is then a loop-invariant lane offset plus an immediate, where a runtime stage base cost seven or eight
  // vector adds per k-tile.
  auto ktile = [&](int kt, auto stg) {
    constexpr int S = decltype(stg)::value;
    // tile kt landed (for this wave's own DMA) once at most one younger tile is outstanding
    if (NS > 2 && kt + 1 < nk) {
      static_assert(LPT == 6 || LPT == 5 || LPT == 4 || LPT == 3, "vmcnt immediates below");
      if (LPT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (LPT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (LPT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (!std::is_same<Hook, NoHook>::value) {
      if (kt == hook->prefetch1_at(nk)) hook->prefetch1(g, batch, tid);   // requests that depend on prefetch()
    }
    const float* as = smem + S * STAGE;
    const float* bs = as + A_FLOATS;
    if constexpr (BF16IN == 6 || BF16IN == 9) {
      // fp32 on the bf16 pipe, exact three-plane split (experiment, x3_mode): the lane splits the eight k it feeds into
      // hi / mid / lo planes and issues the cross products smallest first -- nothing is rounded before the accumulator
      if (kt + NS - 1 < nk) issue(kt + NS - 1, std::integral_constant<int, (S + NS - 1) % NS>{});
#pragma unroll
      for (int gk = 0; gk < 2; ++gk) {
        bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = wm * WTM + i * 32 + l31;
          float x[8];
          if (A_KC) {
            const int sw = (m >> 1) & 7;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(as + (m * 8 + ((4 * gk + 2 * h) ^ sw)) * 4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(as + (m * 8 + ((4 * gk + 2 * h + 1) ^ sw)) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = v0[j]; x[4 + j] = v1[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = as[(16 * gk + 8 * h + j) * BM + m];
          }
          split3_bf16(x, ah[i], am[i], al[i]);
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const int m = wn * WTN + n * 32 + l31;
          float x[8];
          if (B_KC) {
            const int sw = (m >> 1) & 7;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(bs + (m * 8 + ((4 * gk + 2 * h) ^ sw)) * 4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(bs + (m * 8 + ((4 * gk + 2 * h + 1) ^ sw)) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = v0[j]; x[4 + j] = v1[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = bs[(16 * gk + 8 * h + j) * BN + m];
          }
          split3_bf16(x, bh[n], bm[n], bl[n]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n) {
            f32x16 c = acc[i][n];
            if constexpr (BF16IN == 9) {
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bl[n], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bm[n], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bl[n], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[n], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[n], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[n], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[n], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[n], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[n], c, 0, 0, 0);
            acc[i][n] = c;
          }
      }
    } else if constexpr (BF16IN != 0) {
      if (kt + NS - 1 < nk) issue(kt + NS - 1, std::integral_constant<int, (S + NS - 1) % NS>{});  // into stage (kt-1)%NS: every wave is past its reads of it
#pragma unroll
      for (int gk = 0; gk < 2; ++gk) {   // the two 16-k halves of the tile; lanes 0-31 feed k 0-7, lanes 32-63 k 8-15
        bf16x8 av[TM], bv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = wm * WTM + i * 32 + l31;
          if (A_KC) {
            const int sw = (m >> 1) & 7;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(as + (m * 8 + ((4 * gk + 2 * h) ^ sw)) * 4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(as + (m * 8 + ((4 * gk + 2 * h + 1) ^ sw)) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { av[i][j] = (__bf16)v0[j]; av[i][4 + j] = (__bf16)v1[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) av[i][j] = (__bf16)as[(16 * gk + 8 * h + j) * BM + m];
          }
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const int m = wn * WTN + n * 32 + l31;
          if (B_KC) {
            const int sw = (m >> 1) & 7;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(bs + (m * 8 + ((4 * gk + 2 * h) ^ sw)) * 4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(bs + (m * 8 + ((4 * gk + 2 * h + 1) ^ sw)) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { bv[n][j] = (__bf16)v0[j]; bv[n][4 + j] = (__bf16)v1[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[n][j] = (__bf16)bs[(16 * gk + 8 * h + j) * BN + m];
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[n], acc[i][n], 0, 0, 0);
      }
    } else {
    // operand fragments of one 8-k group; the first group of a tile is requested BEFORE the next tile's DMA is
    // issued (address math + 3-6 DMA instructions then run under the LDS latency instead of in front of it),
    // group c+1 is requested before the MFMAs of group c
    auto load_frag = [&](int c, float (&a)[TM][4], float (&b)[TN][4]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = wm * WTM + i * 32 + l31;
        if (A_KC) {
          const int p = (2 * c + h) ^ ((m >> 1) & 7);
          const f32x4 v = *reinterpret_cast<const f32x4*>(as + (m * 8 + p) * 4);
          a[i][0] = v[0]; a[i][1] = v[1]; a[i][2] = v[2]; a[i][3] = v[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[i][j] = as[(8 * c + 4 * h + j) * BM + m];
        }
      }
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int m = wn * WTN + n * 32 + l31;
        if (B_KC) {
          const int p = (2 * c + h) ^ ((m >> 1) & 7);
          const f32x4 v = *reinterpret_cast<const f32x4*>(bs + (m * 8 + p) * 4);
          b[n][0] = v[0]; b[n][1] = v[1]; b[n][2] = v[2]; b[n][3] = v[3];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[n][j] = bs[(8 * c + 4 * h + j) * BN + m];
        }
      }
    };
    float fa[2][TM][4], fb[2][TN][4];
    constexpr int NC = 4 / KG;                 // 8-k groups of a k-tile this wave multiplies: kg * NC .. + NC - 1
    const int c0 = kg * NC;
    load_frag(c0, fa[0], fb[0]);
    if (kt + NS - 1 < nk) issue(kt + NS - 1, std::integral_constant<int, (S + NS - 1) % NS>{});  // into stage (kt-1)%NS: every wave is past its reads of it
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c < NC - 1) load_frag(c0 + c + 1, fa[(c + 1) & 1], fb[(c + 1) & 1]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][i][j], fb[c & 1][n][j], acc[i][n], 0, 0, 0);
    }
    }
    if (do_bsum) {  // wgrad: the dZ operand is reduction-major here, column `tid` of its tile
      if (g.bias_from_b) {
#pragma unroll 8
        for (int k = 0; k < DMA_BK; ++k) bsum += bs[k * BN + tid];
      } else {
#pragma unroll 8
        for (int k = 0; k < DMA_BK; ++k) bsum += as[k * BM + tid];
      }
    }
  };
  {
    int kt = 0;
    for (; kt + NS <= nk; kt += NS) {
      ktile(kt, std::integral_constant<int, 0>{});
      ktile(kt + 1, std::integral_constant<int, 1>{});
      if constexpr (NS > 2) ktile(kt + 2, std::integral_constant<int, 2>{});
    }
    if (kt < nk) ktile(kt, std::integral_constant<int, 0>{});
    if (NS > 2 && kt + 1 < nk) ktile(kt + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue
  if constexpr (!std::is_same<Hook, NoHook>::value) {
    static_assert(TM == 1 && TN == 1 && WGM == 2 && WGN == 4, "the hook sees 32 x 32 wave tiles of a 64 x 128 tile");
    hook->epilogue(acc, smem, g, m0, mt, batch, tid, wave, lane, wm, wn);
    return;
  }
  float* C = g.C + batch * g.sC + split * g.sCsplit;
  const float* bias = g.bias ? g.bias + batch * g.sBias : nullptr;
  const float* aux = g.aux ? g.aux + batch * g.sAux : nullptr;
  int ldc_e = g.ldc, ldaux_e = g.ldaux, m0_e = m0, M_e = g.M;
  if constexpr (GATHER == 4) {   // tile row j = image blk * BM + j at position pm.pos: one image between rows
    C += ((long long)pm.blk * BM * g.conv.OHW + pm.pos) * g.ldc;
    if (aux) aux += ((long long)pm.blk * BM * g.conv.OHW + pm.pos) * g.ldaux;
    ldc_e = g.conv.OHW * g.ldc; ldaux_e = g.conv.OHW * g.ldaux; m0_e = 0; M_e = BM;
  }
  if constexpr (KG == 2) {
    // the two k-groups' partial tiles meet: group 1 parks its accumulators in its partner's slot, group 0 adds them
    // (one fixed order: own + partner) and stores; wide_epi is a launch condition of this tile
    constexpr int EPLD = WTN + 4;
    __syncthreads();  // every wave is done reading the ring; no DMA is in flight (vmcnt(0) above)
    float* ep = smem + wl * (WTM * EPLD);
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ep[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * EPLD + l31] = acc[i][0][r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* q = ep + (i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * EPLD + l31;
          *q = acc[i][0][r] + *q;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      epilogue_rows<EPI_STORE, WTM, WTN>(ep, C, ldc_e, bias, aux, ldaux_e, m0_e + wm * WTM, n0 + wn * WTN, M_e, g.N, lane);
    }
    if (do_bsum) {
      const int idx = g.bias_from_b ? n0 + tid : m0 + tid;
      if (idx < (g.bias_from_b ? g.N : g.M)) g.Cbias[batch * g.sCbias + split * g.sCbiasSplit + idx] = bsum;
    }
    return;
  }
  if (g.wide_epi) {
    constexpr int EPLD = WTN + 4;
    __syncthreads();  // every wave is done reading the ring; no DMA is in flight (vmcnt(0) above)
    float* ep = smem + wave * (WTM * EPLD);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ep[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * EPLD + n * 32 + l31] = acc[i][n][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS writes precede the read-back
    const int row0 = m0_e + wm * WTM, col0 = n0 + wn * WTN;
#define IGI_EPI_ROWS(E) epilogue_rows<E, WTM, WTN>(ep, C, ldc_e, bias, aux, ldaux_e, row0, col0, M_e, g.N, lane)
    if (HEAD) {
      epilogue_rows<EPI_BIAS_TANH, WTM, WTN, true>(ep, C, g.ldc, bias, aux, g.ldaux, row0, col0, g.M, g.N, lane);
      // head weights into the LDS left over behind the eight staging slices; wave q then owns head output q
      // (uniform weight address = LDS broadcast), lane = row of the tile
      float* wsh = smem + DMA_WAVES * (WTM * EPLD);
      for (int e = tid; e < g.head_n * g.N; e += DMA_THREADS) wsh[e] = g.head_W[e];
      __syncthreads();
      static_assert(!HEAD || BM == 64, "one lane per tile row");
      const int r = lane, q = wave;
      if (q < g.head_n && m0 + r < g.M) {
        const int swm = r / WTM, rl = r - swm * WTM;
        const float* wq = wsh + q * g.N;
        float hacc = 0.f;
        // same accumulation order as the MFMA k-loop: pairs (k, k+4) inside every group of eight
        for (int c8 = 0; c8 < g.N; c8 += 8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int c = c8 + 4 * hh + j;
              hacc = fmaf(smem[(swm * WGN + c / WTN) * (WTM * EPLD) + rl * EPLD + (c % WTN)], wq[c], hacc);
            }
          }
        }
        g.head_out[(long long)(m0 + r) * g.head_ld + q] = fast_tanh(hacc + g.head_b[q]);
      }
      return;
    }
    if (STORE_ONLY) IGI_EPI_ROWS(EPI_STORE);
    else if (TANHGRAD_ONLY) {
      if (g.rowdot_out) {
        // the finished dZ tile stays in this wave's LDS slice (KEEP); lane = one of the slice's WTM (= 64) rows
        epilogue_rows<EPI_TANHGRAD, WTM, WTN, true, !LOWW>(ep, C, g.ldc, bias, aux, g.ldaux, row0, col0, g.M, g.N, lane);
        static_assert(!TANHGRAD_ONLY || WTM == 64, "one lane per slice row");
        static_assert(!TANHGRAD_ONLY || WTN == 32, "two 16-wide k halves");
        // On the matrix pipe (v_mfma_f32_16x16x4_f32, as k_latent_bwd did): for each of the slice's four 16-row blocks
        // and each 16-column half, lane (m = lane & 15, kq = lane >> 4) feeds A[m][4 kq + t] = dZ[row m][..] (one
        // 16-byte LDS read) and B[4 kq + t][n = m] = W[n][..] (n < 8, else 0) to the t-th of four instructions.
        typedef float f32x4r __attribute__((ext_vector_type(4)));
        const int fm = lane & 15, fq = lane >> 4;
        float4 wv[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          const int c = col0 + 16 * kh + 4 * fq;
          wv[kh] = (fm < 8 && c < g.N)
                       ? *reinterpret_cast<const float4*>(g.rowdot_W + (long long)fm * g.rowdot_ld + batch * g.rowdot_kz + c)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4r dacc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          dacc[rb] = f32x4r{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            float4 x = *reinterpret_cast<const float4*>(ep + (rb * 16 + fm) * EPLD + 16 * kh + 4 * fq);
            if (col0 + 16 * kh + 4 * fq >= g.N) x = make_float4(0.f, 0.f, 0.f, 0.f);   // columns beyond N hold no dZ
            dacc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, wv[kh].x, dacc[rb], 0, 0, 0);
            dacc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, wv[kh].y, dacc[rb], 0, 0, 0);
            dacc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, wv[kh].z, dacc[rb], 0, 0, 0);
            dacc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, wv[kh].w, dacc[rb], 0, 0, 0);
          }
        }
        if constexpr (LOWW) {
          // dW_below[n][j] += sum_r dZ[r][n] * X[r][j] over the slice's 64 rows: A(m = n, k = r) from the slice in LDS
          // (lane = column: conflict-free 4-byte reads), B(n = j, k = r) = the layer below's input row, 128 contiguous
          // bytes per row straight from L2 (2 MB, read by every tile of these rows); lane j == 31 multiplies by ONE instead
          // (that input column is zero padding): column 31 of the product is the bias gradient sum_r dZ[r][n].
          // MFMA step s takes rows 2 s + h.  All 32 input values are requested before the first MFMA.
          const float* xr = g.lw_X + (long long)(row0 + h) * g.lw_ldx + l31;
          // (eight input values in flight beside eight MFMAs: all 32 at once took the kernel to 128 VGPRs + scratch)
          float xb[2][8];
          auto xload = [&](int b8, float (&d)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = (l31 == 31) ? 1.0f : xr[(long long)(2 * (8 * b8 + u)) * g.lw_ldx];
          };
          xload(0, xb[0]);
          const float* ea = ep + h * EPLD + l31;
          f32x16 la = *lw_acc;
#pragma unroll
          for (int b8 = 0; b8 < 4; ++b8) {
            if (b8 < 3) xload(b8 + 1, xb[(b8 + 1) & 1]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
              la = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[(2 * (8 * b8 + u)) * EPLD], xb[b8 & 1][u], la, 0, 0, 0);
          }
          *lw_acc = la;
        }
        // the slice is dead now: park the sums at its start as [64 rows][8] (register r of lane (n, q) is row 4q + r,
        // output n), then the wn == 0 wave of each row group adds the WGN column slices in fixed order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (fm < 8) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[(rb * 16 + 4 * fq + r) * 8 + fm] = dacc[rb][r];
        }
        // LDS-only rendezvous: the tile's global stores stay in flight (a __syncthreads() would drain them: +3 us per tile)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wn == 0) {
          const int row = row0 + lane;
          if (row < g.M) {
            float sum[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) sum[q] = 0.f;
            for (int w2 = 0; w2 < WGN; ++w2) {
              const float* ps = smem + (wm * WGN + w2) * (WTM * EPLD) + lane * 8;
#pragma unroll
              for (int q = 0; q < 8; ++q) sum[q] += ps[q];
            }
            float* o = g.rowdot_out + (((long long)batch * n_tiles + nt) * g.M + row) * 8;
            *reinterpret_cast<float4*>(o) = make_float4(sum[0], sum[1], sum[2], sum[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(sum[4], sum[5], sum[6], sum[7]);
          }
        }
      } else {
        IGI_EPI_ROWS(EPI_TANHGRAD);
      }
    }
    else if (g.epilogue == EPI_TANHGRAD) IGI_EPI_ROWS(EPI_TANHGRAD);
    else if (g.epilogue == EPI_BIAS_TANH) IGI_EPI_ROWS(EPI_BIAS_TANH);
    else if (g.epilogue == EPI_BIAS) IGI_EPI_ROWS(EPI_BIAS);
    else if (GATHER == 6 && BM == 256 && BN == 64 && g.epilogue == EPI_BIAS_RELU && g.ssa_part) {
      // last tactile convolution: the activated tile stays in LDS (KEEP) and every 32-row group emits its soft-argmax
      // partial per channel -- the feature map is not read again by a soft-argmax forward kernel (two passes over it)
      epilogue_rows<EPI_BIAS_RELU, WTM, WTN, true>(ep, C, ldc_e, bias, aux, ldaux_e, row0, col0, M_e, g.N, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();     // LDS-only rendezvous: the tile's global stores stay in flight
      asm volatile("" ::: "memory");
      if constexpr (GATHER == 6 && BM == 256 && BN == 64) ssa_tile_partials<WTM, WTN, WGN>(g, smem, m0, n0, tid);
    }
    else if (g.epilogue == EPI_BIAS_RELU) IGI_EPI_ROWS(EPI_BIAS_RELU);
    else if (g.epilogue == EPI_RELUGRAD) IGI_EPI_ROWS(EPI_RELUGRAD);
    else if (g.epilogue == EPI_BIAS_ELU) IGI_EPI_ROWS(EPI_BIAS_ELU);
    else if (g.epilogue == EPI_ELUGRAD) IGI_EPI_ROWS(EPI_ELUGRAD);
    else IGI_EPI_ROWS(EPI_STORE);
#undef IGI_EPI_ROWS
    if (do_bsum) {
      const int idx = g.bias_from_b ? n0 + tid : m0 + tid;
      if (idx < (g.bias_from_b ? g.N : g.M)) g.Cbias[batch * g.sCbias + split * g.sCbiasSplit + idx] = bsum;
    }
    return;
  }
#define IGI_EPI_CALL(E, ACC)                                                                   \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int n = 0; n < TN; ++n) \
      epilogue_tile<E, ACC>(acc[i][n], C, g.ldc, bias, aux, g.ldaux, m0 + wm * WTM + i * 32 + 4 * h, \
                            n0 + wn * WTN + n * 32 + l31, g.M, g.N)
  if (STORE_ONLY) { IGI_EPI_CALL(EPI_STORE, false); }
  else if (TANHGRAD_ONLY) { IGI_EPI_CALL(EPI_TANHGRAD, false); }
  else if (g.epilogue == EPI_TANHGRAD) { IGI_EPI_CALL(EPI_TANHGRAD, false); }
  else if (g.epilogue == EPI_BIAS_TANH) { IGI_EPI_CALL(EPI_BIAS_TANH, false); }
  else if (g.epilogue == EPI_BIAS) { IGI_EPI_CALL(EPI_BIAS, false); }
  else if (g.epilogue == EPI_BIAS_RELU) { IGI_EPI_CALL(EPI_BIAS_RELU, false); }
  else if (g.epilogue == EPI_RELUGRAD) { IGI_EPI_CALL(EPI_RELUGRAD, false); }
  else if (g.epilogue == EPI_BIAS_ELU) { IGI_EPI_CALL(EPI_BIAS_ELU, false); }
  else if (g.epilogue == EPI_ELUGRAD) { IGI_EPI_CALL(EPI_ELUGRAD, false); }
  else { IGI_EPI_CALL(EPI_STORE, false); }
#undef IGI_EPI_CALL
  if (do_bsum) {
    const int idx = g.bias_from_b ? n0 + tid : m0 + tid;
    if (idx < (g.bias_from_b ? g.N : g.M)) g.Cbias[batch * g.sCbias + split * g.sCbiasSplit + idx] = bsum;
  }
}

// XCD-aware tile order: blocks b and b+8 share an XCD/L2, so give each XCD a contiguous run of tile
// ids (n fastest): the tiles that re-read the same A rows / the same k-chunk hit in L2.
// Any grid size (bijective): XCD x (the blocks with bid % 8 == x) owns the next ceil / floor(total / 8) tile ids.  Before,
// grids that are not a multiple of 8 kept the round-robin order -- e.g. the conv3 weight gradient (5 tap tiles x 102
// k-chunks = 510 workgroups) put the five tiles that stream the SAME rows on five different XCDs: 4.2 GB fetched per
// launch for 0.95 GB of operands, L2 hit rate 0.025 (profiles/r03_student_c3_hbm_traffic.json, first collection).
__device__ __forceinline__ int xcd_remap(int bid, int total) {
  const int q = total >> 3, r = total & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int BN, bool A_KC, bool B_KC, int GATHER = 0, int NS = DMA_NS, int BM = DMA_BM>
__global__ __launch_bounds__(DMA_THREADS) void gemm_dma_kernel(const GemmArgs g, int n_tiles, int m_tiles) {
  gemm_dma_body<BN, A_KC, B_KC, GATHER, NS, BM>(g, n_tiles, m_tiles, xcd_remap(blockIdx.x, gridDim.x));
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(DMA_THREADS) void gemm_dma_bf16_kernel(const GemmArgs g, int n_tiles, int m_tiles) {
  gemm_dma_body<128, A_KC, B_KC, 0, 2, DMA_BM, false, false, 1>(g, n_tiles, m_tiles, xcd_remap(blockIdx.x, gridDim.x));
}

// EXPERIMENT (x3_mode): the k-contiguous forward product with every operand split into three exact bf16 planes in the loop
template <int PRODUCTS>
__global__ __launch_bounds__(DMA_THREADS) void gemm_dma_x3_kernel(const GemmArgs g, int n_tiles, int m_tiles) {
  gemm_dma_body<128, true, true, 0, 2, DMA_BM, false, false, PRODUCTS>(g, n_tiles, m_tiles, xcd_remap(blockIdx.x, gridDim.x));
}

// Grouped launch: up to DMA_GROUP_MAX independent problems of the same operand layout share one
// grid (problem p owns tile ids [tile_end[p-1], tile_end[p])).  Used for the weight-gradient products
// of one backward pass, which depend on nothing but each other's inputs: one launch pays the
// ~15-25 us fill/drain once instead of once per layer, and the short problems fill the tail of the
// long ones.
constexpr int DMA_GROUP_MAX = 4;
struct GemmGroup {
  GemmArgs g[DMA_GROUP_MAX];
  int tile_end[DMA_GROUP_MAX];
  int n_tiles[DMA_GROUP_MAX], m_tiles[DMA_GROUP_MAX];
  int n = 0;
};

template <int BN, bool A_KC, bool B_KC, int NS, bool BF16IN = false>
__global__ __launch_bounds__(DMA_THREADS) void gemm_dma_group_kernel(const GemmGroup gr) {
  // The XCD remap is applied PER PROBLEM: remapping the whole grid would hand each XCD a contiguous run
  // of tile ids, i.e. all of the long problem to some XCDs and only short ones to the others.
  const int bid = blockIdx.x;
  int p = 0;
#pragma unroll
  for (int q = 0; q < DMA_GROUP_MAX - 1; ++q)
    if (q + 1 < gr.n && bid >= gr.tile_end[q]) p = q + 1;
  const int start = (p > 0 ? gr.tile_end[p - 1] : 0);
  int local = bid - start;
  if ((start & 7) == 0) local = xcd_remap(local, gr.tile_end[p] - start);
  // problem p is wave-uniform; index the by-value struct with a uniform switch (no scratch copy)
  switch (p) {
    case 0: gemm_dma_body<BN, A_KC, B_KC, 0, NS, DMA_BM, true, false, BF16IN>(gr.g[0], gr.n_tiles[0], gr.m_tiles[0], local); break;
    case 1: gemm_dma_body<BN, A_KC, B_KC, 0, NS, DMA_BM, true, false, BF16IN>(gr.g[1], gr.n_tiles[1], gr.m_tiles[1], local); break;
    case 2: gemm_dma_body<BN, A_KC, B_KC, 0, NS, DMA_BM, true, false, BF16IN>(gr.g[2], gr.n_tiles[2], gr.m_tiles[2], local); break;
    default: gemm_dma_body<BN, A_KC, B_KC, 0, NS, DMA_BM, true, false, BF16IN>(gr.g[3], gr.n_tiles[3], gr.m_tiles[3], local); break;
  }
}

// All weight-gradient products of a backward pass in ONE grid, both tile widths: the problem table stays in the
// kernarg segment and is indexed there (uniform address -> scalar loads), so the body is instantiated once per tile
// width, not once per problem slot (the by-value table of gemm_dma_group_kernel had to be indexed with a uniform
// switch to stay out of scratch), and a 64-wide problem (input width <= 64) no longer needs a launch of its own:
// one fill / drain instead of two, the short workgroups fill the tail of the long ones.
constexpr int DMA_MULTI_MAX = 6;
struct GemmMulti {
  GemmArgs g[DMA_MULTI_MAX];
  int tile_end[DMA_MULTI_MAX];
  int n_tiles[DMA_MULTI_MAX], m_tiles[DMA_MULTI_MAX];
  int chain = 1;             // data-gradient tiles per workgroup (consecutive row tiles of one column tile), IGI_DGRAD_CHAIN
  int kind[DMA_MULTI_MAX];   // weight gradients (reduction-major operands, plain store): 0 = 128 x 128 tiles, 1 = 128 x 64,
                             // 3 = 256 x 32 (<= 32 input columns: the zero-padded first trunk layer, no padded MFMA columns);
                             // 2 = data gradient dZ.W times tanh' (A k-contiguous, B reduction-major), 128 x 128 tiles
  int n = 0;
};
typedef const __attribute__((address_space(4))) GemmMulti* gemm_multi_cptr;

// (two workgroups per CU: four waves per SIMD, i.e. at most 128 registers)
__global__ __launch_bounds__(DMA_THREADS, 4) void gemm_dma_wgrad_multi_kernel(const GemmMulti table_in_kernarg) {
  (void)table_in_kernarg;
  gemm_multi_cptr gr = (gemm_multi_cptr)__builtin_amdgcn_kernarg_segment_ptr();
  const int bid = blockIdx.x;
  const int n = gr->n;
  int p = 0;
#pragma unroll
  for (int q = 0; q < DMA_MULTI_MAX - 1; ++q)
    if (q + 1 < n && bid >= gr->tile_end[q]) p = q + 1;
  p = __builtin_amdgcn_readfirstlane(p);
  const int start = (p > 0 ? gr->tile_end[p - 1] : 0);
  int local = bid - start;
  if ((start & 7) == 0) local = xcd_remap(local, gr->tile_end[p] - start);   // per problem, as in the grouped kernel
  const GemmArgs& g = *(const GemmArgs*)&gr->g[p];   // constant -> generic: the loads stay scalar after address-space inference
  const int kind = gr->kind[p];
  if (kind == 0) gemm_dma_body<128, false, false, 0, 2, DMA_BM, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else if (kind == 1) gemm_dma_body<64, false, false, 0, 2, DMA_BM, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else if (kind == 3) gemm_dma_body<32, false, false, 0, 2, 256, true>(g, gr->n_tiles[p], gr->m_tiles[p], local);
  else {
    // chain > 1: a workgroup computes `chain` consecutive row tiles of ONE column tile back to back (same weight slice:
    // it stays in this XCD's L2 and in the CU's L1), so the grid holds chain x fewer, longer data-gradient workgroups
    const int chain = gr->chain, nt_ = gr->n_tiles[p], mt_ = gr->m_tiles[p];
    if (chain <= 1 && kind != 6) {
      gemm_dma_body<128, true, false, 0, 2, DMA_BM, false, false, false, true>(g, nt_, mt_, local);
      return;
    }
    const int mg = mt_ / chain;                        // row-tile groups per batch entry
    const int ntq = local % nt_, grp = local / nt_;
    const int z = grp / mg, m0t = (grp - z * mg) * chain;
    if (kind != 6) {
      for (int t = 0; t < chain; ++t) {
        if (t > 0) __syncthreads();                    // the previous tile's epilogue is done with the LDS the ring reuses
        gemm_dma_body<128, true, false, 0, 2, DMA_BM, false, false, false, true>(g, nt_, mt_, (z * mt_ + m0t + t) * nt_ + ntq);
      }
      return;
    }
    // kind 6: the tiles also accumulate the weight gradient of the layer below (GemmArgs::lw_*) over the chain; one
    // partial record per workgroup: [128 columns of this column tile][32 inputs] + the bias gradient from column 31
    f32x16 lw;
#pragma unroll
    for (int r = 0; r < 16; ++r) lw[r] = 0.f;
    for (int t = 0; t < chain; ++t) {
      if (t > 0) __syncthreads();
      gemm_dma_body<128, true, false, 0, 2, DMA_BM, false, false, false, true, NoHook, true>(g, nt_, mt_, (z * mt_ + m0t + t) * nt_ + ntq,
                                                                                            nullptr, &lw);
    }
    {
      extern __shared__ __attribute__((aligned(1024))) float smem[];
      const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, h = lane >> 5;
      __syncthreads();
      float* ex = smem + wn * (32 * 33);               // the two row halves (wm = 0, 1) of a column slice meet here
      if (wm == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = lw[r];
      }
      __syncthreads();
      if (wm == 0) {
        // accumulator layout: lane l31 = input column j, register r = output (r & 3) + 8 (r >> 2) + 4 h of the slice
        const int part = grp - z * mg;
        float* out = g.lw_out + part * g.lw_sPart + z * g.lw_sNet + (long long)(ntq * 128 + wn * 32) * 32;
        float* bo = g.lw_bias + part * g.lw_bsPart + z * g.lw_bsNet + ntq * 128 + wn * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = lw[r] + ex[c * 33 + l31];
          out[c * 32 + l31] = (l31 == 31) ? 0.f : v;   // (column 31 of the slab: a padding input column, never summed)
          if (l31 == 31) bo[c] = v;
        }
      }
    }
  }
}

// Tile width: 256 keeps each A row-tile read once, but only if that still yields one workgroup
// per CU; narrow outputs (padded first layer, latent) take the 64-wide tile.
static inline int dma_pick_bn(int M, int N, int zcount) {
  if (N <= 64) return 64;
  const long long mt = (M + DMA_BM - 1) / DMA_BM;
  if ((N % 256 == 0 || N > 256) && mt * ((N + 255) / 256) * zcount >= 256) return 256;
  return 128;
}

// split-k factor for a weight-gradient product (reduction over the minibatch): one workgroup per CU,
// at least 4 k-tiles (128 rows) per split, preferring the 256-wide tile when 8 k-tiles remain.
static inline int dma_choose_splitk(int M, int N, int K, int nbatch, int bm = DMA_BM) {
  const long long mt = (M + bm - 1) / bm;
  int bn = (N <= 64) ? 64 : ((N % 256 == 0 || N > 256) ? 256 : 128);
  long long tiles = mt * ((N + bn - 1) / bn) * nbatch;
  // never MORE workgroups than CUs x residency: 260 workgroups on 256 CUs put two on four CUs, and
  // those four finish twice as late as the rest (measured: conv3 weight gradient at 44 instead of 88 TF)
  auto fill = [&](long long target) { return (int)(target / tiles > 1 ? target / tiles : 1); };
  int sk = fill(256);
  if (bn == 256 && sk > K / 256) {
    bn = 128;
    tiles = mt * ((N + bn - 1) / bn) * nbatch;
    sk = fill(256);
  }
  // very long reductions (convolution weight gradients: K = images x pixels): two resident workgroups
  // per CU, as long as each still runs >= 128 k-tiles
  if (bn <= 128 && (long long)K / fill(512) >= 128LL * DMA_BK) sk = fill(512);
  const int maxsk = K / 128 > 1 ? K / 128 : 1;
  if (sk > maxsk) sk = maxsk;
  return sk < 1 ? 1 : sk;
}

static inline bool dma_eligible(const GemmArgs& g, bool akc, bool bkc) {
  if (g.M < 4 || g.N < 4 || g.K < DMA_BK || g.accumulate) return false;
  if (g.gather) {  // implicit-GEMM convolution: only this kernel implements it
    const ConvDesc& c = g.conv;
    if (c.planar) {   // forward of an unpadded 8 x 8 kernel over an NCHW tensor of < 4 GB (the test-free loader only)
      const long long images = (g.M + c.OHW - 1) / c.OHW;
      if (g.gather != 1 || c.pad != 0 || c.KH != 8 || c.KW != 8 || c.ntaps != 64 * c.C || g.K != c.ntaps ||
          images * c.IH * c.IW * c.C * 4 >= (1LL << 32))
        return false;
    } else if (!c.zero || !(c.C == 4 ? c.KW == 8 : (c.C % 32 == 0))) return false;
    // (the planar loader's 16-byte requests start on any 4-byte boundary: tools/probes/dma_align.hip)
    if (g.gather == 1 && !(akc && (c.planar || aligned16(g.A)) && aligned16(g.B) && (g.ldb & 3) == 0)) return false;
    if (g.gather == 2 && !(!akc && !bkc && aligned16(g.A) && aligned16(g.B) && (g.lda & 3) == 0 && (g.M & 3) == 0 && (g.N & 3) == 0)) return false;
    if (g.gather == 3 && !(!akc && !bkc && aligned16(g.A) && aligned16(g.B) && (g.ldb & 3) == 0 && (g.M & 3) == 0 && (g.N & 3) == 0)) return false;
    const int kr = (g.splitk > 1) ? g.kchunk : g.K;
    // the loaders index the input with 32-bit element offsets
    const long long images = ((g.gather == 1 ? g.M : g.K) + c.OHW - 1) / c.OHW;
    if (images * c.IH * c.IW * c.C >= (1LL << 31)) return false;
    return kr % DMA_BK == 0 && g.K % DMA_BK == 0 && (long long)g.M * g.ldc < (1LL << 31) &&
           (long long)g.M * (g.ldaux + 1) < (1LL << 31);
  }
  if ((long long)g.M * g.ldc >= (1LL << 31) || (long long)g.M * (g.ldaux + 1) >= (1LL << 31)) return false;
  if (!aligned16(g.A) || !aligned16(g.B) || (g.lda & 3) || (g.ldb & 3) || (g.sA & 3) || (g.sB & 3)) return false;
  if (g.lda >= (1 << 22) || g.ldb >= (1 << 22)) return false;   // 32-bit byte offsets inside a tile (DmaPtrs)
  const int kr = (g.splitk > 1) ? g.kchunk : g.K;
  if (kr % DMA_BK != 0 || g.K % DMA_BK != 0) return false;
  if (!akc && (g.M & 3)) return false;   // reduction-major rows are fetched 4 wide
  if (!bkc && (g.N & 3)) return false;
  if (g.Cbias && (g.bias_from_b ? bkc : akc)) return false;  // bias-sum reads a reduction-major image
  return true;
}

// Can the im2col product `g` (gather == 1) run on position-major tiles (GATHER == 4)?  A padded stride-1 correlation
// (= the data gradient of an unpadded convolution), whole blocks of 256 images, 16-byte aligned output / aux rows.
static inline bool conv_pmajor_ok(const GemmArgs& g, bool bkc) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_CONV_PMAJOR"); on = e ? atoi(e) : 1; }
  const ConvDesc& c = g.conv;
  if (!on || g.gather != 1 || !bkc || c.pad <= 0 || c.KH <= 0 || c.KH * c.KW > 32 || (c.C & 31) || g.splitk > 1 ||
      g.nbatch != 1 || c.OHW <= 0)
    return false;
  const long long images = g.M / c.OHW;
  if (images * c.OHW != g.M || (images & 255)) return false;
  if (256LL * c.IH * c.IW * c.C * 4 >= (1LL << 32)) return false;
  const bool wide = aligned16(g.C) && (g.ldc & 3) == 0 && (g.N & 3) == 0 &&
                    (!g.bias || aligned16(g.bias)) && (!g.aux || (aligned16(g.aux) && (g.ldaux & 3) == 0));
  return wide && (long long)c.OHW * g.ldc < (1 << 24) && (long long)c.OHW * g.ldaux < (1 << 24);
}

// IGI_CONV_BM192=0: the 576-tap weight gradient on five 128-tap tiles (A/B)
static inline bool conv_bm192_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_CONV_BM192"); on = e ? atoi(e) : 1; }
  return on != 0;
}

// Can the weight gradient `g` (gather == 3) walk its reduction position-major (GATHER == 5)?  Unpadded convolution, whole
// groups of 32 images, every split's k-range whole k-tiles (the launcher guarantees that already).
static inline bool conv_pw_ok(const GemmArgs& g) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_CONV_PW"); on = e ? atoi(e) : 1; }
  const ConvDesc& c = g.conv;
  if (!on || g.gather != 3 || c.pad != 0 || c.OHW <= 0 || (c.C & 3) || g.nbatch != 1) return false;
  const long long images = g.K / c.OHW;
  if (images * c.OHW != g.K || (images & 31)) return false;
  if (32LL * c.IH * c.IW * c.C * 4 >= (1LL << 32) || 32LL * c.OHW * g.ldb * 4 >= (1LL << 32)) return false;
  return images * c.IH * c.IW * c.C < (1LL << 40);
}

static inline void dma_set_divs(GemmArgs& g, int n_tiles, int m_tiles) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_TILE_DIVS"); on = e ? atoi(e) : 1; }   // 0: the kernels divide (A/B)
  if (!on) return;
  g.dNT = make_fastdiv((unsigned)n_tiles); g.dMT = make_fastdiv((unsigned)m_tiles);
  g.dSK = make_fastdiv((unsigned)(g.splitk < 1 ? 1 : g.splitk));
}

template <int BN, int NS = DMA_NS, int BM = DMA_BM>
static hipError_t launch_dma_cfg(const GemmArgs& g, bool akc, bool bkc, hipStream_t s) {
  const int n_tiles = (g.N + BN - 1) / BN, m_tiles = (g.M + BM - 1) / BM;
  const int total = n_tiles * m_tiles * g.nbatch * g.splitk;
  const size_t shm_max = sizeof(float) * NS * (BM + BN) * DMA_BK;
  // short reductions do not use the whole ring: a smaller LDS footprint lets more workgroups share
  // a CU, which is what hides the (then dominant) epilogue latency
  const int kr = (g.splitk > 1) ? g.kchunk : g.K;
  const int stages = kr / DMA_BK < NS ? (kr / DMA_BK < 1 ? 1 : kr / DMA_BK) : NS;
  constexpr int KG_ = (BM == 192) ? 2 : 1;
  constexpr int WGM_ = KG_ == 2 ? 2 : ((BN == 32) ? 8 : ((BN == 64) ? 4 : 2)), WGN_ = DMA_WAVES / KG_ / WGM_;
  constexpr size_t EPI_BYTES = sizeof(float) * (DMA_WAVES / KG_) * (BM / WGM_) * (BN / WGN_ + 4);
  size_t shm = sizeof(float) * stages * (BM + BN) * DMA_BK;
  GemmArgs gg = g;
  dma_set_divs(gg, n_tiles, m_tiles);
  if (g.gather == 1) {   // unpadded convolution over an image tensor of < 4 GB: the loader's test-free path
    const ConvDesc& c = g.conv;
    const long long images = (g.M + c.OHW - 1) / c.OHW;
    gg.conv.fast32 = (c.pad == 0 && images * c.IH * c.IW * c.C * 4 < (1LL << 32)) ? 1 : 0;
  }
  // wide (LDS-staged, 16 B per lane) epilogue needs float4-aligned C / bias / aux
  gg.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.sCsplit & 3) == 0 && (g.N & 3) == 0 &&
            (!g.bias || (aligned16(g.bias) && (g.sBias & 3) == 0)) &&
            (!g.aux || (aligned16(g.aux) && (g.ldaux & 3) == 0 && (g.sAux & 3) == 0));
  if (gg.wide_epi && shm < EPI_BYTES) shm = EPI_BYTES;
  const size_t shm_cap = shm_max > EPI_BYTES ? shm_max : EPI_BYTES;
  dim3 grid(total), block(DMA_THREADS);
#define IGI_DMA_LAUNCH(AK, BK_, GA)                                                                    \
  do {                                                                                                 \
    static bool attr_set = false;                                                                      \
    if (!attr_set) {                                                                                   \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_kernel<BN, AK, BK_, GA, NS, BM>,        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_cap);    \
      if (e != hipSuccess) return e;                                                                   \
      attr_set = true;                                                                                 \
    }                                                                                                  \
    IGI_LAUNCH((gemm_dma_kernel<BN, AK, BK_, GA, NS, BM>), grid, block, shm, s, gg, n_tiles, m_tiles); \
  } while (0)
  if constexpr (BM == 192) {
    // three 192-tap tiles for a 576-tap weight gradient (conv3 of the tactile CNN): the position-major reduction only
    if (!(g.gather == 3 && conv_pw_ok(g) && gg.wide_epi && g.epilogue == EPI_STORE && !g.accumulate && BN == 64)) return hipErrorInvalidValue;
    gg.conv.pmajor = 2;
    gg.conv.nb32 = (g.K / g.conv.OHW) / 32;
    gg.conv.dNB32 = make_fastdiv((unsigned)gg.conv.nb32);
    IGI_DMA_LAUNCH(false, false, 5);
    return hipGetLastError();
  } else {   // (nothing below is instantiated for the 192-row tile)
  if constexpr (BM == 256) {
    if (conv_pmajor_ok(g, bkc)) {   // position-major data-gradient tiles
      gg.conv.pmajor = 1;
      IGI_DMA_LAUNCH(true, true, 4);
      return hipGetLastError();
    }
  }
  if (g.gather == 1) {
    if constexpr (BM == 256 && BN == 64) {
      if (g.ssa_part && bkc) {   // conv3 of the tactile CNN: soft-argmax partials from the tiles (gemm() checked the shape)
        IGI_DMA_LAUNCH(true, true, 6);
        return hipGetLastError();
      }
    }
    if (g.ssa_part) return hipErrorInvalidValue;
    if (bkc) IGI_DMA_LAUNCH(true, true, 1); else IGI_DMA_LAUNCH(true, false, 1);
  } else if (g.gather == 3) {
    if constexpr (BN <= 64) {
      if (conv_pw_ok(g)) {   // reduction over position-major rows: no per-request row decomposition
        gg.conv.pmajor = 2;
        gg.conv.nb32 = (g.K / g.conv.OHW) / 32;
        gg.conv.dNB32 = make_fastdiv((unsigned)gg.conv.nb32);
        IGI_DMA_LAUNCH(false, false, 5);
        return hipGetLastError();
      }
    }
    IGI_DMA_LAUNCH(false, false, 3);
  } else if constexpr (BM != DMA_BM) {
    return hipErrorInvalidValue;  // the tall tile is built for the im2col products only
  } else if (g.gather == 2) {
    IGI_DMA_LAUNCH(false, false, 2);
  } else if (akc && bkc) IGI_DMA_LAUNCH(true, true, 0);
  else if (akc && !bkc) IGI_DMA_LAUNCH(true, false, 0);
  else if (!akc && !bkc) IGI_DMA_LAUNCH(false, false, 0);
  else IGI_DMA_LAUNCH(false, true, 0);
  return hipGetLastError();
  }
#undef IGI_DMA_LAUNCH
}

template <bool B_KC>
__global__ __launch_bounds__(DMA_THREADS) void gemm_dma_head_kernel(const GemmArgs g, int n_tiles, int m_tiles) {
  gemm_dma_body<128, true, B_KC, 0, 2, 64, false, true>(g, n_tiles, m_tiles, xcd_remap(blockIdx.x, gridDim.x));
}

// bias+tanh layer (N <= 128, a multiple of 8) + fused row head; returns hipErrorInvalidValue when the shape does
// not fit (the caller then runs the two layers separately)
static hipError_t gemm_with_head(GemmArgs g, hipStream_t s) {
  if (g.splitk < 1) g.splitk = 1;
  const bool wide = aligned16(g.C) && (g.ldc & 3) == 0 && (g.N & 3) == 0 && g.bias && aligned16(g.bias);
  if (!dma_eligible(g, true, true) || g.gather || g.splitk != 1 || g.nbatch != 1 || g.N > 128 || (g.N & 7) ||
      g.epilogue != EPI_BIAS_TANH || !wide || g.head_n < 1 || g.head_n > 8 || !g.head_W || !g.head_b || !g.head_out)
    return hipErrorInvalidValue;
  g.wide_epi = 1;
  constexpr int BM = 64;
  const int m_tiles = (g.M + BM - 1) / BM;
  constexpr size_t ring = sizeof(float) * 2 * (BM + 128) * DMA_BK;
  constexpr size_t epi = sizeof(float) * (DMA_WAVES * (BM / 2) * (32 + 4) + 8 * 128);  // staging + head weights
  constexpr size_t shm = ring > epi ? ring : epi;
  const double fl = 2.0 * g.M * (double)g.N * (g.K + g.head_n);
  const double by = 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N);
  ProfScope ps(PC_DMA_HEAD, s, fl, by);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_head_kernel<true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    attr = true;
  }
  IGI_LAUNCH((gemm_dma_head_kernel<true>), dim3(m_tiles), dim3(DMA_THREADS), shm, s, g, 1, m_tiles);
  return hipGetLastError();
}

// Will gemm() run the im2col forward product (M rows, <= 64 output channels) on the tall 256 x 64 tile that can emit the
// soft-argmax partials (GemmArgs::ssa_part)?  The tactile plan asks before it sets ssa_part.
static inline bool conv_ssa_fusable(long long M, int N, int P) {
  static int on = -1, tall = -1;
  if (on < 0) { const char* e = getenv("IGI_SSA_FUSE"); on = e ? atoi(e) : 1; }
  if (tall < 0) { const char* e = getenv("IGI_CONV_TALL"); tall = e ? atoi(e) : 3; }
  return on && tall && N > 32 && N <= 64 && (P & 31) == 0 && (M & 255) == 0 && M / 256 >= 512;
}

// Front door used by the C ABI and the teacher/student orchestration.
static hipError_t gemm(GemmArgs g, bool akc, bool bkc, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return hipSuccess;
  if (g.splitk < 1) g.splitk = 1;
  if (g.splitk > 1 && g.kchunk <= 0) {
    int c = (g.K + g.splitk - 1) / g.splitk;
    g.kchunk = (c + DMA_BK - 1) / DMA_BK * DMA_BK;
  }
  if (!dma_eligible(g, akc, bkc)) return g.gather ? hipErrorInvalidValue : launch_gemm(g, akc, bkc, s);
  const double fl = 2.0 * g.M * g.N * (double)g.K * g.nbatch * g.flop_credit;
  const double by = 4.0 * g.nbatch * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * g.splitk);
  // Default policy (measured on the teacher update, 4096 x 32): 128-wide tiles with a 2-stage ring
  // = 64-72 KB of LDS, so TWO workgroups share a CU and one computes while the other is in its DMA
  // prologue / store epilogue; the 256-wide, 3-stage variant (one workgroup per CU) was 4 % slower
  // end to end.  IGI_DMA_MODE=0 selects the latter for experiments.
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("IGI_DMA_MODE"); mode = e ? atoi(e) : 1; }
  int bn = dma_pick_bn(g.M, g.N, g.nbatch * g.splitk);
  // im2col forward / dgrad with <= 64 output channels (every tactile convolution): a 256 x 64 tile gives
  // each wave the same 64 x 32 sub-tile (two independent accumulator chains, 32 MFMAs per barrier) as the
  // 128 x 128 configuration; 2 stages x 40 KB so two workgroups still share a CU.
  static int tall = -1;
  if (tall < 0) { const char* e = getenv("IGI_CONV_TALL"); tall = e ? atoi(e) : 3; }
  const bool tall_fwd = tall && g.gather == 1 && bn == 64 && g.splitk == 1 && (long long)((g.M + 255) / 256) * g.nbatch >= 512;
  if (g.ssa_part && !(tall_fwd && g.N > 32 && g.N <= 64 && (g.M & 255) == 0 && !conv_pmajor_ok(g, bkc)))
    return hipErrorInvalidValue;   // the caller plans the fused soft-argmax with conv_ssa_fusable(): only that tile emits it
  if (tall_fwd) {
    // <= 32 output channels (conv1 forward, conv2 data gradient): a 32-wide tile, no padded MFMA columns
    const bool pmj = conv_pmajor_ok(g, bkc);
    if (g.N <= 32 && tall > 1) {
      ProfScope ps(pmj ? PC_CONV_PM32 : (bkc ? PC_CONV_TALL32_TT : PC_CONV_TALL32_TF), s, fl, by);
      return launch_dma_cfg<32, 2, 256>(g, akc, bkc, s);
    }
    ProfScope ps(pmj ? PC_CONV_PM64 : (g.ssa_part ? PC_CONV_TALL64_SSA : (bkc ? PC_CONV_TALL64_TT : PC_CONV_TALL64_TF)), s, fl, by);
    return launch_dma_cfg<64, 2, 256>(g, akc, bkc, s);
  }
  if (tall > 1 && g.gather == 3 && g.N <= 32 && g.M % 256 == 0) {  // conv1 weight gradient: 32 output channels
    ProfScope ps(conv_pw_ok(g) ? PC_CONV_PW32 : PC_CONV_WG_TALL32, s, fl, by);
    return launch_dma_cfg<32, 2, 256>(g, akc, bkc, s);
  }
  if (tall > 2 && g.gather == 3 && g.N > 32 && g.N <= 64 && g.M % 192 == 0 && g.M % 256 != 0 && conv_bm192_on() && conv_pw_ok(g) &&
      aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.sCsplit & 3) == 0 && (g.N & 3) == 0 && g.epilogue == EPI_STORE) {
    // 576 taps = 3 x 192: no padded tap rows (five 128-tap tiles multiplied 640), 24 MFMAs per wave and barrier
    ProfScope ps(PC_CONV_PW64_192, s, fl, by);
    return launch_dma_cfg<64, 2, 192>(g, akc, bkc, s);
  }
  if (tall > 2 && g.gather == 3 && g.N <= 64 && g.M % 256 == 0) {  // 64-channel weight gradients with whole 256-tap tiles
    ProfScope ps(conv_pw_ok(g) ? PC_CONV_PW64 : PC_CONV_WG_TALL64, s, fl, by);
    return launch_dma_cfg<64, 2, 256>(g, akc, bkc, s);
  }
  bool two_stage = (mode == 1 && bn >= 128);
  if (two_stage) bn = 128;
  // a 128-wide grid that covers at most half the CUs (the 256 -> 128 env_mlp layer: 128 workgroups)
  // runs on 64-wide tiles instead: twice the workgroups, +1.2 % on the update.
  static int fill = -1;
  if (fill < 0) { const char* e = getenv("IGI_BN64_FILL"); fill = e ? atoi(e) : 128; }
  if (two_stage && (long long)((g.M + DMA_BM - 1) / DMA_BM) * ((g.N + 127) / 128) * g.nbatch * g.splitk <= fill) {
    two_stage = false;
    bn = 64;
  }
  // one-k-tile products (the first trunk layer: K = 32, 67 MB of tanh outputs): the launch is epilogue + store, and the
  // 128-wide tile's 73 KB of epilogue staging allows two workgroups per CU; 64-wide tiles stage 37 KB (IGI_K1_BN64, A/B)
  static int k1bn64 = -1;
  if (k1bn64 < 0) { const char* e = getenv("IGI_K1_BN64"); k1bn64 = e ? atoi(e) : 1; }   // 20.6 -> 19.4 us for the first trunk layer
  if (k1bn64 && two_stage && !g.gather && g.K <= DMA_BK) { two_stage = false; bn = 64; }
  const int lay = akc ? (bkc ? 0 : 1) : (bkc ? 3 : 2);
  ProfScope ps((bn == 256 ? PC_DMA_256_TT : (bn == 128 ? PC_DMA_128_TT : PC_DMA_64_TT)) + lay, s, fl, by);
  if (two_stage && x3_mode() && !bf16_mode() && !g.gather && akc && bkc && g.K >= 256) {
    // experiment: exact three-plane bf16 split (same tiles, loaders and epilogues; only the inner loop differs)
    const int n_tiles = (g.N + 127) / 128, m_tiles = (g.M + DMA_BM - 1) / DMA_BM;
    GemmArgs gg = g;
    gg.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.sCsplit & 3) == 0 && (g.N & 3) == 0 &&
                  (!g.bias || (aligned16(g.bias) && (g.sBias & 3) == 0)) &&
                  (!g.aux || (aligned16(g.aux) && (g.ldaux & 3) == 0 && (g.sAux & 3) == 0));
    constexpr size_t ring = sizeof(float) * 2 * (DMA_BM + 128) * DMA_BK, epi = sizeof(float) * DMA_WAVES * 64 * (32 + 4);
    constexpr size_t shm = ring > epi ? ring : epi;
    const dim3 grid(n_tiles * m_tiles * g.nbatch * g.splitk);
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_x3_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_dma_x3_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (e != hipSuccess) return e;
      attr = true;
    }
    if (x3_mode() == 9) IGI_LAUNCH((gemm_dma_x3_kernel<9>), grid, dim3(DMA_THREADS), shm, s, gg, n_tiles, m_tiles);
    else IGI_LAUNCH((gemm_dma_x3_kernel<6>), grid, dim3(DMA_THREADS), shm, s, gg, n_tiles, m_tiles);
    return hipGetLastError();
  }
  if (two_stage && bf16_mode() && !g.gather && akc) {
    // opt-in bf16-input mode: same tiles, same loaders, same epilogues
    const int n_tiles = (g.N + 127) / 128, m_tiles = (g.M + DMA_BM - 1) / DMA_BM;
    GemmArgs gg = g;
    gg.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.sCsplit & 3) == 0 && (g.N & 3) == 0 &&
                  (!g.bias || (aligned16(g.bias) && (g.sBias & 3) == 0)) &&
                  (!g.aux || (aligned16(g.aux) && (g.ldaux & 3) == 0 && (g.sAux & 3) == 0));
    constexpr size_t ring = sizeof(float) * 2 * (DMA_BM + 128) * DMA_BK, epi = sizeof(float) * DMA_WAVES * 64 * (32 + 4);
    constexpr size_t shm = ring > epi ? ring : epi;
    const dim3 grid(n_tiles * m_tiles * g.nbatch * g.splitk);
    if (bkc) {
      static bool attr = false;
      if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_bf16_kernel<true, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
        attr = true;
      }
      IGI_LAUNCH((gemm_dma_bf16_kernel<true, true>), grid, dim3(DMA_THREADS), shm, s, gg, n_tiles, m_tiles);
    } else {
      static bool attr = false;
      if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_bf16_kernel<true, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
        attr = true;
      }
      IGI_LAUNCH((gemm_dma_bf16_kernel<true, false>), grid, dim3(DMA_THREADS), shm, s, gg, n_tiles, m_tiles);
    }
    return hipGetLastError();
  }
  if (two_stage) return launch_dma_cfg<128, 2>(g, akc, bkc, s);
  if (bn == 256) return launch_dma_cfg<256>(g, akc, bkc, s);
  if (bn == 128) return launch_dma_cfg<128>(g, akc, bkc, s);
  return launch_dma_cfg<64>(g, akc, bkc, s);
}

// Launch a set of independent weight-gradient products (reduction-major operands) as grouped grids.
// Problems are bucketed by tile width; anything the LDS-DMA kernel cannot take runs on its own.
static hipError_t gemm(GemmArgs g, bool akc, bool bkc, hipStream_t s);
// dgrad (optional): a data-gradient product (A k-contiguous, B reduction-major, tanh' epilogue) sharing the grid; its
// (shorter) tiles follow the weight-gradient tiles, see below.  Returns hipErrorNotSupported when dgrad cannot ride (the caller launches it on its own).
// Can the data-gradient product `g` (k-contiguous dZ times reduction-major W, tanh' epilogue) ride in the grid of
// gemm_dma_wgrad_multi_kernel -- and, when it carries rowdot_out, emit the row dots from its tiles?  The ONE predicate
// the launcher below and the teacher's planner (latent_rowdot) both use, so a plan never asks for row dots the launch
// would decline.  Sets g.wide_epi.
static inline bool gemm_multi_dgrad_ok(GemmArgs& g) {
  const long long mtl = (g.M + DMA_BM - 1) / DMA_BM, ntl = (g.N + 127) / 128;
  if (g.epilogue != EPI_TANHGRAD || g.splitk > 1 || g.gather || !dma_eligible(g, true, false) ||
      mtl * ntl * g.nbatch > (1 << 20))
    return false;
  g.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.N & 3) == 0 &&
               (!g.aux || (aligned16(g.aux) && (g.ldaux & 3) == 0 && (g.sAux & 3) == 0));
  if (g.rowdot_out && (!g.wide_epi || !g.rowdot_W || !aligned16(g.rowdot_out))) return false;
  return true;
}

static hipError_t gemm_wgrad_multi(GemmArgs* list, int count, hipStream_t s, const GemmArgs* dgrad = nullptr) {
  GemmMulti mt_;
  double fl = 0, by = 0;
  static int dgrad_first = -1;
  if (dgrad_first < 0) { const char* e = getenv("IGI_MULTI_DGRAD_FIRST"); dgrad_first = e ? atoi(e) : 0; }
  if (dgrad) {
    GemmArgs g = *dgrad;
    if (g.splitk < 1) g.splitk = 1;
    const long long mtl = (g.M + DMA_BM - 1) / DMA_BM, ntl = (g.N + 127) / 128;
    if (!gemm_multi_dgrad_ok(g)) return hipErrorNotSupported;
    dma_set_divs(g, (int)ntl, (int)mtl);
    static int chain = -1;
    if (chain < 0) { const char* e = getenv("IGI_DGRAD_CHAIN"); chain = e ? atoi(e) : 1; if (chain < 1) chain = 1; }
    mt_.chain = (chain > 1 && mtl % chain == 0 && count > 0) ? chain : 1;
    bool loww = false;
    if (g.lw_out) {   // the planner asked for the layer below's weight gradient from these tiles: its conditions, or nothing
      // (lw_xw <= 31: lane 31 of the padded input row feeds the ONE of the bias gradient -- a 32-wide real input has no
      //  free column and must take the separate weight-gradient launch)
      if (!g.rowdot_out || !g.lw_X || !g.lw_bias || g.lw_ldx != 32 || g.lw_xw < 1 || g.lw_xw > 31 || g.lw_chain < 1 ||
          mtl % g.lw_chain != 0 || (g.M % DMA_BM) != 0 || (g.N % 128) != 0)
        return hipErrorInvalidValue;
      mt_.chain = g.lw_chain;
      loww = true;
    }
    mt_.g[0] = g; mt_.n_tiles[0] = (int)ntl; mt_.m_tiles[0] = (int)mtl; mt_.kind[0] = loww ? 6 : 2;
    mt_.tile_end[0] = (int)(mtl * ntl * g.nbatch) / mt_.chain;
    mt_.n = 1;
    fl += 2.0 * g.M * g.N * (double)g.K * g.nbatch * g.flop_credit;
    if (g.rowdot_out) fl += 2.0 * g.M * (double)g.N * 8 * g.nbatch;   // the eight extra columns of the same contraction
    if (loww) fl += 2.0 * g.M * (double)g.N * g.lw_xw * g.nbatch;      // the layer below's weight gradient (real input columns only)
    by += 4.0 * g.nbatch * ((double)g.M * g.K + (double)g.N * g.K + 2.0 * (double)g.M * g.N);
  }
  for (int i = 0; i < count; ++i) {
    GemmArgs& g = list[i];
    if (g.M <= 0 || g.N <= 0) continue;
    if (g.splitk < 1) g.splitk = 1;
    if (g.splitk > 1 && g.kchunk <= 0) {
      int c = (g.K + g.splitk - 1) / g.splitk;
      g.kchunk = (c + DMA_BK - 1) / DMA_BK * DMA_BK;
    }
    if (!dma_eligible(g, false, false) || g.gather || mt_.n >= DMA_MULTI_MAX || g.epilogue != EPI_STORE) {
      hipError_t e = gemm(g, false, false, s);
      if (e != hipSuccess) return e;
      continue;
    }
    static int narrow = -1;
    // 256 x 32 tiles for a <= 32-wide input (the zero-padded first trunk layer): slower while that product shared the
    // env level's grid with two heavier ones (round 3), faster once the level runs as the row-block kernel and the
    // product is left with the first env layer's in the step's last launch (24.0 vs 29.7 us, round 5)
    if (narrow < 0) { const char* e = getenv("IGI_WGRAD_N32"); narrow = e ? atoi(e) : (rb_level_enabled() ? 1 : 0); }
    const bool n32 = narrow && g.N <= 32 && (g.M % 256) == 0;   // 256 x 32 tiles: no padded columns for a <= 32-wide input
    const int bn = n32 ? 32 : ((g.N <= 64) ? 64 : 128);
    const int bm = n32 ? 256 : DMA_BM;
    const int nt = (g.N + bn - 1) / bn, mt = (g.M + bm - 1) / bm;
    const int tiles = nt * mt * g.nbatch * g.splitk;
    GemmArgs gg = g;
    gg.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.sCsplit & 3) == 0 && (g.N & 3) == 0;
    dma_set_divs(gg, nt, mt);
    const int k = mt_.n++;
    mt_.g[k] = gg; mt_.n_tiles[k] = nt; mt_.m_tiles[k] = mt; mt_.kind[k] = n32 ? 3 : (bn == 64 ? 1 : 0);
    mt_.tile_end[k] = (k > 0 ? mt_.tile_end[k - 1] : 0) + tiles;
    fl += 2.0 * g.M * g.N * (double)g.K * g.nbatch * g.flop_credit;
    by += 4.0 * g.nbatch * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * g.splitk);
  }
  if (mt_.n == 0) return hipSuccess;
  if (dgrad && mt_.n > 1 && !dgrad_first) {
    // longest first: a weight-gradient workgroup runs two to four times as many k-tiles as a data-gradient one, and
    // there is one of them per CU -- started first they share their CU with a stream of short data-gradient tiles and
    // finish with them; started last they ran alone at the end (fused trunk-2 level 150 -> 142 us with its 256
    // 32-k-tile workgroups, the 768-tile levels 46.2 -> 43.5)
    const GemmArgs g0 = mt_.g[0];
    const int nt0 = mt_.n_tiles[0], mt0 = mt_.m_tiles[0], k0 = mt_.kind[0], t0 = mt_.tile_end[0];
    for (int k = 0; k + 1 < mt_.n; ++k) {
      mt_.g[k] = mt_.g[k + 1]; mt_.n_tiles[k] = mt_.n_tiles[k + 1]; mt_.m_tiles[k] = mt_.m_tiles[k + 1];
      mt_.kind[k] = mt_.kind[k + 1]; mt_.tile_end[k] = mt_.tile_end[k + 1] - t0;
    }
    const int l = mt_.n - 1;
    mt_.g[l] = g0; mt_.n_tiles[l] = nt0; mt_.m_tiles[l] = mt0; mt_.kind[l] = k0;
    mt_.tile_end[l] = (l > 0 ? mt_.tile_end[l - 1] : 0) + t0;
  }
  constexpr size_t ring = sizeof(float) * 2 * (DMA_BM + 128) * DMA_BK, epi = sizeof(float) * DMA_WAVES * 64 * (32 + 4);
  constexpr size_t shm = ring > epi ? ring : epi;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_wgrad_multi_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    attr = true;
  }
  ProfScope ps(PC_WGRAD_MULTI + (g_multi_level >= 0 && g_multi_level <= 5 ? g_multi_level : 4), s, fl, by);   // rocprofv3 reports one symbol; the class carries the level
  IGI_LAUNCH(gemm_dma_wgrad_multi_kernel, dim3(mt_.tile_end[mt_.n - 1]), dim3(DMA_THREADS), shm, s, mt_);
  return hipGetLastError();
}

static hipError_t gemm_wgrad_group(GemmArgs* list, int count, hipStream_t s) {
  static int multi = -1;
  if (multi < 0) { const char* e = getenv("IGI_WGRAD_MULTI"); multi = e ? atoi(e) : 1; }
  if (multi && !bf16_mode()) return gemm_wgrad_multi(list, count, s);
  GemmGroup grp[2];   // [0]: 128-wide two-stage tiles, [1]: 64-wide three-stage tiles
  double fl[2] = {0, 0}, by[2] = {0, 0};
  for (int i = 0; i < count; ++i) {
    GemmArgs& g = list[i];
    if (g.M <= 0 || g.N <= 0) continue;
    if (g.splitk < 1) g.splitk = 1;
    if (g.splitk > 1 && g.kchunk <= 0) {
      int c = (g.K + g.splitk - 1) / g.splitk;
      g.kchunk = (c + DMA_BK - 1) / DMA_BK * DMA_BK;
    }
    const int bn = (g.N <= 64) ? 64 : 128;
    GemmGroup& G = grp[bn == 64 ? 1 : 0];
    if (!dma_eligible(g, false, false) || g.gather || G.n >= DMA_GROUP_MAX || g.epilogue != EPI_STORE) {
      hipError_t e = gemm(g, false, false, s);
      if (e != hipSuccess) return e;
      continue;
    }
    const int nt = (g.N + bn - 1) / bn, mt = (g.M + DMA_BM - 1) / DMA_BM;
    const int tiles = nt * mt * g.nbatch * g.splitk;
    GemmArgs gg = g;
    gg.wide_epi = aligned16(g.C) && (g.ldc & 3) == 0 && (g.sC & 3) == 0 && (g.sCsplit & 3) == 0 && (g.N & 3) == 0;
    dma_set_divs(gg, nt, mt);
    const int k = G.n++;
    G.g[k] = gg; G.n_tiles[k] = nt; G.m_tiles[k] = mt;
    G.tile_end[k] = (k > 0 ? G.tile_end[k - 1] : 0) + tiles;
    fl[bn == 64 ? 1 : 0] += 2.0 * g.M * g.N * (double)g.K * g.nbatch * g.flop_credit;
    by[bn == 64 ? 1 : 0] += 4.0 * g.nbatch * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * g.splitk);
  }
  if (grp[0].n > 0) {
    constexpr size_t shm = sizeof(float) * DMA_WAVES * 64 * (32 + 4) > sizeof(float) * 2 * (DMA_BM + 128) * DMA_BK
                               ? sizeof(float) * DMA_WAVES * 64 * (32 + 4) : sizeof(float) * 2 * (DMA_BM + 128) * DMA_BK;
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_group_kernel<128, false, false, 2>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (e != hipSuccess) return e;
      attr = true;
    }
    ProfScope ps(PC_GROUP_128_FF, s, fl[0], by[0]);
    if (bf16_mode()) {
      static bool attr2 = false;
      if (!attr2) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_group_kernel<128, false, false, 2, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
        attr2 = true;
      }
      IGI_LAUNCH((gemm_dma_group_kernel<128, false, false, 2, true>), dim3(grp[0].tile_end[grp[0].n - 1]),
                         dim3(DMA_THREADS), shm, s, grp[0]);
    } else
    IGI_LAUNCH((gemm_dma_group_kernel<128, false, false, 2>), dim3(grp[0].tile_end[grp[0].n - 1]),
                       dim3(DMA_THREADS), shm, s, grp[0]);
  }
  if (grp[1].n > 0) {
    constexpr size_t shm = sizeof(float) * DMA_NS * (DMA_BM + 64) * DMA_BK;
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_dma_group_kernel<64, false, false, DMA_NS>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (e != hipSuccess) return e;
      attr = true;
    }
    ProfScope ps(PC_GROUP_64_FF, s, fl[1], by[1]);
    IGI_LAUNCH((gemm_dma_group_kernel<64, false, false, DMA_NS>), dim3(grp[1].tile_end[grp[1].n - 1]),
                       dim3(DMA_THREADS), shm, s, grp[1]);
  }
  return hipGetLastError();
}

// One level of a backward chain: the data gradient `dgrad` (k-contiguous dZ times reduction-major W, tanh' epilogue)
// together with the weight-gradient products that are ready at this point, in one grid when the shapes allow it.
static inline bool gemm_level_enabled() {
  static int fuse = -1, multi = -1;
  if (fuse < 0) { const char* e = getenv("IGI_LEVEL_FUSE"); fuse = e ? atoi(e) : 1; }
  if (multi < 0) { const char* e = getenv("IGI_WGRAD_MULTI"); multi = e ? atoi(e) : 1; }
  return fuse && multi && !bf16_mode();
}
static hipError_t gemm_level(const GemmArgs& dgrad, GemmArgs* wgrads, int count, hipStream_t s) {
  if (gemm_level_enabled() && count > 0) {
    hipError_t e = gemm_wgrad_multi(wgrads, count, s, &dgrad);
    if (e != hipErrorNotSupported) return e;
  }
  if (dgrad.rowdot_out) return hipErrorNotSupported;   // only the fused tile emits the row dots: the caller planned for them
  hipError_t e = gemm(dgrad, true, false, s);
  if (e != hipSuccess) return e;
  return count > 0 ? gemm_wgrad_group(wgrads, count, s) : hipSuccess;
}

}  // namespace igi

