// Teacher PPO update on gfx950: rollout post-processing (GAE, advantage / value normalisation),
// minibatch gather + running-stat update, fused policy/value heads + PPO loss + head backward,
// deterministic split-K gradient reduction, global-norm clip + Adam.
//
// Data layout in HBM
//   * rollout arena stays time-major [t][n][.] exactly as play_steps wrote it; the reference's
//     env-major sample id b = n*T + t (experience.py:39-46) is mapped to element t*N + n on the fly,
//     so the 11 transpose-copies of prepare_training never happen;
//   * parameters / gradients / Adam moments are single flat fp32 vectors in state_dict order;
//   * per-minibatch activations are row-major [mb][width4] (width rounded up to 4 floats so rows
//     are 16-byte aligned), actor and critic stacked [2][mb][width4] so one batched launch serves both.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/igi_ppo.h"
#include "gemm_dma.h"
#include "rowblock.h"
#include "env_mlp.h"
#include "policy_fwd.h"
#include "fwd12.h"
#include "gemm_f32.h"
#include "rollout.h"

namespace igi {

#define IGI_HIP_TRY(expr)                      \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return (int)_e;      \
  } while (0)

static inline int ru4(int x) { return (x + 3) & ~3; }
static inline long long ru64(long long x) { return (x + 63) & ~63LL; }  // 256-byte granules

constexpr int PREP_THREADS = 256;
constexpr int GS_THREADS = 256;
constexpr int LOSS_THREADS = 256;
constexpr int LOSS_BLOCKS_MAX = 1024;
constexpr int LOSS_BLOCKS_DEFAULT = 512;  // measured: 256 -> 55 us, 512 -> 35 us, 1024 -> 37 us per launch
constexpr int RED_THREADS = 256;
constexpr int SUMSQ_BLOCKS = 512;
constexpr int MAX_SEG = 32;
constexpr int SLAB_GX_MAX = 2048;   // blocks per segment of k_slab_reduce (IGI_SLAB_GX is clamped to it)
constexpr int ADAM_BLOCKS_MAX = 1024;

// ---------------------------------------------------------------------------------------------
// plan: parameter offsets + workspace carve-up, recomputed from the cfg on every call (pure
// integer arithmetic on the host).
// ---------------------------------------------------------------------------------------------
struct TeacherPlan {
  int obs, priv, act, npl, nl;
  int pu[IGI_MAX_LAYERS], u[IGI_MAX_LAYERS];
  int N, T, E, mb, nmb;
  long long Bsz;
  int latent, xw, xld;  // xcat = [obs_n | latent | 0...], width xw, leading dim xld (multiple of 32)
  int u0p;              // first trunk width rounded up to 4
  // parameter offsets (floats) in the flat vector
  long long o_sigma, o_envW[IGI_MAX_LAYERS], o_envB[IGI_MAX_LAYERS];
  long long o_acW[IGI_MAX_LAYERS], o_acB[IGI_MAX_LAYERS];  // actor; critic = + ac_block
  long long ac_block, o_valW, o_valB, o_muW, o_muB, P;
  // workspace offsets (bytes)
  size_t w_prep_part, w_prep_coef, w_rms_part, w_norm_coef, w_priv, w_xcat, w_dxcat, w_w1p;
  size_t w_moments, w_traj_coef, w_traj_state, w_lat_part, w_wlat, w_lat_rowdot;
  int lat_tiles;
  int lat_fused, lat_blocks, lat_rpw;  // fused latent / last-env-layer backward (k_latent_bwd)  // per-minibatch batch moments; per-step normaliser trajectory
  size_t w_e[IGI_MAX_LAYERS], w_de[IGI_MAX_LAYERS], w_h[IGI_MAX_LAYERS], w_dh[IGI_MAX_LAYERS];
  size_t w_loss_part, w_head_slab, w_slab, w_sumsq, w_gpart, w_ppart, w_scal, w_total;
  int gae_blocks, gs_rows, gs_blocks, loss_blocks, loss_rpw;
  int loss_fused;  // heads + loss + head backward ride in the last trunk layer's forward (k_trunk_loss); loss_blocks = its m-tiles
  // backward levels that run as ONE persistent row-block kernel (rowblock.h) instead of data-gradient + weight-gradient
  // tiles: rb_ac[l] / rb_env[l] = row ranges (= weight-gradient partials) of trunk / env_mlp layer l, 0 = tile kernels
  int rb_ac[IGI_MAX_LAYERS], rb_env[IGI_MAX_LAYERS];
  // the FIRST trunk layer's weight gradient comes from the data-gradient tiles that produce its dZ (GemmArgs::lw_*):
  // lw_chain consecutive row tiles per workgroup, lw_parts partial records per net; dZ of that layer is never written
  int lw_chain, lw_parts;
  int lx_env;   // the FIRST env layer's weight gradient rides in the env level's row-block kernel (rowblock.h, LOWX); its dZ is never written
  int latz;     // round 6 (IGI_LATZ_FUSE=0 turns it off): k_latent_bwd's work at the head of the env level's blocks (rowblock.h, LATZ); de2 is never written
  int head_count;  // muW, muB, valW, valB, sigma partial vector length
  // wgrad split factors and slab offsets (floats, relative to w_slab)
  int sk_env[IGI_MAX_LAYERS], sk_ac[IGI_MAX_LAYERS];
  long long s_envW[IGI_MAX_LAYERS], s_envB[IGI_MAX_LAYERS], s_acW[IGI_MAX_LAYERS], s_acB[IGI_MAX_LAYERS];
  long long slab_floats;
};

static inline int env_in(const TeacherPlan& p, int l) { return l == 0 ? p.priv : p.pu[l - 1]; }
static inline int ac_in(const TeacherPlan& p, int l) { return l == 0 ? p.xw : p.u[l - 1]; }
static inline bool H_last_is_128(const TeacherPlan* p) { return p->u[p->nl - 1] == 128; }

static int choose_splitk(int M, int N, int K, int nbatch) {
  if (M >= 4 && N >= 4 && (M & 3) == 0 && (N & 3) == 0) {
    // the weight gradients run in gemm_dma_wgrad_multi_kernel: 128 x 128 tiles (128 x 64 up to 64 input columns).  One
    // workgroup per CU and product: with the tile width that kernel really uses (the generic planner assumes 256-wide
    // tiles for wide layers and gave the 512 -> 256 layer 32 splits = 512 workgroups of 16 k-tiles; 16 splits = 256
    // workgroups of 32 k-tiles, first in the grid, halve its slab and run 7 us shorter)
    const int bn = N <= 64 ? 64 : 128;
    // (<= 32 input columns and whole 256-row tiles: gemm_wgrad_multi runs 256 x 32 tiles, kind 3)
    static int narrow = -1;
    if (narrow < 0) { const char* e = getenv("IGI_WGRAD_N32"); narrow = e ? atoi(e) : (rb_level_enabled() ? 1 : 0); }   // see gemm_wgrad_multi
    const int bm = (narrow && N <= 32 && M % 256 == 0) ? 256 : DMA_BM;
    const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * nbatch;
    int sk = (int)(256 / tiles > 1 ? 256 / tiles : 1);
    const int maxsk = K / 128 > 1 ? K / 128 : 1;
    return sk > maxsk ? maxsk : sk;
  }
  int bm, bn;
  gemm_tile_for(M, N, &bm, &bn);
  long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * nbatch;
  int sk = (int)((256 + tiles - 1) / tiles);
  int maxsk = K / 128;
  if (sk > maxsk) sk = maxsk;
  if (sk < 1) sk = 1;
  return sk;
}

// igi_teacher_set_latz_fuse / IGI_LATZ_FUSE (initial value).  Default ON since the rank-8 weight gradient of the block runs
// on the matrix pipe (rowblock.h, latz_head): alternating A/B on one box (profiles/r06_latz_ab.log) 38.71 / 39.40 / 39.87
// updates/s on against 38.60 / 38.52 / 38.92 off -- the 9.0 us launch it removes (k_latent_bwd) and ~1.3 us of the slab sum /
// Adam pass against +4.5 us in the env level (36.1 - 39.3 -> 39.5 - 44.9 us; the transform of the staged image is vector work
// that the four column slices of a row range repeat, behind two more barriers per block).  With that phase as vector code
// (64 LDS reads and ~45 address instructions per 8 rows and thread) the two were even: 39.16 / 39.25 on, 39.16 / 39.23 off.
// The workspace carve-up does not depend on the switch.
static inline int& latz_fuse_ref() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_LATZ_FUSE"); on = e ? (atoi(e) != 0) : 1; }
  return on;
}

static int make_plan(const igi_teacher_cfg* c, TeacherPlan* p) {
  memset(p, 0, sizeof(*p));
  if (!c) return IGI_E_BADARG;
  if (c->n_priv_layers < 1 || c->n_priv_layers > IGI_MAX_LAYERS || c->n_layers < 1 ||
      c->n_layers > IGI_MAX_LAYERS || c->act_dim < 1 || c->act_dim > IGI_MAX_ACT ||
      c->obs_dim < 1 || c->priv_dim < 1 || c->num_envs < 1 || c->horizon < 1 || c->mini_epochs < 1)
    return IGI_E_BADARG;
  p->obs = c->obs_dim; p->priv = c->priv_dim; p->act = c->act_dim;
  p->npl = c->n_priv_layers; p->nl = c->n_layers;
  for (int i = 0; i < p->npl; ++i) { p->pu[i] = c->priv_units[i]; if (p->pu[i] < 1) return IGI_E_BADARG; }
  for (int i = 0; i < p->nl; ++i) { p->u[i] = c->units[i]; if (p->u[i] < 1) return IGI_E_BADARG; }
  if (p->u[p->nl - 1] > 256) return IGI_E_UNSUPPORTED;  // head kernel keeps <=4 columns per lane
  p->N = c->num_envs; p->T = c->horizon; p->E = c->mini_epochs;
  p->Bsz = (long long)p->N * p->T;
  p->mb = (int)(p->Bsz / p->E);
  if (p->mb < 2) return IGI_E_BADARG;
  p->nmb = (int)(p->Bsz / p->mb);
  p->latent = p->pu[p->npl - 1];
  p->xw = p->obs + p->latent;
  p->xld = (p->xw + 31) & ~31;  // zero-padded to the LDS-DMA kernel's k-tile
  p->u0p = ru4(p->u[0]);

  // every tensor starts on a 16-byte boundary (gaps stay zero) so weight tiles load as float4
  long long o = 0;
  auto put = [&](long long n) { long long at = o; o = (o + n + 3) & ~3LL; return at; };
  p->o_sigma = put(p->act);
  for (int l = 0; l < p->npl; ++l) {
    p->o_envW[l] = put((long long)p->pu[l] * env_in(*p, l));
    p->o_envB[l] = put(p->pu[l]);
  }
  long long ac0 = o;
  for (int l = 0; l < p->nl; ++l) {
    p->o_acW[l] = put((long long)p->u[l] * ac_in(*p, l));
    p->o_acB[l] = put(p->u[l]);
  }
  p->ac_block = o - ac0;
  o += p->ac_block;  // critic: same shapes, same internal offsets
  const int H = p->u[p->nl - 1];
  p->o_valW = put(H);
  p->o_valB = put(1);
  p->o_muW = put((long long)p->act * H);
  p->o_muB = put(p->act);
  p->P = o;

  // ---- workspace
  const long long mb = p->mb;
  size_t w = 0;
  auto take = [&](size_t bytes) { size_t at = w; w += (size_t)ru64((long long)bytes); return at; };
  p->gae_blocks = (p->N + PREP_THREADS - 1) / PREP_THREADS;
  p->w_prep_part = take(sizeof(double) * 6 * p->gae_blocks);
  p->w_prep_coef = take(sizeof(float) * 8);
  const int D = p->obs + p->priv;
  p->gs_rows = 32;
  while (p->gs_rows > 8 && (size_t)p->gs_rows * (D + 2) * sizeof(float) > 48 * 1024) p->gs_rows /= 2;
  p->gs_blocks = (int)((mb + p->gs_rows - 1) / p->gs_rows);
  p->w_rms_part = take(sizeof(double) * 2 * D * p->gs_blocks * p->nmb);
  p->w_norm_coef = take(sizeof(float) * 2 * D);
  p->w_moments = take(sizeof(float) * 2 * D * p->nmb);
  p->w_traj_coef = take(sizeof(float) * 2 * D * p->E * p->nmb);
  p->w_traj_state = take(sizeof(double) * (2 * D + 2) * p->E * p->nmb);
  {
    const int K2 = 2 * ru4(p->u[0]);
    // LDS: transposed weight slice 8 x (K2p+4) + 32 x 260 staging tile (also hosts the 4 x (8*H2+8) final reduction)
    p->lat_fused = (p->latent == 8 && p->npl >= 2 && p->pu[p->npl - 2] <= 256 && K2 <= 2048) ? 1 : 0;
    p->lat_rpw = 0;
    p->lat_blocks = (int)((mb + 31) / 32);
    p->w_lat_part = p->lat_fused ? take(sizeof(float) * (size_t)p->lat_blocks * (8 * p->pu[p->npl - 2] + 8)) : 0;
    // row dots of dZ1 with the eight latent columns, one partial per 128-column tile and net (see k_latent_bwd<., true>)
    p->lat_tiles = 2 * ((p->u[0] + 127) / 128);
    p->w_lat_rowdot = p->lat_fused ? take(sizeof(float) * (size_t)p->lat_tiles * mb * 8) : 0;
    p->w_wlat = p->lat_fused ? take(sizeof(float) * 8 * (size_t)((K2 + 255) / 256 * 256)) : 0;  // [8][K2p], see k_latent_bwd
  }
  p->w_priv = take(sizeof(float) * mb * ru4(p->priv));
  p->w_xcat = take(sizeof(float) * mb * p->xld);
  p->w_dxcat = take(sizeof(float) * mb * p->xld);
  p->w_w1p = take(sizeof(float) * 2 * p->u0p * p->xld);
  for (int l = 0; l < p->npl; ++l) {
    // rows rounded up to the fused env_mlp kernel's 64-row blocks: it stores whole blocks (env_mlp.h)
    p->w_e[l] = (l < p->npl - 1) ? take(sizeof(float) * ((mb + 63) / 64 * 64) * ru4(p->pu[l])) : 0;
    p->w_de[l] = (l < p->npl - 1) ? take(sizeof(float) * mb * ru4(p->pu[l])) : 0;  // last: dxcat[:, obs:]
  }
  for (int l = 0; l < p->nl; ++l) {
    p->w_h[l] = take(sizeof(float) * 2 * mb * ru4(p->u[l]));
    p->w_dh[l] = take(sizeof(float) * 2 * mb * ru4(p->u[l]));
  }
  // loss kernel: one wave per row, loss_rpw rows per wave
  long long waves_needed = mb;
  int blocks = (int)((waves_needed + 4 * 4 - 1) / (4 * 4));
  int max_blocks = LOSS_BLOCKS_DEFAULT;
  if (const char* e = getenv("IGI_LOSS_BLOCKS")) {  // tuning knob (power of two, <= 1024)
    const int v = atoi(e);
    if (v >= 1 && v <= LOSS_BLOCKS_MAX) max_blocks = v;
  }
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks < 1) blocks = 1;
  p->loss_blocks = blocks;
  p->loss_rpw = (int)((mb + (long long)blocks * 4 - 1) / ((long long)blocks * 4));
  {
    // the fused last-layer forward + loss (k_trunk_loss): decided from the shapes alone, so that the workspace carve-up,
    // the slab sums and the statistics kernel agree on the number of partial records (one per 64-row m-tile)
    static int fused = -1;
    if (fused < 0) { const char* e = getenv("IGI_LOSS_FUSED"); fused = e ? atoi(e) : 1; }
    // (the launch site re-checks dma_eligible and the 16-byte alignment of bias / dh: every offset and stride that enters
    // those checks is tested HERE, so that a shape which passes keeps passing at the launch and one which does not keeps
    // the two-launch path -- only a misaligned base pointer of the caller remains an error there)
    const int ll = p->nl - 1;
    const bool aligned = (p->o_acW[ll] & 3) == 0 && (p->o_acB[ll] & 3) == 0 && (p->ac_block & 3) == 0 &&
                         (ru4(p->u[ll]) & 3) == 0 && (ru4(p->u[ll - (ll > 0)]) & 3) == 0 && ((long long)p->mb * ru4(p->u[ll]) & 3) == 0;
    p->loss_fused = (fused && p->nl >= 2 && H_last_is_128(p) && p->act <= 7 && (p->u[p->nl - 2] % DMA_BK) == 0 &&
                     p->Bsz < (1LL << 31) && !bf16_mode() && aligned) ? 1 : 0;
    if (p->loss_fused) p->loss_blocks = (int)((mb + 63) / 64);   // TrunkLossHook::TILE_M
  }
  p->w_loss_part = take(sizeof(double) * 8 * p->loss_blocks);
  p->head_count = p->act * H + p->act + H + 1 + p->act;
  p->w_head_slab = take(sizeof(float) * (size_t)p->head_count * p->loss_blocks);
  // wgrad slabs.  Every product picks its split factor as if it had the chip to itself (power-of-two k-chunks).
  // Planning the factors jointly for the shared grid (equal work per workgroup, the trunk products filling the 512
  // workgroup slots exactly once: 84 -> ~40 MB of slabs) was measured and is SLOWER: 141 us + 14 us reduce against
  // 131.5 + 18.5 (`tools/scratch`-style sweep over slot targets 256..2048): uneven k-chunks lose the
  // per-problem XCD grouping and long workgroups run below the k-loop's steady rate.
  // IGI_SK_OVERRIDE="e0,e1,e2,a0,a1,a2" (0 = keep): split factors of the env / trunk weight gradients, for A/B runs
  int sk_over[2 * IGI_MAX_LAYERS] = {0};
  if (const char* e = getenv("IGI_SK_OVERRIDE")) {
    int i = 0;
    for (const char* q = e; *q && i < 2 * IGI_MAX_LAYERS; ++i) {
      sk_over[i] = atoi(q);
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
  }
  auto pick = [&](int dflt, int over, int K) {
    if (over <= 0) return dflt;
    const int maxsk = K / 128 > 1 ? K / 128 : 1;
    return over > maxsk ? maxsk : over;
  };
  long long s = 0;
  for (int l = 0; l < p->npl; ++l) {
    int sk = choose_splitk(p->pu[l], env_in(*p, l), p->mb, 1);
    // the first env layer's weight gradient is a launch of its own and has to fill the chip; the later ones share the
    // env-level launch with two other products: half the split (8 k-tiles per workgroup instead of 4, half the slab) --
    // env level 44.3 -> 42.4 us, slab sum 17.2 -> 15.4 us, A/B of tools/probes/sk_ab.sh (every other factor: slower)
    if (l > 0 && sk > 1) sk /= 2;
    p->sk_env[l] = pick(sk, sk_over[l], p->mb);
    // layer l's weight gradient and the data gradient into layer l - 1 as one row-block kernel (the last layer's
    // backward is k_latent_bwd's when lat_fused): decided from the shapes alone, like every other plan entry
    if (l >= 1 && l < p->npl - p->lat_fused && rb_level_shape_ok(p->mb, p->pu[l], p->pu[l - 1], 1)) {
      p->rb_env[l] = rb_level_ranges(p->mb, p->pu[l - 1], 1);
      p->sk_env[l] = p->rb_env[l];
    }
    p->s_envW[l] = s; s += (long long)p->sk_env[l] * p->pu[l] * env_in(*p, l);
    p->s_envB[l] = s; s += (long long)p->sk_env[l] * p->pu[l];
    s = (s + 3) & ~3LL;
  }
  {
    static int lx_on = -1;
    if (lx_on < 0) { const char* e = getenv("IGI_LOWX_FUSE"); lx_on = e ? atoi(e) : 1; }
    if (lx_on && p->npl >= 2 && p->rb_env[1] && p->priv == 64) {
      // env layer 0's weight gradient from the data-gradient tiles of the level above: rb_env[1] partial records; the slab
      // offsets of the layers behind it move accordingly (recomputed below)
      p->lx_env = 1;
      p->sk_env[0] = p->rb_env[1];
      s = 0;
      for (int l = 0; l < p->npl; ++l) {
        p->s_envW[l] = s; s += (long long)p->sk_env[l] * p->pu[l] * env_in(*p, l);
        p->s_envB[l] = s; s += (long long)p->sk_env[l] * p->pu[l];
        s = (s + 3) & ~3LL;
      }
    }
  }
  {
    const int latz_on = latz_fuse_ref();
    // shapes only: the fused latent path with the row dots from the dZ1 tiles, the env level as the row-block kernel with the
    // first env layer's weight gradient in it, the reference's 128-wide second env layer
    p->latz = (latz_on && p->lat_fused && p->npl == 3 && p->rb_env[1] && p->lx_env && p->pu[1] == 128 && p->lat_tiles <= 8 &&
               p->lat_blocks >= p->rb_env[1]) ? 1 : 0;
  }
  for (int l = 0; l < p->nl; ++l) {
    const int inw = (l == 0) ? p->xld : ac_in(*p, l);  // layer 0 multiplies the padded xcat
    p->sk_ac[l] = pick(choose_splitk(p->u[l], inw, p->mb, 2), sk_over[p->npl + l], p->mb);
    // (l >= 2: the data gradient into trunk layer 0 keeps its interleaved layout and the latent row dots)
    if (l >= 2 && rb_level_shape_ok(p->mb, p->u[l], p->u[l - 1], 2)) {
      p->rb_ac[l] = rb_level_ranges(p->mb, p->u[l - 1], 2);
      p->sk_ac[l] = p->rb_ac[l];
    }
    if (l == 0) {
      static int lw_on = -1;
      if (lw_on < 0) { const char* e = getenv("IGI_LOWW_FUSE"); lw_on = e ? atoi(e) : 1; }
      const int mt128 = p->mb / DMA_BM;
      const int chain = (mt128 % 4 == 0) ? 4 : ((mt128 % 2 == 0) ? 2 : 1);
      // shapes only (the launch re-checks pointers): row dots from those tiles, 32-wide padded input with a FREE last
      // column (xw <= 31: column 31 carries the ONE of the bias gradient; obs + latent == 32 takes the separate
      // weight-gradient launch), whole 128-row / 128-column tiles, the level-fused grid
      if (lw_on && p->nl >= 2 && p->lat_fused && p->xld == 32 && p->xw < 32 && (p->mb % DMA_BM) == 0 && (p->u[0] % 128) == 0 &&
          gemm_level_enabled() && p->mb >= 4) {
        p->lw_chain = chain;
        p->lw_parts = mt128 / chain;
        p->sk_ac[0] = p->lw_parts;
      }
    }
    // layout [split][net][...]: split stride = 2*size so the batch stride stays the net size
    p->s_acW[l] = s; s += (long long)p->sk_ac[l] * 2 * p->u[l] * inw;
    p->s_acB[l] = s; s += (long long)p->sk_ac[l] * 2 * p->u[l];
    s = (s + 3) & ~3LL;
  }
  p->slab_floats = s;
  p->w_slab = take(sizeof(float) * (size_t)s);
  p->w_sumsq = take(sizeof(double) * 2 * SUMSQ_BLOCKS);
  p->w_gpart = take(sizeof(double) * (size_t)SLAB_GX_MAX * MAX_SEG);   // k_slab_reduce's per-block gradient sums of squares (norm fusion)
  p->w_ppart = take(sizeof(double) * 2 * ADAM_BLOCKS_MAX);             // the Adam blocks' parameter sums of squares, two alternating sets
  p->w_scal = take(sizeof(float) * 8);
  p->w_total = w;
  return 0;
}

template <typename T>
static inline T* wsp(const igi_teacher_state* st, size_t off) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(st->workspace) + off);
}

// ---------------------------------------------------------------------------------------------
// block reduction helpers (deterministic: fixed shuffle tree + fixed wave order)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// fp32 wave-wide sum, result in every lane.  DPP lane permutes inside each 16-lane row
// (quad_perm xor-1 / xor-2, row_half_mirror, row_mirror: 4 VALU ops, no LDS crossbar) and four
// v_readlane across the rows, instead of six dependent ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror  -> every lane of a row holds the row's sum
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}

// Eight wave-wide sums at once: after the call lane l holds the sum over all 64 lanes of v[l & 7].
// Each butterfly step halves the number of live registers by keeping, per lane, only the value its low lane
// bits select (8 -> 4 -> 2 -> 1), so the whole thing is ~30 instructions instead of 8 x 11.
__device__ __forceinline__ float wave_sum8(const float (&v)[8], int lane) {
  float w[4], u[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j] + dpp_mov<0xB1>(v[2 * j]);          // lanes l, l^1
    const float b = v[2 * j + 1] + dpp_mov<0xB1>(v[2 * j + 1]);
    w[j] = (lane & 1) ? b : a;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float a = w[2 * i] + dpp_mov<0x4E>(w[2 * i]);          // lanes l, l^2
    const float b = w[2 * i + 1] + dpp_mov<0x4E>(w[2 * i + 1]);
    u[i] = (lane & 2) ? b : a;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {                                  // the four quads of a 16-lane row
    u[i] += dpp_mov<0x128>(u[i]);                                // row_ror:8
    u[i] += dpp_mov<0x124>(u[i]);                                // row_ror:4
  }
  float z = (lane & 4) ? u[1] : u[0];
  z += __shfl_xor(z, 16, 64);                                    // the four rows
  z += __shfl_xor(z, 32, 64);
  return z;
}
// sum over the first 16-lane row (the callers' values live in lanes 0..7, the rest of the row is zero),
// returned wave-uniform
__device__ __forceinline__ float row0_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x128>(v);   // row_ror:8 then row_ror:4: all four quads, whatever the rotate direction
  v += dpp_mov<0x124>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
}

// the same sum for every 16-lane row at once, left in all lanes of the row (same DPP order as row0_sum)
__device__ __forceinline__ float rows_sum_ror(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x128>(v);
  v += dpp_mov<0x124>(v);
  return v;
}

// ---------------------------------------------------------------------------------------------
// 1. GAE + returns (experience.py:242-255): one thread per env walks T backwards over
//    [t][env]-coalesced loads.  Also accumulates the six fp64 sums that the advantage
//    normalisation (experience.py:261-262) and the two value_mean_std updates
//    (frozen_ppo.py:719-723) need.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PREP_THREADS) void k_gae(const float* __restrict__ rewards,
                                                       const float* __restrict__ values,
                                                       const uint8_t* __restrict__ dones,
                                                       const float* __restrict__ last_values,
                                                       float* __restrict__ returns_raw, int N, int T,
                                                       float gamma, float gamma_tau,
                                                       double* __restrict__ partials) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  double s[6] = {0, 0, 0, 0, 0, 0};
  if (n < N) {
    float lam = 0.f;
    float nextv = last_values[n];
    for (int t = T - 1; t >= 0; --t) {
      const long long i = (long long)t * N + n;
      const float v = values[i];
      const float nn = 1.0f - (float)dones[i];
      // rewards + gamma*next_values*nn - values, each product/sum rounded (no contraction)
      const float delta = (rewards[i] + (gamma * nextv) * nn) - v;
      lam = delta + (gamma_tau * nn) * lam;
      const float ret = lam + v;
      returns_raw[i] = ret;
      const float adv = ret - v;
      s[0] += adv; s[1] += (double)adv * adv;
      s[2] += v;   s[3] += (double)v * v;
      s[4] += ret; s[5] += (double)ret * ret;
      nextv = v;
    }
  }
  __shared__ double red[PREP_THREADS / 64][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double r = wave_sum(s[j]);
    if (lane == 0) red[wave][j] = r;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double r = 0;
    for (int w = 0; w < PREP_THREADS / 64; ++w) r += red[w][threadIdx.x];
    partials[blockIdx.x * 6 + threadIdx.x] = r;
  }
}

__device__ __forceinline__ void chan_merge(double& mean, double& var, double& count, float b_mean,
                                           float b_var, double n) {
  // running_mean_std.py:48-58 with fp32 batch moments promoted to fp64
  const double delta = (double)b_mean - mean;
  const double tot = count + n;
  const double new_mean = mean + delta * n / tot;
  const double m2 = var * count + (double)b_var * n + delta * delta * count * n / tot;
  mean = new_mean;
  var = m2 / tot;
  count = tot;
}

// coef: [adv_mean, adv_std+1e-8, v_mean, v_den, r_mean, r_den]
__global__ void k_prep_final(const double* __restrict__ partials, int nblocks, long long B,
                             double* __restrict__ rms_value, float eps, float* __restrict__ coef,
                             int normalize_value) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int b = 0; b < nblocks; ++b)
    for (int j = 0; j < 6; ++j) s[j] += partials[b * 6 + j];
  const double n = (double)B;
  float mean[3], var[3];
  for (int q = 0; q < 3; ++q) {
    const double m = s[2 * q] / n;
    double v = (s[2 * q + 1] - n * m * m) / (n - 1.0);  // unbiased (torch.std / var default)
    if (v < 0) v = 0;
    mean[q] = (float)m;
    var[q] = (float)v;
  }
  coef[0] = mean[0];
  coef[1] = sqrtf(var[0]) + 1e-8f;
  if (!normalize_value) return;
  double rm = rms_value[0], rv = rms_value[1], rc = rms_value[2];
  chan_merge(rm, rv, rc, mean[1], var[1], n);       // value_mean_std(values)  (train)
  coef[2] = (float)rm;
  coef[3] = sqrtf((float)rv + eps);
  chan_merge(rm, rv, rc, mean[2], var[2], n);       // value_mean_std(returns) (train)
  coef[4] = (float)rm;
  coef[5] = sqrtf((float)rv + eps);
  rms_value[0] = rm; rms_value[1] = rv; rms_value[2] = rc;
}

__device__ __forceinline__ float clamp5(float y) { return fminf(fmaxf(y, -5.0f), 5.0f); }

__global__ __launch_bounds__(PREP_THREADS) void k_prep_norm(
    const float* __restrict__ values, const float* __restrict__ returns_raw,
    const float* __restrict__ mus, const float* __restrict__ sigmas, const float* __restrict__ coef,
    float* __restrict__ adv, float* __restrict__ values_n, float* __restrict__ returns_n,
    float* __restrict__ mus_w, float* __restrict__ sigmas_w, long long B, int act,
    int normalize_value) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const float am = coef[0], ad = coef[1], vm = coef[2], vd = coef[3], rm = coef[4], rd = coef[5];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += stride) {
    const float v = values[i], r = returns_raw[i];
    adv[i] = ((r - v) - am) / ad;
    values_n[i] = normalize_value ? clamp5((v - vm) / vd) : v;
    returns_n[i] = normalize_value ? clamp5((r - rm) / rd) : r;
  }
  const long long BA = B * act;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < BA; i += stride) {
    mus_w[i] = mus[i];
    sigmas_w[i] = sigmas[i];
  }
}

// ---------------------------------------------------------------------------------------------
// 2. minibatch gather (experience.py:207-226) + column statistics for the two in-loop
//    RunningMeanStd updates (frozen_ppo.py:521-522).  Each block stages gs_rows gathered rows in
//    LDS, writes them out raw, then one thread per column sums that tile in fp64.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GS_THREADS) void k_gather_stats(
    const float* __restrict__ obses, const float* __restrict__ priv_info,
    const int64_t* __restrict__ perm, long long start, int mb, int N, int T, int obs, int priv,
    int rows_per_block, float* __restrict__ xcat, int xld, float* __restrict__ priv_g, int pld,
    double* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int D = obs + priv;
  const int LD = D + 1;
  const int r0 = blockIdx.x * rows_per_block;
  const int nrows = min(rows_per_block, mb - r0);
  int* rowi = reinterpret_cast<int*>(tile + rows_per_block * LD);  // time-major element index per row
  for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
    const long long b = perm[start + r0 + r];
    const int n = (int)(b / T);
    rowi[r] = (int)(b - (long long)n * T) * N + n;   // b = n*T + t  ->  t*N + n
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nrows * obs; e += blockDim.x) {
    const int r = e / obs, c = e - r * obs;
    const float x = obses[(long long)rowi[r] * obs + c];
    tile[r * LD + c] = x;
    xcat[(long long)(r0 + r) * xld + c] = x;
  }
  for (int e = threadIdx.x; e < nrows * priv; e += blockDim.x) {
    const int r = e / priv, c = e - r * priv;
    const float x = priv_info[(long long)rowi[r] * priv + c];
    tile[r * LD + obs + c] = x;
    priv_g[(long long)(r0 + r) * pld + c] = x;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    double s = 0, s2 = 0;
    for (int r = 0; r < nrows; ++r) {
      const double x = tile[r * LD + c];
      s += x;
      s2 += x * x;
    }
    partials[((long long)blockIdx.x * D + c) * 2 + 0] = s;
    partials[((long long)blockIdx.x * D + c) * 2 + 1] = s2;
  }
}

// one block of 1024 threads: thread (c = tid % 128, j = tid / 128) sums the per-block partials
// b = j, j+8, ... of column c in fixed order; 8 sub-sums are then combined in fixed order, merged
// into the fp64 running state and the fp32 (mean, sqrt(var+eps)) pair of the normalise pass emitted.
constexpr int RMSF_THREADS = 1024;
__global__ __launch_bounds__(RMSF_THREADS) void k_rms_final(const double* __restrict__ partials,
                                                             int nblocks, int rows, int obs, int priv,
                                                             double* __restrict__ rms_obs,
                                                             double* __restrict__ rms_priv, float eps,
                                                             float* __restrict__ coef) {
  const int D = obs + priv;
  __shared__ double sh[2][8][128];
  __shared__ double cnt[2];
  if (threadIdx.x == 0) { cnt[0] = rms_obs[2 * obs]; cnt[1] = rms_priv[2 * priv]; }
  const double n = (double)rows;
  const int cl = threadIdx.x & 127, j = threadIdx.x >> 7;
  for (int c0 = 0; c0 < D; c0 += 128) {
    const int c = c0 + cl;
    double s = 0, s2 = 0;
    if (c < D) {
      for (int b = j; b < nblocks; b += 8) {
        s += partials[((long long)b * D + c) * 2 + 0];
        s2 += partials[((long long)b * D + c) * 2 + 1];
      }
    }
    sh[0][j][cl] = s;
    sh[1][j][cl] = s2;
    __syncthreads();
    if (j == 0 && c < D) {
      s = 0; s2 = 0;
      for (int q = 0; q < 8; ++q) { s += sh[0][q][cl]; s2 += sh[1][q][cl]; }
      const double m = s / n;
      double v = (s2 - n * m * m) / (n - 1.0);
      if (v < 0) v = 0;
      const bool is_obs = c < obs;
      double* stt = is_obs ? rms_obs : rms_priv;
      const int d = is_obs ? obs : priv;
      const int cc = is_obs ? c : c - obs;
      double mean = stt[cc], var = stt[d + cc], count = cnt[is_obs ? 0 : 1];
      chan_merge(mean, var, count, (float)m, (float)v, n);
      stt[cc] = mean;
      stt[d + cc] = var;
      coef[2 * c + 0] = (float)mean;
      coef[2 * c + 1] = sqrtf((float)var + eps);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { rms_obs[2 * obs] = cnt[0] + n; rms_priv[2 * priv] = cnt[1] + n; }
}

// eval-mode coefficients straight from the running state (model_act path)
__global__ void k_rms_coef(int obs, int priv, const double* __restrict__ rms_obs,
                           const double* __restrict__ rms_priv, float eps, float* __restrict__ coef) {
  const int D = obs + priv;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const bool is_obs = c < obs;
    const double* stt = is_obs ? rms_obs : rms_priv;
    const int d = is_obs ? obs : priv;
    const int cc = is_obs ? c : c - obs;
    coef[2 * c + 0] = (float)stt[cc];
    coef[2 * c + 1] = sqrtf((float)stt[d + cc] + eps);
  }
}

// ---- normaliser trajectory ---------------------------------------------------------------------
// The two in-loop RunningMeanStd updates (frozen_ppo.py:521-522) depend only on the rollout and the
// fixed permutation, never on the parameters: minibatch i has the same batch moments in every
// mini-epoch.  So the whole sequence of E*n_mb Chan merges is evaluated ONCE per update (8 moment
// reductions + one 79-thread scan) and every optimizer step just looks its (mean, sqrt(var+eps)) up:
// the per-step critical path loses a grid-wide reduction and two launches.
__global__ __launch_bounds__(RMSF_THREADS) void k_mb_moments(const double* __restrict__ partials, int nblocks,
                                                             int rows, int D, float* __restrict__ moments) {
  __shared__ double sh[2][8][128];
  const double* part = partials + (long long)blockIdx.x * nblocks * D * 2;
  float* mom = moments + (long long)blockIdx.x * D * 2;
  const double n = (double)rows;
  const int cl = threadIdx.x & 127, j = threadIdx.x >> 7;
  for (int c0 = 0; c0 < D; c0 += 128) {
    const int c = c0 + cl;
    double s = 0, s2 = 0;
    if (c < D) {
      for (int b = j; b < nblocks; b += 8) {
        s += part[((long long)b * D + c) * 2 + 0];
        s2 += part[((long long)b * D + c) * 2 + 1];
      }
    }
    sh[0][j][cl] = s;
    sh[1][j][cl] = s2;
    __syncthreads();
    if (j == 0 && c < D) {
      s = 0; s2 = 0;
      for (int q = 0; q < 8; ++q) { s += sh[0][q][cl]; s2 += sh[1][q][cl]; }
      const double m = s / n;
      double v = (s2 - n * m * m) / (n - 1.0);
      if (v < 0) v = 0;
      mom[2 * c] = (float)m;       // fp32 batch moments, as x.mean(0) / x.var(0) are
      mom[2 * c + 1] = (float)v;
    }
    __syncthreads();
  }
}

// thread c scans column c through all steps; state row layout: [obs mean, obs var, obs count | priv ...]
__global__ void k_rms_traj(const float* __restrict__ moments, int nmb, int steps, int rows, int obs, int priv,
                           const double* __restrict__ rms_obs, const double* __restrict__ rms_priv, float eps,
                           float* __restrict__ traj_coef, double* __restrict__ traj_state) {
  const int D = obs + priv;
  const int srow = 2 * D + 2;
  const double n = (double)rows;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const bool is_obs = c < obs;
    const double* stt = is_obs ? rms_obs : rms_priv;
    const int d = is_obs ? obs : priv;
    const int cc = is_obs ? c : c - obs;
    const int base = is_obs ? 0 : 2 * obs + 1;
    double mean = stt[cc], var = stt[d + cc], count = stt[2 * d];
    for (int k = 0; k < steps; ++k) {
      const float* mom = moments + ((long long)(k % nmb) * D + c) * 2;
      chan_merge(mean, var, count, mom[0], mom[1], n);
      traj_coef[((long long)k * D + c) * 2] = (float)mean;
      traj_coef[((long long)k * D + c) * 2 + 1] = sqrtf((float)var + eps);
      double* row = traj_state + (long long)k * srow + base;
      row[cc] = mean;
      row[d + cc] = var;
      if (cc == 0) row[2 * d] = count;
    }
  }
}

// per step: gather the minibatch rows, normalise with the step's looked-up statistics, publish the
// step's running state, and refresh the zero-padded first-layer weight (extra blocks).
struct GatherArgs {
  const float* obses; const float* priv_info; const int64_t* perm;
  long long start; int mb, N, T, obs, priv, rows_per_block, gather_blocks;
  const float* coef; const double* state_row; double* rms_obs; double* rms_priv;
  float* xcat; int xld, xw; float* priv_g; int pld;
  const float* params; long long o_w, ac_block; int u0, u0p; float* w1p; float* wlat; int K2p;
};

// bid / nblocks: this body's block index and block count (it also runs as the second half of k_adam_gather)
__device__ __forceinline__ void gather_normalize_body(const GatherArgs& a, int bid, int nblocks) {
  const float* __restrict__ obses = a.obses; const float* __restrict__ priv_info = a.priv_info;
  const int64_t* __restrict__ perm = a.perm;
  const long long start = a.start; const int mb = a.mb, N = a.N, T = a.T, obs = a.obs, priv = a.priv;
  const int rows_per_block = a.rows_per_block, gather_blocks = a.gather_blocks;
  const float* __restrict__ coef = a.coef; const double* __restrict__ state_row = a.state_row;
  double* __restrict__ rms_obs = a.rms_obs; double* __restrict__ rms_priv = a.rms_priv;
  float* __restrict__ xcat = a.xcat; const int xld = a.xld, xw = a.xw; float* __restrict__ priv_g = a.priv_g;
  const int pld = a.pld; const float* __restrict__ params = a.params; const long long o_w = a.o_w, ac_block = a.ac_block;
  const int u0 = a.u0, u0p = a.u0p; float* __restrict__ w1p = a.w1p; float* __restrict__ wlat = a.wlat;
  const int K2p = a.K2p;
  if (bid >= gather_blocks) {  // W1p[net][o][c] refresh (see k_pad_w1)
    const int total = 2 * u0p * xld;
    const int nb = nblocks - gather_blocks;
    for (int e = (bid - gather_blocks) * blockDim.x + threadIdx.x; e < total; e += nb * blockDim.x) {
      const int c = e % xld;
      const int o = (e / xld) % u0p;
      const int net = e / (xld * u0p);
      const float v = (c < xw && o < u0) ? params[o_w + net * ac_block + (long long)o * xw + c] : 0.f;
      w1p[e] = v;
      // compact, transposed copy of the 8 latent columns for k_latent_bwd: wlat[j][k], k = net*u0p + o
      if (wlat && c >= obs && c < obs + 8) wlat[(long long)(c - obs) * K2p + net * u0p + o] = v;
    }
    if (wlat) {  // zero tail k in [2*u0p, K2p)
      const int tail = K2p - 2 * u0p;
      for (int e = (bid - gather_blocks) * blockDim.x + threadIdx.x; e < 8 * tail; e += nb * blockDim.x)
        wlat[(long long)(e / tail) * K2p + 2 * u0p + e % tail] = 0.f;
    }
    return;
  }
  __shared__ int rowi[64];
  if (bid == 0) {  // the running state after this step (what RunningMeanStd would now hold)
    for (int e = threadIdx.x; e < 2 * obs + 1; e += blockDim.x) rms_obs[e] = state_row[e];
    for (int e = threadIdx.x; e < 2 * priv + 1; e += blockDim.x) rms_priv[e] = state_row[2 * obs + 1 + e];
  }
  const int r0 = bid * rows_per_block;
  const int nrows = min(rows_per_block, mb - r0);
  for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
    const long long b = perm[start + r0 + r];
    const int n = (int)(b / T);
    rowi[r] = (int)(b - (long long)n * T) * N + n;
  }
  __syncthreads();
  // wave w takes rows w, w+4, ...; lanes stride the columns (no per-element integer division)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // a wave's rows eight at a time: the eight gathered loads go out together (one per trip was eight dependent round
  // trips per wave -- the whole 12 us of this body)
  for (int c = lane; c < priv; c += 64) {
    const float m = coef[2 * (obs + c)], d = coef[2 * (obs + c) + 1];
    for (int rb = wave; rb < nrows; rb += 8 * nw) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = priv_info[(long long)rowi[min(rb + u * nw, nrows - 1)] * priv + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u * nw;
        if (r < nrows) priv_g[(long long)(r0 + r) * pld + c] = clamp5((x[u] - m) / d);
      }
    }
  }
  for (int c = lane; c < xld; c += 64) {
    if (c < obs) {
      const float m = coef[2 * c], d = coef[2 * c + 1];
      for (int rb = wave; rb < nrows; rb += 8 * nw) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = obses[(long long)rowi[min(rb + u * nw, nrows - 1)] * obs + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = rb + u * nw;
          if (r < nrows) xcat[(long long)(r0 + r) * xld + c] = clamp5((x[u] - m) / d);
        }
      }
    } else if (c >= xw) {
      for (int r = wave; r < nrows; r += nw) xcat[(long long)(r0 + r) * xld + c] = 0.f;  // keep the padding zero
    }
  }
}

__global__ __launch_bounds__(GS_THREADS) void k_gather_normalize(const GatherArgs a) {
  gather_normalize_body(a, (int)blockIdx.x, (int)gridDim.x);
}

__global__ __launch_bounds__(256) void k_normalize(float* __restrict__ xcat, int xld, int xw,
                                                   float* __restrict__ priv_g, int pld, int rows,
                                                   int obs, int priv,
                                                   const float* __restrict__ coef) {
  const int D = obs + priv;
  const long long total = (long long)rows * D;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const long long r = e / D;
    const int c = (int)(e - r * D);
    const float m = coef[2 * c], d = coef[2 * c + 1];
    float* p = (c < obs) ? &xcat[r * xld + c] : &priv_g[r * pld + (c - obs)];
    *p = clamp5((*p - m) / d);
  }
  // keep the zero padding of xcat zero (columns xw..xld-1 feed the padded first layer)
  const int padw = xld - xw;
  const long long ptotal = (long long)rows * padw;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < ptotal; e += stride) {
    const long long r = e / padw;
    xcat[r * xld + xw + (int)(e - r * padw)] = 0.f;
  }
}

// raw (un-gathered) rows -> workspace, for the inference path
__global__ __launch_bounds__(256) void k_copy_rows(const float* __restrict__ obs_in,
                                                   const float* __restrict__ priv_in, int rows, int obs,
                                                   int priv, float* __restrict__ xcat, int xld,
                                                   float* __restrict__ priv_g, int pld) {
  const int D = obs + priv;
  const long long total = (long long)rows * D;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const long long r = e / D;
    const int c = (int)(e - r * D);
    if (c < obs) xcat[r * xld + c] = obs_in[r * obs + c];
    else priv_g[r * pld + (c - obs)] = priv_in[r * priv + (c - obs)];
  }
}

// ---------------------------------------------------------------------------------------------
// 3. heads + PPO loss + head backward (models_split.py:222-250; frozen_ppo.py:543-570, 618).
//    One wave per minibatch row: lane l holds columns l, l+64, ... of the last hidden layer of
//    actor and critic, so mu/value are a wave reduction, the loss scalars are computed redundantly
//    on every lane, and d(hidden) leaves as one coalesced row.  Head weight gradients accumulate in
//    registers over the wave's rows and leave as one per-block partial (reduced later in fixed
//    order -> bitwise reproducible).
// ---------------------------------------------------------------------------------------------
struct LossArgs {
  const float* h;        // [2][mb][ldh] last hidden (actor, critic)
  float* dh;             // [2][mb][ldh] d(pre-activation) of the last hidden layer
  long long net_stride;  // mb*ldh
  int ldh, H;
  int ld_dh;             // layout of dh (may be the interleaved [row][net][u0p] form)
  long long net_stride_dh;
  const float* Wmu; const float* bmu; const float* Wv; const float* bv; const float* logstd;
  const float* actions; const float* neglogpacs;                  // rollout (time-major)
  const float* adv; const float* values_n; const float* returns_n;  // prepared
  float* mus_w; float* sigmas_w;
  const int64_t* perm;
  long long start;
  int mb, N, T, act, rows_per_wave;
  float e_clip, critic_coef, entropy_coef, bounds_coef;
  double* loss_part;  // [blocks][8]
  float* head_slab;   // [blocks][head_count]
  int head_count;
};

constexpr float LOG_SQRT_2PI_F = 0.918938533204672741780329736406f;

template <int MAXJ>
__global__ __launch_bounds__(LOSS_THREADS) void k_loss(const LossArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H, act = a.act;
  float wmu[IGI_MAX_ACT][MAXJ], wv[MAXJ];
  float gmu[IGI_MAX_ACT][MAXJ], gv[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int k = lane + 64 * j;
    wv[j] = (k < H) ? a.Wv[k] : 0.f;
    gv[j] = 0.f;
#pragma unroll
    for (int q = 0; q < IGI_MAX_ACT; ++q) {
      wmu[q][j] = (q < act && k < H) ? a.Wmu[q * H + k] : 0.f;
      gmu[q][j] = 0.f;
    }
  }
  // lane q (< act) owns action dimension q for the per-action arithmetic
  const bool alane = lane < act;
  const float my_logstd = alane ? a.logstd[lane] : 0.f;
  const float my_sig = expf(my_logstd);
  const float my_logsc = logf(my_sig);  // Normal.log_prob uses scale.log() (torch/distributions/normal.py)
  const float my_var = my_sig * my_sig;
  const float my_bmu = alane ? a.bmu[lane] : 0.f;
  float gbmu = 0.f, gsig = 0.f;         // lane q accumulates d(bias_mu[q]), d(sigma[q])
  const float bv = a.bv[0];
  float gbv = 0.f;
  double s_a = 0, s_c = 0, s_b = 0, s_e = 0, s_kl = 0;
  const float inv_mb = 1.0f / (float)a.mb;
  const float lo = 1.0f - a.e_clip, hi = 1.0f + a.e_clip;

  // Rows are processed four at a time: lane r fetches the permutation entry of row r, then lane q
  // fetches the q-th per-sample scalar of each row (actions, old mu, old sigma, advantage, return,
  // old value, old neglogp) and the hidden rows are loaded, all before any arithmetic, so a group of
  // four rows costs two dependent memory latencies instead of eight.  Scalars reach the (redundant,
  // wave-uniform) loss arithmetic through v_readlane.
  const int gw = blockIdx.x * (LOSS_THREADS / 64) + wave;
  const int row_begin = gw * a.rows_per_wave;
  auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
  for (int base = 0; base < a.rows_per_wave; base += 4) {
    const int row0 = row_begin + base;
    int nrows = a.rows_per_wave - base;
    if (nrows > 4) nrows = 4;
    if (nrows > a.mb - row0) nrows = a.mb - row0;
    if (nrows <= 0) break;  // wave-uniform
    int my_i = 0;
    if (lane < nrows) {
      const long long b = a.perm[a.start + row0 + lane];
      const int n = (int)(b / a.T);
      my_i = (int)(b - (long long)n * a.T) * a.N + n;  // b = n*T + t  ->  t*N + n
    }
    int irow[4];
    float d[4];
    float ha[4][MAXJ], hc[4][MAXJ];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      irow[r] = __builtin_amdgcn_readlane(my_i, r);
      const long long i = irow[r];
      // branch-free address select: ONE predicated load per row (a chain of divergent
      // `if (lane ...) load` arms would serialise seven dependent memory round trips)
      const float* src = a.actions + i * act + lane;
      src = (lane >= act) ? a.mus_w + i * act + (lane - act) : src;
      src = (lane >= 2 * act) ? a.sigmas_w + i * act + (lane - 2 * act) : src;
      src = (lane == 3 * act) ? a.adv + i : src;
      src = (lane == 3 * act + 1) ? a.returns_n + i : src;
      src = (lane == 3 * act + 2) ? a.values_n + i : src;
      src = (lane == 3 * act + 3) ? a.neglogpacs + i : src;
      d[r] = (r < nrows && lane < 3 * act + 4) ? *src : 0.f;
      const float* ha_p = a.h + (long long)(row0 + r) * a.ldh;
      const float* hc_p = ha_p + a.net_stride;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int k = lane + 64 * j;
        ha[r][j] = (r < nrows && k < H) ? ha_p[k] : 0.f;
        hc[r][j] = (r < nrows && k < H) ? hc_p[k] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r >= nrows) break;  // wave-uniform
      const int row = row0 + r;
      const long long i = irow[r];
      float pm[IGI_MAX_ACT], pv = 0.f;
#pragma unroll
      for (int q = 0; q < IGI_MAX_ACT; ++q) pm[q] = 0.f;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        pv = fmaf(hc[r][j], wv[j], pv);   // explicit fma: -ffp-contract=off would issue mul + add
#pragma unroll
        for (int q = 0; q < IGI_MAX_ACT; ++q) pm[q] = fmaf(ha[r][j], wmu[q][j], pm[q]);
      }
      // lane q < act ends up with its own mu[q], lane 7 with the value (IGI_MAX_ACT == 8; act <= 7 here)
      float red8[8];
#pragma unroll
      for (int q = 0; q < 7; ++q) red8[q] = pm[q];
      red8[7] = pv;
      float my_pm;
      if (act <= 7) {
        const float z = wave_sum8(red8, lane);
        my_pm = z;
        pv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), 7));
      } else {
        pv = wave_sum(pv);
        my_pm = 0.f;
#pragma unroll
        for (int q = 0; q < IGI_MAX_ACT; ++q) {
          const float t = wave_sum(pm[q]);
          my_pm = (lane == q) ? t : my_pm;
        }
      }

      const float v = pv + bv;
      const float adv = rl(d[r], 3 * act), R = rl(d[r], 3 * act + 1), vp = rl(d[r], 3 * act + 2),
                  old_nlp = rl(d[r], 3 * act + 3);
      // per-action terms on lane q: select this lane's mu from the (wave-uniform) reductions and pull
      // the old mu / sigma of action q over from lanes act+q / 2*act+q
      const float my_mu = my_pm + my_bmu;
      const float ac = d[r];
      const float omu = __shfl(d[r], lane + act, 64), osig = __shfl(d[r], lane + 2 * act, 64);
      const float x = ac - my_mu;
      const float bh = fminf(my_mu - 1.1f, 0.f), blo = fminf(-my_mu + 1.1f, 0.f);
      const float dm = omu - my_mu;
      float t_nlp = (x * x) / (2.0f * my_var) + my_logsc + LOG_SQRT_2PI_F;
      float t_ent = 0.5f + LOG_SQRT_2PI_F + my_logsc;
      float t_bl = blo * blo + bh * bh;
      // policy_kl(new, old) frozen_ppo.py:854-860
      float t_kl = (logf(osig / my_sig + 1e-5f) + (my_var + dm * dm) / (2.0f * (osig * osig + 1e-5f))) - 0.5f;
      if (!alane) { t_nlp = 0.f; t_ent = 0.f; t_bl = 0.f; t_kl = 0.f; }
      float nlp, ent, bl, kl;
      if (act <= 8) { nlp = row0_sum(t_nlp); ent = row0_sum(t_ent); bl = row0_sum(t_bl); kl = row0_sum(t_kl); }
      else { nlp = wave_sum(t_nlp); ent = wave_sum(t_ent); bl = wave_sum(t_bl); kl = wave_sum(t_kl); }
      // actor loss (frozen_ppo.py:544-547)
      const float ratio = expf(old_nlp - nlp);
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float s1 = -(adv * ratio), s2 = -(adv * rc);
      const float a_loss = fmaxf(s1, s2);
      const float d1 = adv * ratio;  // d s1 / d nlp
      const float d2 = (ratio >= lo && ratio <= hi) ? d1 : 0.f;
      const float da = (s1 > s2) ? d1 : ((s1 < s2) ? d2 : 0.5f * (d1 + d2));
      const float g_nlp = da * inv_mb;
      // critic loss (frozen_ppo.py:549-552)
      const float dvp = v - vp;
      const float vclip = vp + fminf(fmaxf(dvp, -a.e_clip), a.e_clip);
      const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
      const float c_loss = fmaxf(l1, l2);
      const float g1 = 2.0f * (v - R);
      const float g2 = (dvp >= -a.e_clip && dvp <= a.e_clip) ? 2.0f * (vclip - R) : 0.f;
      const float dc = (l1 > l2) ? g1 : ((l1 < l2) ? g2 : 0.5f * (g1 + g2));
      const float dv = dc * (0.5f * a.critic_coef * inv_mb);

      float my_dmu = g_nlp * (-(x / my_var)) + (a.bounds_coef * inv_mb) * (2.0f * bh - 2.0f * blo);
      if (!alane) my_dmu = 0.f;
      if (alane) {
        gsig += g_nlp * (1.0f - (x * x) / my_var) - a.entropy_coef * inv_mb;
        gbmu += my_dmu;
      }
      float dmu[IGI_MAX_ACT];
#pragma unroll
      for (int q = 0; q < IGI_MAX_ACT; ++q) dmu[q] = rl(my_dmu, q);   // back to wave-uniform for the row products
      gbv += dv;
      s_a += a_loss; s_c += c_loss; s_b += bl; s_e += ent; s_kl += kl;

      // d(hidden pre-activation) rows + head weight gradients
      float* dha_p = a.dh + (long long)row * a.ld_dh;
      float* dhc_p = dha_p + a.net_stride_dh;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int k = lane + 64 * j;
        float da3 = 0.f;
#pragma unroll
        for (int q = 0; q < IGI_MAX_ACT; ++q) {
          da3 = fmaf(dmu[q], wmu[q][j], da3);
          gmu[q][j] = fmaf(dmu[q], ha[r][j], gmu[q][j]);
        }
        gv[j] = fmaf(dv, hc[r][j], gv[j]);
        if (k < H) {
          dha_p[k] = da3 * (1.0f - ha[r][j] * ha[r][j]);
          dhc_p[k] = (dv * wv[j]) * (1.0f - hc[r][j] * hc[r][j]);
        }
      }
      // update_mu_sigma (experience.py:228-233): scatter the new mu / sigma
      if (alane) {
        a.mus_w[i * act + lane] = my_mu;
        a.sigmas_w[i * act + lane] = my_sig;
      }
    }
  }

  // ---- block partials: [muW (act*H) | muB (act) | valW (H) | valB (1) | sigma (act)]
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][head_count]
  float* mine = red + wave * a.head_count;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int k = lane + 64 * j;
    if (k < H) {
#pragma unroll
      for (int q = 0; q < IGI_MAX_ACT; ++q)
        if (q < act) mine[q * H + k] = gmu[q][j];
      mine[act * H + act + k] = gv[j];
    }
  }
  if (alane) {
    mine[act * H + lane] = gbmu;
    mine[act * H + act + H + 1 + lane] = gsig;
  }
  if (lane == 0) mine[act * H + act + H] = gbv;
  __shared__ double sred[LOSS_THREADS / 64][5];
  if (lane == 0) {
    sred[wave][0] = s_a; sred[wave][1] = s_c; sred[wave][2] = s_b; sred[wave][3] = s_e; sred[wave][4] = s_kl;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < a.head_count; e += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LOSS_THREADS / 64; ++w) s += red[w * a.head_count + e];
    a.head_slab[(long long)blockIdx.x * a.head_count + e] = s;
  }
  if (threadIdx.x < 5) {
    double s = 0;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) s += sred[w][threadIdx.x];
    a.loss_part[blockIdx.x * 8 + threadIdx.x] = s;
  }
}

// k_loss with the per-sample scalar arithmetic done ONCE for a group of four rows: 16-lane row rr of the wave holds
// row rr's actions / old mu / old sigma / advantage ..., its head sums land there straight out of wave_sum8, and the
// row-local DPP sums give every row its neglogp / entropy / bounds / KL at once (the divisions, logs and exp of that
// section were ~55 % of the per-row instruction count).  act <= 7.
template <int MAXJ>
__global__ __launch_bounds__(LOSS_THREADS) void k_loss_packed(const LossArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rr = lane >> 4, qi = lane & 15;   // scalar section: 16-lane row rr works on row rr of a group of four
  const int H = a.H, act = a.act;
  float wmu[IGI_MAX_ACT][MAXJ], wv[MAXJ];
  float gmu[IGI_MAX_ACT][MAXJ], gv[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int k = lane + 64 * j;
    wv[j] = (k < H) ? a.Wv[k] : 0.f;
    gv[j] = 0.f;
#pragma unroll
    for (int q = 0; q < IGI_MAX_ACT; ++q) {
      wmu[q][j] = (q < act && k < H) ? a.Wmu[q * H + k] : 0.f;
      gmu[q][j] = 0.f;
    }
  }
  // lane qi (< act) of each 16-lane row owns action dimension qi for the per-action arithmetic
  const bool alane = qi < act;
  const float my_logstd = alane ? a.logstd[qi] : 0.f;
  const float my_sig = expf(my_logstd);
  const float my_logsc = logf(my_sig);  // Normal.log_prob uses scale.log() (torch/distributions/normal.py)
  const float my_var = my_sig * my_sig;
  const float my_bmu = alane ? a.bmu[qi] : 0.f;
  float gbmu = 0.f, gsig = 0.f;         // lane q accumulates d(bias_mu[q]), d(sigma[q])
  const float bv = a.bv[0];
  float gbv = 0.f;
  double s_a = 0, s_c = 0, s_b = 0, s_e = 0, s_kl = 0;
  const float inv_mb = 1.0f / (float)a.mb;
  const float lo = 1.0f - a.e_clip, hi = 1.0f + a.e_clip;

  // Rows are processed four at a time: lane r fetches the permutation entry of row r, then lane q
  // fetches the q-th per-sample scalar of each row (actions, old mu, old sigma, advantage, return,
  // old value, old neglogp) and the hidden rows are loaded, all before any arithmetic, so a group of
  // four rows costs two dependent memory latencies instead of eight.  Scalars reach the (redundant,
  // wave-uniform) loss arithmetic through v_readlane.
  const int gw = blockIdx.x * (LOSS_THREADS / 64) + wave;
  const int row_begin = gw * a.rows_per_wave;
  auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
  // One group of four rows: everything it reads.  The group AFTER the one being worked on is requested first (two
  // register sets, the loop below is unrolled by two), so its two dependent round trips (permutation entry -> per-sample
  // scalars; the hidden rows do not depend on it) run under the arithmetic of the current group; only the first group
  // of a wave waits for memory.  (The mu / sigma rows written below belong to other samples than any row read later:
  // the permutation visits each sample once per pass.)
  struct LossRows {
    long long ip;
    bool okrow, aok;
    int nrows, row0;
    float ac, omu, osig, adv, R, vp, old_nlp;
    float ha[4][MAXJ], hc[4][MAXJ];
  };
  auto load_group = [&](int base, LossRows& g) {
    const int row0 = row_begin + base;
    int nrows = a.rows_per_wave - base;
    if (nrows > 4) nrows = 4;
    if (nrows > a.mb - row0) nrows = a.mb - row0;
    g.nrows = nrows; g.row0 = row0;
    if (nrows <= 0) return;  // wave-uniform
    int my_i = 0;
    if (lane < nrows) {
      const long long b = a.perm[a.start + row0 + lane];
      const int n = (int)(b / a.T);
      my_i = (int)(b - (long long)n * a.T) * a.N + n;  // b = n*T + t  ->  t*N + n
    }
    // the hidden rows first: their addresses do not wait for the permutation entry
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* ha_p = a.h + (long long)(row0 + r) * a.ldh;
      const float* hc_p = ha_p + a.net_stride;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int k = lane + 64 * j;
        g.ha[r][j] = (r < nrows && k < H) ? ha_p[k] : 0.f;
        g.hc[r][j] = (r < nrows && k < H) ? hc_p[k] : 0.f;
      }
    }
    // per-sample scalars in the packed layout: row rr of the group lives in 16-lane row rr
    g.okrow = rr < nrows;
    const long long ip = __shfl(my_i, rr, 64);
    g.ip = ip;
    g.aok = g.okrow && alane;
    g.ac = g.aok ? a.actions[ip * act + qi] : 0.f;
    g.omu = g.aok ? a.mus_w[ip * act + qi] : 0.f;
    g.osig = g.aok ? a.sigmas_w[ip * act + qi] : 0.f;
    g.adv = g.okrow ? a.adv[ip] : 0.f;
    g.R = g.okrow ? a.returns_n[ip] : 0.f;
    g.vp = g.okrow ? a.values_n[ip] : 0.f;
    g.old_nlp = g.okrow ? a.neglogpacs[ip] : 0.f;
  };
  auto compute_group = [&](const LossRows& g) {
    const int row0 = g.row0, nrows = g.nrows;
    const bool okrow = g.okrow, aok = g.aok;
    const long long ip = g.ip;
    const float ac = g.ac, omu = g.omu, osig = g.osig, adv = g.adv, R = g.R, vp = g.vp, old_nlp = g.old_nlp;
    const float (&ha)[4][MAXJ] = g.ha;
    const float (&hc)[4][MAXJ] = g.hc;
    // head dot products: after wave_sum8 EVERY lane l holds the total of value l & 7 (mu_0..mu_6, value), so row r's
    // totals are already in place for 16-lane row r -- keep them there
    float p_z = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pm[IGI_MAX_ACT], pv = 0.f;
#pragma unroll
      for (int q = 0; q < IGI_MAX_ACT; ++q) pm[q] = 0.f;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        pv = fmaf(hc[r][j], wv[j], pv);   // explicit fma: -ffp-contract=off would issue mul + add
#pragma unroll
        for (int q = 0; q < IGI_MAX_ACT; ++q) pm[q] = fmaf(ha[r][j], wmu[q][j], pm[q]);
      }
      float red8[8];
#pragma unroll
      for (int q = 0; q < 7; ++q) red8[q] = pm[q];
      red8[7] = pv;
      const float z = wave_sum8(red8, lane);
      p_z = (rr == r) ? z : p_z;
    }
    // ---- scalar section, once for the four rows (lanes qi >= 8 of a row mirror lanes qi - 8: harmless)
    float my_dmu, dv;
    {
      const float pv = __shfl(p_z, (lane & 48) | 7, 64);
      const float v = pv + bv;
      const float my_mu = p_z + my_bmu;
      const float x = ac - my_mu;
      const float bh = fminf(my_mu - 1.1f, 0.f), blo = fminf(-my_mu + 1.1f, 0.f);
      const float dm = omu - my_mu;
      float t_nlp = (x * x) / (2.0f * my_var) + my_logsc + LOG_SQRT_2PI_F;
      float t_ent = 0.5f + LOG_SQRT_2PI_F + my_logsc;
      float t_bl = blo * blo + bh * bh;
      // policy_kl(new, old) frozen_ppo.py:854-860
      float t_kl = (logf(osig / my_sig + 1e-5f) + (my_var + dm * dm) / (2.0f * (osig * osig + 1e-5f))) - 0.5f;
      if (!aok) { t_nlp = 0.f; t_ent = 0.f; t_bl = 0.f; t_kl = 0.f; }
      const float nlp = rows_sum_ror(t_nlp), ent = rows_sum_ror(t_ent), bl = rows_sum_ror(t_bl), kl = rows_sum_ror(t_kl);
      // actor loss (frozen_ppo.py:544-547)
      const float ratio = expf(old_nlp - nlp);
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float s1 = -(adv * ratio), s2 = -(adv * rc);
      const float a_loss = fmaxf(s1, s2);
      const float d1 = adv * ratio;  // d s1 / d nlp
      const float d2 = (ratio >= lo && ratio <= hi) ? d1 : 0.f;
      const float da = (s1 > s2) ? d1 : ((s1 < s2) ? d2 : 0.5f * (d1 + d2));
      const float g_nlp = da * inv_mb;
      // critic loss (frozen_ppo.py:549-552)
      const float dvp = v - vp;
      const float vclip = vp + fminf(fmaxf(dvp, -a.e_clip), a.e_clip);
      const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
      const float c_loss = fmaxf(l1, l2);
      const float g1 = 2.0f * (v - R);
      const float g2 = (dvp >= -a.e_clip && dvp <= a.e_clip) ? 2.0f * (vclip - R) : 0.f;
      const float dc = (l1 > l2) ? g1 : ((l1 < l2) ? g2 : 0.5f * (g1 + g2));
      dv = okrow ? dc * (0.5f * a.critic_coef * inv_mb) : 0.f;
      my_dmu = g_nlp * (-(x / my_var)) + (a.bounds_coef * inv_mb) * (2.0f * bh - 2.0f * blo);
      if (!aok) my_dmu = 0.f;
      if (aok) {
        gsig += g_nlp * (1.0f - (x * x) / my_var) - a.entropy_coef * inv_mb;
        gbmu += my_dmu;
        // update_mu_sigma (experience.py:228-233): scatter the new mu / sigma
        a.mus_w[ip * act + qi] = my_mu;
        a.sigmas_w[ip * act + qi] = my_sig;
      }
      if (okrow && qi == 0) { s_a += a_loss; s_c += c_loss; s_b += bl; s_e += ent; s_kl += kl; gbv += dv; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r >= nrows) break;  // wave-uniform
      const int row = row0 + r;
      float dmu[IGI_MAX_ACT];
#pragma unroll
      for (int q = 0; q < IGI_MAX_ACT; ++q) dmu[q] = rl(my_dmu, 16 * r + q);   // wave-uniform for the row products
      const float dvr = rl(dv, 16 * r);
      // d(hidden pre-activation) rows + head weight gradients
      float* dha_p = a.dh + (long long)row * a.ld_dh;
      float* dhc_p = dha_p + a.net_stride_dh;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int k = lane + 64 * j;
        float da3 = 0.f;
#pragma unroll
        for (int q = 0; q < IGI_MAX_ACT; ++q) {
          da3 = fmaf(dmu[q], wmu[q][j], da3);
          gmu[q][j] = fmaf(dmu[q], ha[r][j], gmu[q][j]);
        }
        gv[j] = fmaf(dvr, hc[r][j], gv[j]);
        if (k < H) {
          dha_p[k] = da3 * (1.0f - ha[r][j] * ha[r][j]);
          dhc_p[k] = (dvr * wv[j]) * (1.0f - hc[r][j] * hc[r][j]);
        }
      }
    }
  };
  {
    LossRows gA, gB;
    load_group(0, gA);
    for (int base = 0; base < a.rows_per_wave; base += 8) {
      if (gA.nrows <= 0) break;
      load_group(base + 4, gB);
      compute_group(gA);
      if (gB.nrows <= 0) break;
      load_group(base + 8, gA);
      compute_group(gB);
    }
  }
  // fold the four 16-lane rows' accumulators (lanes l, l ^ 16, l ^ 32, l ^ 48)
  gbmu += __shfl_xor(gbmu, 16, 64); gbmu += __shfl_xor(gbmu, 32, 64);
  gsig += __shfl_xor(gsig, 16, 64); gsig += __shfl_xor(gsig, 32, 64);
  gbv += __shfl_xor(gbv, 16, 64); gbv += __shfl_xor(gbv, 32, 64);
  s_a += __shfl_xor(s_a, 16, 64); s_a += __shfl_xor(s_a, 32, 64);
  s_c += __shfl_xor(s_c, 16, 64); s_c += __shfl_xor(s_c, 32, 64);
  s_b += __shfl_xor(s_b, 16, 64); s_b += __shfl_xor(s_b, 32, 64);
  s_e += __shfl_xor(s_e, 16, 64); s_e += __shfl_xor(s_e, 32, 64);
  s_kl += __shfl_xor(s_kl, 16, 64); s_kl += __shfl_xor(s_kl, 32, 64);

  // ---- block partials: [muW (act*H) | muB (act) | valW (H) | valB (1) | sigma (act)]
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][head_count]
  float* mine = red + wave * a.head_count;
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int k = lane + 64 * j;
    if (k < H) {
#pragma unroll
      for (int q = 0; q < IGI_MAX_ACT; ++q)
        if (q < act) mine[q * H + k] = gmu[q][j];
      mine[act * H + act + k] = gv[j];
    }
  }
  if (lane < act) {
    mine[act * H + lane] = gbmu;
    mine[act * H + act + H + 1 + lane] = gsig;
  }
  if (lane == 0) mine[act * H + act + H] = gbv;
  __shared__ double sred[LOSS_THREADS / 64][5];
  if (lane == 0) {
    sred[wave][0] = s_a; sred[wave][1] = s_c; sred[wave][2] = s_b; sred[wave][3] = s_e; sred[wave][4] = s_kl;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < a.head_count; e += blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LOSS_THREADS / 64; ++w) s += red[w * a.head_count + e];
    a.head_slab[(long long)blockIdx.x * a.head_count + e] = s;
  }
  if (threadIdx.x < 5) {
    double s = 0;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) s += sred[w][threadIdx.x];
    a.loss_part[blockIdx.x * 8 + threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// 3b. The same loss stage fused behind the LAST trunk layer's forward (models_split.py:222-250 right behind the last
//     Linear + Tanh of :27-38; frozen_ppo.py:543-570, 618).  k_loss re-reads the 2 x mb x 128 hidden rows that layer has
//     just written (16.8 MB out, 16.8 MB in again, 16 MB of d(hidden) out) and spends one WAVE per row; here the
//     64 x 128 output tile of one net (rows m0..m0+63, actor or critic; 2 x mb / 64 workgroups, two or three per CU,
//     an actor tile paired with a critic tile) never leaves the CU:
//       A  accumulators -> LDS (the waves' 32 x 32 slices), bias + tanh in place; the head weights -> registers
//       B  head products on the matrix pipe (v_mfma_f32_16x16x4_f32): 16 rows x (<= 7 mu | value) per wave pair
//       C  the results leave the pipe as (row, action) per lane: Normal log-prob / entropy / KL / bounds with the sums over
//          the actions as 16-lane DPP sums (k_loss_packed's layout), clipped surrogate or clipped value loss with the
//          per-row scalars requested under the last k-tile, d(loss)/d(mu) | d(loss)/d(value) -> LDS, update_mu_sigma
//          write-back, bias / sigma gradient and fp64 loss sums per wave
//       D  d(pre-activation) of the layer = (d(head) . W_head) * (1 - h^2): the only large thing written to HBM (16-byte
//          row segments); head weight gradients of the tile on the matrix pipe (A = d(head)^T, B = the tanh'd tile in LDS),
//          stored straight into the tile's partial record
//       E  the waves' bias / sigma / loss partials in wave order -> the same record (mb / 64 records per minibatch)
//     The hidden layer itself is not stored (nothing reads it: the data gradient below needs tanh' of the layer BELOW).
//     Same formulas, expression by expression, as k_loss; the head sums run in the MFMA's k order.  H == 128, act <= 7;
//     other shapes keep the two launches (IGI_LOSS_FUSED=0 forces them).  Same box, A/B: 21.6 + 15.4 -> 30.6 us per step.
// ---------------------------------------------------------------------------------------------
struct TrunkLossHook {
  const LossArgs& a;
  // Per-sample scalars, requested in two steps (permutation entries up front, the rows' values under the last k-tile).  Thread (wave w, q = lane & 15, fq = lane >> 4)
  // owns action q of rows m0 + 16 (w & 3) + 4 fq + 2 (w >> 2) + r, r < 2 -- the layout in which the head products leave
  // the matrix pipe (waves w and w + 4 both compute the 16 x 16 block of rows 16 (w & 3) .. +15 and halve its rows).
  unsigned pb[2];                                   // permutation entries (step 0), then arena rows t*N + n
  float ac[2], omu[2], osig[2], s0[2], s1[2];       // (s0, s1) = (advantage, old neglogp) | (return, old value)

  static constexpr int TILE_M = 64;                 // rows per tile: 2 x mb / 64 workgroups, two (or three) per CU
  static constexpr int EPLD = 36, SLICE = 32 * EPLD;
  static constexpr int O_DM = 8 * SLICE, O_RED = O_DM + TILE_M * 8, O_LSUM = O_RED + 8 * 16, LDS_FLOATS = O_LSUM + 8 * 8;

  __device__ __forceinline__ explicit TrunkLossHook(const LossArgs& a_) : a(a_) {}

  // step 0, in front of the first tile's DMA requests: the permutation entries
  __device__ __forceinline__ void prefetch(const GemmArgs& g, int m0, int batch, int tid) {
    const int w = tid >> 6;
    const int row0 = m0 + 16 * (w & 3) + 4 * ((tid & 63) >> 4) + 2 * (w >> 2);
#pragma unroll
    for (int r = 0; r < 2; ++r) pb[r] = (unsigned)a.perm[a.start + min(row0 + r, g.M - 1)];   // < 2^31 (launcher)
  }
  // which k-tile carries step 1: the LAST one.  Requests return in order, so gathers issued in front of a tile's DMA
  // hold that tile's vmcnt wait until they have landed (issued with the first tile: +3 us per launch, with the last: +1.5;
  // measured with early exits from the kernel) -- behind the last DMA they fly under the last MFMAs and phases A / B.
  __device__ __forceinline__ int prefetch1_at(int nk) const { return nk - 1; }
  // step 1, behind the barrier of the k-tile prefetch1_at() names (the entries landed long ago)
  __device__ __forceinline__ void prefetch1(const GemmArgs& g, int batch, int tid) {
    const int q = tid & 15, act = a.act;
    // the net is wave-uniform: its two per-row arrays are picked on the scalar unit (written as an if / else over the four
    // loads the compiler built a table of the four pointers in SCRATCH and indexed it: a memory round trip in front of
    // the gathers)
    const bool actor = batch == 0;
    const float* p0 = uniform_ptr(actor ? a.adv : a.returns_n);
    const float* p1 = uniform_ptr(actor ? a.neglogpacs : a.values_n);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const unsigned n = pb[r] / (unsigned)a.T;
      pb[r] = (pb[r] - n * (unsigned)a.T) * (unsigned)a.N + n;   // b = n*T + t  ->  t*N + n
      const long long i = pb[r];
      ac[r] = 0.f; omu[r] = 0.f; osig[r] = 1.f;
      if (actor && q < act) {
        ac[r] = a.actions[i * act + q];
        omu[r] = a.mus_w[i * act + q];
        osig[r] = a.sigmas_w[i * act + q];
      }
      s0[r] = p0[i];
      s1[r] = p1[i];
    }
  }

  __device__ __forceinline__ void epilogue(f32x16 (&acc)[1][1], float* smem, const GemmArgs& g, int m0, int mt, int batch,
                                           int tid, int wave, int lane, int wm, int wn) {
    typedef float f32x4r __attribute__((ext_vector_type(4)));
    const bool actor = batch == 0;
    const int act = a.act;
    const int nq = actor ? act : 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int fm = lane & 15, fq = lane >> 4;   // MFMA 16x16x4 operand / result coordinates
    const int c4 = lane & 7, rl = lane >> 3;    // row-major passes over a 64 x 32 slice: 16-byte column group, row
    float* dm = smem + O_DM;                    // [64 rows][8]: d(loss)/d(head output)
    float* red = smem + O_RED;                  // [8 waves][16]: bias / sigma gradient partials
    double* lsum = reinterpret_cast<double*>(smem + O_LSUM);   // [8 waves][4]
    const float* W = actor ? a.Wmu : a.Wv;      // [nq][128]

    // ---- A: accumulators -> this wave's slice, bias + tanh in place; meanwhile the head weights arrive in registers
    __syncthreads();   // every wave is done reading the ring
    float* ep = smem + wave * SLICE;
#pragma unroll
    for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * h) * EPLD + l31] = acc[0][0][r];
    // B operand of the head product: lane (n = fm = head output, fq) feeds W[n][16 kh + 4 fq + t] to step (kh, t)
    float4 wv[8];
#pragma unroll
    for (int kh = 0; kh < 8; ++kh)
      wv[kh] = (fm < nq) ? *reinterpret_cast<const float4*>(W + fm * 128 + 16 * kh + 4 * fq) : make_float4(0.f, 0.f, 0.f, 0.f);
    // per-action constants of this lane's action
    const bool alane = actor && fm < act;
    const float my_logstd = alane ? a.logstd[fm] : 0.f;
    const float my_bmu = alane ? a.bmu[fm] : 0.f;
    const float bvv = a.bv[0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const float4 b = *reinterpret_cast<const float4*>(g.bias + batch * g.sBias + wn * 32 + 4 * c4);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float4* p = reinterpret_cast<float4*>(ep + (it * 8 + rl) * EPLD + 4 * c4);
        float4 v = *p;
        v.x = fast_tanh(v.x + b.x); v.y = fast_tanh(v.y + b.y); v.z = fast_tanh(v.z + b.z); v.w = fast_tanh(v.w + b.w);
        *p = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- B: head products of rows 16 (w & 3) .. +15 on the matrix pipe (v_mfma_f32_16x16x4_f32, k = 128):
    //         A[m = fm][4 fq + t] = h[row 16 (w & 3) + fm][16 kh + 4 fq + t] (one 16-byte LDS read per four instructions)
    f32x4r hacc = f32x4r{0.f, 0.f, 0.f, 0.f};
    {
      const int row = 16 * (wave & 3) + fm;
      const float* hrow = smem + (row >> 5) * 4 * SLICE + (row & 31) * EPLD + 4 * fq;
#pragma unroll
      for (int kh = 0; kh < 8; ++kh) {
        const float4 x = *reinterpret_cast<const float4*>(hrow + (kh >> 1) * SLICE + 16 * (kh & 1));
        hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, wv[kh].x, hacc, 0, 0, 0);
        hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, wv[kh].y, hacc, 0, 0, 0);
        hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, wv[kh].z, hacc, 0, 0, 0);
        hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, wv[kh].w, hacc, 0, 0, 0);
      }
    }
    // ---- C: hacc[2 (w >> 2) + r] = head output fm of row 16 (w & 3) + 4 fq + 2 (w >> 2) + r: this lane's action of its
    //         two rows.  The sums over the actions of a row are sums over the 16-lane row (DPP), as in k_loss_packed.
    float gb = 0.f, gs = 0.f;            // d(bias_mu[fm]) | d(bias_v), d(sigma[fm]) over this lane's rows
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    {
      const float inv_mb = 1.0f / (float)a.mb;
      const float my_sig = expf(my_logstd);
      const float my_logsc = logf(my_sig);  // Normal.log_prob uses scale.log() (torch/distributions/normal.py)
      const float my_var = my_sig * my_sig;
      const float lo = 1.0f - a.e_clip, hi = 1.0f + a.e_clip;
      const int hi2 = wave >> 2;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int rowt = 16 * (wave & 3) + 4 * fq + 2 * hi2 + r;           // row of the tile
        const bool okrow = m0 + rowt < g.M;
        const float hout = hi2 ? hacc[2 + r] : hacc[r];
        float my_d = 0.f;
        if (actor) {
          const bool aok = okrow && alane;
          const float my_mu = hout + my_bmu;
          const float x = ac[r] - my_mu;
          const float bh = fminf(my_mu - 1.1f, 0.f), blo = fminf(-my_mu + 1.1f, 0.f);
          const float dmo = omu[r] - my_mu;
          float t_nlp = (x * x) / (2.0f * my_var) + my_logsc + LOG_SQRT_2PI_F;
          float t_ent = 0.5f + LOG_SQRT_2PI_F + my_logsc;
          float t_bl = blo * blo + bh * bh;
          // policy_kl(new, old) frozen_ppo.py:854-860
          float t_kl = (logf(osig[r] / my_sig + 1e-5f) + (my_var + dmo * dmo) / (2.0f * (osig[r] * osig[r] + 1e-5f))) - 0.5f;
          if (!aok) { t_nlp = 0.f; t_ent = 0.f; t_bl = 0.f; t_kl = 0.f; }
          const float nlp = rows_sum_ror(t_nlp), ent = rows_sum_ror(t_ent), bl = rows_sum_ror(t_bl), kl = rows_sum_ror(t_kl);
          // actor loss (frozen_ppo.py:544-547)
          const float adv = s0[r], old_nlp = s1[r];
          const float ratio = expf(old_nlp - nlp);
          const float rc = fminf(fmaxf(ratio, lo), hi);
          const float sa1 = -(adv * ratio), sa2 = -(adv * rc);
          const float a_loss = fmaxf(sa1, sa2);
          const float d1 = adv * ratio;  // d s1 / d nlp
          const float d2 = (ratio >= lo && ratio <= hi) ? d1 : 0.f;
          const float da = (sa1 > sa2) ? d1 : ((sa1 < sa2) ? d2 : 0.5f * (d1 + d2));
          const float g_nlp = da * inv_mb;
          if (aok) {
            my_d = g_nlp * (-(x / my_var)) + (a.bounds_coef * inv_mb) * (2.0f * bh - 2.0f * blo);
            gs += g_nlp * (1.0f - (x * x) / my_var) - a.entropy_coef * inv_mb;
            gb += my_d;
            // update_mu_sigma (experience.py:228-233): scatter the new mu / sigma
            a.mus_w[(long long)pb[r] * act + fm] = my_mu;
            a.sigmas_w[(long long)pb[r] * act + fm] = my_sig;
          }
          if (okrow && fm == 0) { t0 += a_loss; t1 += bl; t2 += ent; t3 += kl; }
        } else {
          // critic loss (frozen_ppo.py:549-552)
          const float v = hout + bvv;
          const float R = s0[r], vp = s1[r];
          const float dvp = v - vp;
          const float vclip = vp + fminf(fmaxf(dvp, -a.e_clip), a.e_clip);
          const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
          const float c_loss = fmaxf(l1, l2);
          const float g1 = 2.0f * (v - R);
          const float g2 = (dvp >= -a.e_clip && dvp <= a.e_clip) ? 2.0f * (vclip - R) : 0.f;
          const float dc = (l1 > l2) ? g1 : ((l1 < l2) ? g2 : 0.5f * (g1 + g2));
          if (okrow && fm == 0) {
            my_d = dc * (0.5f * a.critic_coef * inv_mb);
            gb += my_d;
            t0 += c_loss;
          }
        }
        if (fm < 8) dm[rowt * 8 + fm] = my_d;
      }
      // this wave's 8 rows: lanes fm, fm + 16, fm + 32, fm + 48
      gb += __shfl_xor(gb, 16, 64); gb += __shfl_xor(gb, 32, 64);
      gs += __shfl_xor(gs, 16, 64); gs += __shfl_xor(gs, 32, 64);
      t0 += __shfl_xor(t0, 16, 64); t0 += __shfl_xor(t0, 32, 64);
      if (actor) {
        t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
        t2 += __shfl_xor(t2, 16, 64); t2 += __shfl_xor(t2, 32, 64);
        t3 += __shfl_xor(t3, 16, 64); t3 += __shfl_xor(t3, 32, 64);
      }
      if (lane < 8) { red[wave * 16 + lane] = gb; red[wave * 16 + 8 + lane] = gs; }
      if (lane == 0) { lsum[wave * 4 + 0] = t0; lsum[wave * 4 + 1] = t1; lsum[wave * 4 + 2] = t2; lsum[wave * 4 + 3] = t3; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // LDS-only rendezvous: the mu / sigma stores stay in flight
    asm volatile("" ::: "memory");

    // ---- D1: d(pre-activation) of this wave's 32 x 32 slice = (d(head) . W_head) * (1 - h^2), 16-byte row segments
    {
      float4 w[7];
#pragma unroll
      for (int q = 0; q < 7; ++q)
        w[q] = (q < nq) ? *reinterpret_cast<const float4*>(W + q * 128 + wn * 32 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      float* dst = a.dh + batch * a.net_stride_dh + (long long)(m0 + wm * 32) * a.ld_dh + wn * 32 + 4 * c4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + rl;
        const float4 hv = *reinterpret_cast<const float4*>(ep + r * EPLD + 4 * c4);
        const float4 d0 = *reinterpret_cast<const float4*>(dm + (wm * 32 + r) * 8);
        const float4 d1 = *reinterpret_cast<const float4*>(dm + (wm * 32 + r) * 8 + 4);
        const float d[7] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z};
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 7; ++q)
          if (q < nq) {
            z.x = fmaf(d[q], w[q].x, z.x); z.y = fmaf(d[q], w[q].y, z.y);
            z.z = fmaf(d[q], w[q].z, z.z); z.w = fmaf(d[q], w[q].w, z.w);
          }
        z.x = z.x * (1.0f - hv.x * hv.x); z.y = z.y * (1.0f - hv.y * hv.y);
        z.z = z.z * (1.0f - hv.z * hv.z); z.w = z.w * (1.0f - hv.w * hv.w);
        if (m0 + wm * 32 + r < g.M) *reinterpret_cast<float4*>(dst + (long long)r * a.ld_dh) = z;
      }
    }
    // ---- D2: head weight gradients of columns 16 w .. 16 w + 15 over the tile's 64 rows, on the matrix pipe:
    //          out[q][col] = sum_row dm[row][q] * h[row][col];  A[m = q][4 fq + t] = dm[16 st + 4 fq + t][q],
    //          B[4 fq + t][n = col] = h[16 st + 4 fq + t][16 w + n]
    {
      f32x4r gacc = f32x4r{0.f, 0.f, 0.f, 0.f};
      const int col = 16 * wave + fm;
      const float* hcol = smem + (col >> 5) * SLICE + (col & 31);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        float av[4], bvv4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = 16 * st + 4 * fq + t;
          av[t] = (fm < 8) ? dm[row * 8 + fm] : 0.f;
          bvv4[t] = hcol[(row >> 5) * 4 * SLICE + (row & 31) * EPLD];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bvv4[t], gacc, 0, 0, 0);
      }
      // gacc[r] = out[q = 4 fq + r][col]; record mt: [muW (act*H) | muB (act) | valW (H) | valB (1) | sigma (act)]
      float* rec = a.head_slab + (long long)mt * a.head_count;
      if (actor) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * fq + r < act) rec[(4 * fq + r) * 128 + col] = gacc[r];
      } else if (fq == 0) {
        rec[act * 128 + act + col] = gacc[0];
      }
    }
    // ---- E: the waves' bias / sigma / loss partials in wave order
    if (tid < 16) {
      float sb = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sb += red[w8 * 16 + tid];
      float* rec = a.head_slab + (long long)mt * a.head_count;
      if (actor) {
        if (tid < act) rec[act * 128 + tid] = sb;                                          // muB
        else if (tid >= 8 && tid - 8 < act) rec[act * 128 + act + 128 + 1 + (tid - 8)] = sb;   // sigma
      } else if (tid == 0) {
        rec[act * 128 + act + 128] = sb;                                                    // valB
      }
    } else if (tid >= 64 && tid < 68) {
      const int j = tid - 64;
      double sl = 0;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sl += lsum[w8 * 4 + j];
      double* lp = a.loss_part + (long long)mt * 8;
      if (actor) lp[j == 0 ? 0 : j + 1] = sl;     // a_loss, bounds, entropy, kl -> slots 0, 2, 3, 4
      else if (j == 0) lp[1] = sl;                // c_loss -> slot 1
    }
  }
};

__global__ __launch_bounds__(DMA_THREADS, 4) void k_trunk_loss(const GemmArgs g, const LossArgs a, int m_tiles) {
  TrunkLossHook hook(a);
  // no XCD remap: workgroup b and b + m_tiles (the same rows of the other net) land on the same XCD, and the round-robin
  // placement pairs an actor tile with a critic tile on a CU (the actor's scalar section is the longer one)
  gemm_dma_body<128, true, true, 0, 2, TrunkLossHook::TILE_M, false, false, false, false, TrunkLossHook>(
      g, 1, m_tiles, (int)blockIdx.x, &hook);
}

// inference heads: mu (rows,act), value (rows,1)
template <int MAXJ>
__global__ __launch_bounds__(256) void k_heads_infer(const float* __restrict__ h, long long net_stride,
                                                     int ldh, int H, const float* __restrict__ Wmu,
                                                     const float* __restrict__ bmu,
                                                     const float* __restrict__ Wv,
                                                     const float* __restrict__ bv, int rows, int act,
                                                     float* __restrict__ mu_out,
                                                     float* __restrict__ v_out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  for (int row = gw; row < rows; row += nw) {
    const float* ha_p = h + (long long)row * ldh;
    const float* hc_p = ha_p + net_stride;
    float pm[IGI_MAX_ACT], pv = 0.f;
#pragma unroll
    for (int q = 0; q < IGI_MAX_ACT; ++q) pm[q] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int k = lane + 64 * j;
      if (k < H) {
        const float ha = ha_p[k], hc = hc_p[k];
        pv += hc * Wv[k];
#pragma unroll
        for (int q = 0; q < IGI_MAX_ACT; ++q)
          if (q < act) pm[q] += ha * Wmu[q * H + k];
      }
    }
    pv = wave_sum(pv);
#pragma unroll
    for (int q = 0; q < IGI_MAX_ACT; ++q)
      if (q < act) pm[q] = wave_sum(pm[q]);
    if (v_out && lane == 0) v_out[row] = pv + bv[0];
    if (mu_out) {
#pragma unroll
      for (int q = 0; q < IGI_MAX_ACT; ++q)
        if (q < act && lane == q) mu_out[(long long)row * act + q] = pm[q] + bmu[q];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 4. gradient assembly: fixed-order sum of the split-K slabs / per-block head partials into the
//    flat gradient vector (bitwise reproducible: no atomics anywhere on the path).
// ---------------------------------------------------------------------------------------------
struct Segment {
  long long dst;      // offset in the flat gradient
  const float* src;   // first partial
  long long stride;   // between partials
  int count;          // elements = rows * cols
  int cols, src_ld;   // element e lives at (e / cols) * src_ld + e % cols of a partial (src_ld 0: dense)
  int nparts;
};
struct SegTable {
  Segment s[MAX_SEG];
  int n;
  int wide = 0;   // 1: dense 16-byte-aligned segments with >= 64 partials take the 16-byte part-group path (the student's
                  // per-workgroup gradient records: 170 - 512 partials of 16 - 80 K floats)
  // Norm fusion (the teacher's one-call update on one GPU, steps >= 1; frozen_ppo.py:605-608): every block also leaves the
  // sum of squares (fp64) of the gradient elements IT wrote in norm_part[blockIdx.y * gridDim.x + blockIdx.x], and the
  // extra grid row y == n turns the loss partials into the step's statistics row -- k_sumsq_stats then has nothing left
  // to do and is not launched (the Adam blocks add the partials in index order: clip_adam_body, NormSrc).
  double* norm_part = nullptr;
  const double* loss_part = nullptr; int loss_blocks = 0, mb = 0; float* stats_row = nullptr;
};

// strided fixed-order sums of the loss partial records -> the statistics row (means over the minibatch); one block
__device__ __forceinline__ void stats_row_block(const double* __restrict__ loss_part, int loss_blocks, int mb,
                                                float* __restrict__ stats_row) {
  // thread (q = tid&7, j = tid>>3): strided fixed-order partial sums, then a fixed tree in LDS
  __shared__ double sh[256];
  const int q = threadIdx.x & 7, j = threadIdx.x >> 3;
  double s = 0;
  for (int b0 = j; b0 < loss_blocks; b0 += 32 * 8) {  // 8 independent loads in flight, fixed order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 32 * u;
      v[u] = (b < loss_blocks) ? loss_part[b * 8 + q] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 5) {
    double t = 0;
    for (int jj = 0; jj < 32; ++jj) t += sh[jj * 8 + threadIdx.x];
    stats_row[threadIdx.x] = (float)(t / (double)mb);
  }
}

constexpr int SLAB_GX = 256;
// grid (SLAB_GX, segments).  A block covers 256/G consecutive elements with G part-groups: thread
// (e = tid % (256/G), grp = tid / (256/G)) sums parts grp, grp+G, ... with 4 independent
// accumulators; groups are combined through LDS in fixed order.  G grows with the number of
// partials so long part lists (per-block head partials) are not a serial chain.
// NORM: the norm-fusion variant (its own kernel, k_slab_reduce_norm: the block reduction costs two registers over the 64 that
// keep eight waves per SIMD, and every other caller runs the plain one)
template <bool NORM>
__device__ __forceinline__ void slab_reduce_body(const SegTable& t, float* __restrict__ grads) {
  __shared__ float sh[RED_THREADS];
  if (NORM && (int)blockIdx.y == t.n) {      // the statistics row of this step
    if (blockIdx.x == 0 && t.stats_row) stats_row_block(t.loss_part, t.loss_blocks, t.mb, t.stats_row);
    return;
  }
  const Segment sg = t.s[blockIdx.y];
  const bool norm = NORM && t.norm_part != nullptr;
  double nsq = 0.0;                  // sum of squares of the elements this thread wrote
  // fixed-order block sum of nsq -> this block's slot (every path below ends here)
  __shared__ double nred[RED_THREADS / 64];
  auto leave_norm = [&]() {
    if (!norm) return;
    nsq = wave_sum(nsq);
    if ((threadIdx.x & 63) == 0) nred[threadIdx.x >> 6] = nsq;
    __syncthreads();
    if (threadIdx.x == 0) {
      double v = nred[0];
#pragma unroll
      for (int w = 1; w < RED_THREADS / 64; ++w) v += nred[w];
      t.norm_part[(long long)blockIdx.y * gridDim.x + blockIdx.x] = v;
    }
  };
  // 16-byte path (dense, aligned segments with few partials = the big split-K slabs): four elements per
  // thread and part, four parts in flight -> 16x the bytes in flight of the scalar path below
  if (sg.src_ld == 0 && sg.nparts < 64 && (sg.count & 3) == 0 && (sg.stride & 3) == 0 && (sg.dst & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(sg.src) & 15) == 0 && (reinterpret_cast<uintptr_t>(grads) & 15) == 0) {
    const int n4 = sg.count >> 2;
    for (int e = blockIdx.x * RED_THREADS + threadIdx.x; e < n4; e += gridDim.x * RED_THREADS) {
      const float4* p = reinterpret_cast<const float4*>(sg.src) + e;
      const long long st4 = sg.stride >> 2;
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
      int k = 0;
      for (; k + 3 < sg.nparts; k += 4) {
        const float4 a0 = p[k * st4], a1 = p[(k + 1) * st4], a2 = p[(k + 2) * st4], a3 = p[(k + 3) * st4];
        s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
        s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
        s2.x += a2.x; s2.y += a2.y; s2.z += a2.z; s2.w += a2.w;
        s3.x += a3.x; s3.y += a3.y; s3.z += a3.z; s3.w += a3.w;
      }
      for (; k < sg.nparts; ++k) {
        const float4 a0 = p[k * st4];
        s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
      }
      float4 o;
      o.x = (s0.x + s1.x) + (s2.x + s3.x); o.y = (s0.y + s1.y) + (s2.y + s3.y);
      o.z = (s0.z + s1.z) + (s2.z + s3.z); o.w = (s0.w + s1.w) + (s2.w + s3.w);
      *reinterpret_cast<float4*>(grads + sg.dst + 4 * (long long)e) = o;
      if (norm) nsq += ((double)o.x * (double)o.x + (double)o.y * (double)o.y) + ((double)o.z * (double)o.z + (double)o.w * (double)o.w);
    }
    leave_norm();
    return;
  }
  if (t.wide && sg.src_ld == 0 && sg.nparts >= 64 && (sg.count & 3) == 0 && (sg.stride & 3) == 0 && (sg.dst & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(sg.src) & 15) == 0 && (reinterpret_cast<uintptr_t>(grads) & 15) == 0) {
    // many partials of a wide record: a block covers 64 consecutive elements (16 lanes x 16 bytes: 256 contiguous bytes
    // per partial) with 16 part-groups; group g sums partials g, g + 16, ... with four independent accumulators, the
    // groups meet in LDS in fixed order.  (The scalar path below gave every group 8 elements = 32 bytes per partial.)
    __shared__ float4 sh4[RED_THREADS];
    const int e4l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int n4 = sg.count >> 2;
    const long long st4 = sg.stride >> 2;
    for (int b0 = blockIdx.x * 16; b0 < n4; b0 += gridDim.x * 16) {
      const int e = b0 + e4l;
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
      if (e < n4) {
        const float4* p = reinterpret_cast<const float4*>(sg.src) + e;
        int k = grp;
        for (; k + 48 < sg.nparts; k += 64) {
          const float4 a0 = p[k * st4], a1 = p[(k + 16) * st4], a2 = p[(k + 32) * st4], a3 = p[(k + 48) * st4];
          s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
          s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
          s2.x += a2.x; s2.y += a2.y; s2.z += a2.z; s2.w += a2.w;
          s3.x += a3.x; s3.y += a3.y; s3.z += a3.z; s3.w += a3.w;
        }
        for (; k < sg.nparts; k += 16) {
          const float4 a0 = p[k * st4];
          s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
        }
      }
      float4 o;
      o.x = (s0.x + s1.x) + (s2.x + s3.x); o.y = (s0.y + s1.y) + (s2.y + s3.y);
      o.z = (s0.z + s1.z) + (s2.z + s3.z); o.w = (s0.w + s1.w) + (s2.w + s3.w);
      sh4[threadIdx.x] = o;
      __syncthreads();
      if (grp == 0 && e < n4) {
        float4 v = sh4[e4l];
#pragma unroll
        for (int q = 1; q < 16; ++q) {
          const float4 u = sh4[q * 16 + e4l];
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        *reinterpret_cast<float4*>(grads + sg.dst + 4 * (long long)e) = v;
        if (norm) nsq += ((double)v.x * (double)v.x + (double)v.y * (double)v.y) + ((double)v.z * (double)v.z + (double)v.w * (double)v.w);
      }
      __syncthreads();
    }
    leave_norm();
    return;
  }
  const int G = sg.nparts >= 256 ? 32 : (sg.nparts >= 64 ? 8 : 1);
  const int epb = RED_THREADS / G;
  const int el = threadIdx.x % epb, grp = threadIdx.x / epb;
  for (int e0 = blockIdx.x * epb; e0 < sg.count; e0 += gridDim.x * epb) {
    const int e = e0 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < sg.count) {
      const long long off = sg.src_ld ? (long long)(e / sg.cols) * sg.src_ld + (e % sg.cols) : e;
      const float* p = sg.src + off;
      const long long st = sg.stride * G;
      int k = grp;
      for (; k + 3 * G < sg.nparts; k += 4 * G) {
        const float* q = p + (long long)k * sg.stride;
        s0 += q[0]; s1 += q[st]; s2 += q[2 * st]; s3 += q[3 * st];
      }
      for (; k < sg.nparts; k += G) s0 += p[(long long)k * sg.stride];
    }
    float v = (s0 + s1) + (s2 + s3);
    if (G > 1) {
      sh[threadIdx.x] = v;
      __syncthreads();
      if (grp == 0) {
        v = 0.f;
        for (int q = 0; q < G; ++q) v += sh[q * epb + el];
      }
      __syncthreads();
    }
    if (grp == 0 && e < sg.count) {
      grads[sg.dst + e] = v;
      if (norm) nsq += (double)v * (double)v;
    }
  }
  leave_norm();
}
__global__ __launch_bounds__(RED_THREADS) void k_slab_reduce(const SegTable t, float* __restrict__ grads) {
  slab_reduce_body<false>(t, grads);
}
__global__ __launch_bounds__(RED_THREADS) void k_slab_reduce_norm(const SegTable t, float* __restrict__ grads) {
  slab_reduce_body<true>(t, grads);
}

// sum of squares of (grad*scale) and of the parameters, per block, in fp64; the extra last block
// turns the loss partials into the stats row (means over the minibatch).
__global__ __launch_bounds__(256) void k_sumsq_stats(const float* __restrict__ grads,
                                                     const float* __restrict__ params, long long P,
                                                     float scale, double* __restrict__ part,
                                                     const double* __restrict__ loss_part,
                                                     int loss_blocks, int mb,
                                                     float* __restrict__ stats_row) {
  __shared__ double red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == SUMSQ_BLOCKS) {
    stats_row_block(loss_part, loss_blocks, mb, stats_row);
    return;
  }
  double sg = 0, sp = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P;
       i += (long long)SUMSQ_BLOCKS * blockDim.x) {
    const double g = (double)(grads[i] * scale);
    const double p = (double)params[i];
    sg += g * g;
    sp += p * p;
  }
  sg = wave_sum(sg);
  sp = wave_sum(sp);
  if (lane == 0) { red[0][wave] = sg; red[1][wave] = sp; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    part[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// clip_grad_norm_ + torch.optim.Adam single-tensor step (frozen_ppo.py:608-610).
// Where the parameters of the first trunk layer sit and where their zero-padded / transposed working copies go
// (see k_pad_w1 and the extra blocks of k_gather_normalize): the fused tail keeps those copies current from inside
// the Adam pass, so the next step needs no refresh launch.
struct W1Mirror {
  float* w1p; float* wlat;
  long long o_w, ac_block; int u0, u0p, xw, xld, obs, K2p;
};

// Where the two norms of a step come from when k_sumsq_stats is not launched (the one-call update on one GPU, steps >= 1):
// the gradient's sum of squares from k_slab_reduce's per-block partials (SegTable::norm_part), the parameters' from the
// partials the PREVIOUS step's Adam blocks left of the parameters they had just written (pp_out of that pass = pp_in of
// this one; two buffers alternate, a block reads all of pp_in while others already write pp_out).
struct NormSrc {
  const double* gpart = nullptr; int n_g = 0;
  const double* pp_in = nullptr; int n_p = 0;
  double* pp_out = nullptr;        // one fp64 per Adam block: sum of squares of the UPDATED parameters it wrote (may be set alone)
};

__device__ __forceinline__ void clip_adam_body(float* __restrict__ params, const float* __restrict__ grads,
                                               float* __restrict__ m, float* __restrict__ v, long long P,
                                               const double* __restrict__ part, float scale, float max_norm, float w1,
                                               float beta2, float w2, float step_size, float bc2_sqrt, float eps,
                                               float* __restrict__ stats_row, float decay, float l2, int bid,
                                               int nblocks, const W1Mirror* mir, const NormSrc* ns = nullptr) {
  __shared__ float s_coef;
  __shared__ double s_part[2][64];
  const bool fused_norm = ns && ns->gpart;
  if (fused_norm) {
    // every thread adds its strided share of the partials in index order (eight loads in flight), the waves meet through
    // the fixed butterfly of wave_sum, wave sums in wave order: the same value in every block, launch after launch
    double sg = 0, sp = 0;
    for (int b0 = threadIdx.x; b0 < ns->n_g; b0 += 256 * 8) {
      double q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int b = b0 + 256 * u; q[u] = b < ns->n_g ? ns->gpart[b] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) sg += q[u];
    }
    for (int b = threadIdx.x; b < ns->n_p; b += 256) sp += ns->pp_in[b];
    sg = wave_sum(sg);
    sp = wave_sum(sp);
    if ((threadIdx.x & 63) == 0) { s_part[0][threadIdx.x >> 6] = sg; s_part[1][threadIdx.x >> 6] = sp; }
  } else if (threadIdx.x < 64) {
    // 128 partial pairs: lane b of the first wave adds pairs b and b + 64, lane 0 finishes in lane order
    // (every block repeats this, so a serial chain of 256 loads sat in front of each block's real work)
    double sg = 0, sp = 0;
#pragma unroll 2
    for (int b = threadIdx.x; b < SUMSQ_BLOCKS; b += 64) { sg += part[2 * b]; sp += part[2 * b + 1]; }
    s_part[0][threadIdx.x] = sg;
    s_part[1][threadIdx.x] = sp;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sg = 0, sp = 0;
    const int nsum = fused_norm ? 4 : 64;
#pragma unroll 4
    for (int b = 0; b < nsum; ++b) { sg += s_part[0][b]; sp += s_part[1][b]; }
    const float total = (float)sqrt(sg);
    float coef = 1.0f;
    if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.0f);
    s_coef = coef;
    if (bid == 0 && stats_row) {
      stats_row[5] = total;
      stats_row[6] = (float)sqrt(sp);  // the reference logs the PARAMETER norm as "grad_norms"
      stats_row[7] = coef;
    }
  }
  __syncthreads();
  const float coef = s_coef;
  double psq = 0.0;                  // sum of squares of the parameters this thread has written (for the NEXT step's log)
  // one element: exactly torch's single-tensor Adam arithmetic (each line one rounding, -ffp-contract=off)
  auto one = [&](long long i, float gi, float& pi, float& mi, float& vi) {
    float g = (gi * scale) * coef;
    if (l2 != 0.f) g += l2 * pi;        // torch.optim.Adam(weight_decay=...): L2 term joins the clipped gradient
    mi = mi + w1 * (g - mi);            // exp_avg.lerp_(grad, 1-beta1)
    vi = vi * beta2 + (w2 * g) * g;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    // AdamW: param.mul_(1 - lr * weight_decay) first (decay == 1 for plain Adam: exact no-op)
    pi = pi * decay + (-step_size) * (mi / denom);  // param.addcdiv_(exp_avg, denom, -step_size)
    psq += (double)pi * (double)pi;
    if (mir) {  // W1p[net][o][c] and the transposed latent columns follow the parameter they copy
      long long rel = i - mir->o_w;
      int net = 0;
      if (rel >= mir->ac_block) { rel -= mir->ac_block; net = 1; }
      if (rel >= 0 && rel < (long long)mir->u0 * mir->xw) {
        const int o = (int)(rel / mir->xw), c = (int)(rel - (long long)o * mir->xw);
        mir->w1p[((long long)net * mir->u0p + o) * mir->xld + c] = pi;
        if (mir->wlat && c >= mir->obs && c < mir->obs + 8)
          mir->wlat[(long long)(c - mir->obs) * mir->K2p + net * mir->u0p + o] = pi;
      }
    }
  };
  const bool vec = (P & 3) == 0 && ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
                                     reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  if (vec) {  // 16-byte accesses: four consecutive elements per thread and trip
    const long long P4 = P >> 2;
#pragma unroll 1
    for (long long q = (long long)bid * blockDim.x + threadIdx.x; q < P4; q += (long long)nblocks * blockDim.x) {
      const float4 g4 = reinterpret_cast<const float4*>(grads)[q];
      float4 p4 = reinterpret_cast<float4*>(params)[q];
      float4 m4 = reinterpret_cast<float4*>(m)[q], v4 = reinterpret_cast<float4*>(v)[q];
      one(4 * q + 0, g4.x, p4.x, m4.x, v4.x);
      one(4 * q + 1, g4.y, p4.y, m4.y, v4.y);
      one(4 * q + 2, g4.z, p4.z, m4.z, v4.z);
      one(4 * q + 3, g4.w, p4.w, m4.w, v4.w);
      reinterpret_cast<float4*>(params)[q] = p4;
      reinterpret_cast<float4*>(m)[q] = m4;
      reinterpret_cast<float4*>(v)[q] = v4;
    }
  } else {
#pragma unroll 1
    for (long long i = (long long)bid * blockDim.x + threadIdx.x; i < P; i += (long long)nblocks * blockDim.x) {
      float pi = params[i], mi = m[i], vi = v[i];
      one(i, grads[i], pi, mi, vi);
      params[i] = pi;
      m[i] = mi;
      v[i] = vi;
    }
  }
  if (ns && ns->pp_out) {   // this block's share of |params|^2 after the step, for the next step's statistics row
    psq = wave_sum(psq);
    __syncthreads();        // (s_part was read by thread 0 above)
    if ((threadIdx.x & 63) == 0) s_part[0][threadIdx.x >> 6] = psq;
    __syncthreads();
    if (threadIdx.x == 0) ns->pp_out[bid] = (s_part[0][0] + s_part[0][1]) + (s_part[0][2] + s_part[0][3]);
  }
}

__global__ __launch_bounds__(256) void k_clip_adam(float* __restrict__ params,
                                                   const float* __restrict__ grads,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   long long P, const double* __restrict__ part,
                                                   float scale, float max_norm, float w1, float beta2,
                                                   float w2, float step_size, float bc2_sqrt, float eps,
                                                   float* __restrict__ stats_row, float decay = 1.0f,
                                                   float l2 = 0.0f, const NormSrc ns = NormSrc()) {
  clip_adam_body(params, grads, m, v, P, part, scale, max_norm, w1, beta2, w2, step_size, bc2_sqrt, eps, stats_row,
                 decay, l2, (int)blockIdx.x, (int)gridDim.x, nullptr, &ns);
}

// Tail of optimizer step s fused with the head of step s+1: blocks [0, adam_blocks) run clip + Adam (and keep the
// padded first-layer weight copies current), the remaining blocks gather + normalise the NEXT minibatch -- it
// depends only on the rollout, the permutation and the pre-scanned normaliser trajectory, never on the parameters,
// and the activations it overwrites were last read by this step's weight-gradient launches (earlier in the stream).
// One launch and one kernel boundary less per optimizer step; same arithmetic as the two separate kernels.
struct AdamArgs {
  float* params; const float* grads; float* m; float* v; long long P; const double* part;
  float scale, max_norm, w1, beta2, w2, step_size, bc2_sqrt, eps; float* stats_row;
};
__global__ __launch_bounds__(256) void k_adam_gather(const AdamArgs a, const W1Mirror mir, const GatherArgs g,
                                                     int adam_blocks, const NormSrc ns) {
  // the gather blocks come first in the grid (the longer dependent chain: index -> row -> store), so that both kinds
  // are resident from the start
  const int gblocks = (int)gridDim.x - adam_blocks;
  if ((int)blockIdx.x < gblocks)
    gather_normalize_body(g, (int)blockIdx.x, gblocks);
  else
    clip_adam_body(a.params, a.grads, a.m, a.v, a.P, a.part, a.scale, a.max_norm, a.w1, a.beta2, a.w2, a.step_size,
                   a.bc2_sqrt, a.eps, a.stats_row, 1.0f, 0.0f, (int)blockIdx.x - gblocks, adam_blocks, &mir, &ns);
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static int check_state(const TeacherPlan& p, const igi_teacher_state* st) {
  if (!st || !st->params || !st->workspace) return IGI_E_BADARG;
  if (st->workspace_bytes < p.w_total) return IGI_E_WORKSPACE;
  return 0;
}

static int teacher_prepare(const igi_teacher_cfg* c, const igi_rollout* ro,
                           const igi_teacher_state* st, int normalize_value, hipStream_t s) {
  TeacherPlan p;
  int rc = make_plan(c, &p);
  if (rc) return rc;
  if ((rc = check_state(p, st))) return rc;
  if (!ro || !ro->rewards || !ro->values || !ro->dones || !ro->last_values || !ro->mus ||
      !ro->sigmas || !st->returns_raw || !st->advantages || !st->values_n || !st->returns_n ||
      !st->mus_w || !st->sigmas_w || (normalize_value && !st->rms_value))
    return IGI_E_BADARG;
  double* part = wsp<double>(st, p.w_prep_part);
  float* coef = wsp<float>(st, p.w_prep_coef);
  const float gamma = (float)c->gamma;
  const float gamma_tau = (float)((double)c->gamma * (double)c->tau);
  ProfScope ps(PC_PREPARE, s, 0.0, 17.0 * (double)p.Bsz + 8.0 * p.act * (double)p.Bsz * 2, /*ext=*/false);
  IGI_LAUNCH(k_gae, dim3(p.gae_blocks), dim3(PREP_THREADS), 0, s, ro->rewards, ro->values,
                     ro->dones, ro->last_values, st->returns_raw, p.N, p.T, gamma, gamma_tau, part);
  IGI_LAUNCH(k_prep_final, dim3(1), dim3(64), 0, s, part, p.gae_blocks, p.Bsz,
                     st->rms_value, c->rms_eps, coef, normalize_value);
  int nb = (int)((p.Bsz * p.act + PREP_THREADS - 1) / PREP_THREADS);
  if (nb > 2048) nb = 2048;
  IGI_LAUNCH(k_prep_norm, dim3(nb), dim3(PREP_THREADS), 0, s, ro->values, st->returns_raw,
                     ro->mus, ro->sigmas, coef, st->advantages, st->values_n, st->returns_n,
                     st->mus_w, st->sigmas_w, p.Bsz, p.act, normalize_value);
  // normaliser trajectory for the E*n_mb optimizer steps of this update (see k_rms_traj)
  if (ro->obses && ro->priv_info && st->perm && st->rms_obs && st->rms_priv) {
    const int D = p.obs + p.priv;
    double* rpart = wsp<double>(st, p.w_rms_part);
    for (int i = 0; i < p.nmb; ++i)
      IGI_LAUNCH(k_gather_stats, dim3(p.gs_blocks), dim3(GS_THREADS),
                         (size_t)p.gs_rows * (D + 2) * sizeof(float), s, ro->obses, ro->priv_info, st->perm,
                         (long long)i * p.mb, p.mb, p.N, p.T, p.obs, p.priv, p.gs_rows, wsp<float>(st, p.w_xcat),
                         p.xld, wsp<float>(st, p.w_priv), ru4(p.priv), rpart + (long long)i * p.gs_blocks * D * 2);
    IGI_LAUNCH(k_mb_moments, dim3(p.nmb), dim3(RMSF_THREADS), 0, s, rpart, p.gs_blocks, p.mb, D,
                       wsp<float>(st, p.w_moments));
    IGI_LAUNCH(k_rms_traj, dim3(1), dim3(128), 0, s, wsp<float>(st, p.w_moments), p.nmb, p.E * p.nmb,
                       p.mb, p.obs, p.priv, st->rms_obs, st->rms_priv, c->rms_eps, wsp<float>(st, p.w_traj_coef),
                       wsp<double>(st, p.w_traj_state));
  }
  return (int)hipGetLastError();
}

// W1p[net][o][c] = first trunk layer weight padded to xld columns (zeros) so the layer runs on
// the LDS-DMA kernel (k-tile 32) and its dgrad / wgrad see aligned 16-byte rows.
__global__ __launch_bounds__(256) void k_pad_w1(const float* __restrict__ params, long long o_w, long long ac_block,
                                                int u0, int u0p, int xw, int xld, float* __restrict__ w1p) {
  const int total = 2 * u0p * xld;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int c = e % xld;
    const int o = (e / xld) % u0p;
    const int net = e / (xld * u0p);
    w1p[e] = (c < xw && o < u0) ? params[o_w + net * ac_block + (long long)o * xw + c] : 0.f;
  }
}

// d(latent pre-activation) = ([dZ1_actor | dZ1_critic] . W1p[:, obs:obs+LAT]) * (1 - latent^2).
// Only LAT (= 8) of the first layer's input columns need a gradient, so this is a skinny
// (mb x K) . (K x 8) product: one wave per row, each lane keeps its 16 x 8 slice of the weight in
// registers for all of its rows and the 8 row sums are wave reductions -- HBM-bound on reading dZ1
// once (a 128 x 64 MFMA tile would spend 8x the useful FLOPs on padding here).
template <int KQ, int LAT>
__global__ __launch_bounds__(256) void k_latent_dgrad(const float* __restrict__ dz, int ldz,
                                                      const float* __restrict__ w1p, int xld, int obs,
                                                      const float* __restrict__ xcat, float* __restrict__ dxcat,
                                                      int mb, int rows_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w[KQ][4][LAT];
#pragma unroll
  for (int q = 0; q < KQ; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 4 * lane + 256 * q + i;
#pragma unroll
      for (int j = 0; j < LAT; ++j) w[q][i][j] = w1p[(long long)k * xld + obs + j];
    }
  const int gw = blockIdx.x * 4 + wave;
  for (int it = 0; it < rows_per_wave; it += 4) {  // four rows in flight: their loads are independent
    const int row0 = gw * rows_per_wave + it;
    if (row0 >= mb) break;
    float4 v[4][KQ];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(row0 + r, mb - 1);
#pragma unroll
      for (int q = 0; q < KQ; ++q)
        v[r][q] = *reinterpret_cast<const float4*>(dz + (long long)row * ldz + 4 * lane + 256 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + r;
      float acc[LAT];
#pragma unroll
      for (int j = 0; j < LAT; ++j) acc[j] = 0.f;
#pragma unroll
      for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int j = 0; j < LAT; ++j)
          acc[j] += ((v[r][q].x * w[q][0][j] + v[r][q].y * w[q][1][j]) + v[r][q].z * w[q][2][j]) + v[r][q].w * w[q][3][j];
#pragma unroll
      for (int j = 0; j < LAT; ++j) acc[j] = wave_sum(acc[j]);
      if (row < mb && it + r < rows_per_wave) {
#pragma unroll
        for (int j = 0; j < LAT; ++j) {
          if (lane == j) {
            const float t = xcat[(long long)row * xld + obs + j];
            dxcat[(long long)row * xld + obs + j] = acc[j] * (1.0f - t * t);
          }
        }
      }
    }
  }
}

// Fused tail of the backward pass around the 8-wide latent: the latent data gradient, plus the last
// env_mlp layer's data gradient (d pre-activation of the previous env layer, times tanh') and its
// weight / bias gradient.  All are rank-8 products over rows the block already holds, so they ride in
// one HBM-bound kernel instead of an MFMA launch padded 8x plus two latency-bound generic GEMMs.
//   dze3[j]     = (sum_k dZ1[row][k] * W1p[k][obs+j]) * (1 - latent_j^2)
//   dze2[row][c]= (sum_j dze3[j] * We3[j][c]) * (1 - e2[row][c]^2)
//   dWe3[j][c] += dze3[j] * e2[row][c] ;  dbe3[j] += dze3[j]      (per-block partials, reduced later)
// A block takes 32 rows.  Phase 1 is a matrix-pipe product: the dZ1 rows stream through LDS in 256-column chunks
// (coalesced 16-byte loads, next chunk prefetched into registers), the eight weight rows sit in LDS transposed,
// and each wave multiplies 16 rows by the eight outputs with v_mfma_f32_16x16x4_f32 (see the loop).  Measured: the
// chunk loop runs at the stream's rate (67 MB in ~14 us); the other ~13 us of the kernel are launch, the weight
// copy, phase 2 and the block reduction.  Phase 2 gives each wave its 8 rows for the rank-8 updates; its operands
// are requested before phase 1.
constexpr int LATB_ROWS = 32;
constexpr int LATB_CH = 256;            // columns per staged chunk
constexpr int LATB_LD = LATB_CH + 4;    // padded row stride (floats): rows land on distinct LDS banks
// PARTS: phase 1 already happened inside the data-gradient tiles that produced dZ1 (GemmArgs::rowdot_*): `dz` then
// points at their per-tile row dots [ldz tiles][mb][8], which are added in tile order; dZ1 is not read again.
template <int MAXJ, bool PARTS = false>
__global__ __launch_bounds__(256) void k_latent_bwd(const float* __restrict__ dz, int ldz, int K2,
                                                    const float* __restrict__ wlat, int xld, int obs,
                                                    const float* __restrict__ xcat, float* __restrict__ dxcat,
                                                    const float* __restrict__ e2, int lde, int H2,
                                                    const float* __restrict__ We3, float* __restrict__ dze2,
                                                    float* __restrict__ partial, int mb) {
  constexpr int LAT = 8;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K2p = (K2 + LATB_CH - 1) / LATB_CH * LATB_CH;
  const int wld = K2p + 4;
  float* wt = sm;                               // [LAT][wld]: latent columns of W1p, transposed, zero beyond K2
  float* tile = wt + LAT * wld;                 // [LATB_ROWS][LATB_LD]; reused for the final block reduction
  float* psum = tile + LATB_ROWS * LATB_LD;     // [LATB_ROWS][LAT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (!PARTS)
  for (int e = tid; e < LAT * K2p / 4; e += 256) {  // wlat is [LAT][K2p], already zero beyond K2: coalesced copy
    const int j = e / (K2p / 4), k4 = e - j * (K2p / 4);
    *reinterpret_cast<float4*>(wt + j * wld + 4 * k4) = *reinterpret_cast<const float4*>(wlat + (long long)j * K2p + 4 * k4);
  }
  float we[LAT][MAXJ], gw[LAT][MAXJ];
#pragma unroll
  for (int j = 0; j < LAT; ++j)
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
      const int c = lane + 64 * jj;
      we[j][jj] = (c < H2) ? We3[j * H2 + c] : 0.f;
      gw[j][jj] = 0.f;
    }
  float gb = 0.f;  // lane j (< 8) accumulates dbe3[j]
  auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
  const int row0 = blockIdx.x * LATB_ROWS;
  // phase 1 on the matrix pipe: per staged 256-column chunk, wave w multiplies rows 16 (w & 1) .. +15 by the eight
  // weight rows over every second 16-k slice (starting at w >> 1) with v_mfma_f32_16x16x4_f32: lane (m = lane & 15,
  // q = lane >> 4) supplies A[m][4q + t] and B[4q + t][n = m] to the t-th instruction of a slice, i.e. one 16-byte
  // LDS read of the staged dZ1 row m and one of weight row n per four MFMAs (the VALU version needed two 16-byte
  // reads per four FMAs per thread).  Outputs n >= 8 multiply zeros.  The two k-halves meet in LDS afterwards.
  // phase-2 operands of this wave's eight rows are requested now: their latency hides under phase 1
  float tl8[8], ee8[8][MAXJ];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = min(row0 + wave * 8 + r, mb - 1);
    tl8[r] = xcat[(long long)row * xld + obs + (lane & 7)];
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
      const int c = lane + 64 * jj;
      ee8[r][jj] = (c < H2) ? e2[(long long)row * lde + c] : 0.f;
    }
  }
  if constexpr (!PARTS) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const int fm = lane & 15, fq = lane >> 4, rh = wave & 1, kh = wave >> 1;
  const bool bvalid = fm < LAT;
  f32x4_t facc = {0.f, 0.f, 0.f, 0.f};
  float4 ld[8];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx >> 6, c4 = idx & 63;
      const int row = min(row0 + r, mb - 1);
      const int k = c0 + 4 * c4;
      ld[i] = (k < K2) ? *reinterpret_cast<const float4*>(dz + (long long)row * ldz + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < K2p; c0 += LATB_CH) {
    __syncthreads();  // previous chunk consumed (and wt written, first time round)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      *reinterpret_cast<float4*>(tile + (idx >> 6) * LATB_LD + 4 * (idx & 63)) = ld[i];
    }
    __syncthreads();
    if (c0 + LATB_CH < K2p) fetch(c0 + LATB_CH);  // in flight during the arithmetic below
    const float* xr = tile + (rh * 16 + fm) * LATB_LD + 4 * fq;
    const float* wr = wt + min(fm, LAT - 1) * wld + c0 + 4 * fq;
#pragma unroll
    for (int c = 0; c < LATB_CH / 32; ++c) {       // this wave's 16-k slices: kh, kh + 2, ...
      const int off = 16 * (kh + 2 * c);
      const float4 x = *reinterpret_cast<const float4*>(xr + off);
      float4 wv = *reinterpret_cast<const float4*>(wr + off);
      if (!bvalid) wv = make_float4(0.f, 0.f, 0.f, 0.f);
      facc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, wv.x, facc, 0, 0, 0);
      facc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, wv.y, facc, 0, 0, 0);
      facc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, wv.z, facc, 0, 0, 0);
      facc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, wv.w, facc, 0, 0, 0);
    }
  }
  // C/D layout: register r of lane (m, q) is D[row 4q + r][output m]; k-half 1 parks its part in LDS, half 0 adds
  __syncthreads();                                 // the staging tile is free
  float* khalf = tile;                             // [LATB_ROWS][LAT]
  if (kh == 1 && bvalid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) khalf[(rh * 16 + 4 * fq + r) * LAT + fm] = facc[r];
  }
  __syncthreads();
  if (kh == 0 && bvalid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = (rh * 16 + 4 * fq + r) * LAT + fm;
      psum[e] = facc[r] + khalf[e];
    }
  }
  __syncthreads();
  } else {
    // thread (row = tid / 8, j = tid % 8) adds the tiles' row dots in tile order
    const int r_ = tid >> 3, j_ = tid & 7;
    const int row_ = min(row0 + r_, mb - 1);
    float acc_ = 0.f;
    for (int t0 = 0; t0 < ldz; t0 += 8) {   // eight tiles' partials per round trip (one each was the kernel's duration)
      float v_[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v_[u] = dz[((long long)min(t0 + u, ldz - 1) * mb + row_) * LAT + j_];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t0 + u < ldz) acc_ += v_[u];
    }
    psum[r_ * LAT + j_] = acc_;
    __syncthreads();
  }
  // ---- phase 2: wave handles rows wave*8 .. wave*8+7
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int rr = wave * 8 + r;
    const int row = row0 + rr;
    const bool live = row < mb;
    const float ps = psum[rr * LAT + (lane & 7)];
    const float dl = (live && lane < LAT) ? ps * (1.0f - tl8[r] * tl8[r]) : 0.f;  // lane j: dze3[j]
    if (live && lane < LAT) {
      dxcat[(long long)row * xld + obs + lane] = dl;
      gb += dl;
    }
    float d[LAT];
#pragma unroll
    for (int j = 0; j < LAT; ++j) d[j] = rl(dl, j);
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
      const int c = lane + 64 * jj;
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < LAT; ++j) {
        sacc = fmaf(d[j], we[j][jj], sacc);
        gw[j][jj] = fmaf(d[j], ee8[r][jj], gw[j][jj]);
      }
      if (live && c < H2) dze2[(long long)row * lde + c] = sacc * (1.0f - ee8[r][jj] * ee8[r][jj]);
    }
  }
  // block partial [LAT*H2 | LAT] (the staging tile is free now)
  __syncthreads();
  float* red = tile;
  const int pc = LAT * H2 + LAT;
  float* mine_p = red + wave * pc;
#pragma unroll
  for (int j = 0; j < LAT; ++j)
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
      const int c = lane + 64 * jj;
      if (c < H2) mine_p[j * H2 + c] = gw[j][jj];
    }
  if (lane < LAT) mine_p[LAT * H2 + lane] = gb;
  __syncthreads();
  for (int e = threadIdx.x; e < pc; e += blockDim.x)
    partial[(long long)blockIdx.x * pc + e] = (red[e] + red[pc + e]) + (red[2 * pc + e] + red[3 * pc + e]);
}

// forward through env_mlp -> xcat -> actor/critic trunk for `rows` rows already staged
// (normalised) in priv_g / xcat.
static int trunk_forward(const TeacherPlan& p, const igi_teacher_state* st, int rows, bool pad_w1, hipStream_t s,
                         int nl_run = -1) {   // nl_run: trunk layers to run (the training step leaves the last one to k_trunk_loss)
  if (nl_run < 0) nl_run = p.nl;
  const float* P = st->params;
  float* priv_g = wsp<float>(st, p.w_priv);
  float* xcat = wsp<float>(st, p.w_xcat);
  float* w1p = wsp<float>(st, p.w_w1p);
  const long long mbs = p.mb;
  if (pad_w1) {
    ProfScope ps(PC_OTHER, s, 0.0, 8.0 * 2 * p.u0p * p.xld);
    IGI_LAUNCH(k_pad_w1, dim3((2 * p.u0p * p.xld + 255) / 256), dim3(256), 0, s, P, p.o_acW[0],
                       p.ac_block, p.u[0], p.u0p, p.xw, p.xld, w1p);
  }
  // env_mlp: tanh after every layer, the last one lands in xcat[:, obs:]
  const float* in = priv_g;
  int ldin = ru4(p.priv);
  static int fuse_head = -1;
  if (fuse_head < 0) { const char* e = getenv("IGI_FUSE_HEAD"); fuse_head = e ? atoi(e) : 1; }
  static int env_fused = -1;
  if (env_fused < 0) { const char* e = getenv("IGI_ENV_FUSED"); env_fused = e ? atoi(e) : 1; }
  bool env_done = false;
  int first_trunk_layer = 0;
  // round 6: env_mlp AND the first trunk layer of both nets as one persistent launch (fwd12.h); IGI_FWD12=0 keeps
  // k_env_fwd + the layer's own launch below
  if (fwd12_enabled() && env_fused && p.npl == 3 && nl_run >= 1 && rows >= 2048 && !bf16_mode() &&
      fwd12_supported(p.priv, p.pu[0], p.pu[1], p.pu[2], p.obs, p.xld, p.u[0]) && p.u0p == p.u[0]) {
    Fwd12Args f;
    f.priv = priv_g; f.ldp = ldin; f.xcat = xcat; f.ldx = p.xld; f.M = rows; f.obs = p.obs;
    f.eW1 = P + p.o_envW[0]; f.eb1 = P + p.o_envB[0]; f.eW2 = P + p.o_envW[1]; f.eb2 = P + p.o_envB[1];
    f.eW3 = P + p.o_envW[2]; f.eb3 = P + p.o_envB[2];
    f.w1p = w1p; f.tb1 = P + p.o_acB[0]; f.ac_block = p.ac_block;
    f.e1 = wsp<float>(st, p.w_e[0]); f.lde1 = ru4(p.pu[0]);
    f.e2 = wsp<float>(st, p.w_e[1]); f.lde2 = ru4(p.pu[1]);
    f.h1 = wsp<float>(st, p.w_h[0]); f.ldh = ru4(p.u[0]); f.sH = mbs * ru4(p.u[0]);
    if (fwd12_forward(f, s) == hipSuccess) { env_done = true; first_trunk_layer = 1; }
  }
  // (the fused kernel is one workgroup per 64 rows with a fixed ~20 us chain: at the rollout's 4096 rows it fills a
  // quarter of the chip and the per-layer launches win -- measured 8.4 -> 8.0 ms per 32-step rollout)
  if (!env_done && env_fused && p.npl == 3 && rows >= 8192) {
    // the whole env_mlp of a 64-row block in one workgroup (env_mlp.h); other shapes run layer by layer below
    EnvFwdArgs a;
    a.priv = priv_g; a.ldp = ldin;
    a.W1 = P + p.o_envW[0]; a.b1 = P + p.o_envB[0];
    a.W2 = P + p.o_envW[1]; a.b2 = P + p.o_envB[1];
    a.W3 = P + p.o_envW[2]; a.b3 = P + p.o_envB[2];
    a.e1 = wsp<float>(st, p.w_e[0]); a.lde1 = ru4(p.pu[0]);
    a.e2 = wsp<float>(st, p.w_e[1]); a.lde2 = ru4(p.pu[1]);
    a.out = xcat + p.obs; a.ldo = p.xld;
    a.M = rows; a.K1 = p.priv; a.N1 = p.pu[0]; a.N2 = p.pu[1]; a.N3 = p.pu[2]; a.ldw2 = p.pu[0];
    env_done = env_mlp_forward(a, s) == hipSuccess;
  }
  for (int l = 0; l < p.npl && !env_done; ++l) {
    GemmArgs g;
    g.A = in; g.lda = ldin;
    g.B = P + p.o_envW[l]; g.ldb = env_in(p, l);
    g.bias = P + p.o_envB[l];
    g.M = rows; g.N = p.pu[l]; g.K = env_in(p, l);
    if (l == p.npl - 1) { g.C = xcat + p.obs; g.ldc = p.xld; }
    else { g.C = wsp<float>(st, p.w_e[l]); g.ldc = ru4(p.pu[l]); }
    g.epilogue = EPI_BIAS_TANH;
    if (fuse_head && l == p.npl - 2 && p.pu[l + 1] <= 8) {
      // the <= 8-wide latent layer rides in this layer's epilogue (one launch less per step)
      GemmArgs gh = g;
      gh.head_W = P + p.o_envW[l + 1]; gh.head_b = P + p.o_envB[l + 1];
      gh.head_out = xcat + p.obs; gh.head_ld = p.xld; gh.head_n = p.pu[l + 1];
      if (gemm_with_head(gh, s) == hipSuccess) break;
    }
    IGI_HIP_TRY(gemm(g, true, true, s));
    in = g.C; ldin = g.ldc;
  }
  // actor + critic, batched (critic parameters sit ac_block floats after the actor's)
  in = xcat; ldin = p.xld;
  long long sIn = 0;
  if (first_trunk_layer == 1) { in = wsp<float>(st, p.w_h[0]); ldin = ru4(p.u[0]); sIn = mbs * ru4(p.u[0]); }
  for (int l = first_trunk_layer; l < nl_run; ++l) {
    GemmArgs g;
    g.A = in; g.lda = ldin; g.sA = sIn;
    if (l == 0) { g.B = w1p; g.ldb = p.xld; g.sB = (long long)p.u0p * p.xld; g.K = p.xld; g.flop_credit = (double)p.xw / p.xld; }
    else { g.B = P + p.o_acW[l]; g.ldb = ac_in(p, l); g.sB = p.ac_block; g.K = ac_in(p, l); }
    g.bias = P + p.o_acB[l]; g.sBias = p.ac_block;
    g.M = rows; g.N = p.u[l];
    g.C = wsp<float>(st, p.w_h[l]); g.ldc = ru4(p.u[l]); g.sC = mbs * ru4(p.u[l]);
    g.nbatch = 2;
    g.epilogue = EPI_BIAS_TANH;
    IGI_HIP_TRY(gemm(g, true, true, s));
    in = g.C; ldin = g.ldc; sIn = g.sC;
  }
  return 0;
}

// phase -1: the whole step.  Data-parallel runs split it so that the gradient all-reduce of the big
// bucket overlaps the rest of backward (frozen_ppo.py:586-603 reduces everything after backward).  The cut follows the
// backward levels, so both phases launch exactly the kernels of the unsplit step (the only extra launch is the second
// call of k_slab_reduce):
//   phase 0: gather, forward, loss, trunk backward down to dZ of the FIRST trunk layer, sum of the split-K slabs of the
//            heads and of trunk layers >= 1       -> the EARLY bucket is final: actor layers >= 1 | critic layers >= 1,
//            value, mu  (two ranges of the flat gradient: the critic's first layer lies between them)
//   phase 1: latent + env_mlp backward with the first trunk layer's weight gradient riding in the env level's grid,
//            sum of the remaining slabs          -> the LATE bucket: sigma, env_mlp, actor layer 0 | critic layer 0
struct GradBuckets { long long off[4], len[4]; };   // [0], [1]: early; [2], [3]: late (a length may be 0)
static GradBuckets grad_buckets(const TeacherPlan& p) {
  const long long a0 = p.o_acW[0], blk = p.ac_block;
  const long long rel1 = p.nl > 1 ? p.o_acW[1] - p.o_acW[0] : blk;   // first trunk layer's share of a net's block
  GradBuckets b;
  b.off[0] = a0 + rel1;        b.len[0] = blk - rel1;                 // actor layers >= 1
  b.off[1] = a0 + blk + rel1;  b.len[1] = p.P - b.off[1];             // critic layers >= 1, value, mu
  b.off[2] = 0;                b.len[2] = a0 + rel1;                  // sigma, env_mlp, actor layer 0
  b.off[3] = a0 + blk;         b.len[3] = rel1;                       // critic layer 0
  return b;
}

static GatherArgs gather_args(const TeacherPlan& p, const igi_rollout* ro, const igi_teacher_state* st, int mb_index,
                              int step_slot) {
  const int D = p.obs + p.priv;
  GatherArgs a;
  a.obses = ro->obses; a.priv_info = ro->priv_info; a.perm = st->perm;
  a.start = (long long)mb_index * p.mb; a.mb = p.mb; a.N = p.N; a.T = p.T; a.obs = p.obs; a.priv = p.priv;
  a.rows_per_block = p.gs_rows; a.gather_blocks = p.gs_blocks;
  a.coef = wsp<float>(st, p.w_traj_coef) + (long long)step_slot * 2 * D;
  a.state_row = wsp<double>(st, p.w_traj_state) + (long long)step_slot * (2 * D + 2);
  a.rms_obs = st->rms_obs; a.rms_priv = st->rms_priv;
  a.xcat = wsp<float>(st, p.w_xcat); a.xld = p.xld; a.xw = p.xw;
  a.priv_g = wsp<float>(st, p.w_priv); a.pld = ru4(p.priv);
  a.params = st->params; a.o_w = p.o_acW[0]; a.ac_block = p.ac_block; a.u0 = p.u[0]; a.u0p = p.u0p;
  a.w1p = wsp<float>(st, p.w_w1p);
  a.wlat = p.lat_fused ? wsp<float>(st, p.w_wlat) : (float*)nullptr;
  a.K2p = (2 * p.u0p + 255) / 256 * 256;
  return a;
}

// Data gradient of trunk layer l (> 0) into layer l-1: dZ_{l-1} = (dZ_l . W_l) * tanh'(h_{l-1}), both nets batched.
// with_rowdot (l == 1): the dZ1 tiles also emit their share of dZ1 . W1[:, latent columns] (GemmArgs::rowdot_*).
static GemmArgs trunk_dgrad_args(const TeacherPlan& p, const igi_teacher_state* st, int l, bool with_rowdot) {
  const long long mbs = p.mb;
  auto dz_ld = [&](int k) { return k == 0 ? 2 * p.u0p : ru4(p.u[k]); };
  auto dz_stride = [&](int k) { return k == 0 ? (long long)p.u0p : mbs * ru4(p.u[k]); };
  GemmArgs g;
  g.A = wsp<float>(st, p.w_dh[l]); g.lda = dz_ld(l); g.sA = dz_stride(l);
  g.B = st->params + p.o_acW[l]; g.ldb = ac_in(p, l); g.sB = p.ac_block;
  g.M = p.mb; g.N = ac_in(p, l); g.K = p.u[l];
  g.C = wsp<float>(st, p.w_dh[l - 1]); g.ldc = dz_ld(l - 1); g.sC = dz_stride(l - 1);
  g.aux = wsp<float>(st, p.w_h[l - 1]); g.ldaux = ru4(p.u[l - 1]); g.sAux = mbs * ru4(p.u[l - 1]);
  g.nbatch = 2;
  g.epilogue = EPI_TANHGRAD;
  if (with_rowdot) {
    g.rowdot_W = wsp<float>(st, p.w_wlat); g.rowdot_ld = (2 * p.u0p + 255) / 256 * 256; g.rowdot_kz = p.u0p;
    g.rowdot_out = wsp<float>(st, p.w_lat_rowdot);
    if (p.lw_parts > 0 && l == 1) {
      float* slab = wsp<float>(st, p.w_slab);
      const int out0 = p.u[0];
      g.lw_X = wsp<float>(st, p.w_xcat); g.lw_ldx = p.xld; g.lw_xw = p.xw; g.lw_chain = p.lw_chain;
      g.lw_out = slab + p.s_acW[0]; g.lw_sPart = 2LL * out0 * p.xld; g.lw_sNet = (long long)out0 * p.xld;
      g.lw_bias = slab + p.s_acB[0]; g.lw_bsPart = 2LL * out0; g.lw_bsNet = out0;
    }
  }
  return g;
}

// The latent gradient's K = 2*u0 contraction can ride in the tiles of the data gradient that produces dZ1 (no second
// pass over its 67 MB): needs the level-fused multi kernel for that product.  Decided from the plan and the state's
// addresses alone, with the very predicate the launcher applies (gemm_multi_dgrad_ok), so that both halves of a phased
// step agree and the launch can never decline row dots the plan counted on (it falls back to k_latent_bwd<., false>).
static bool latent_rowdot(const TeacherPlan& p, const igi_teacher_state* st) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_LAT_ROWDOT"); on = e ? atoi(e) : 1; }
  if (!(on && p.lat_fused && p.nl >= 2 && gemm_level_enabled() && p.mb >= 4)) return false;
  GemmArgs g = trunk_dgrad_args(p, st, 1, true);
  return gemm_multi_dgrad_ok(g);
}

// skip_gather: the previous step's fused tail (k_adam_gather) already gathered + normalised this minibatch
// norm_parts (phase -1 only): non-null turns the norm fusion on -- k_slab_reduce also leaves the gradient's sum-of-squares
// partials (their count comes back in *norm_parts) and writes this step's statistics row; the caller then runs
// teacher_apply(..., norm_mode 2), which does not launch k_sumsq_stats
static int teacher_fwd_bwd(const igi_teacher_cfg* c, const igi_rollout* ro,
                           const igi_teacher_state* st, int mb_index, int step_slot, hipStream_t s,
                           int phase = -1, bool skip_gather = false, int* norm_parts = nullptr) {
  TeacherPlan p;
  int rc = make_plan(c, &p);
  if (rc) return rc;
  if ((rc = check_state(p, st))) return rc;
  if (!ro || !ro->obses || !ro->priv_info || !ro->actions || !ro->neglogpacs || !st->grads ||
      !st->perm || !st->rms_obs || !st->rms_priv || !st->stats || !st->advantages ||
      mb_index < 0 || mb_index >= p.nmb || step_slot < 0)
    return IGI_E_BADARG;
  const float* P = st->params;
  const int mb = p.mb;
  const long long mbs = mb;
  float* priv_g = wsp<float>(st, p.w_priv);
  float* xcat = wsp<float>(st, p.w_xcat);
  float* dxcat = wsp<float>(st, p.w_dxcat);
  float* w1p = wsp<float>(st, p.w_w1p);
  const int pld = ru4(p.priv);
  const int D = p.obs + p.priv;

  // ---- gather + normalise with this step's pre-scanned running statistics (experience.py:207-226;
  //      frozen_ppo.py:521-522) + publish the running state + refresh the padded first-layer weight
  if (mb_index != step_slot % p.nmb || step_slot >= p.E * p.nmb) return IGI_E_BADARG;  // canonical step order
  if (phase < -1 || phase > 1) return IGI_E_BADARG;
  const bool do0 = phase != 1, do1 = phase != 0;
  if (do0 && !skip_gather) {
    ProfScope ps(PC_GATHER_NORMALIZE, s, 0.0, 8.0 * (double)mbs * D + 8.0 * mbs);
    const int pad_blocks = 16;
    const GatherArgs ga = gather_args(p, ro, st, mb_index, step_slot);
    IGI_LAUNCH(k_gather_normalize, dim3(p.gs_blocks + pad_blocks), dim3(GS_THREADS), 0, s, ga);
  }
  // ---- forward trunk (models_split.py:166-232)
  if (do0 && (rc = trunk_forward(p, st, mb, false, s, p.loss_fused ? p.nl - 1 : p.nl))) return rc;

  // ---- heads + loss + head backward.  d(pre-activation) of the FIRST trunk layer is kept
  // interleaved [row][net][u0p] so that the dgrad into xcat is one contraction over both nets.
  const int H = p.u[p.nl - 1];
  const int ldh = ru4(H);
  auto dz_ld = [&](int l) { return l == 0 ? 2 * p.u0p : ru4(p.u[l]); };
  auto dz_stride = [&](int l) { return l == 0 ? (long long)p.u0p : mbs * ru4(p.u[l]); };
  if (do0) {
    LossArgs a;
    a.h = wsp<float>(st, p.w_h[p.nl - 1]);
    a.dh = wsp<float>(st, p.w_dh[p.nl - 1]);
    a.net_stride = mbs * ldh; a.ldh = ldh; a.H = H;
    a.ld_dh = dz_ld(p.nl - 1); a.net_stride_dh = dz_stride(p.nl - 1);
    a.Wmu = P + p.o_muW; a.bmu = P + p.o_muB; a.Wv = P + p.o_valW; a.bv = P + p.o_valB;
    a.logstd = P + p.o_sigma;
    a.actions = ro->actions; a.neglogpacs = ro->neglogpacs;
    a.adv = st->advantages; a.values_n = st->values_n; a.returns_n = st->returns_n;
    a.mus_w = st->mus_w; a.sigmas_w = st->sigmas_w;
    a.perm = st->perm; a.start = (long long)mb_index * mb;
    a.mb = mb; a.N = p.N; a.T = p.T; a.act = p.act; a.rows_per_wave = p.loss_rpw;
    a.e_clip = c->e_clip; a.critic_coef = c->critic_coef; a.entropy_coef = c->entropy_coef;
    a.bounds_coef = c->bounds_loss_coef;
    a.loss_part = wsp<double>(st, p.w_loss_part);
    a.head_slab = wsp<float>(st, p.w_head_slab);
    a.head_count = p.head_count;
    if (p.loss_fused) {
      // last trunk layer (both nets) with the heads, the loss and the head backward in its tiles' epilogue
      const int l = p.nl - 1, in = ac_in(p, l);
      GemmArgs g;
      g.A = wsp<float>(st, p.w_h[l - 1]); g.lda = ru4(in); g.sA = mbs * ru4(in);
      g.B = P + p.o_acW[l]; g.ldb = in; g.sB = p.ac_block; g.K = in;
      g.bias = P + p.o_acB[l]; g.sBias = p.ac_block;
      g.M = mb; g.N = H; g.nbatch = 2;
      g.epilogue = EPI_BIAS_TANH;
      if (!dma_eligible(g, true, true) || !aligned16(g.bias) || (g.sBias & 3) || !aligned16(a.dh) || (a.ld_dh & 3) ||
          (a.net_stride_dh & 3))
        return IGI_E_UNSUPPORTED;
      const int m_tiles = (mb + TrunkLossHook::TILE_M - 1) / TrunkLossHook::TILE_M;
      dma_set_divs(g, 1, m_tiles);
      constexpr size_t ring = sizeof(float) * 2 * (TrunkLossHook::TILE_M + 128) * DMA_BK;
      constexpr size_t shm = sizeof(float) * TrunkLossHook::LDS_FLOATS > ring ? sizeof(float) * TrunkLossHook::LDS_FLOATS : ring;
      static bool attr = false;
      if (!attr) {
        IGI_HIP_TRY(hipFuncSetAttribute((const void*)k_trunk_loss, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr = true;
      }
      ProfScope ps(PC_TRUNK_LOSS, s, 2.0 * 2 * (double)mbs * H * in + 2.0 * 3 * (double)mbs * H * (p.act + 1),
                   4.0 * (2.0 * mbs * in + 2.0 * H * in + 2.0 * mbs * H + (double)mbs * (4 * p.act + 6)));
      IGI_LAUNCH(k_trunk_loss, dim3(2 * m_tiles), dim3(DMA_THREADS), shm, s, g, a, m_tiles);
    } else {
    ProfScope ps(PC_LOSS, s, 2.0 * 3 * (double)mbs * H * (p.act + 1),
                 4.0 * (double)mbs * (4.0 * ldh + 4 * p.act + 6));
    const size_t shm = sizeof(float) * 4 * p.head_count;
    const int maxj = (H + 63) / 64;
    static int packed = -1;
    if (packed < 0) { const char* e = getenv("IGI_LOSS_PACKED"); packed = e ? atoi(e) : 1; }
    if (packed && p.act <= 7) {   // scalar section once per four rows
      if (maxj <= 1) IGI_LAUNCH(k_loss_packed<1>, dim3(p.loss_blocks), dim3(LOSS_THREADS), shm, s, a);
      else if (maxj == 2) IGI_LAUNCH(k_loss_packed<2>, dim3(p.loss_blocks), dim3(LOSS_THREADS), shm, s, a);
      else IGI_LAUNCH(k_loss_packed<4>, dim3(p.loss_blocks), dim3(LOSS_THREADS), shm, s, a);
    } else if (maxj <= 1) IGI_LAUNCH(k_loss<1>, dim3(p.loss_blocks), dim3(LOSS_THREADS), shm, s, a);
    else if (maxj == 2) IGI_LAUNCH(k_loss<2>, dim3(p.loss_blocks), dim3(LOSS_THREADS), shm, s, a);
    else IGI_LAUNCH(k_loss<4>, dim3(p.loss_blocks), dim3(LOSS_THREADS), shm, s, a);
    }
  }

  // ---- backward through the actor / critic trunk.  Level fusion: the weight gradient of layer l and the data
  //      gradient INTO layer l-1 both consume dZ_l and are independent of each other, so they share one grid
  //      (gemm_level -> gemm_dma_wgrad_multi_kernel: the data-gradient tiles lead, the weight-gradient workgroups fill
  //      their fill / drain bubbles).  The first trunk layer's weight gradient rides with the env_mlp data gradient.
  float* slab = wsp<float>(st, p.w_slab);
  GemmArgs wgrads[2 * IGI_MAX_LAYERS];
  int n_wgrads = 0;
  for (int l = p.nl - 1; l >= 0; --l) {
    if (l > 0 && !do0) continue;
    const bool do_wgrad = l > 0 ? do0 : do1;   // the first layer's weight gradient belongs to phase 1 (env level's grid)
    const int out = p.u[l];
    const int in = (l == 0) ? p.xld : ac_in(p, l);  // layer 0 sees the zero-padded xcat
    const float* dz = wsp<float>(st, p.w_dh[l]);
    const int ldz = dz_ld(l);
    const long long sZ = dz_stride(l);
    const float* x = (l == 0) ? xcat : wsp<float>(st, p.w_h[l - 1]);
    const int ldx = (l == 0) ? p.xld : ru4(p.u[l - 1]);
    const long long sX = (l == 0) ? 0 : mbs * ldx;
    if (p.rb_ac[l]) {
      // this level as one persistent row-block kernel: dZ_l and h_{l-1} cross the chip once, the weight gradient leaves
      // as rb_ac[l] partials per net (rowblock.h)
      RbLevelArgs r;
      r.dZ = dz; r.ldz = ldz; r.sZ = sZ;
      r.W = P + p.o_acW[l]; r.ldw = in; r.sW = p.ac_block;
      r.X = x; r.ldx = ldx; r.sX = sX;
      r.dX = wsp<float>(st, p.w_dh[l - 1]); r.lddx = dz_ld(l - 1); r.sdX = dz_stride(l - 1);
      r.dWp = slab + p.s_acW[l]; r.ldwp = in; r.sWpart = 2LL * out * in; r.sWnet = (long long)out * in;
      r.dBp = slab + p.s_acB[l]; r.sBpart = 2LL * out; r.sBnet = out;
      r.rows = mb; r.IN = in; r.nets = 2; r.ranges = p.rb_ac[l];
      const hipError_t e = rb_level_backward(r, s, PC_RB_TRUNK);
      if (e == hipErrorNotSupported) return IGI_E_UNSUPPORTED;   // (alignment: the plan cannot see the caller's pointers)
      IGI_HIP_TRY(e);
      continue;
    }
    if (do_wgrad && !(l == 0 && p.lw_parts > 0 && latent_rowdot(p, st))) {  // wgrad: dW[out][in] = dZ^T X, bias = column sums of dZ
      GemmArgs g;
      g.A = dz; g.lda = ldz; g.sA = sZ;
      g.B = x; g.ldb = ldx; g.sB = sX;
      g.M = out; g.N = in; g.K = mb;
      g.C = slab + p.s_acW[l]; g.ldc = in; g.sC = (long long)out * in;
      g.Cbias = slab + p.s_acB[l]; g.sCbias = out;
      g.nbatch = 2; g.splitk = p.sk_ac[l];
      if (l == 0) g.flop_credit = (double)p.xw / p.xld;   // the zero-padded input columns carry no algorithmic work
      g.sCsplit = 2LL * out * in; g.sCbiasSplit = 2LL * out;
      wgrads[n_wgrads++] = g;
    }
    if (l > 0) {  // dgrad into the previous hidden layer, times tanh'
      // (l == 1 with row dots: the dZ1 tiles also emit their share of dZ1 . W1[:, latent columns])
      const GemmArgs g = trunk_dgrad_args(p, st, l, l == 1 && latent_rowdot(p, st));
      // this layer's weight gradient needs the same dZ: it shares the data gradient's launch (gemm_level)
      g_multi_level = (p.nl == 3) ? (l == 2 ? 0 : 1) : 4;
      IGI_HIP_TRY(gemm_level(g, wgrads, n_wgrads, s));
      n_wgrads = 0;
    } else if (do1) {
      // d(xcat) = [dZ1_actor | dZ1_critic] . [W1a ; W1c] (one contraction, K = 2*u0p), times tanh'
      // of xcat: columns obs..obs+latent-1 are d(pre-activation) of the last env_mlp layer; the
      // other columns (observations, padding) are never read.
      const int K2 = 2 * p.u0p;
      if (p.lat_fused && p.latz && latent_rowdot(p, st)) {
        // (LATZ: the env level's row-block kernel forms dZ of the second env layer from the row dots itself)
      } else if (p.lat_fused) {
        const int H2 = p.pu[p.npl - 2];
        const bool parts_path = latent_rowdot(p, st);   // the K2-wide contraction then happened in the dZ1 tiles
        ProfScope ps(PC_LATENT_BWD, s, (parts_path ? 0.0 : 2.0 * mbs * K2 * 8) + 6.0 * mbs * 8 * H2,
                     4.0 * mbs * ((parts_path ? 8.0 * p.lat_tiles : (double)K2) + 3 * H2));
        const int maxj = (H2 + 63) / 64;
        const int K2p = (K2 + LATB_CH - 1) / LATB_CH * LATB_CH;
        size_t tile_f = (size_t)LATB_ROWS * LATB_LD;
        if (tile_f < (size_t)4 * (8 * H2 + 8)) tile_f = (size_t)4 * (8 * H2 + 8);
        const size_t shm = sizeof(float) * (8 * (size_t)(K2p + 4) + tile_f + LATB_ROWS * 8);
        float* part = wsp<float>(st, p.w_lat_part);
#define IGI_LATB(MJ_)                                                                                        \
  do {                                                                                                       \
    static bool attr = false;                                                                                \
    if (!attr) {                                                                                             \
      IGI_HIP_TRY(hipFuncSetAttribute((const void*)k_latent_bwd<MJ_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024));                                                          \
      attr = true;                                                                                           \
    }                                                                                                        \
    IGI_LAUNCH((k_latent_bwd<MJ_>), dim3(p.lat_blocks), dim3(256), shm, s, dz, ldz, K2,               \
                       wsp<float>(st, p.w_wlat), p.xld,                                                      \
                       p.obs, xcat, dxcat, wsp<float>(st, p.w_e[p.npl - 2]), ru4(H2), H2,                    \
                       P + p.o_envW[p.npl - 1], wsp<float>(st, p.w_de[p.npl - 2]), part, mb);                \
  } while (0)
        if (parts_path) {
          const float* parts = wsp<float>(st, p.w_lat_rowdot);
#define IGI_LATP(MJ_)                                                                                        \
  do {                                                                                                       \
    static bool attr = false;                                                                                \
    if (!attr) {                                                                                             \
      IGI_HIP_TRY(hipFuncSetAttribute((const void*)k_latent_bwd<MJ_, true>,                                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));              \
      attr = true;                                                                                           \
    }                                                                                                        \
    IGI_LAUNCH((k_latent_bwd<MJ_, true>), dim3(p.lat_blocks), dim3(256), shm, s, parts, p.lat_tiles, K2,     \
                       wsp<float>(st, p.w_wlat), p.xld,                                                      \
                       p.obs, xcat, dxcat, wsp<float>(st, p.w_e[p.npl - 2]), ru4(H2), H2,                    \
                       P + p.o_envW[p.npl - 1], wsp<float>(st, p.w_de[p.npl - 2]), part, mb);                \
  } while (0)
          if (maxj <= 2) IGI_LATP(2); else IGI_LATP(4);
#undef IGI_LATP
        } else if (maxj <= 2) IGI_LATB(2); else IGI_LATB(4);
#undef IGI_LATB
      } else if (p.latent == 8 && K2 % 256 == 0 && K2 <= 1024) {
        ProfScope ps(PC_OTHER, s, 2.0 * mbs * K2 * 8, 4.0 * mbs * K2);
        const int rpw = 8;
        const int nb = (mb + 4 * rpw - 1) / (4 * rpw);
        const int kq = K2 / 256;
#define IGI_LAT(KQ_) IGI_LAUNCH((k_latent_dgrad<KQ_, 8>), dim3(nb), dim3(256), 0, s, dz, ldz, w1p, p.xld, \
                                        p.obs, xcat, dxcat, mb, rpw)
        if (kq == 1) IGI_LAT(1); else if (kq == 2) IGI_LAT(2); else if (kq == 3) IGI_LAT(3); else IGI_LAT(4);
#undef IGI_LAT
      } else {
        GemmArgs g;
        g.A = dz; g.lda = ldz;
        g.B = w1p; g.ldb = p.xld;
        g.M = mb; g.N = p.xld; g.K = K2;
        g.C = dxcat; g.ldc = p.xld;
        g.aux = xcat; g.ldaux = p.xld;
        g.epilogue = EPI_TANHGRAD;
        IGI_HIP_TRY(gemm(g, true, false, s));
      }
    }
  }
  // ---- backward through env_mlp (its last layer is already done when k_latent_bwd ran)
  for (int l = p.npl - 1 - p.lat_fused; l >= 0 && do1; --l) {
    const int out = p.pu[l], in = env_in(p, l);
    const bool last = (l == p.npl - 1);
    const float* dz = last ? dxcat + p.obs : wsp<float>(st, p.w_de[l]);
    const int ldz = last ? p.xld : ru4(out);
    const float* x = (l == 0) ? priv_g : wsp<float>(st, p.w_e[l - 1]);
    const int ldx = (l == 0) ? pld : ru4(p.pu[l - 1]);
    if (l == 0 && p.lx_env) continue;   // done by the level above (LOWX)
    if (p.rb_env[l]) {
      RbLevelArgs r;
      r.dZ = dz; r.ldz = ldz;
      r.W = P + p.o_envW[l]; r.ldw = in;
      r.X = x; r.ldx = ldx;
      r.dX = wsp<float>(st, p.w_de[l - 1]); r.lddx = ru4(in);
      r.dWp = slab + p.s_envW[l]; r.ldwp = in; r.sWpart = (long long)out * in;
      r.dBp = slab + p.s_envB[l]; r.sBpart = out;
      r.rows = mb; r.IN = in; r.nets = 1; r.ranges = p.rb_env[l];
      if (p.lx_env && l == 1) {   // + the first env layer's weight / bias gradient; d(pre-activation) of that layer is not written
        r.lx_X = priv_g; r.lx_ld = pld;
        r.lx_W = slab + p.s_envW[0]; r.lx_ldw = env_in(p, 0); r.lx_sPart = (long long)p.pu[0] * env_in(p, 0);
        r.lx_B = slab + p.s_envB[0]; r.lx_bsPart = p.pu[0];
        if (p.latz && latent_rowdot(p, st)) {
          // dZ of this layer is formed in the kernel from the latent row dots: it stages the layer's OUTPUT activations instead
          r.dZ = wsp<float>(st, p.w_e[1]); r.ldz = ru4(p.pu[1]);
          r.lz_parts = wsp<float>(st, p.w_lat_rowdot); r.lz_tiles = p.lat_tiles; r.lz_tstride = (long long)mb * 8;
          r.lz_lat = xcat + p.obs; r.lz_ldlat = p.xld;
          r.lz_W3 = P + p.o_envW[p.npl - 1];
          r.lz_rec = wsp<float>(st, p.w_lat_part); r.lz_srec = 8 * p.pu[1] + 8;
        }
      }
      const hipError_t e = rb_level_backward(r, s, PC_RB_ENV);
      if (e == hipErrorNotSupported) return IGI_E_UNSUPPORTED;
      IGI_HIP_TRY(e);
      continue;      // (weight gradients still pending -- the first trunk layer's -- go out with the last launch below)
    }
    {
      GemmArgs g;
      g.A = dz; g.lda = ldz;
      g.B = x; g.ldb = ldx;
      g.M = out; g.N = in; g.K = mb;
      g.C = slab + p.s_envW[l]; g.ldc = in;
      g.Cbias = slab + p.s_envB[l];
      g.splitk = p.sk_env[l];
      g.sCsplit = (long long)out * in; g.sCbiasSplit = out;
      wgrads[n_wgrads++] = g;
    }
    if (l > 0) {
      GemmArgs g;
      g.A = dz; g.lda = ldz;
      g.B = P + p.o_envW[l]; g.ldb = in;
      g.M = mb; g.N = in; g.K = out;
      g.C = wsp<float>(st, p.w_de[l - 1]); g.ldc = ru4(in);
      g.aux = wsp<float>(st, p.w_e[l - 1]); g.ldaux = ru4(in);
      g.epilogue = EPI_TANHGRAD;
      g_multi_level = (p.npl == 3 && l == 1) ? 2 : 4;
      IGI_HIP_TRY(gemm_level(g, wgrads, n_wgrads, s));   // + this layer's (and the first trunk layer's) weight gradient
      n_wgrads = 0;
    }
  }

  g_multi_level = (p.npl == 3 && phase != 0) ? ((p.rb_env[1] && !(p.lw_parts > 0 && latent_rowdot(p, st))) ? 5 : 3) : 4;
  IGI_HIP_TRY(gemm_wgrad_group(wgrads, n_wgrads, s));

  // ---- assemble the flat gradient
  SegTable t;
  t.n = 0;
  auto add = [&](long long dst, const float* src, long long stride, int rows, int cols, int src_ld,
                 int nparts) {
    Segment& sg = t.s[t.n++];
    sg.dst = dst; sg.src = src; sg.stride = stride; sg.count = rows * cols; sg.cols = cols;
    sg.src_ld = src_ld; sg.nparts = nparts;
  };
  const float* hs = wsp<float>(st, p.w_head_slab);
  const int hc = p.head_count;
  if (do0) {
    add(p.o_muW, hs, hc, 1, p.act * H, 0, p.loss_blocks);
    add(p.o_muB, hs + p.act * H, hc, 1, p.act, 0, p.loss_blocks);
    add(p.o_valW, hs + p.act * H + p.act, hc, 1, H, 0, p.loss_blocks);
    add(p.o_valB, hs + p.act * H + p.act + H, hc, 1, 1, 0, p.loss_blocks);
  }
  if (do1) add(p.o_sigma, hs + p.act * H + p.act + H + 1, hc, 1, p.act, 0, p.loss_blocks);
  for (int l = 0; l < p.npl - p.lat_fused && do1; ++l) {
    const int out = p.pu[l], in = env_in(p, l);
    add(p.o_envW[l], slab + p.s_envW[l], (long long)out * in, 1, out * in, 0, p.sk_env[l]);
    add(p.o_envB[l], slab + p.s_envB[l], out, 1, out, 0, p.sk_env[l]);
  }
  if (p.lat_fused && do1) {
    const int H2 = p.pu[p.npl - 2], pc = 8 * H2 + 8;
    const float* part = wsp<float>(st, p.w_lat_part);
    const int nrec = (p.latz && latent_rowdot(p, st)) ? p.rb_env[1] : p.lat_blocks;   // LATZ: one record per row range
    add(p.o_envW[p.npl - 1], part, pc, 1, 8 * H2, 0, nrec);
    add(p.o_envB[p.npl - 1], part + 8 * H2, pc, 1, 8, 0, nrec);
  }
  for (int l = 0; l < p.nl; ++l) {
    if (!(l > 0 ? do0 : do1)) continue;
    const int out = p.u[l], in = ac_in(p, l);
    const int inw = (l == 0) ? p.xld : in;  // slab rows are inw wide; the parameter rows are `in` wide
    for (int net = 0; net < 2; ++net) {
      add(p.o_acW[l] + net * p.ac_block, slab + p.s_acW[l] + (long long)net * out * inw,
          2LL * out * inw, out, in, inw, p.sk_ac[l]);
      add(p.o_acB[l] + net * p.ac_block, slab + p.s_acB[l] + (long long)net * out, 2LL * out, 1, out, 0,
          p.sk_ac[l]);
    }
  }
  {
    ProfScope ps(PC_SLAB_REDUCE, s, 0.0, 4.0 * ((double)p.slab_floats + (double)hc * p.loss_blocks + p.P));
    static int gx = -1;
    // blocks per segment: 64 / 128 / 256 / 512 / 1024 / 2048 -> 29.5 / 19.2 / 15.4 / 14.3 / 15.0 / 17.9 us (IGI_SLAB_GX)
    // (with the row-block levels' fewer, larger partial sets: 256 -> 13.0 us, 384 -> 13.9, 512 -> 14.2)
    if (gx < 0) { const char* e = getenv("IGI_SLAB_GX"); gx = e ? atoi(e) : (rb_level_enabled() ? SLAB_GX : 2 * SLAB_GX); if (gx < 1) gx = 1; if (gx > SLAB_GX_MAX) gx = SLAB_GX_MAX; }
    int gy = t.n;
    if (norm_parts) {
      if (phase != -1 || !st->stats) return IGI_E_BADARG;
      t.norm_part = wsp<double>(st, p.w_gpart);
      t.loss_part = wsp<double>(st, p.w_loss_part); t.loss_blocks = p.loss_blocks; t.mb = p.mb;
      t.stats_row = st->stats + (long long)step_slot * IGI_STATS_PER_STEP;
      *norm_parts = gx * t.n;
      gy = t.n + 1;     // + the statistics row
    }
    if (norm_parts) IGI_LAUNCH(k_slab_reduce_norm, dim3(gx, gy), dim3(RED_THREADS), 0, s, t, st->grads);
    else IGI_LAUNCH(k_slab_reduce, dim3(gx, gy), dim3(RED_THREADS), 0, s, t, st->grads);
  }
  return (int)hipGetLastError();
}

// next_ro != NULL: fuse the gather + normalise of optimizer step (next_mb, next_slot) into this step's Adam launch
// norm_mode 0: k_sumsq_stats computes both norms and the statistics row (any caller, any grad_scale);
//           1: the same, and the Adam blocks leave the updated parameters' sum-of-squares partials for the next step;
//           2: no k_sumsq_stats -- the gradient norm comes from the norm_parts partials of this step's k_slab_reduce, the
//              parameter norm from the previous step's Adam partials (that step ran mode 1 or 2), the statistics row was
//              written by k_slab_reduce; the Adam blocks leave their partials again.  grad_scale must be 1.
static int teacher_apply(const igi_teacher_cfg* c, const igi_teacher_state* st, int step_slot,
                         int64_t adam_t, float grad_scale, hipStream_t s, const igi_rollout* next_ro = nullptr,
                         int next_mb = 0, int next_slot = 0, int norm_mode = 0, int norm_parts = 0) {
  TeacherPlan p;
  int rc = make_plan(c, &p);
  if (rc) return rc;
  if ((rc = check_state(p, st))) return rc;
  if (!st->grads || !st->adam_m || !st->adam_v || adam_t < 1) return IGI_E_BADARG;
  if (norm_mode < 0 || norm_mode > 2 || (norm_mode == 2 && (grad_scale != 1.0f || norm_parts < 1 || !st->stats))) return IGI_E_BADARG;
  double* part = wsp<double>(st, p.w_sumsq);
  float* row = st->stats ? st->stats + (long long)step_slot * IGI_STATS_PER_STEP : nullptr;
  if (norm_mode != 2) {
  ProfScope ps(PC_SUMSQ, s, 0.0, 8.0 * (double)p.P);
  IGI_LAUNCH(k_sumsq_stats, dim3(SUMSQ_BLOCKS + (row ? 1 : 0)), dim3(256), 0, s, st->grads,
                     st->params, p.P, grad_scale, part, wsp<double>(st, p.w_loss_part), p.loss_blocks,
                     p.mb, row);
  }
  // torch.optim.Adam (_single_tensor_adam): python-double scalars, cast to fp32 at the tensor op
  const double b1 = c->beta1, b2 = c->beta2;
  const double bc1 = 1.0 - pow(b1, (double)adam_t);
  const double bc2 = 1.0 - pow(b2, (double)adam_t);
  const float step_size = (float)((double)c->lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float w1 = (float)(1.0 - b1), w2 = (float)(1.0 - b2);
  int nb = (int)((p.P / 4 + 255) / 256);   // four elements per thread and trip
  if (nb > ADAM_BLOCKS_MAX) nb = ADAM_BLOCKS_MAX;
  if (nb < 1) nb = 1;
  NormSrc ns;
  if (norm_mode >= 1) ns.pp_out = wsp<double>(st, p.w_ppart) + (size_t)(step_slot & 1) * ADAM_BLOCKS_MAX;
  if (norm_mode == 2) {
    ns.gpart = wsp<double>(st, p.w_gpart); ns.n_g = norm_parts;
    ns.pp_in = wsp<double>(st, p.w_ppart) + (size_t)((step_slot + 1) & 1) * ADAM_BLOCKS_MAX; ns.n_p = nb;
  }
  if (next_ro) {
    if (!next_ro->obses || !next_ro->priv_info || !st->perm || !st->rms_obs || !st->rms_priv) return IGI_E_BADARG;
    const int D = p.obs + p.priv;
    ProfScope ps(PC_ADAM_GATHER, s, 0.0, 28.0 * (double)p.P + 8.0 * (double)p.mb * D + 8.0 * p.mb);
    AdamArgs aa;
    aa.params = st->params; aa.grads = st->grads; aa.m = st->adam_m; aa.v = st->adam_v; aa.P = p.P; aa.part = part;
    aa.scale = grad_scale; aa.max_norm = c->grad_norm; aa.w1 = w1; aa.beta2 = (float)b2; aa.w2 = w2;
    aa.step_size = step_size; aa.bc2_sqrt = bc2_sqrt; aa.eps = (float)c->adam_eps; aa.stats_row = row;
    const GatherArgs ga = gather_args(p, next_ro, st, next_mb, next_slot);
    W1Mirror mir;
    mir.w1p = ga.w1p; mir.wlat = ga.wlat; mir.o_w = ga.o_w; mir.ac_block = ga.ac_block; mir.u0 = ga.u0;
    mir.u0p = ga.u0p; mir.xw = p.xw; mir.xld = p.xld; mir.obs = p.obs; mir.K2p = ga.K2p;
    IGI_LAUNCH(k_adam_gather, dim3(nb + p.gs_blocks), dim3(256), 0, s, aa, mir, ga, nb, ns);
    return (int)hipGetLastError();
  }
  ProfScope ps(PC_ADAM, s, 0.0, 28.0 * (double)p.P);  // 16 B read + 12 B written per parameter
  IGI_LAUNCH(k_clip_adam, dim3(nb), dim3(256), 0, s, st->params, st->grads, st->adam_m,
                     st->adam_v, p.P, part, grad_scale, c->grad_norm, w1, (float)b2, w2, step_size,
                     bc2_sqrt, (float)c->adam_eps, row, 1.0f, 0.0f, ns);
  return (int)hipGetLastError();
}

// igi_teacher_set_norm_fusion / IGI_NORM_FUSE (initial value).  Default OFF: measured slower (profiles/r06_norm_fuse_ab.log,
// A/B on one box, two rounds: 25.74 / 25.74 ms per update off, 26.05 / 25.90 on) -- the 4.2 us kernel it removes comes back
// as +1.4 us in k_slab_reduce (a block reduction in each of its ~6,700 blocks), +1.0 us in the Adam blocks (every block
// re-adds the partials) and ~1 % lower clocks in the matrix kernels around it (the chip is power-limited: a short
// bandwidth-bound kernel between two MFMA-dense ones is not dead time for the clocks).
static inline int& norm_fusion_ref() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("IGI_NORM_FUSE"); on = e ? (atoi(e) != 0) : 0; }
  return on;
}

static int teacher_update(const igi_teacher_cfg* c, const igi_rollout* ro,
                          const igi_teacher_state* st, int64_t adam_t0, hipStream_t s) {
  TeacherPlan p;
  int rc = make_plan(c, &p);
  if (rc) return rc;
  static int fuse_tail = -1;
  if (fuse_tail < 0) { const char* e = getenv("IGI_FUSE_TAIL"); fuse_tail = e ? atoi(e) : 1; }
  // igi_teacher_set_norm_fusion(0) / IGI_NORM_FUSE=0: k_sumsq_stats on every step (A/B; default off, see norm_fusion_ref).  With it on, from the second step of the update on, the two norms
  // and the statistics row come from k_slab_reduce's and the previous Adam pass's partials (one launch less per step);
  // the FIRST step keeps k_sumsq_stats -- the parameters may have been replaced since the last update's partials were left
  const bool nf = norm_fusion_ref() && st->stats;
  const int total = p.E * p.nmb;
  int slot = 0;
  for (int e = 0; e < p.E; ++e) {
    for (int i = 0; i < p.nmb; ++i, ++slot) {
      // from the second step on the minibatch was gathered by the previous step's fused tail
      int nparts = 0;
      const bool fused = nf && slot > 0;
      if ((rc = teacher_fwd_bwd(c, ro, st, i, slot, s, -1, fuse_tail && slot > 0, fused ? &nparts : nullptr))) return rc;
      const bool more = fuse_tail && slot + 1 < total;
      if ((rc = teacher_apply(c, st, slot, adam_t0 + slot + 1, 1.0f, s, more ? ro : nullptr, (slot + 1) % p.nmb,
                              slot + 1, nf ? (fused ? 2 : 1) : 0, nparts)))
        return rc;
    }
  }
  return 0;
}

// data-parallel update: the same 64 steps with the two-bucket gradient exchange driven through a callback
// (frozen_ppo.py:586-603); one host call per update, the callback only enqueues collectives / stream waits
static int teacher_update_dp(const igi_teacher_cfg* c, const igi_rollout* ro, const igi_teacher_state* st,
                             int64_t adam_t0, float grad_scale, igi_reduce_fn reduce, void* user, hipStream_t s) {
  TeacherPlan p;
  int rc = make_plan(c, &p);
  if (rc) return rc;
  const int total = p.E * p.nmb;
  int slot = 0;
  for (int e = 0; e < p.E; ++e) {
    for (int i = 0; i < p.nmb; ++i, ++slot) {
      if ((rc = teacher_fwd_bwd(c, ro, st, i, slot, s, 0, slot > 0))) return rc;
      if (reduce(user, 0, slot)) return IGI_E_CALLBACK;
      if ((rc = teacher_fwd_bwd(c, ro, st, i, slot, s, 1))) return rc;
      if (reduce(user, 1, slot)) return IGI_E_CALLBACK;
      if (reduce(user, 2, slot)) return IGI_E_CALLBACK;
      const bool more = slot + 1 < total;
      if ((rc = teacher_apply(c, st, slot, adam_t0 + slot + 1, grad_scale, s, more ? ro : nullptr,
                              (slot + 1) % p.nmb, slot + 1)))
        return rc;
    }
  }
  return 0;
}

static int teacher_infer(const igi_teacher_cfg* c, const igi_teacher_state* st, const float* obs,
                         const float* priv, int64_t rows, int normalize, float* mu, float* value,
                         float* latent, hipStream_t s) {
  TeacherPlan p;
  int rc = make_plan(c, &p);
  if (rc) return rc;
  if ((rc = check_state(p, st))) return rc;
  if (!obs || !priv || rows < 0 || (normalize && (!st->rms_obs || !st->rms_priv))) return IGI_E_BADARG;
  float* priv_g = wsp<float>(st, p.w_priv);
  float* xcat = wsp<float>(st, p.w_xcat);
  float* ncoef = wsp<float>(st, p.w_norm_coef);
  const int pld = ru4(p.priv);
  const int D = p.obs + p.priv;
  const int H = p.u[p.nl - 1];
  const int ldh = ru4(H);
  if (normalize)
    IGI_LAUNCH(k_rms_coef, dim3(1), dim3(128), 0, s, p.obs, p.priv, st->rms_obs, st->rms_priv,
                       c->rms_eps, ncoef);
  for (int64_t r0 = 0; r0 < rows; r0 += p.mb) {
    const int nr = (int)((rows - r0 < p.mb) ? rows - r0 : p.mb);
    long long tot = (long long)nr * D;
    int nb = (int)((tot + 255) / 256);
    if (nb > 2048) nb = 2048;
    IGI_LAUNCH(k_copy_rows, dim3(nb), dim3(256), 0, s, obs + r0 * p.obs, priv + r0 * p.priv, nr,
                       p.obs, p.priv, xcat, p.xld, priv_g, pld);
    if (normalize)
      IGI_LAUNCH(k_normalize, dim3(nb), dim3(256), 0, s, xcat, p.xld, p.xw, priv_g, pld, nr, p.obs,
                         p.priv, ncoef);
    if ((rc = trunk_forward(p, st, nr, true, s))) return rc;
    if (latent)
      IGI_HIP_TRY(hipMemcpy2DAsync(latent + r0 * p.latent, sizeof(float) * p.latent, xcat + p.obs,
                                   sizeof(float) * p.xld, sizeof(float) * p.latent, nr,
                                   hipMemcpyDeviceToDevice, s));
    if (mu || value) {
      const float* P = st->params;
      int hb = (nr + 3) / 4;
      if (hb > 1024) hb = 1024;
      const float* h = wsp<float>(st, p.w_h[p.nl - 1]);
      const long long ns = (long long)p.mb * ldh;
      float* mo = mu ? mu + r0 * p.act : nullptr;
      float* vo = value ? value + r0 : nullptr;
      const int maxj = (H + 63) / 64;
      if (maxj <= 1)
        IGI_LAUNCH(k_heads_infer<1>, dim3(hb), dim3(256), 0, s, h, ns, ldh, H, P + p.o_muW,
                           P + p.o_muB, P + p.o_valW, P + p.o_valB, nr, p.act, mo, vo);
      else if (maxj == 2)
        IGI_LAUNCH(k_heads_infer<2>, dim3(hb), dim3(256), 0, s, h, ns, ldh, H, P + p.o_muW,
                           P + p.o_muB, P + p.o_valW, P + p.o_valB, nr, p.act, mo, vo);
      else
        IGI_LAUNCH(k_heads_infer<4>, dim3(hb), dim3(256), 0, s, h, ns, ldh, H, P + p.o_muW,
                           P + p.o_muB, P + p.o_valW, P + p.o_valB, nr, p.act, mo, vo);
    }
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Rollout-side policy step (frozen_ppo.py:343-366 + 655-665): ONE host call and 7-8 launches per environment step
// instead of igi_teacher_infer (10 launches) + a torch randn + igi_rollout_act_store:
//   k_policy_stage   : raw obs / priv -> arena slot t (raw copies), normalised xcat / priv_g with the CURRENT running
//                      statistics (eval mode), zero padding, refresh of the padded first-layer weight
//   trunk_forward    : env_mlp + actor / critic trunk (the launches of the training forward)
//   k_heads_act_store: mu / value heads, action = mu + sigma * noise, neglogp, value de-normalisation, arena writes,
//                      clamp(action, +-1) for env.step
// Same arithmetic, in the same order, as the separate kernels (k_rms_coef + k_copy_rows + k_normalize, k_heads_infer,
// k_rollout_act_store): bit-identical outputs.
// ---------------------------------------------------------------------------------------------
struct PolicyStageArgs {
  const float* obs; const float* priv; int rows, obs_dim, priv_dim;
  const double* rms_obs; const double* rms_priv; float eps; int normalize;
  float* xcat; int xld, xw; float* priv_g; int pld;
  float* obses_t; float* priv_t;            // arena slot (raw copies) or NULL
  const float* params; long long o_w, ac_block; int u0, u0p; float* w1p;
  int stage_blocks;
};

__global__ __launch_bounds__(256) void k_policy_stage(const PolicyStageArgs a) {
  if ((int)blockIdx.x >= a.stage_blocks) {  // W1p[net][o][c] refresh (see k_pad_w1)
    const int total = 2 * a.u0p * a.xld;
    const int nb = (int)gridDim.x - a.stage_blocks;
    for (int e = ((int)blockIdx.x - a.stage_blocks) * blockDim.x + threadIdx.x; e < total; e += nb * blockDim.x) {
      const int c = e % a.xld;
      const int o = (e / a.xld) % a.u0p;
      const int net = e / (a.xld * a.u0p);
      a.w1p[e] = (c < a.xw && o < a.u0) ? a.params[a.o_w + net * a.ac_block + (long long)o * a.xw + c] : 0.f;
    }
    return;
  }
  const int D = a.obs_dim + a.priv_dim;
  const long long total = (long long)a.rows * D;
  const long long stride = (long long)a.stage_blocks * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const long long r = e / D;
    const int c = (int)(e - r * D);
    const bool is_obs = c < a.obs_dim;
    const int cc = is_obs ? c : c - a.obs_dim;
    const float x = is_obs ? a.obs[r * a.obs_dim + cc] : a.priv[r * a.priv_dim + cc];
    if (is_obs) { if (a.obses_t) a.obses_t[r * a.obs_dim + cc] = x; }
    else if (a.priv_t) a.priv_t[r * a.priv_dim + cc] = x;
    float y = x;
    if (a.normalize) {  // k_rms_coef + k_normalize: fp32 (mean, sqrt(var + eps)) from the fp64 running state
      const double* stt = is_obs ? a.rms_obs : a.rms_priv;
      const int d = is_obs ? a.obs_dim : a.priv_dim;
      const float m = (float)stt[cc], den = sqrtf((float)stt[d + cc] + a.eps);
      y = clamp5((x - m) / den);
    }
    if (is_obs) a.xcat[r * a.xld + cc] = y;
    else a.priv_g[r * a.pld + cc] = y;
  }
  const int padw = a.xld - a.xw;   // keep the zero padding of xcat zero (feeds the padded first layer)
  const long long ptotal = (long long)a.rows * padw;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < ptotal; e += stride) {
    const long long r = e / padw;
    a.xcat[r * a.xld + a.xw + (int)(e - r * padw)] = 0.f;
  }
}

struct ActStoreArgs {
  const float* logstd; const float* noise; const double* rms_value; float eps;
  float* actions_t; float* nlp_t; float* values_t; float* mus_t; float* sigmas_t; float* actions_clamped;
  float* values_out;
};

// one wave per row: the heads exactly as k_heads_infer computes them, then lane q owns action q
template <int MAXJ>
__global__ __launch_bounds__(256) void k_heads_act_store(const float* __restrict__ h, long long net_stride, int ldh,
                                                         int H, const float* __restrict__ Wmu,
                                                         const float* __restrict__ bmu, const float* __restrict__ Wv,
                                                         const float* __restrict__ bv, int rows, int act,
                                                         const ActStoreArgs a) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  float vm = 0.f, vd = 1.f;
  if (a.rms_value) { vm = (float)a.rms_value[0]; vd = sqrtf((float)a.rms_value[1] + a.eps); }
  const float my_logstd = lane < act ? a.logstd[lane] : 0.f;
  const float my_bmu = lane < act ? bmu[lane] : 0.f;
  for (int row = gw; row < rows; row += nw) {
    const float* ha_p = h + (long long)row * ldh;
    const float* hc_p = ha_p + net_stride;
    float pm[IGI_MAX_ACT], pv = 0.f;
#pragma unroll
    for (int q = 0; q < IGI_MAX_ACT; ++q) pm[q] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int k = lane + 64 * j;
      if (k < H) {
        const float ha = ha_p[k], hc = hc_p[k];
        pv += hc * Wv[k];
#pragma unroll
        for (int q = 0; q < IGI_MAX_ACT; ++q)
          if (q < act) pm[q] += ha * Wmu[q * H + k];
      }
    }
    pv = wave_sum(pv);
    float my_pm = 0.f;
#pragma unroll
    for (int q = 0; q < IGI_MAX_ACT; ++q)
      if (q < act) { const float t = wave_sum(pm[q]); my_pm = (lane == q) ? t : my_pm; }
    // per-action arithmetic of k_rollout_act_store on lane q
    const float m = my_pm + my_bmu;
    const float sig = expf(m * 0.f + my_logstd);
    const long long ia = (long long)row * act + lane;
    const float nz = lane < act ? a.noise[ia] : 0.f;
    const float av = m + sig * nz;
    const float x = av - m;
    const float term = ((x * x) / (2.0f * (sig * sig)) + logf(sig)) + ROLL_LOG_SQRT_2PI;
    if (lane < act) {
      a.actions_t[ia] = av;
      a.mus_t[ia] = m;
      a.sigmas_t[ia] = sig;
      a.actions_clamped[ia] = fminf(fmaxf(av, -1.0f), 1.0f);
    }
    float nlp = 0.f;   // summed in action order, as the one-thread-per-env kernel does
    for (int q = 0; q < act; ++q) nlp += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), q));
    if (lane == 0) {
      float v = pv + bv[0];
      if (a.rms_value) v = vd * fminf(fmaxf(v, -5.0f), 5.0f) + vm;
      a.nlp_t[row] = nlp;
      a.values_t[row] = v;
      a.values_out[row] = v;
    }
  }
}

static int teacher_policy_step(const igi_teacher_cfg* c, const igi_teacher_state* st, const float* obs,
                               const float* priv, int64_t rows, int normalize, const float* noise,
                               const double* rms_value, float* obses_t, float* priv_t, float* actions_t, float* nlp_t,
                               float* values_t, float* mus_t, float* sigmas_t, float* actions_clamped,
                               float* values_out, hipStream_t s) {
  TeacherPlan p;
  int rc = make_plan(c, &p);
  if (rc) return rc;
  if ((rc = check_state(p, st))) return rc;
  if (!obs || !priv || rows < 1 || !noise || !actions_t || !nlp_t || !values_t || !mus_t || !sigmas_t ||
      !actions_clamped || !values_out || (normalize && (!st->rms_obs || !st->rms_priv)))
    return IGI_E_BADARG;
  if (p.act > 64) return IGI_E_UNSUPPORTED;
  const float* P = st->params;
  const int H = p.u[p.nl - 1];
  const int ldh = ru4(H);
  const int D = p.obs + p.priv;
  for (int64_t r0 = 0; r0 < rows; r0 += p.mb) {
    const int nr = (int)((rows - r0 < p.mb) ? rows - r0 : p.mb);
    PolicyStageArgs a;
    a.obs = obs + r0 * p.obs; a.priv = priv + r0 * p.priv; a.rows = nr; a.obs_dim = p.obs; a.priv_dim = p.priv;
    a.rms_obs = st->rms_obs; a.rms_priv = st->rms_priv; a.eps = c->rms_eps; a.normalize = normalize;
    a.xcat = wsp<float>(st, p.w_xcat); a.xld = p.xld; a.xw = p.xw; a.priv_g = wsp<float>(st, p.w_priv);
    a.pld = ru4(p.priv);
    a.obses_t = obses_t ? obses_t + r0 * p.obs : nullptr; a.priv_t = priv_t ? priv_t + r0 * p.priv : nullptr;
    a.params = P; a.o_w = p.o_acW[0]; a.ac_block = p.ac_block; a.u0 = p.u[0]; a.u0p = p.u0p;
    a.w1p = wsp<float>(st, p.w_w1p);
    long long tot = (long long)nr * D;
    int nb = (int)((tot + 255) / 256);
    if (nb > 2048) nb = 2048;
    a.stage_blocks = nb;
    const int pad_blocks = 16;
    {
      ProfScope ps(PC_OTHER, s, 0.0, 8.0 * tot);
      IGI_LAUNCH(k_policy_stage, dim3(nb + pad_blocks), dim3(256), 0, s, a);
    }
    if (policy_fwd_enabled() && !bf16_mode() && policy_fwd_shape_ok(p.obs, p.priv, p.act, p.npl, p.pu, p.nl, p.u) && p.xld == 32) {
      // env_mlp, both trunks, the heads, the sample and the arena writes of these rows as ONE persistent launch (policy_fwd.h)
      PolicyFwdArgs f;
      f.priv = a.priv_g; f.ldp = a.pld; f.xcat = a.xcat; f.ldx = p.xld; f.rows = nr; f.obs = p.obs; f.act = p.act;
      f.eW1 = P + p.o_envW[0]; f.eb1 = P + p.o_envB[0]; f.eW2 = P + p.o_envW[1]; f.eb2 = P + p.o_envB[1];
      f.eW3 = P + p.o_envW[2]; f.eb3 = P + p.o_envB[2];
      f.w1p = a.w1p; f.tb1 = P + p.o_acB[0]; f.tW2 = P + p.o_acW[1]; f.tb2 = P + p.o_acB[1]; f.tW3 = P + p.o_acW[2];
      f.tb3 = P + p.o_acB[2]; f.ac_block = p.ac_block;
      f.Wmu = P + p.o_muW; f.bmu = P + p.o_muB; f.Wv = P + p.o_valW; f.bv = P + p.o_valB; f.logstd = P + p.o_sigma;
      f.noise = noise + r0 * p.act; f.rms_value = rms_value; f.eps = c->rms_eps;
      f.actions_t = actions_t + r0 * p.act; f.nlp_t = nlp_t + r0; f.values_t = values_t + r0;
      f.mus_t = mus_t + r0 * p.act; f.sigmas_t = sigmas_t + r0 * p.act;
      f.actions_clamped = actions_clamped + r0 * p.act; f.values_out = values_out + r0;
      const hipError_t e = policy_forward(f, s);
      if (e == hipSuccess) continue;
      if (e != hipErrorInvalidValue) return (int)e;      // (alignment of the caller's buffers: the per-layer launches below)
    }
    if ((rc = trunk_forward(p, st, nr, false, s))) return rc;
    ActStoreArgs t;
    t.logstd = P + p.o_sigma; t.noise = noise + r0 * p.act; t.rms_value = rms_value; t.eps = c->rms_eps;
    t.actions_t = actions_t + r0 * p.act; t.nlp_t = nlp_t + r0; t.values_t = values_t + r0;
    t.mus_t = mus_t + r0 * p.act; t.sigmas_t = sigmas_t + r0 * p.act;
    t.actions_clamped = actions_clamped + r0 * p.act; t.values_out = values_out + r0;
    int hb = (nr + 3) / 4;
    if (hb > 1024) hb = 1024;
    const float* h = wsp<float>(st, p.w_h[p.nl - 1]);
    const long long ns = (long long)p.mb * ldh;
    const int maxj = (H + 63) / 64;
    ProfScope ps(PC_OTHER, s, 0.0, 8.0 * nr * H);
    if (maxj <= 1)
      IGI_LAUNCH(k_heads_act_store<1>, dim3(hb), dim3(256), 0, s, h, ns, ldh, H, P + p.o_muW, P + p.o_muB,
                 P + p.o_valW, P + p.o_valB, nr, p.act, t);
    else if (maxj == 2)
      IGI_LAUNCH(k_heads_act_store<2>, dim3(hb), dim3(256), 0, s, h, ns, ldh, H, P + p.o_muW, P + p.o_muB,
                 P + p.o_valW, P + p.o_valB, nr, p.act, t);
    else
      IGI_LAUNCH(k_heads_act_store<4>, dim3(hb), dim3(256), 0, s, h, ns, ldh, H, P + p.o_muW, P + p.o_muB,
                 P + p.o_valW, P + p.o_valB, nr, p.act, t);
  }
  return (int)hipGetLastError();
}

}  // namespace igi
