/*
 * igi_ppo.h -- C ABI of libigi_hip.so: the MI355X (gfx950) learning-side hot path of
 * osheraz/IsaacGymInsertion (teacher PPO update).
 *
 * The reference has no FFI for this path: the boundary it exposes is the Python class API
 * (algo/ppo/frozen_ppo.py::PPO, algo/ppo/experience.py::ExperienceBuffer,
 * algo/models/models_split.py::ActorCriticSplit, algo/models/running_mean_std.py::RunningMeanStd).
 * Each entry point below replaces the chain of ATen ops behind one of those methods; the
 * citation on each declaration names the reference lines it replaces (paths relative to the
 * reference root).  isaacgyminsertion_amd/ binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - no entry point allocates, frees, retains or synchronises: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream), memory is owned by the caller;
 *   - return value: 0 on success, a positive hipError_t, or a negative IGI_E_* code;
 *     igi_last_error() returns a static string describing the last failure on this thread;
 *   - all arithmetic is fp32 (exact-fp32 MFMA, v_mfma_f32_32x32x2_f32) except normaliser state
 *     and reductions feeding it, which are fp64 as in running_mean_std.py:44-46.
 */
#ifndef IGI_PPO_H
#define IGI_PPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGI_ABI_VERSION 3
#define IGI_MAX_LAYERS 4
#define IGI_MAX_ACT 8

#define IGI_E_BADARG (-1)
#define IGI_E_WORKSPACE (-2)
#define IGI_E_CALLBACK (-5)
#define IGI_E_UNSUPPORTED (-3)
#define IGI_E_COMM (-6)        /* an RCCL call failed: igi_comm_last_error() has the text */

typedef void* igi_stream_t;

int igi_abi_version(void);
/* sha256 prefix (32 hex digits) of the sources this library was compiled from (__graft_entry__.source_hash) */
const char* igi_build_info(void);
const char* igi_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Measurement hook (new; the reference only has wall-clock accumulators, frozen_ppo.py:272-274,
 * 500, 645).  After igi_prof_enable(1) every kernel launch of this library is bracketed by HIP
 * events on its own stream; igi_prof_read synchronises those events and returns one entry per
 * kernel class with the number of launches, their summed duration and the ALGORITHMIC flops /
 * bytes of those launches (SURVEY.md section 8d conventions).  igi_prof_enable(0|1) resets.
 * ---------------------------------------------------------------------------------------- */
typedef struct igi_prof_entry {
  const char* name;
  int64_t launches;
  double total_ms;
  double flops;
  double bytes;
} igi_prof_entry;
int igi_prof_enable(int on);
int igi_prof_read(igi_prof_entry* out_host, int max_entries);
/* The box's own fp32 matrix rate, measured ("re-measure on the box with a microbench and state
 * both"; the reference has no counterpart): one launch of `blocks` x 256 threads, each wave issuing iters x 16
 * MFMAs from registers only (no LDS, no memory) on non-trivial operands; shape 0 = v_mfma_f32_32x32x2_f32 (4096 flop
 * each: the instruction every fp32 product here runs on), 1 = v_mfma_f32_16x16x4_f32 (2048), 2 = v_mfma_f32_32x32x16_bf16
 * (32768) -> flops per launch = blocks * 4 * iters * 16 * flop; the caller times the launch.  clocks_dev (device, 2 * blocks uint64): per block the shader
 * cycles and the 100 MHz real-time ticks its first wave spent in the loop (clock = cycles / ticks * 100 MHz);
 * sink_dev: one device float that is never written in practice (keeps the loop alive). */
int igi_mfma_peak_probe(int shape, int blocks, int iters, uint64_t* clocks_dev, float* sink_dev, igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Generic exact-fp32 MFMA GEMM used by every Linear forward / dgrad / wgrad on the path.
 *   C[m][n] (+)= sum_k A(m,k) * B(n,k)
 *   a_kcontig: A(m,k) = A[m*lda + k]  else A[k*lda + m];   b_kcontig likewise for B(n,k).
 * epilogue: 0 store | 1 tanh(acc + bias[n]) | 2 acc * (1 - aux[m][n]^2) | 3 acc + bias[n]
 *           4 relu(acc + bias[n]) | 5 acc * (aux[m][n] > 0)
 *           6 elu(acc + bias[n]) | 7 acc * elu'(.) with aux = the ELU OUTPUT (a > 0 ? 1 : a + 1)   [alpha = 1]
 * accumulate != 0 adds the previous contents of C before the epilogue.
 * Replaces torch.nn.Linear + nn.Tanh forward and their autograd backward
 * (algo/models/models_split.py:27-38, 222-228).
 * ---------------------------------------------------------------------------------------- */
/* Opt-in reduced-precision mode of the large Linear products (forward, data gradient and grouped weight gradient
 * of the 128-wide tiles): operands are rounded to bf16 (nearest even) as they are fed to v_mfma_f32_32x32x16_bf16,
 * accumulation stays fp32.  16x the matrix rate, ~3 significant digits per product -- NOT reference arithmetic:
 * off by default (also settable with IGI_GEMM_BF16=1), parity tests and the headline benchmark run with it off.
 * Returns the previous setting. */
int igi_gemm_set_bf16_inputs(int on);
/* EXPERIMENT, off by default (IGI_GEMM_X3 = 6 | 9 in the environment starts it on): fp32 products on the bf16 matrix pipe by
 * an exact three-plane split of every operand element (x = hi + mid + lo in bf16: each plane product is exact in fp32, the
 * MFMA accumulates in fp32) for the large k-contiguous forward products (K >= 256).  products = 9: all nine cross products,
 * nothing dropped; 6: without mid*lo, lo*mid, lo*lo (each below 2^-24 of the leading product); 0: off.  Returns the previous
 * setting.  Not the arithmetic of the headline number: bench.py reports it beside it. */
int igi_gemm_set_bf16x3(int products);

int igi_gemm_f32(int a_kcontig, int b_kcontig, int M, int N, int K,
                 const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                 const float* bias, const float* aux, int ldaux, int epilogue, int accumulate,
                 igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Running mean / std  (algo/models/running_mean_std.py:23-93).
 * state = [mean[D], var[D], count] as D+D+1 doubles.  train != 0: merge the batch moments of
 * x (rows x D, unbiased variance) into state first (Chan), then y = clamp((x-mean)/sqrt(var+eps),
 * +-5); unnorm != 0: y = sqrt(var+eps)*clamp(x,+-5)+mean (no update).
 * workspace: igi_rms_workspace_bytes(rows, D).
 * ---------------------------------------------------------------------------------------- */
size_t igi_rms_workspace_bytes(int64_t rows, int D);
int igi_rms_forward(const float* x, float* y, int64_t rows, int D, double* state, float eps,
                    int train, int unnorm, void* workspace, size_t workspace_bytes,
                    igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Teacher PPO update (algo/ppo/frozen_ppo.py:495-646, 714-725; algo/ppo/experience.py:199-263).
 * ---------------------------------------------------------------------------------------- */
typedef struct igi_teacher_cfg {
  int32_t obs_dim, priv_dim, act_dim;
  int32_t n_priv_layers;
  int32_t priv_units[IGI_MAX_LAYERS]; /* env_mlp widths; last = latent (models_split.py:73-76) */
  int32_t n_layers;
  int32_t units[IGI_MAX_LAYERS];      /* actor_mlp / critic_mlp widths (models_split.py:100-102) */
  int32_t num_envs, horizon, mini_epochs; /* N, T, E; minibatch = N*T/E (frozen_ppo.py:213-215) */
  int32_t _pad0;
  /* Python-float (double) hyper-parameters: the reference multiplies them as doubles before the
   * tensor op casts to fp32 (e.g. gamma*tau, experience.py:254; Adam bias corrections). */
  double gamma, tau;                  /* experience.py:242-255 */
  double lr, beta1, beta2, adam_eps;  /* torch.optim.Adam (frozen_ppo.py:192-194) */
  float e_clip, critic_coef, entropy_coef, bounds_loss_coef; /* frozen_ppo.py:543-564 */
  float grad_norm;                    /* clip_grad_norm_ max norm; <=0: no clipping (:608-609) */
  float rms_eps;                      /* running_mean_std.py:24 (1e-5) */
} igi_teacher_cfg;

/* Time-major rollout arena, exactly what play_steps stores (frozen_ppo.py:655-683;
 * experience.py:163-185): obses (T,N,obs) priv_info (T,N,priv) rewards/values (T,N,1)
 * neglogpacs (T,N) dones (T,N) u8 actions/mus/sigmas (T,N,act); last_values (N,1). */
typedef struct igi_rollout {
  const float* obses;
  const float* priv_info;
  const float* rewards;
  const float* values;
  const float* neglogpacs;
  const uint8_t* dones;
  const float* actions;
  const float* mus;
  const float* sigmas;
  const float* last_values;
} igi_rollout;

/* Trainer-owned persistent state. */
typedef struct igi_teacher_state {
  float* params;      /* flat fp32, ActorCriticSplit.state_dict() order, every
                         tensor starting on a 16-byte boundary: see igi_teacher_param_offsets */
  float* grads;       /* flat fp32 gradient of the last minibatch (pre-clip), same layout */
  float* adam_m;      /* exp_avg */
  float* adam_v;      /* exp_avg_sq */
  double* rms_obs;    /* [mean(obs_dim), var(obs_dim), count] */
  double* rms_priv;   /* [mean(priv_dim), var(priv_dim), count] */
  double* rms_value;  /* [mean, var, count] */
  const int64_t* perm;/* fixed permutation of env-major sample ids (experience.py:202) */
  /* prepared per-update data, time-major index t*N+n (written by igi_teacher_prepare): */
  float* returns_raw; /* (T,N) GAE returns before normalisation */
  float* advantages;  /* (T,N) normalised advantages */
  float* values_n;    /* (T,N) normalised old values */
  float* returns_n;   /* (T,N) normalised returns */
  float* mus_w;       /* (T,N,act) working copy, overwritten by update_mu_sigma */
  float* sigmas_w;    /* (T,N,act) */
  float* stats;       /* [E*E][IGI_STATS_PER_STEP] per-optimizer-step scalars, see below */
  void* workspace;
  size_t workspace_bytes;
} igi_teacher_state;

/* stats row: a_loss, c_loss, b_loss, entropy, kl, grad_total_norm (pre-clip), param_norm, 0 */
#define IGI_STATS_PER_STEP 8

/* Flat parameter vector: tensors in state_dict order (sigma, env_mlp.mlp.{0,2,..}.{weight,bias},
 * actor_mlp..., critic_mlp..., value.{weight,bias}, mu.{weight,bias}); each tensor starts at a
 * multiple of 4 floats (gaps are zero and stay zero).  igi_teacher_param_count = padded length;
 * igi_teacher_param_offsets fills offsets_host[i] / sizes_host[i] for tensor i and returns the
 * number of tensors (or a negative error); either array may be NULL. */
int64_t igi_teacher_param_count(const igi_teacher_cfg* cfg);
int igi_teacher_param_offsets(const igi_teacher_cfg* cfg, int64_t* offsets_host, int64_t* sizes_host,
                              int max_tensors);
size_t igi_teacher_workspace_bytes(const igi_teacher_cfg* cfg);

/* computer_return + prepare_training + value normalisation tail
 * (experience.py:242-263, frozen_ppo.py:717-725).  No transposed copies are made: sample id
 * b = n*T+t of the reference's env-major flattening addresses element t*N+n.
 * normalize_value == 0 (ppo.normalize_value False, frozen_ppo.py:719) leaves rms_value untouched
 * and copies values / returns through un-normalised. */
int igi_teacher_prepare(const igi_teacher_cfg* cfg, const igi_rollout* ro,
                        const igi_teacher_state* st, int normalize_value, igi_stream_t stream);

/* One optimizer step of the minibatch loop, split so a gradient all-reduce can sit between
 * (frozen_ppo.py:518-584 = fwd_bwd; :586-603 = caller's all-reduce on st->grads; :605-618 = apply).
 *   mb_index : which minibatch of the fixed permutation (experience.py:207-226)
 *   step_slot: row of st->stats to fill
 *   adam_t   : 1-based Adam step count;  grad_scale: 1/world_size folded into the optimizer. */
int igi_teacher_fwd_bwd(const igi_teacher_cfg* cfg, const igi_rollout* ro,
                        const igi_teacher_state* st, int mb_index, int step_slot,
                        igi_stream_t stream);
int igi_teacher_apply(const igi_teacher_cfg* cfg, const igi_teacher_state* st, int step_slot,
                      int64_t adam_t, float grad_scale, igi_stream_t stream);

/* Data-parallel variant of igi_teacher_fwd_bwd in two phases, so that the all-reduce of the large gradient
 * bucket overlaps the rest of backward (the reference reduces all gradients after backward,
 * frozen_ppo.py:586-603; BASELINE north_star asks for the overlap).  The cut follows the backward levels, so the two
 * phases launch exactly the kernels of igi_teacher_fwd_bwd plus one more call of the slab reduction:
 *   phase 0: gather, forward, losses, trunk backward down to dZ of the first trunk layer, heads.  On return (in stream
 *            order) the EARLY bucket is final -> start its all-reduce: actor layers >= 1 | critic layers >= 1, value, mu.
 *   phase 1: latent and env_mlp backward + the first trunk layer's weight gradient.  The LATE bucket is final:
 *            sigma, env_mlp, actor layer 0 | critic layer 0.
 * igi_teacher_grad_buckets writes the four ranges (offset, length in floats) of the flat gradient: [0], [1] early,
 * [2], [3] late (the critic's first layer sits between the two early ranges; a length may be 0) and returns 4.
 * Both phases of a step take the same (mb_index, step_slot); results equal igi_teacher_fwd_bwd bit for bit. */
int igi_teacher_fwd_bwd_phase(const igi_teacher_cfg* cfg, const igi_rollout* ro,
                              const igi_teacher_state* st, int mb_index, int step_slot, int phase,
                              igi_stream_t stream);
int igi_teacher_grad_buckets(const igi_teacher_cfg* cfg, int64_t* offsets, int64_t* lengths);

/* Whole single-GPU update: mini_epochs x n_minibatch (fwd_bwd + apply), enqueued back to back
 * with no host synchronisation (frozen_ppo.py:508-640).  adam_t0 = steps taken before. */
int igi_teacher_update(const igi_teacher_cfg* cfg, const igi_rollout* ro,
                       const igi_teacher_state* st, int64_t adam_t0, igi_stream_t stream);
/* Norm fusion of igi_teacher_update (default OFF -- measured slower, profiles/r06_norm_fuse_ab.log; IGI_NORM_FUSE=1 in the
 * environment starts it on): from the second
 * optimizer step of an update on, the gradient norm of clip_grad_norm_ (frozen_ppo.py:608) is summed from per-block partials
 * the gradient assembly leaves behind and the logged parameter norm (frozen_ppo.py:605-606) from partials of the previous
 * step's Adam pass -- one launch less per step.  Same values up to the order of the fp64 additions; with 0 every step runs
 * the separate norm kernel, and igi_teacher_update is then bit-identical to the igi_teacher_fwd_bwd / _apply loop and to
 * the data-parallel updates on one rank (which always run it: the norm there is taken AFTER the all-reduce).  Returns the
 * previous setting. */
int igi_teacher_set_norm_fusion(int on);
/* Latent-gradient fusion of the teacher backward (default ON; IGI_LATZ_FUSE=0 in the environment starts it off;
 * profiles/r06_latz_ab.log): the backward of env_mlp's last (8-wide) layer -- d(latent) from the first trunk layer's row dots,
 * its rank-8 weight / bias gradient and dZ of the 128-wide layer below (autograd of models_split.py:185-232 as called from
 * frozen_ppo.py:583-585) -- happens at the head of every row block of the env level's row-block kernel instead of in a launch
 * of its own (9 launches per optimizer step instead of 10, dZ of that layer never reaches HBM).  dZ is formed by the same
 * expressions (bit-identical); the latent layer's weight gradient is summed on the matrix pipe per row range instead of per 32
 * rows, so it differs in the last bits.  Reference shapes only (priv_units [256, 128, 8]); anything else keeps the separate
 * launch whatever the setting.  Returns the previous setting. */
int igi_teacher_set_latz_fuse(int on);

/* Whole DATA-PARALLEL update as ONE host call (frozen_ppo.py:508-640 with the gradient exchange of :586-603):
 * per optimizer step the library enqueues phase 0, calls reduce(user, 0, step) -- the caller starts the all-reduce
 * (SUM) of grads[grad_split : param_count) on its communication stream and returns without blocking --, enqueues
 * phase 1, calls reduce(user, 1, step) for grads[0 : grad_split), then reduce(user, 2, step) -- the caller makes
 * `stream` wait (stream-ordered, no host block) for both collectives -- and enqueues clip + Adam with grad_scale
 * (= 1 / world_size).  reduce returns 0, or non-zero to abort (returned as IGI_E_CALLBACK).  The callback is the
 * only thing that is not a kernel launch: there is no per-step host <-> library round trip besides it. */
typedef int (*igi_reduce_fn)(void* user, int bucket, int step);
int igi_teacher_update_dp(const igi_teacher_cfg* cfg, const igi_rollout* ro, const igi_teacher_state* st,
                          int64_t adam_t0, float grad_scale, igi_reduce_fn reduce, void* user, igi_stream_t stream);

/* ---- RCCL communicator owned by the library (replaces dist.init_process_group("nccl") + the per-step
 * torch.cat / dist.all_reduce / copy-back of frozen_ppo.py:116-126, 586-603 and ext_adapt.py:833-851).
 * One process per GPU.  Rank 0 draws an id (igi_comm_unique_id, 128 bytes), hands it to every rank by any means
 * (the Python layer uses torch.distributed's store), and every rank calls igi_comm_create on ITS device: the object
 * holds the ncclComm_t, a communication stream and the events that fence it against the compute stream
 * (system-scope events when world > 1, device-scope on a one-rank communicator; IGI_EVENT_SYSFENCE=0 / 1 overrides).
 * A failing igi_comm_create releases everything it had built, sets *out = NULL and leaves the reason in
 * igi_comm_last_error(NULL) (per thread). */
#define IGI_COMM_ID_BYTES 128
typedef struct igi_comm* igi_comm_t;
int igi_comm_unique_id(void* id128);
int igi_comm_create(const void* id128, int rank, int world, igi_comm_t* out);
int igi_comm_destroy(igi_comm_t comm);
int igi_comm_rank(igi_comm_t comm);
int igi_comm_world(igi_comm_t comm);
/* What RCCL itself says about the communicator: ncclCommCount (the number of ranks RCCL connected -- igi_comm_world
 * returns the value the caller passed to igi_comm_create) or a negative IGI_E_* code; and RCCL's version
 * (ncclGetVersion: major * 10000 + minor * 100 + patch), no communicator needed.  A benchmark record that carries both can
 * show that an N-GPU run really exchanged gradients among N ranks (frozen_ppo.py:119-121 reads WORLD_SIZE only). */
int igi_comm_count(igi_comm_t comm);
int igi_rccl_version(void);
const char* igi_comm_last_error(igi_comm_t comm);
/* in place, SUM, enqueued on `stream` (stream-ordered; the host does not block) */
int igi_comm_all_reduce_sum_f32(igi_comm_t comm, float* buf, int64_t n, igi_stream_t stream);
/* The same reduction on the communicator's own stream, ordered behind everything enqueued on `compute_stream` so far
 * (event fence; the host does not block): work enqueued on `compute_stream` afterwards overlaps the collective -- the
 * student's early gradient bucket (decoder side, final first) under the encoders' backward; ext_adapt.py:833-851 reduces
 * everything after backward.  igi_comm_join makes `compute_stream` wait for the last such reduction.  One in flight at
 * a time per communicator. */
int igi_comm_all_reduce_async_f32(igi_comm_t comm, float* buf, int64_t n, igi_stream_t compute_stream);
int igi_comm_join(igi_comm_t comm, igi_stream_t compute_stream);
/* parameter broadcast at the start of training (frozen_ppo.py:376-381 pickles a state_dict; here: the flat vector) */
int igi_comm_broadcast(igi_comm_t comm, void* buf, int64_t bytes, int root, igi_stream_t stream);

/* Whole data-parallel update as ONE host call with the gradient exchange issued by the library: per optimizer step
 * phase 0 -> [comm stream] all-reduce of the early bucket -> phase 1 (runs meanwhile) -> all-reduce of the late bucket
 * -> clip + Adam with grad_scale = 1 / world; events only, no callback, no host synchronisation.  overlap == 0: the
 * reference's serial schedule (one all-reduce of the whole flat gradient after backward, on `stream`).
 * stats_sum: NULL, or mini_epochs * n_minibatch * IGI_STATS_PER_STEP floats receiving st->stats summed over the
 * ranks (the per-mini-epoch KL all-reduce of frozen_ppo.py:624-627 and the loss aggregation of :387-396 as one
 * collective per update).  The learning-rate broadcast of :632-637 has nothing to send: the scheduler call is
 * commented out in the reference (:630), the rate is constant. */
int igi_teacher_update_dp_rccl(const igi_teacher_cfg* cfg, const igi_rollout* ro, const igi_teacher_state* st,
                               int64_t adam_t0, igi_comm_t comm, int overlap, float* stats_sum, igi_stream_t stream);

/* Inference forward used by model_act / act_inference (models_split.py:120-164; frozen_ppo.py:343-366).
 * normalize != 0: obs/priv are raw and are normalised with the CURRENT running stats (eval mode,
 * frozen_ppo.py:345-346); normalize == 0: they are used as given (ActorCriticSplit.act receives
 * already-processed inputs).  Writes mu (rows,act), value (rows,1) (normalised scale) and latent
 * (rows, priv_units[-1]).  Any output may be NULL. */
int igi_teacher_infer(const igi_teacher_cfg* cfg, const igi_teacher_state* st, const float* obs,
                      const float* priv, int64_t rows, int normalize, float* mu, float* value,
                      float* latent, igi_stream_t stream);

/* The policy side of ONE environment step of play_steps as one call (frozen_ppo.py:343-366 model_act + :655-665 buffer
 * writes): igi_teacher_infer (normalise with the current running statistics when normalize != 0, env_mlp, trunk, heads)
 * and igi_rollout_act_store (sample with the caller's noise, neglogp, value de-normalisation with rms_value
 * ([mean, var, count], may be NULL), arena-slot writes, clamped actions) fused: 7-8 launches, bit-identical results.
 * obses_t / priv_t (raw copies into the arena slot) may be NULL; the other outputs are required:
 * actions_t / mus_t / sigmas_t / actions_clamped (rows, act), neglogp_t / values_t / values_out (rows). */
int igi_rollout_policy_step(const igi_teacher_cfg* cfg, const igi_teacher_state* st, const float* obs,
                            const float* priv, int64_t rows, int normalize, const float* noise,
                            const double* rms_value, float* obses_t, float* priv_t, float* actions_t,
                            float* neglogp_t, float* values_t, float* mus_t, float* sigmas_t, float* actions_clamped,
                            float* values_out, igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Rollout-side bookkeeping of PPO.play_steps, two launches per environment step.
 * igi_rollout_act_store (frozen_ppo.py:343-366, 655-665): from the policy outputs of N environments (mu (N,act),
 * value_n (N,1) on the normalised scale, logstd (act), noise (N,act) ~ N(0,1) supplied by the caller's generator):
 * sigma = exp(logstd), action = mu + sigma*noise, neglogp = -Normal(mu,sigma).log_prob(action).sum(-1), value =
 * sqrt(var+eps)*clamp(value_n,+-5)+mean when rms_value ([mean,var,count]) is given, else value_n.  Writes slot t of
 * the time-major arena -- the *_t pointers address that slot: obses_t (N,obs) priv_t (N,priv) actions_t / mus_t /
 * sigmas_t (N,act) neglogp_t (N) values_t (N) -- plus actions_clamped (N,act) = clamp(action,+-1) for env.step and
 * values_out (N).
 * igi_rollout_env_store (frozen_ppo.py:671-700): dones_t = dones; rewards_t = 0.01 r + gamma*value*time_out when
 * bootstrap != 0 and time_outs != NULL, else r; cur_rewards / cur_lengths / cur_success (N) accumulate r / 1 /
 * successes and are zeroed where done; meter[0..3] += {sum reward, sum length, sum success, count} of the episodes
 * that ended in this step (caller zeroes meter).
 * ---------------------------------------------------------------------------------------- */
int igi_rollout_act_store(int64_t n_envs, int obs_dim, int priv_dim, int act_dim, const float* obs,
                          const float* priv, const float* mu, const float* value_n, const float* logstd,
                          const float* noise, const double* rms_value, float eps, float* obses_t, float* priv_t,
                          float* actions_t, float* neglogp_t, float* values_t, float* mus_t, float* sigmas_t,
                          float* actions_clamped, float* values_out, igi_stream_t stream);
int igi_rollout_env_store(int64_t n_envs, const float* rewards, const uint8_t* dones, const float* values,
                          const uint8_t* time_outs, const float* successes, float gamma, int bootstrap,
                          float* rewards_t, uint8_t* dones_t, float* cur_rewards, float* cur_lengths,
                          float* cur_success, float* meter, igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Student distillation loss (ext_adapt.py:812-819): loss[0] = sum over rows and action dims of
 * weights[q] * (clamp(mu,+-1) - clamp(teacher,+-1))^2 (a SUM; the reference's trailing .mean() acts on a scalar), and,
 * when dmu != NULL, dmu = d loss / d mu = 2 weights[q] (clamp(mu) - clamp(teacher)) where -1 <= mu <= 1, else 0.
 * Fixed-order fp64 partial sums.  workspace >= igi_bc_loss_workspace_bytes().
 * ---------------------------------------------------------------------------------------- */
size_t igi_bc_loss_workspace_bytes(void);
int igi_bc_loss(const float* mu, const float* teacher_actions, const float* weights, int64_t rows, int act_dim,
                float* loss, float* dmu, void* workspace, size_t workspace_bytes, igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * clip_grad_norm_ + torch.optim.Adam step on one flat fp32 vector (frozen_ppo.py:608-610;
 * ext_adapt.py:853-855): grads are scaled by grad_scale (1/world after an all-reduce SUM), clipped to
 * global L2 norm max_norm (<= 0: no clipping) and applied with Adam's single-tensor update rule
 * (bias corrections from the 1-based step t).  stats_out (8 floats, may be NULL) receives
 * [.., .., .., .., .., total_norm, param_norm, clip_coef].  workspace: igi_clip_adam_workspace_bytes().
 * ---------------------------------------------------------------------------------------- */
size_t igi_clip_adam_workspace_bytes(void);
int igi_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float max_norm, double lr, double beta1, double beta2, double eps, int64_t t,
                  float grad_scale, void* workspace, size_t workspace_bytes, float* stats_out,
                  igi_stream_t stream);

/* Same with decoupled weight decay = torch.optim.AdamW's single-tensor rule: param *= 1 - lr*weight_decay
 * before the Adam update (offline supervised student: runner.py:481, AdamW(lr, weight_decay=1e-6)). */
int igi_clip_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float max_norm, double lr, double beta1, double beta2, double eps, double weight_decay,
                   int64_t t, float grad_scale, void* workspace, size_t workspace_bytes, float* stats_out,
                   igi_stream_t stream);
/* As igi_clip_adamw plus torch.optim.Adam's COUPLED weight decay (ext_adapt.py:1139, phase-3 optimizer
 * Adam(lr=1e-3, weight_decay=1e-6)): after clipping, grad += l2 * param. */
int igi_clip_adam_l2(float* params, const float* grads, float* m, float* v, int64_t n, float max_norm, double lr,
                     double beta1, double beta2, double eps, double weight_decay, double l2, int64_t t,
                     float grad_scale, void* workspace, size_t workspace_bytes, float* stats_out,
                     igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One backward LEVEL of a Linear + Tanh stack as ONE persistent row-block kernel (csrc/rowblock.h): for the layer
 * Z = X W^T + b whose input X is the tanh output of the layer below,
 *     dx = (dz . weight) * (1 - x^2)      [nets][rows][in]     data gradient into the layer below, times tanh'
 *     dweight_partials, dbias_partials    [parts][nets][out][in] / [parts][nets][out]: fixed-order partial sums over
 *                                         row ranges; the caller adds the `parts` partials (igi_level_backward_parts)
 * Replaces autograd through nn.Linear + nn.Tanh (algo/models/models_split.py:27-38, reached from loss.backward(),
 * algo/ppo/frozen_ppo.py:583-585) for the 128-wide layers of the teacher: trunk layer 3 (both nets) and env_mlp layer 2.
 * Shapes: out == 128, in a multiple of 64 (<= 1024), rows a multiple of 64 (>= 256), nets 1 or 2; dense rows
 * (ld = width), 16-byte aligned pointers.  Other shapes: IGI_E_UNSUPPORTED (igi_gemm_f32 covers them).
 * dx is bit-identical to igi_gemm_f32(epilogue 2) on the same operands.
 * ---------------------------------------------------------------------------------------- */
int igi_level_backward_parts(int64_t rows, int in_features, int nets);   /* 0 when the shape is not supported */
int igi_level_backward(const float* dz, const float* weight, const float* x, float* dx, float* dweight_partials,
                       float* dbias_partials, int64_t rows, int in_features, int out_features, int nets,
                       igi_stream_t stream);
/* The same level (nets == 1) with the weight / bias gradient of the layer BELOW computed from the finished data-gradient
 * tiles instead of dx being written (csrc/rowblock.h, LOWX; the teacher's env_mlp backward, models_split.py:27-38):
 *     below_dweight_partials [parts][in][64] : sums over row ranges of dx^T . x_below,   x_below [rows][64] = the input
 *     below_dbias_partials   [parts][in]     : sums over row ranges of dx                 rows of the layer below
 * dx = (dz . weight) * (1 - x^2) is consumed element by element as the left operand of that product and never stored. */
int igi_level_backward_below(const float* dz, const float* weight, const float* x, const float* x_below,
                             float* dweight_partials, float* dbias_partials, float* below_dweight_partials,
                             float* below_dbias_partials, int64_t rows, int in_features, int out_features,
                             igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * nn.Linear with a fused activation, forward and backward, for the student's small MLPs: lin encoder
 * Linear(15,64)-ReLU-Linear(64,32) (tact.py:337-339), point-cloud compress (tact.py:367-369), MLPDecoder /
 * MultiLayerDecoder output stack (tact.py:137-158, 197-212), action head Linear(32,6)+Tanh (tact.py:407-410)
 * and the offline supervised loop built on them (runner.py:194-304).
 *   forward : y[rows][out] = act(x[rows][in] . weight[out][in]^T + bias)     activation: 0 none, 1 tanh, 2 relu
 *   backward: dy is the gradient w.r.t. y; dz = dy * act'(y); dx = dz . weight (dx may be NULL);
 *             dweight = dz^T x (may be NULL with dbias NULL: frozen layer, data gradient only); dbias = column sums
 *             of dz (may be NULL).  Sums over rows are split and
 *             added in a fixed order (deterministic).  bias may be NULL only when activation == 0.
 * ld* are row strides in floats.  workspace: igi_linear_workspace_bytes(rows, in, out).
 * ---------------------------------------------------------------------------------------- */
size_t igi_linear_workspace_bytes(int64_t rows, int in_features, int out_features);
int igi_linear_forward(const float* x, int ldx, const float* weight, const float* bias, float* y, int ldy,
                       int64_t rows, int in_features, int out_features, int activation, igi_stream_t stream);
int igi_linear_backward(const float* x, int ldx, const float* weight, const float* y, int ldy, const float* dy,
                        int lddy, float* dx, int lddx, float* dweight, float* dbias, int64_t rows,
                        int in_features, int out_features, int activation, void* workspace,
                        size_t workspace_bytes, igi_stream_t stream);

/* Backward of a CHAIN of such layers (an nn.Sequential of Linear + activation: tact.py:137-158, 196-212, 337-339, 367-369,
 * 407-410 under loss.backward()) as one call: dims[0 .. n_layers] are the widths (layer l: dims[l] -> dims[l + 1]),
 * acts[l] its activation, weight[l] its (out, in) matrix, y[l] its saved OUTPUT [rows][dims[l + 1]] (dense rows),
 * x the chain's input, dy the gradient w.r.t. the chain's output.  Per layer ONE grid computes the weight gradient and
 * the data gradient into the layer below, whose epilogue applies that layer's act'; every layer's split-row partials are
 * summed by one launch at the end.  Results are bit-identical to n_layers calls of igi_linear_backward.
 *   dx      : [rows][dims[0]] or NULL
 *   grads   : ONE flat buffer of igi_mlp_grad_floats() floats; dweight of layer l at w_offsets[l], dbias at b_offsets[l]
 *             (multiples of four floats)
 *   need_w  : NULL, or per layer 0 = frozen (its gradient range is left untouched) */
#define IGI_MLP_MAX_LAYERS 8
/* The same chain FORWARD as one launch (csrc/mlp_fwd.h; tact.py:137-212, 337-339, 367-369, 407-410 under forward()):
 * a workgroup carries 32 rows through every layer, hidden activations stay in LDS, y[l] = the layer outputs
 * [rows][dims[l + 1]] (row pitch ldy[l], NULL = dense) are each written once -- the tensors igi_mlp_backward reads.
 * bias[l] may be NULL only where acts[l] == 0.  Bit-identical to n_layers calls of igi_linear_forward.  Layers wider
 * than 256 outputs (and the bf16-input mode): IGI_E_UNSUPPORTED, run the layers one by one. */
int igi_mlp_forward(const float* x, int ldx, int64_t rows, int n_layers, const int32_t* dims, const int32_t* acts,
                    const float* const* weight, const float* const* bias, float* const* y, const int32_t* ldy,
                    igi_stream_t stream);
int64_t igi_mlp_grad_floats(int n_layers, const int32_t* dims, int64_t* w_offsets, int64_t* b_offsets);
size_t igi_mlp_workspace_bytes(int64_t rows, int n_layers, const int32_t* dims);
int igi_mlp_backward(const float* x, int ldx, int64_t rows, int n_layers, const int32_t* dims, const int32_t* acts,
                     const float* const* weight, const float* const* y, const float* dy, float* dx, float* grads,
                     const int32_t* need_w, void* workspace, size_t workspace_bytes, igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Depth / segmentation image encoder: DepthOnlyFCBackbone54x96 (algo/models/transformer/tact.py:81-113),
 * forward and backward.  x is (batch, 1, 54, 96) fp32; y is (batch, latent_dim) BEFORE the optional output
 * activation (identity in the reference's use, tact.py:305, 323).  params / grads: flat vector in state_dict order
 * (image_compression.0.weight [32,1,5,5], .0.bias, .3.weight [64,32,3,3], .3.bias, .6.weight [128,64768], .6.bias,
 * .8.weight [latent,128], .8.bias).  backward needs the workspace forward filled and the same x; the input gets
 * no gradient (it is an observation).  batch must be a multiple of 32, latent_dim of 4.
 * ---------------------------------------------------------------------------------------- */
typedef struct igi_depth_cfg {
  int32_t batch, latent_dim;
} igi_depth_cfg;
int64_t igi_depth_param_count(const igi_depth_cfg* cfg);
size_t igi_depth_workspace_bytes(const igi_depth_cfg* cfg);
int igi_depth_forward(const igi_depth_cfg* cfg, const float* x, const float* params, float* y, void* workspace,
                      size_t workspace_bytes, igi_stream_t stream);
int igi_depth_backward(const igi_depth_cfg* cfg, const float* x, const float* dy, const float* params, float* grads,
                       void* workspace, size_t workspace_bytes, igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Token decoder of the student: `layers` x nn.TransformerEncoderLayer(d_model 32, nhead 2, dim_feedforward ff,
 * activation "gelu", batch_first, norm_first) over `seq` <= 8 tokens per sample
 * (algo/models/transformer/tact.py:137-158: MultiLayerDecoder.sa_decoder), forward and backward.
 * x, y, dy, dx: (batch, seq, 32) fp32.  params / grads: layers x igi_token_param_count()/layers floats, each layer
 * in nn.TransformerEncoderLayer's parameter order (in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias,
 * linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight, norm1.bias, norm2.weight, norm2.bias).
 * training != 0 applies the layer's four dropouts (attention probabilities, both residual branches, after the
 * activation) with probability `dropout`; masks are a counter-based function of `seed` (pass the same seed and the
 * same workspace to backward; nothing else is retained between the two calls).
 * ---------------------------------------------------------------------------------------- */
typedef struct igi_token_cfg {
  int32_t batch, seq, d_model, nhead, ff, layers;
  float dropout;
  int32_t training;
} igi_token_cfg;
int64_t igi_token_param_count(const igi_token_cfg* cfg);
size_t igi_token_workspace_bytes(const igi_token_cfg* cfg);
int igi_token_forward(const igi_token_cfg* cfg, const float* x, const float* params, float* y, void* workspace,
                      size_t workspace_bytes, uint64_t seed, igi_stream_t stream);
int igi_token_backward(const igi_token_cfg* cfg, const float* dy, const float* params, float* dx, float* grads,
                       void* workspace, size_t workspace_bytes, uint64_t seed, igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * AllSight tactile encoder: CNNWithSpatialSoftArgmax (algo/models/transformer/tactile_cnn.py:7-79),
 * forward and backward.  x is (batch, 3, height, width) fp32 NCHW as the reference feeds it
 * (3 fingers' gray images stacked as channels; runner.py:397-400, tact.py:431-432); y is
 * (batch, latent_dim).  params / grads are flat fp32 in the module's state_dict order:
 * cnn.0.weight (32,3,8,8) cnn.0.bias cnn.2.weight (64,32,4,4) cnn.2.bias cnn.4.weight (64,64,3,3)
 * cnn.4.bias cnn.7.weight (latent,128) cnn.7.bias.  batch must be a multiple of 32.
 * igi_tactile_backward must follow igi_tactile_forward on the same workspace (saved activations);
 * it overwrites `grads` with d(loss)/d(params) given dy = d(loss)/d(y).
 * ---------------------------------------------------------------------------------------- */
typedef struct igi_tactile_cfg {
  int32_t batch, height, width, latent_dim;
} igi_tactile_cfg;
int64_t igi_tactile_param_count(const igi_tactile_cfg* cfg);
size_t igi_tactile_workspace_bytes(const igi_tactile_cfg* cfg);
int igi_tactile_forward(const igi_tactile_cfg* cfg, const float* x, const float* params, float* y,
                        void* workspace, size_t workspace_bytes, igi_stream_t stream);
int igi_tactile_backward(const igi_tactile_cfg* cfg, const float* dy, const float* params, float* grads,
                         void* workspace, size_t workspace_bytes, igi_stream_t stream);
/* Where igi_tactile_forward leaves the three activated feature maps in its workspace (diagnostics / parity tests: which
 * side of a ReLU a pre-activation fell on): byte offsets[3] and rows[3] = batch * H_out * W_out of conv 1..3; each map is
 * channels-last (rows, 32 | 64 | 64) fp32.  Returns 0, or an error for a configuration the encoder rejects. */
int igi_tactile_activation_layout(const igi_tactile_cfg* cfg, int64_t offsets[3], int64_t rows[3]);

/* Standalone SpatialSoftArgmax.forward (algo/models/transformer/tactile_cnn.py:47-58) and its gradient, for callers
 * that use the module outside CNNWithSpatialSoftArgmax: x is (rows = B*C, h*w) contiguous (an NCHW tensor),
 * out (rows, 2) = (E[x], E[y]) with the reference's coordinate quirk (flat index k -> grid_w[k / h], grid_h[k % h]),
 * stat (rows, 2) = (row max, sum of exp) kept for the backward.  normalize != 0: linspace(-1, 1) grids, else arange. */
int igi_spatial_softargmax_forward(const float* x, int64_t rows, int h, int w, int normalize, float* out,
                                   float* stat, igi_stream_t stream);
int igi_spatial_softargmax_backward(const float* x, const float* out, const float* stat, const float* dout,
                                    int64_t rows, int h, int w, int normalize, float* dx, igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Segmented point-cloud encoder: PointNet (algo/models/transformer/pointnets.py:12-42), forward
 * (+ argmax over the point axis) and backward.  x (batch, npoints, 3), consecutive clouds x_pitch floats apart (0 = dense;
 * > 3 npoints: a slice of a wider cloud tensor read in place, tact.py:542-566); dy (batch, 256) with rows dy_pitch floats
 * apart (0 = dense: a slice of the gradient of the concatenated encodings); y (batch, 256); argmax
 * (batch, 256) int32 point index of each column's maximum (may be NULL in forward when no backward
 * follows).  params / grads flat fp32 in state_dict order: local_mlp.0.weight (64,3),
 * local_mlp.0.bias (64), local_mlp.2.weight (256,64), local_mlp.2.bias (256) = 16896 floats.
 * Backward overwrites `grads`; workspace: igi_pointnet_workspace_bytes(batch).  1 <= npoints <= 8192 (the reference's
 * clouds hold 400 points per object; more returns IGI_E_UNSUPPORTED).  Ties between points: the first maximum in point
 * order, as torch.max.  erf-GELU: x * Phi(x) with |Phi error| <= 7.3e-8 (csrc/pointnet.h).
 * ---------------------------------------------------------------------------------------- */
#define IGI_POINTNET_PARAMS 16896
size_t igi_pointnet_workspace_bytes(int64_t batch);
int igi_pointnet_forward(const float* x, int64_t x_pitch, int64_t batch, int npoints, const float* params, float* y,
                         int32_t* argmax, igi_stream_t stream);
int igi_pointnet_backward(const float* x, int64_t x_pitch, int64_t batch, int npoints, const float* params, const float* dy,
                          int64_t dy_pitch, const int32_t* argmax, float* grads, void* workspace, size_t workspace_bytes,
                          igi_stream_t stream);
/* Several objects of ONE cloud tensor in one launch (tact.py:542-566: the plug, socket, goal ... PointNets each encode their
 * own slice of obs_pcl; here: one forward and one backward launch per step instead of one per object, and the encodings land
 * concatenated, which is what compress_pcl_enc reads, tact.py:568-571).  1 <= nobj <= 4; object i = points
 * x_off[i] / 3 ... of every cloud (x_off in floats from the cloud's first float), npoints[i] of them, parameters params[i]
 * (host array of device pointers, 16896 floats each).  y and argmax are (batch, nobj * 256): object i in columns 256 i ...;
 * dy rows dy_pitch floats apart (0 = dense nobj * 256); grads (nobj, 16896); workspace:
 * igi_pointnet_workspace_bytes_multi(batch, nobj).  Same arithmetic per object as the single-object entries (bit-identical
 * outputs; the weight-gradient partials are summed over another number of workgroups). */
size_t igi_pointnet_workspace_bytes_multi(int64_t batch, int nobj);
int igi_pointnet_forward_multi(int nobj, const float* x, int64_t x_pitch, int64_t batch, const int32_t* x_off,
                               const int32_t* npoints, const float* const* params, float* y, int32_t* argmax,
                               igi_stream_t stream);
int igi_pointnet_backward_multi(int nobj, const float* x, int64_t x_pitch, int64_t batch, const int32_t* x_off,
                                const int32_t* npoints, const float* const* params, const float* dy, int64_t dy_pitch,
                                const int32_t* argmax, float* grads, void* workspace, size_t workspace_bytes,
                                igi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The student step's data movement between the blocks above (csrc/glue.h) -- pure copies, bit-exact:
 *   igi_gather_rows : dst[k][r][:] = src[k][rows[r]][:] for n <= 8 arenas of width[k] floats per row in ONE launch -- the
 *                     minibatch gather of StudentBuffer.__getitem__ (experience.py:117-139 gathers every key by itself).
 *                     A row number outside [0, rows_total) yields a NaN row.
 *   igi_cat_cols    : cat[b] = [part[0][b] | part[1][b] | ...] (+ add, one row of sum(width) floats added to every row,
 *                     or NULL): torch.cat of the encoders' tokens plus the positional encoding (tact.py:567-571, 126-135)
 *                     and of the point-cloud encodings (tact.py:566).
 *   igi_split_cols  : the reverse (the backward of the concatenation): every part dense, one launch.
 * ---------------------------------------------------------------------------------------- */
#define IGI_GLUE_MAX 8
int igi_gather_rows(int n, const float* const* src, const int64_t* width, float* const* dst, const int64_t* rows,
                    int64_t nrows, int64_t rows_total, igi_stream_t stream);
int igi_cat_cols(int n, const float* const* part, const int64_t* width, float* cat, const float* add, int64_t rows,
                 igi_stream_t stream);
int igi_split_cols(int n, float* const* part, const int64_t* width, const float* cat, int64_t rows, igi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IGI_PPO_H */
