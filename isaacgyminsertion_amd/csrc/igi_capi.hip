// C ABI of libigi_hip.so (declared in include/igi_ppo.h).  Thin argument checking around the
// kernels in gemm_f32.h / teacher.h / rms.h; nothing here allocates or synchronises.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/igi_ppo.h"
#include "gemm_dma.h"
#include "gemm_f32.h"
#include "rms.h"
#include "rollout.h"
#include "pointnet.h"
#include "tactile.h"
#include "teacher.h"
#include "linear.h"
#include "mlp_chain.h"
#include "mlp_fwd.h"
#include "token_encoder.h"
#include "depth.h"
#include "comm.h"
#include "glue.h"
#include "peak_probe.h"

namespace {
thread_local char g_err[256] = "";

int fail(int code, const char* where) {
  if (code > 0)
    snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, code, hipGetErrorString((hipError_t)code));
  else if (code == IGI_E_BADARG)
    snprintf(g_err, sizeof(g_err), "%s: bad argument", where);
  else if (code == IGI_E_WORKSPACE)
    snprintf(g_err, sizeof(g_err), "%s: workspace too small", where);
  else if (code == IGI_E_CALLBACK)
    snprintf(g_err, sizeof(g_err), "%s: the caller's reduce callback failed", where);
  else if (code == IGI_E_UNSUPPORTED)
    snprintf(g_err, sizeof(g_err), "%s: unsupported configuration", where);
  else if (code == IGI_E_COMM)
    snprintf(g_err, sizeof(g_err), "%s: RCCL call failed (igi_comm_last_error)", where);
  else if (code != 0)
    snprintf(g_err, sizeof(g_err), "%s: error %d", where, code);
  return code;
}
inline hipStream_t S(igi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
}  // namespace

extern "C" {

#ifndef IGI_SRC_HASH
#define IGI_SRC_HASH "unknown"
#endif
int igi_abi_version(void) { return IGI_ABI_VERSION; }
// the tag makes the hash findable in the file without loading it (__graft_entry__.library_hash)
static const char g_build_tag[] = "igi-src-hash:" IGI_SRC_HASH;
const char* igi_build_info(void) { return g_build_tag + 13; }
const char* igi_last_error(void) { return g_err; }

int igi_gemm_f32(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda,
                 const float* B, int ldb, float* C, int ldc, const float* bias, const float* aux,
                 int ldaux, int epilogue, int accumulate, igi_stream_t stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0 || epilogue < 0 || epilogue >= igi::EPI_COUNT) return fail(IGI_E_BADARG, "igi_gemm_f32");
  if ((epilogue == igi::EPI_BIAS_TANH || epilogue == igi::EPI_BIAS || epilogue == igi::EPI_BIAS_RELU ||
       epilogue == igi::EPI_BIAS_ELU) && !bias)
    return fail(IGI_E_BADARG, "igi_gemm_f32");
  if ((epilogue == igi::EPI_TANHGRAD || epilogue == igi::EPI_RELUGRAD || epilogue == igi::EPI_ELUGRAD) && !aux)
    return fail(IGI_E_BADARG, "igi_gemm_f32");
  igi::GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = aux;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.epilogue = epilogue; g.accumulate = accumulate;
  return fail((int)igi::gemm(g, a_kcontig != 0, b_kcontig != 0, S(stream)), "igi_gemm_f32");
}

int igi_level_backward_parts(int64_t rows, int in_features, int nets) {
  if (rows < 1 || rows >= (1 << 24) || !igi::rb_level_shape_ok(rows, igi::RB_KO, in_features, nets)) return 0;
  return igi::rb_level_ranges((int)rows, in_features, nets);
}

int igi_level_backward(const float* dz, const float* weight, const float* x, float* dx, float* dweight_partials,
                       float* dbias_partials, int64_t rows, int in_features, int out_features, int nets,
                       igi_stream_t stream) {
  if (!dz || !weight || !x || !dx || !dweight_partials || !dbias_partials) return fail(IGI_E_BADARG, "igi_level_backward");
  if (out_features != igi::RB_KO || rows < 1 || rows >= (1 << 24) ||
      !igi::rb_level_shape_ok(rows, out_features, in_features, nets))
    return fail(IGI_E_UNSUPPORTED, "igi_level_backward");
  igi::RbLevelArgs a;
  a.dZ = dz; a.ldz = out_features; a.sZ = rows * out_features;
  a.W = weight; a.ldw = in_features; a.sW = (long long)out_features * in_features;
  a.X = x; a.ldx = in_features; a.sX = rows * in_features;
  a.dX = dx; a.lddx = in_features; a.sdX = rows * in_features;
  a.dWp = dweight_partials; a.ldwp = in_features; a.sWnet = (long long)out_features * in_features; a.sWpart = nets * a.sWnet;
  a.dBp = dbias_partials; a.sBnet = out_features; a.sBpart = (long long)nets * out_features;
  a.rows = (int)rows; a.IN = in_features; a.nets = nets; a.ranges = igi::rb_level_ranges((int)rows, in_features, nets);
  const hipError_t e = igi::rb_level_backward(a, S(stream), igi::PC_OTHER);
  if (e == hipErrorNotSupported) return fail(IGI_E_UNSUPPORTED, "igi_level_backward");
  return fail((int)e, "igi_level_backward");
}

int igi_level_backward_below(const float* dz, const float* weight, const float* x, const float* x_below,
                             float* dweight_partials, float* dbias_partials, float* below_dweight_partials,
                             float* below_dbias_partials, int64_t rows, int in_features, int out_features,
                             igi_stream_t stream) {
  if (!dz || !weight || !x || !x_below || !dweight_partials || !dbias_partials || !below_dweight_partials || !below_dbias_partials)
    return fail(IGI_E_BADARG, "igi_level_backward_below");
  if (out_features != igi::RB_KO || rows < 1 || rows >= (1 << 24) || !igi::rb_level_shape_ok(rows, out_features, in_features, 1))
    return fail(IGI_E_UNSUPPORTED, "igi_level_backward_below");
  igi::RbLevelArgs a;
  a.dZ = dz; a.ldz = out_features;
  a.W = weight; a.ldw = in_features;
  a.X = x; a.ldx = in_features;
  a.dX = nullptr; a.lddx = in_features;
  a.dWp = dweight_partials; a.ldwp = in_features; a.sWnet = (long long)out_features * in_features; a.sWpart = a.sWnet;
  a.dBp = dbias_partials; a.sBnet = out_features; a.sBpart = out_features;
  a.rows = (int)rows; a.IN = in_features; a.nets = 1; a.ranges = igi::rb_level_ranges((int)rows, in_features, 1);
  a.lx_X = x_below; a.lx_ld = 64;
  a.lx_W = below_dweight_partials; a.lx_ldw = 64; a.lx_sPart = (long long)in_features * 64;
  a.lx_B = below_dbias_partials; a.lx_bsPart = in_features;
  const hipError_t e = igi::rb_level_backward(a, S(stream), igi::PC_OTHER);
  if (e == hipErrorNotSupported) return fail(IGI_E_UNSUPPORTED, "igi_level_backward_below");
  return fail((int)e, "igi_level_backward_below");
}

int igi_teacher_set_norm_fusion(int on) {
  const int prev = igi::norm_fusion_ref();
  igi::norm_fusion_ref() = on != 0;
  return prev;
}

int igi_teacher_set_latz_fuse(int on) {
  const int prev = igi::latz_fuse_ref();
  igi::latz_fuse_ref() = on != 0;
  return prev;
}

int igi_gemm_set_bf16x3(int products) {
  const int prev = igi::x3_mode();
  igi::x3_mode_ref() = (products == 6 || products == 9) ? products : 0;
  return prev;
}

int igi_gemm_set_bf16_inputs(int on) {
  const int prev = igi::bf16_mode();
  igi::bf16_mode_ref() = on != 0;
  return prev;
}

int igi_mfma_peak_probe(int shape, int blocks, int iters, uint64_t* clocks_dev, float* sink_dev, igi_stream_t stream) {
  return fail(igi::mfma_peak_probe(shape, blocks, iters, (unsigned long long*)clocks_dev, sink_dev, S(stream)), "igi_mfma_peak_probe");
}

int igi_prof_enable(int on) {
  igi::Profiler& p = igi::profiler();
  std::lock_guard<std::mutex> g(p.mu);
  for (auto& r : p.recs) { p.pool.push_back(r.a); p.pool.push_back(r.b); }
  p.recs.clear();
  p.on = on != 0;
  return 0;
}

int igi_prof_read(igi_prof_entry* out, int max_entries) {
  igi::Profiler& p = igi::profiler();
  std::lock_guard<std::mutex> g(p.mu);
  igi_prof_entry acc[igi::PC_COUNT];
  for (int i = 0; i < igi::PC_COUNT; ++i) {
    acc[i].name = igi::kProfNames[i];
    acc[i].launches = 0; acc[i].total_ms = 0; acc[i].flops = 0; acc[i].bytes = 0;
  }
  for (auto& r : p.recs) {
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return fail((int)e, "igi_prof_read");
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return fail((int)e, "igi_prof_read");
    acc[r.cls].launches += 1; acc[r.cls].total_ms += ms; acc[r.cls].flops += r.flops; acc[r.cls].bytes += r.bytes;
  }
  int n = 0;
  for (int i = 0; i < igi::PC_COUNT; ++i)
    if (acc[i].launches > 0) { if (out && n < max_entries) out[n] = acc[i]; ++n; }
  return n;
}

size_t igi_rms_workspace_bytes(int64_t rows, int D) {
  if (rows < 1 || D < 1) return 0;
  return igi::rms_workspace_bytes(rows, D);
}

int igi_rms_forward(const float* x, float* y, int64_t rows, int D, double* state, float eps, int train,
                    int unnorm, void* workspace, size_t workspace_bytes, igi_stream_t stream) {
  return fail(igi::rms_forward(x, y, rows, D, state, eps, train, unnorm, workspace, workspace_bytes, S(stream)),
              "igi_rms_forward");
}

int igi_rollout_act_store(int64_t n_envs, int obs_dim, int priv_dim, int act_dim, const float* obs,
                          const float* priv, const float* mu, const float* value_n, const float* logstd,
                          const float* noise, const double* rms_value, float eps, float* obses_t, float* priv_t,
                          float* actions_t, float* neglogp_t, float* values_t, float* mus_t, float* sigmas_t,
                          float* actions_clamped, float* values_out, igi_stream_t stream) {
  return fail(igi::rollout_act_store(n_envs, obs_dim, priv_dim, act_dim, obs, priv, mu, value_n, logstd, noise,
                                     rms_value, eps, obses_t, priv_t, actions_t, neglogp_t, values_t, mus_t, sigmas_t,
                                     actions_clamped, values_out, S(stream)),
              "igi_rollout_act_store");
}

int igi_rollout_env_store(int64_t n_envs, const float* rewards, const uint8_t* dones, const float* values,
                          const uint8_t* time_outs, const float* successes, float gamma, int bootstrap,
                          float* rewards_t, uint8_t* dones_t, float* cur_rewards, float* cur_lengths,
                          float* cur_success, float* meter, igi_stream_t stream) {
  return fail(igi::rollout_env_store(n_envs, rewards, dones, values, time_outs, successes, gamma, bootstrap,
                                     rewards_t, dones_t, cur_rewards, cur_lengths, cur_success, meter, S(stream)),
              "igi_rollout_env_store");
}

size_t igi_bc_loss_workspace_bytes(void) { return sizeof(double) * igi::BC_BLOCKS; }

int igi_bc_loss(const float* mu, const float* teacher_actions, const float* weights, int64_t rows, int act_dim,
                float* loss, float* dmu, void* workspace, size_t workspace_bytes, igi_stream_t stream) {
  return fail(igi::bc_loss(mu, teacher_actions, weights, rows, act_dim, loss, dmu, workspace, workspace_bytes, S(stream)),
              "igi_bc_loss");
}

int64_t igi_teacher_param_count(const igi_teacher_cfg* cfg) {
  igi::TeacherPlan p;
  int rc = igi::make_plan(cfg, &p);
  if (rc) return fail(rc, "igi_teacher_param_count");
  return p.P;
}

int igi_teacher_param_offsets(const igi_teacher_cfg* cfg, int64_t* off, int64_t* sz, int max_tensors) {
  igi::TeacherPlan p;
  int rc = igi::make_plan(cfg, &p);
  if (rc) return fail(rc, "igi_teacher_param_offsets");
  int n = 0;
  auto put = [&](long long o, long long s) {
    if (n < max_tensors) { if (off) off[n] = o; if (sz) sz[n] = s; }
    ++n;
  };
  put(p.o_sigma, p.act);
  for (int l = 0; l < p.npl; ++l) { put(p.o_envW[l], (long long)p.pu[l] * igi::env_in(p, l)); put(p.o_envB[l], p.pu[l]); }
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < p.nl; ++l) {
      put(p.o_acW[l] + net * p.ac_block, (long long)p.u[l] * igi::ac_in(p, l));
      put(p.o_acB[l] + net * p.ac_block, p.u[l]);
    }
  const int H = p.u[p.nl - 1];
  put(p.o_valW, H); put(p.o_valB, 1); put(p.o_muW, (long long)p.act * H); put(p.o_muB, p.act);
  return n;
}

size_t igi_teacher_workspace_bytes(const igi_teacher_cfg* cfg) {
  igi::TeacherPlan p;
  if (igi::make_plan(cfg, &p)) return 0;
  return p.w_total;
}

int igi_teacher_prepare(const igi_teacher_cfg* cfg, const igi_rollout* ro, const igi_teacher_state* st,
                        int normalize_value, igi_stream_t stream) {
  return fail(igi::teacher_prepare(cfg, ro, st, normalize_value, S(stream)), "igi_teacher_prepare");
}

int igi_teacher_fwd_bwd(const igi_teacher_cfg* cfg, const igi_rollout* ro, const igi_teacher_state* st,
                        int mb_index, int step_slot, igi_stream_t stream) {
  return fail(igi::teacher_fwd_bwd(cfg, ro, st, mb_index, step_slot, S(stream)), "igi_teacher_fwd_bwd");
}

int igi_teacher_apply(const igi_teacher_cfg* cfg, const igi_teacher_state* st, int step_slot,
                      int64_t adam_t, float grad_scale, igi_stream_t stream) {
  return fail(igi::teacher_apply(cfg, st, step_slot, adam_t, grad_scale, S(stream)), "igi_teacher_apply");
}

int igi_teacher_fwd_bwd_phase(const igi_teacher_cfg* cfg, const igi_rollout* ro, const igi_teacher_state* st,
                              int mb_index, int step_slot, int phase, igi_stream_t stream) {
  if (phase != 0 && phase != 1) return fail(IGI_E_BADARG, "igi_teacher_fwd_bwd_phase");
  return fail(igi::teacher_fwd_bwd(cfg, ro, st, mb_index, step_slot, S(stream), phase), "igi_teacher_fwd_bwd_phase");
}

int igi_teacher_grad_buckets(const igi_teacher_cfg* cfg, int64_t* offsets, int64_t* lengths) {
  igi::TeacherPlan p;
  int rc = igi::make_plan(cfg, &p);
  if (rc) return fail(rc, "igi_teacher_grad_buckets");
  if (!offsets || !lengths) return fail(IGI_E_BADARG, "igi_teacher_grad_buckets");
  const igi::GradBuckets b = igi::grad_buckets(p);
  for (int i = 0; i < 4; ++i) { offsets[i] = b.off[i]; lengths[i] = b.len[i]; }
  return 4;
}

int igi_comm_unique_id(void* id128) { return fail(igi::comm_unique_id(id128), "igi_comm_unique_id"); }
int igi_comm_create(const void* id128, int rank, int world, igi_comm_t* out) {
  return fail(igi::comm_create(id128, rank, world, out), "igi_comm_create");
}
int igi_comm_destroy(igi_comm_t comm) { return fail(igi::comm_destroy(comm), "igi_comm_destroy"); }
int igi_comm_rank(igi_comm_t comm) { return comm ? comm->rank : -1; }
int igi_comm_world(igi_comm_t comm) { return comm ? comm->world : -1; }
int igi_comm_count(igi_comm_t comm) {
  if (!comm || !comm->comm) return IGI_E_BADARG;
  int n = 0;
  const ncclResult_t r = ncclCommCount(comm->comm, &n);
  if (r != ncclSuccess) { snprintf(comm->err, sizeof(comm->err), "ncclCommCount: %s", ncclGetErrorString(r)); return IGI_E_COMM; }
  return n;
}
int igi_rccl_version(void) {
  int v = 0;
  return ncclGetVersion(&v) == ncclSuccess ? v : IGI_E_COMM;
}
const char* igi_comm_last_error(igi_comm_t comm) { return comm ? comm->err : igi::g_comm_create_err; }
int igi_comm_all_reduce_sum_f32(igi_comm_t comm, float* buf, int64_t n, igi_stream_t stream) {
  return fail(igi::comm_all_reduce_sum(comm, buf, n, S(stream)), "igi_comm_all_reduce_sum_f32");
}
int igi_comm_all_reduce_async_f32(igi_comm_t comm, float* buf, int64_t n, igi_stream_t compute_stream) {
  return fail(igi::comm_all_reduce_async(comm, buf, n, S(compute_stream)), "igi_comm_all_reduce_async_f32");
}
int igi_comm_join(igi_comm_t comm, igi_stream_t compute_stream) {
  return fail(igi::comm_join(comm, S(compute_stream)), "igi_comm_join");
}
int igi_comm_broadcast(igi_comm_t comm, void* buf, int64_t bytes, int root, igi_stream_t stream) {
  return fail(igi::comm_broadcast(comm, buf, bytes, root, S(stream)), "igi_comm_broadcast");
}
int igi_teacher_update_dp_rccl(const igi_teacher_cfg* cfg, const igi_rollout* ro, const igi_teacher_state* st,
                               int64_t adam_t0, igi_comm_t comm, int overlap, float* stats_sum, igi_stream_t stream) {
  return fail(igi::teacher_update_dp_rccl(cfg, ro, st, adam_t0, comm, overlap, stats_sum, S(stream)),
              "igi_teacher_update_dp_rccl");
}

int igi_teacher_update(const igi_teacher_cfg* cfg, const igi_rollout* ro, const igi_teacher_state* st,
                       int64_t adam_t0, igi_stream_t stream) {
  return fail(igi::teacher_update(cfg, ro, st, adam_t0, S(stream)), "igi_teacher_update");
}

int igi_teacher_update_dp(const igi_teacher_cfg* cfg, const igi_rollout* ro, const igi_teacher_state* st,
                          int64_t adam_t0, float grad_scale, igi_reduce_fn reduce, void* user, igi_stream_t stream) {
  if (!reduce) return fail(IGI_E_BADARG, "igi_teacher_update_dp");
  return fail(igi::teacher_update_dp(cfg, ro, st, adam_t0, grad_scale, reduce, user, S(stream)),
              "igi_teacher_update_dp");
}

int igi_teacher_infer(const igi_teacher_cfg* cfg, const igi_teacher_state* st, const float* obs,
                      const float* priv, int64_t rows, int normalize, float* mu, float* value, float* latent,
                      igi_stream_t stream) {
  return fail(igi::teacher_infer(cfg, st, obs, priv, rows, normalize, mu, value, latent, S(stream)),
              "igi_teacher_infer");
}

int igi_rollout_policy_step(const igi_teacher_cfg* cfg, const igi_teacher_state* st, const float* obs,
                            const float* priv, int64_t rows, int normalize, const float* noise,
                            const double* rms_value, float* obses_t, float* priv_t, float* actions_t,
                            float* neglogp_t, float* values_t, float* mus_t, float* sigmas_t, float* actions_clamped,
                            float* values_out, igi_stream_t stream) {
  return fail(igi::teacher_policy_step(cfg, st, obs, priv, rows, normalize, noise, rms_value, obses_t, priv_t,
                                       actions_t, neglogp_t, values_t, mus_t, sigmas_t, actions_clamped, values_out,
                                       S(stream)), "igi_rollout_policy_step");
}

size_t igi_clip_adam_workspace_bytes(void) { return sizeof(double) * 2 * igi::SUMSQ_BLOCKS; }

int igi_clip_adam(float* params, const float* grads, float* m, float* v, int64_t n, float max_norm, double lr,
                  double beta1, double beta2, double eps, int64_t t, float grad_scale, void* workspace,
                  size_t workspace_bytes, float* stats_out, igi_stream_t stream) {
  return igi_clip_adamw(params, grads, m, v, n, max_norm, lr, beta1, beta2, eps, 0.0, t, grad_scale, workspace,
                        workspace_bytes, stats_out, stream);
}

int igi_clip_adamw(float* params, const float* grads, float* m, float* v, int64_t n, float max_norm, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int64_t t, float grad_scale,
                   void* workspace, size_t workspace_bytes, float* stats_out, igi_stream_t stream) {
  return igi_clip_adam_l2(params, grads, m, v, n, max_norm, lr, beta1, beta2, eps, weight_decay, 0.0, t, grad_scale,
                          workspace, workspace_bytes, stats_out, stream);
}

int igi_clip_adam_l2(float* params, const float* grads, float* m, float* v, int64_t n, float max_norm, double lr,
                     double beta1, double beta2, double eps, double weight_decay, double l2, int64_t t,
                     float grad_scale, void* workspace, size_t workspace_bytes, float* stats_out,
                     igi_stream_t stream) {
  if (!params || !grads || !m || !v || n < 1 || t < 1 || !workspace || weight_decay < 0.0 || l2 < 0.0)
    return fail(IGI_E_BADARG, "igi_clip_adam_l2");
  if (workspace_bytes < igi_clip_adam_workspace_bytes()) return fail(IGI_E_WORKSPACE, "igi_clip_adam_l2");
  double* part = reinterpret_cast<double*>(workspace);
  hipStream_t s = S(stream);
  hipLaunchKernelGGL(igi::k_sumsq_stats, dim3(igi::SUMSQ_BLOCKS), dim3(256), 0, s, grads, params, (long long)n,
                     grad_scale, part, (const double*)nullptr, 0, 1, (float*)nullptr);
  const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
  int nb = (int)((n / 4 + 255) / 256);   // four elements per thread and trip on the 16-byte path
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(igi::k_clip_adam, dim3(nb), dim3(256), 0, s, params, grads, m, v, (long long)n, part,
                     grad_scale, max_norm, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                     (float)(lr / bc1), (float)sqrt(bc2), (float)eps, stats_out, (float)(1.0 - lr * weight_decay),
                     (float)l2);
  return fail((int)hipGetLastError(), "igi_clip_adam_l2");
}

size_t igi_linear_workspace_bytes(int64_t rows, int in_features, int out_features) {
  return igi::linear_workspace_bytes(rows, in_features, out_features);
}

int igi_linear_forward(const float* x, int ldx, const float* weight, const float* bias, float* y, int ldy,
                       int64_t rows, int in_features, int out_features, int activation, igi_stream_t stream) {
  return fail(igi::linear_forward(x, ldx, weight, bias, y, ldy, rows, in_features, out_features, activation,
                                  S(stream)), "igi_linear_forward");
}

int igi_linear_backward(const float* x, int ldx, const float* weight, const float* y, int ldy, const float* dy,
                        int lddy, float* dx, int lddx, float* dweight, float* dbias, int64_t rows,
                        int in_features, int out_features, int activation, void* workspace,
                        size_t workspace_bytes, igi_stream_t stream) {
  return fail(igi::linear_backward(x, ldx, weight, y, ldy, dy, lddy, dx, lddx, dweight, dbias, rows, in_features,
                                   out_features, activation, workspace, workspace_bytes, S(stream)),
              "igi_linear_backward");
}

int64_t igi_mlp_grad_floats(int n_layers, const int32_t* dims, int64_t* w_offsets, int64_t* b_offsets) {
  igi::MlpPlan p;
  static_assert(IGI_MLP_MAX_LAYERS == igi::MLP_MAX_LAYERS, "header and kernel agree on the chain length");
  int rc = igi::mlp_plan(1, n_layers, dims, &p);
  if (rc) return fail(rc, "igi_mlp_grad_floats");
  for (int l = 0; l < n_layers; ++l) {
    if (w_offsets) w_offsets[l] = p.o_w[l];
    if (b_offsets) b_offsets[l] = p.o_b[l];
  }
  return p.grad_floats;
}

size_t igi_mlp_workspace_bytes(int64_t rows, int n_layers, const int32_t* dims) {
  igi::MlpPlan p;
  if (igi::mlp_plan(rows, n_layers, dims, &p)) return 0;
  return p.w_total * sizeof(float) + 16;
}

int igi_mlp_backward(const float* x, int ldx, int64_t rows, int n_layers, const int32_t* dims, const int32_t* acts,
                     const float* const* weight, const float* const* y, const float* dy, float* dx, float* grads,
                     const int32_t* need_w, void* workspace, size_t workspace_bytes, igi_stream_t stream) {
  return fail(igi::mlp_backward(x, ldx, rows, n_layers, dims, acts, weight, y, dy, dx, grads, need_w, workspace,
                                workspace_bytes, S(stream)), "igi_mlp_backward");
}

int igi_mlp_forward(const float* x, int ldx, int64_t rows, int n_layers, const int32_t* dims, const int32_t* acts,
                    const float* const* weight, const float* const* bias, float* const* y, const int32_t* ldy,
                    igi_stream_t stream) {
  return fail(igi::mlp_forward(x, ldx, rows, n_layers, dims, acts, weight, bias, y, ldy, S(stream)), "igi_mlp_forward");
}

int64_t igi_depth_param_count(const igi_depth_cfg* cfg) {
  igi::DepthPlan p;
  int rc = igi::make_depth_plan(cfg, &p);
  if (rc) return fail(rc, "igi_depth_param_count");
  return p.P;
}

size_t igi_depth_workspace_bytes(const igi_depth_cfg* cfg) {
  igi::DepthPlan p;
  if (igi::make_depth_plan(cfg, &p)) return 0;
  return p.w_total;
}

int igi_depth_forward(const igi_depth_cfg* cfg, const float* x, const float* params, float* y, void* workspace,
                      size_t workspace_bytes, igi_stream_t stream) {
  return fail(igi::depth_forward(cfg, x, params, y, workspace, workspace_bytes, S(stream)), "igi_depth_forward");
}

int igi_depth_backward(const igi_depth_cfg* cfg, const float* x, const float* dy, const float* params, float* grads,
                       void* workspace, size_t workspace_bytes, igi_stream_t stream) {
  return fail(igi::depth_backward(cfg, x, dy, params, grads, workspace, workspace_bytes, S(stream)),
              "igi_depth_backward");
}

int64_t igi_token_param_count(const igi_token_cfg* cfg) {
  igi::TokenPlan p;
  int rc = igi::make_token_plan(cfg, &p);
  if (rc) return fail(rc, "igi_token_param_count");
  return p.per_layer * p.L;
}

size_t igi_token_workspace_bytes(const igi_token_cfg* cfg) {
  igi::TokenPlan p;
  if (igi::make_token_plan(cfg, &p)) return 0;
  return p.total_bytes;
}

int igi_token_forward(const igi_token_cfg* cfg, const float* x, const float* params, float* y, void* workspace,
                      size_t workspace_bytes, uint64_t seed, igi_stream_t stream) {
  return fail(igi::token_forward(cfg, x, params, y, workspace, workspace_bytes, seed, S(stream)), "igi_token_forward");
}

int igi_token_backward(const igi_token_cfg* cfg, const float* dy, const float* params, float* dx, float* grads,
                       void* workspace, size_t workspace_bytes, uint64_t seed, igi_stream_t stream) {
  return fail(igi::token_backward(cfg, dy, params, dx, grads, workspace, workspace_bytes, seed, S(stream)),
              "igi_token_backward");
}

int64_t igi_tactile_param_count(const igi_tactile_cfg* cfg) {
  igi::TactilePlan p;
  int rc = igi::make_tactile_plan(cfg, &p);
  if (rc) return fail(rc, "igi_tactile_param_count");
  return p.P;
}

size_t igi_tactile_workspace_bytes(const igi_tactile_cfg* cfg) {
  igi::TactilePlan p;
  if (igi::make_tactile_plan(cfg, &p)) return 0;
  return p.w_total;
}

int igi_tactile_forward(const igi_tactile_cfg* cfg, const float* x, const float* params, float* y, void* workspace,
                        size_t workspace_bytes, igi_stream_t stream) {
  return fail(igi::tactile_forward(cfg, x, params, y, workspace, workspace_bytes, S(stream)), "igi_tactile_forward");
}

int igi_tactile_activation_layout(const igi_tactile_cfg* cfg, int64_t offsets[3], int64_t rows[3]) {
  igi::TactilePlan p;
  int rc = igi::make_tactile_plan(cfg, &p);
  if (rc || !offsets || !rows) return fail(rc ? rc : IGI_E_BADARG, "igi_tactile_activation_layout");
  offsets[0] = (int64_t)p.w_a1; offsets[1] = (int64_t)p.w_a2; offsets[2] = (int64_t)p.w_a3;
  rows[0] = p.M1; rows[1] = p.M2; rows[2] = p.M3;
  return 0;
}

int igi_tactile_backward(const igi_tactile_cfg* cfg, const float* dy, const float* params, float* grads,
                         void* workspace, size_t workspace_bytes, igi_stream_t stream) {
  return fail(igi::tactile_backward(cfg, dy, params, grads, workspace, workspace_bytes, S(stream)),
              "igi_tactile_backward");
}

int igi_spatial_softargmax_forward(const float* x, int64_t rows, int h, int w, int normalize, float* out, float* stat,
                                   igi_stream_t stream) {
  return fail(igi::spatial_softargmax_forward(x, rows, h, w, normalize, out, stat, S(stream)),
              "igi_spatial_softargmax_forward");
}

int igi_spatial_softargmax_backward(const float* x, const float* out, const float* stat, const float* dout,
                                    int64_t rows, int h, int w, int normalize, float* dx, igi_stream_t stream) {
  return fail(igi::spatial_softargmax_backward(x, out, stat, dout, rows, h, w, normalize, dx, S(stream)),
              "igi_spatial_softargmax_backward");
}

int igi_gather_rows(int n, const float* const* src, const int64_t* width, float* const* dst, const int64_t* rows,
                    int64_t nrows, int64_t rows_total, igi_stream_t stream) {
  return fail(igi::gather_rows(n, src, width, dst, rows, nrows, rows_total, S(stream)), "igi_gather_rows");
}

int igi_cat_cols(int n, const float* const* part, const int64_t* width, float* cat, const float* add, int64_t rows,
                 igi_stream_t stream) {
  return fail(igi::cat_cols(n, const_cast<float* const*>(part), width, cat, add, rows, false, S(stream)), "igi_cat_cols");
}

int igi_split_cols(int n, float* const* part, const int64_t* width, const float* cat, int64_t rows, igi_stream_t stream) {
  return fail(igi::cat_cols(n, part, width, const_cast<float*>(cat), nullptr, rows, true, S(stream)), "igi_split_cols");
}

size_t igi_pointnet_workspace_bytes(int64_t batch) { return batch < 1 ? 0 : igi::pointnet_workspace_bytes(batch); }

int igi_pointnet_forward(const float* x, int64_t x_pitch, int64_t batch, int npoints, const float* params, float* y,
                         int32_t* argmax, igi_stream_t stream) {
  return fail(igi::pointnet_forward(x, x_pitch, batch, npoints, params, y, argmax, S(stream)), "igi_pointnet_forward");
}

int igi_pointnet_backward(const float* x, int64_t x_pitch, int64_t batch, int npoints, const float* params, const float* dy,
                          int64_t dy_pitch, const int32_t* argmax, float* grads, void* workspace, size_t workspace_bytes,
                          igi_stream_t stream) {
  return fail(igi::pointnet_backward(x, x_pitch, batch, npoints, params, dy, dy_pitch, argmax, grads, workspace, workspace_bytes,
                                     S(stream)), "igi_pointnet_backward");
}

size_t igi_pointnet_workspace_bytes_multi(int64_t batch, int nobj) {
  return (batch < 1 || nobj < 1 || nobj > igi::PN_MAX_OBJ) ? 0 : igi::pointnet_workspace_bytes_multi(batch, nobj);
}

int igi_pointnet_forward_multi(int nobj, const float* x, int64_t x_pitch, int64_t batch, const int32_t* x_off,
                               const int32_t* npoints, const float* const* params, float* y, int32_t* argmax,
                               igi_stream_t stream) {
  return fail(igi::pointnet_forward_multi(nobj, x, x_pitch, batch, x_off, npoints, params, y, argmax, S(stream)),
              "igi_pointnet_forward_multi");
}

int igi_pointnet_backward_multi(int nobj, const float* x, int64_t x_pitch, int64_t batch, const int32_t* x_off,
                                const int32_t* npoints, const float* const* params, const float* dy, int64_t dy_pitch,
                                const int32_t* argmax, float* grads, void* workspace, size_t workspace_bytes,
                                igi_stream_t stream) {
  return fail(igi::pointnet_backward_multi(nobj, x, x_pitch, batch, x_off, npoints, params, dy, dy_pitch, argmax, grads,
                                           workspace, workspace_bytes, S(stream)), "igi_pointnet_backward_multi");
}

}  // extern "C"
