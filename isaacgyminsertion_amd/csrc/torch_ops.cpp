// TORCH_LIBRARY(mi355ppo, ...) for C++ hosts: the dispatcher ops SURVEY.md section 8(b) spells out, registered from
// native code over the C ABI of libigi_hip.so (include/igi_ppo.h) -- a libtorch program that links / dlopens
// libigi_torch_ops.so calls torch.ops.mi355ppo.* without any Python.  Same names, same schemas (mutated arguments
// declared), same argument checks (TORCH_CHECK -> c10::Error / RuntimeError), same C entry points as the Python
// registration in isaacgyminsertion_amd/ops.py, which stays the one the Python package uses (it adds the fake kernels,
// the autograd formulas and the remaining student ops).  One process loads ONE of the two: both define namespace mi355ppo.
//
// Host code only (g++; no device code): tensors come from the caller's allocator, kernels are enqueued on
// the current HIP stream (PyTorch-ROCm's tensors carry device type "cuda": the guard / stream accessors are the
// ...MasqueradingAsCUDA ones), nothing synchronises.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <cstring>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/igi_ppo.h"

namespace {

using at::Tensor;

void check(const Tensor& t, const char* name, at::ScalarType dtype = at::kFloat) {
  TORCH_CHECK(t.defined(), name, ": expected a tensor");
  TORCH_CHECK(t.is_cuda(), name, ": expected a HIP (cuda) tensor, got device ", t.device(), " (there is no CPU path)");
  TORCH_CHECK(t.scalar_type() == dtype, name, ": expected dtype ", dtype, ", got ", t.scalar_type());
  TORCH_CHECK(t.is_contiguous(), name, ": expected a contiguous tensor");
}
// a (rows, ...) tensor whose rows are dense but `pitch` floats apart (a column slice of a wider tensor, read in place)
void check_rows(const Tensor& t, const char* name, at::ScalarType dtype = at::kFloat) {
  TORCH_CHECK(t.defined(), name, ": expected a tensor");
  TORCH_CHECK(t.is_cuda(), name, ": expected a HIP (cuda) tensor, got device ", t.device(), " (there is no CPU path)");
  TORCH_CHECK(t.scalar_type() == dtype, name, ": expected dtype ", dtype, ", got ", t.scalar_type());
  TORCH_CHECK(t.dim() >= 2 && t.size(0) >= 1, name, ": expected at least one row of a (rows, ...) tensor, got shape ", t.sizes());
  TORCH_CHECK(t[0].is_contiguous() && (t.size(0) == 1 || t.stride(0) >= t[0].numel()), name,
              ": expected dense rows (any row pitch)");
}
int64_t pitch_of(const Tensor& t) { return t.size(0) > 1 ? t.stride(0) : 0; }
void rc(int code, const char* what) {
  TORCH_CHECK(code == 0, "libigi_hip ", what, " failed (rc=", code, "): ", igi_last_error());
}
igi_stream_t stream_of(const Tensor& t) {
  return reinterpret_cast<igi_stream_t>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream());
}
float* fp(const Tensor& t) { return t.data_ptr<float>(); }
float* fpo(const c10::optional<Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; }

// ---- teacher marshalling: (int[] icfg, float[] fcfg) <-> struct igi_teacher_cfg, tensor lists <-> the structs
igi_teacher_cfg unpack_cfg(at::IntArrayRef ic, at::ArrayRef<double> fc) {
  constexpr int M = IGI_MAX_LAYERS;
  TORCH_CHECK((int)ic.size() == 8 + 2 * M && fc.size() == 12, "teacher cfg: expected ", 8 + 2 * M, " ints and 12 floats, got ",
              ic.size(), " and ", fc.size());
  igi_teacher_cfg c;
  std::memset(&c, 0, sizeof(c));
  c.obs_dim = (int32_t)ic[0]; c.priv_dim = (int32_t)ic[1]; c.act_dim = (int32_t)ic[2]; c.n_priv_layers = (int32_t)ic[3];
  for (int i = 0; i < M; ++i) { c.priv_units[i] = (int32_t)ic[4 + i]; c.units[i] = (int32_t)ic[5 + M + i]; }
  c.n_layers = (int32_t)ic[4 + M];
  c.num_envs = (int32_t)ic[5 + 2 * M]; c.horizon = (int32_t)ic[6 + 2 * M]; c.mini_epochs = (int32_t)ic[7 + 2 * M];
  c.gamma = fc[0]; c.tau = fc[1]; c.lr = fc[2]; c.beta1 = fc[3]; c.beta2 = fc[4]; c.adam_eps = fc[5];
  c.e_clip = (float)fc[6]; c.critic_coef = (float)fc[7]; c.entropy_coef = (float)fc[8]; c.bounds_loss_coef = (float)fc[9];
  c.grad_norm = (float)fc[10]; c.rms_eps = (float)fc[11];
  TORCH_CHECK(c.obs_dim >= 1 && c.priv_dim >= 1 && c.act_dim >= 1 && c.num_envs >= 1 && c.horizon >= 1 && c.mini_epochs >= 1 &&
                  c.n_layers >= 1 && c.n_layers <= M && c.n_priv_layers >= 1 && c.n_priv_layers <= M,
              "teacher cfg: non-positive dimension or unsupported layer count");
  return c;
}

igi_teacher_state state_struct(at::TensorList st, const igi_teacher_cfg& c) {
  static const char* names[16] = {"params", "grads", "adam_m", "adam_v", "rms_obs", "rms_priv", "rms_value", "perm",
                                  "returns_raw", "advantages", "values_n", "returns_n", "mus_w", "sigmas_w", "stats",
                                  "workspace"};
  TORCH_CHECK(st.size() == 16, "state: expected 16 tensors (struct igi_teacher_state field order), got ", st.size());
  const int64_t P = igi_teacher_param_count(&c);
  TORCH_CHECK(P > 0, "teacher cfg rejected by the library: ", igi_last_error());
  const int64_t T = c.horizon, N = c.num_envs, A = c.act_dim;
  const int64_t want[14] = {P, P, P, P, 2 * c.obs_dim + 1, 2 * c.priv_dim + 1, 3, T * N, T * N, T * N, T * N, T * N,
                            T * N * A, T * N * A};
  for (int i = 0; i < 16; ++i) {
    const at::ScalarType dt = (i >= 4 && i <= 6) ? at::kDouble : (i == 7 ? at::kLong : (i == 15 ? at::kByte : at::kFloat));
    const std::string nm = std::string("state.") + names[i];
    check(st[i], nm.c_str(), dt);
    TORCH_CHECK(st[i].device() == st[0].device(), nm, ": all arguments must share one device");
    if (i < 14) TORCH_CHECK(st[i].numel() == want[i], nm, ": expected ", want[i], " elements, got ", st[i].numel());
  }
  TORCH_CHECK(st[14].dim() == 2 && st[14].size(1) == IGI_STATS_PER_STEP, "state.stats: expected (steps, ", IGI_STATS_PER_STEP, ")");
  const size_t need = igi_teacher_workspace_bytes(&c);
  TORCH_CHECK((size_t)st[15].numel() >= need, "state.workspace: ", st[15].numel(), " bytes, the configuration needs ", need);
  igi_teacher_state s;
  s.params = fp(st[0]); s.grads = fp(st[1]); s.adam_m = fp(st[2]); s.adam_v = fp(st[3]);
  s.rms_obs = st[4].data_ptr<double>(); s.rms_priv = st[5].data_ptr<double>(); s.rms_value = st[6].data_ptr<double>();
  s.perm = st[7].data_ptr<int64_t>();
  s.returns_raw = fp(st[8]); s.advantages = fp(st[9]); s.values_n = fp(st[10]); s.returns_n = fp(st[11]);
  s.mus_w = fp(st[12]); s.sigmas_w = fp(st[13]); s.stats = fp(st[14]);
  s.workspace = st[15].data_ptr(); s.workspace_bytes = (size_t)st[15].numel();
  return s;
}

igi_rollout rollout_struct(at::TensorList ro, const igi_teacher_cfg& c, const at::Device& dev) {
  static const char* names[10] = {"obses", "priv_info", "rewards", "values", "neglogpacs", "dones", "actions", "mus",
                                  "sigmas", "last_values"};
  TORCH_CHECK(ro.size() == 10, "rollout: expected 10 tensors (struct igi_rollout field order), got ", ro.size());
  const int64_t T = c.horizon, N = c.num_envs, A = c.act_dim;
  const int64_t want[10] = {T * N * c.obs_dim, T * N * c.priv_dim, T * N, T * N, T * N, T * N, T * N * A, T * N * A, T * N * A, N};
  for (int i = 0; i < 10; ++i) {
    const std::string nm = std::string("rollout.") + names[i];
    check(ro[i], nm.c_str(), i == 5 ? at::kByte : at::kFloat);
    TORCH_CHECK(ro[i].device() == dev, nm, ": all arguments must share one device");
    TORCH_CHECK(ro[i].numel() == want[i], nm, ": expected ", want[i], " elements, got ", ro[i].numel());
  }
  igi_rollout r;
  r.obses = fp(ro[0]); r.priv_info = fp(ro[1]); r.rewards = fp(ro[2]); r.values = fp(ro[3]); r.neglogpacs = fp(ro[4]);
  r.dones = ro[5].data_ptr<uint8_t>(); r.actions = fp(ro[6]); r.mus = fp(ro[7]); r.sigmas = fp(ro[8]); r.last_values = fp(ro[9]);
  return r;
}

// ---- teacher ops (frozen_ppo.py:495-646, experience.py:242-263)
void gae_advnorm(at::TensorList rollout, at::TensorList state, at::IntArrayRef icfg, at::ArrayRef<double> fcfg,
                 bool normalize_value) {
  const igi_teacher_cfg c = unpack_cfg(icfg, fcfg);
  const igi_teacher_state s = state_struct(state, c);
  const igi_rollout r = rollout_struct(rollout, c, state[0].device());
  c10::hip::HIPGuardMasqueradingAsCUDA g(state[0].device());
  rc(igi_teacher_prepare(&c, &r, &s, normalize_value ? 1 : 0, stream_of(state[0])), "igi_teacher_prepare");
}
void ppo_minibatch_fwd_bwd(at::TensorList rollout, at::TensorList state, at::IntArrayRef icfg, at::ArrayRef<double> fcfg,
                           int64_t mb_index, int64_t step_slot, int64_t phase) {
  const igi_teacher_cfg c = unpack_cfg(icfg, fcfg);
  const igi_teacher_state s = state_struct(state, c);
  const igi_rollout r = rollout_struct(rollout, c, state[0].device());
  TORCH_CHECK(phase >= -1 && phase <= 1, "phase: expected -1, 0 or 1, got ", phase);
  c10::hip::HIPGuardMasqueradingAsCUDA g(state[0].device());
  if (phase < 0) rc(igi_teacher_fwd_bwd(&c, &r, &s, (int)mb_index, (int)step_slot, stream_of(state[0])), "igi_teacher_fwd_bwd");
  else rc(igi_teacher_fwd_bwd_phase(&c, &r, &s, (int)mb_index, (int)step_slot, (int)phase, stream_of(state[0])),
          "igi_teacher_fwd_bwd_phase");
}
void ppo_clip_adam(at::TensorList state, at::IntArrayRef icfg, at::ArrayRef<double> fcfg, int64_t step_slot, int64_t adam_t,
                   double grad_scale) {
  const igi_teacher_cfg c = unpack_cfg(icfg, fcfg);
  const igi_teacher_state s = state_struct(state, c);
  c10::hip::HIPGuardMasqueradingAsCUDA g(state[0].device());
  rc(igi_teacher_apply(&c, &s, (int)step_slot, adam_t, (float)grad_scale, stream_of(state[0])), "igi_teacher_apply");
}
void ppo_update(at::TensorList rollout, at::TensorList state, at::IntArrayRef icfg, at::ArrayRef<double> fcfg, int64_t adam_t0) {
  const igi_teacher_cfg c = unpack_cfg(icfg, fcfg);
  const igi_teacher_state s = state_struct(state, c);
  const igi_rollout r = rollout_struct(rollout, c, state[0].device());
  c10::hip::HIPGuardMasqueradingAsCUDA g(state[0].device());
  rc(igi_teacher_update(&c, &r, &s, adam_t0, stream_of(state[0])), "igi_teacher_update");
}
std::tuple<Tensor, Tensor, Tensor> actor_critic_infer(at::TensorList state, at::IntArrayRef icfg, at::ArrayRef<double> fcfg,
                                                      const Tensor& obs, const Tensor& priv, bool normalize, bool want_latent) {
  const igi_teacher_cfg c = unpack_cfg(icfg, fcfg);
  const igi_teacher_state s = state_struct(state, c);
  check(obs, "obs"); check(priv, "priv");
  TORCH_CHECK(obs.dim() == 2 && obs.size(1) == c.obs_dim, "obs: expected (rows, ", c.obs_dim, ")");
  TORCH_CHECK(priv.dim() == 2 && priv.size(0) == obs.size(0) && priv.size(1) == c.priv_dim, "priv: expected (rows, ", c.priv_dim, ")");
  const int64_t rows = obs.size(0);
  const int lat = c.priv_units[c.n_priv_layers - 1];
  Tensor mu = at::empty({rows, c.act_dim}, obs.options()), val = at::empty({rows, 1}, obs.options());
  Tensor latent = at::empty({want_latent ? rows : 0, lat}, obs.options());
  c10::hip::HIPGuardMasqueradingAsCUDA g(obs.device());
  rc(igi_teacher_infer(&c, &s, fp(obs), fp(priv), rows, normalize ? 1 : 0, fp(mu), fp(val), want_latent ? fp(latent) : nullptr,
                       stream_of(obs)), "igi_teacher_infer");
  return {mu, val, latent};
}
void rollout_policy_step(at::TensorList state, at::IntArrayRef icfg, at::ArrayRef<double> fcfg, const Tensor& obs,
                         const Tensor& priv, bool normalize, const Tensor& noise, const c10::optional<Tensor>& rms_value,
                         const c10::optional<Tensor>& obses_t, const c10::optional<Tensor>& priv_t, Tensor actions_t,
                         Tensor neglogp_t, Tensor values_t, Tensor mus_t, Tensor sigmas_t, Tensor actions_clamped,
                         Tensor values_out) {
  const igi_teacher_cfg c = unpack_cfg(icfg, fcfg);
  const igi_teacher_state s = state_struct(state, c);
  check(obs, "obs"); check(priv, "priv"); check(noise, "noise");
  const int64_t n = obs.size(0), a = c.act_dim;
  TORCH_CHECK(obs.dim() == 2 && obs.size(1) == c.obs_dim && priv.dim() == 2 && priv.size(0) == n && priv.size(1) == c.priv_dim &&
                  noise.numel() == n * a, "obs / priv / noise: expected (n, obs), (n, priv), (n, act)");
  for (const Tensor* t : {&actions_t, &mus_t, &sigmas_t, &actions_clamped}) { check(*t, "action outputs"); TORCH_CHECK(t->numel() == n * a, "action outputs: expected n * act elements"); }
  for (const Tensor* t : {&neglogp_t, &values_t, &values_out}) { check(*t, "per-env outputs"); TORCH_CHECK(t->numel() == n, "per-env outputs: expected n elements"); }
  if (rms_value.has_value() && rms_value->defined()) { check(*rms_value, "rms_value", at::kDouble); TORCH_CHECK(rms_value->numel() == 3, "rms_value: [mean, var, count]"); }
  c10::hip::HIPGuardMasqueradingAsCUDA g(obs.device());
  rc(igi_rollout_policy_step(&c, &s, fp(obs), fp(priv), n, normalize ? 1 : 0, fp(noise),
                             rms_value.has_value() && rms_value->defined() ? rms_value->data_ptr<double>() : nullptr, fpo(obses_t),
                             fpo(priv_t), fp(actions_t), fp(neglogp_t), fp(values_t), fp(mus_t), fp(sigmas_t), fp(actions_clamped),
                             fp(values_out), stream_of(obs)), "igi_rollout_policy_step");
}

// ---- normaliser / optimizer / loss (running_mean_std.py:60-93; ext_adapt.py:812-819, 853-855)
Tensor rms_update_normalize(const Tensor& x, Tensor state, double eps, bool train, bool unnorm) {
  check(x, "x");
  TORCH_CHECK(x.dim() == 2, "x: expected (rows, D)");
  const int64_t rows = x.size(0), D = x.size(1);
  check(state, "state", at::kDouble);
  TORCH_CHECK(state.numel() == 2 * D + 1 && state.device() == x.device(), "state: expected 2 * D + 1 doubles on x's device");
  Tensor y = at::empty_like(x);
  c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
  Tensor ws = at::empty({(int64_t)igi_rms_workspace_bytes(rows > 0 ? rows : 1, (int)D)}, x.options().dtype(at::kByte));
  if (rows > 0)
    rc(igi_rms_forward(fp(x), fp(y), rows, (int)D, state.data_ptr<double>(), (float)eps, train ? 1 : 0, unnorm ? 1 : 0,
                       ws.data_ptr(), (size_t)ws.numel(), stream_of(x)), "igi_rms_forward");
  return y;
}
void clip_adam_step(Tensor params, const Tensor& grads, Tensor exp_avg, Tensor exp_avg_sq, double max_norm, double lr,
                    double beta1, double beta2, double eps, double weight_decay, double l2, int64_t t, double grad_scale,
                    Tensor stats) {
  check(params, "params"); check(grads, "grads"); check(exp_avg, "exp_avg"); check(exp_avg_sq, "exp_avg_sq"); check(stats, "stats");
  const int64_t n = params.numel();
  TORCH_CHECK(grads.numel() == n && exp_avg.numel() == n && exp_avg_sq.numel() == n && stats.numel() >= 8, "clip_adam_step: sizes");
  c10::hip::HIPGuardMasqueradingAsCUDA g(params.device());
  Tensor ws = at::empty({(int64_t)igi_clip_adam_workspace_bytes()}, params.options().dtype(at::kByte));
  rc(igi_clip_adam_l2(fp(params), fp(grads), fp(exp_avg), fp(exp_avg_sq), n, (float)max_norm, lr, beta1, beta2, eps,
                      weight_decay, l2, t, (float)grad_scale, ws.data_ptr(), (size_t)ws.numel(), fp(stats), stream_of(params)),
     "igi_clip_adam_l2");
}
std::tuple<Tensor, Tensor> bc_loss_fwd_bwd(const Tensor& mu, const Tensor& teacher_actions, const Tensor& weights, bool want_grad) {
  check(mu, "mu"); check(teacher_actions, "teacher_actions"); check(weights, "weights");
  TORCH_CHECK(mu.dim() == 2 && teacher_actions.sizes() == mu.sizes() && weights.numel() == mu.size(1), "bc_loss: shapes");
  Tensor loss = at::empty({1}, mu.options());
  Tensor dmu = want_grad ? at::empty_like(mu) : at::empty({0, mu.size(1)}, mu.options());
  c10::hip::HIPGuardMasqueradingAsCUDA g(mu.device());
  Tensor ws = at::empty({(int64_t)igi_bc_loss_workspace_bytes()}, mu.options().dtype(at::kByte));
  rc(igi_bc_loss(fp(mu), fp(teacher_actions), fp(weights), mu.size(0), (int)mu.size(1), fp(loss), want_grad ? fp(dmu) : nullptr,
                 ws.data_ptr(), (size_t)ws.numel(), stream_of(mu)), "igi_bc_loss");
  return {loss.reshape({}), dmu};
}

// ---- encoders (tactile_cnn.py:7-79, pointnets.py:12-42)
std::tuple<Tensor, Tensor> tactile_cnn_fwd(const Tensor& x, const Tensor& params, int64_t latent_dim) {
  check(x, "x"); check(params, "params");
  TORCH_CHECK(x.dim() == 4 && x.size(1) == 3 && x.size(0) % 32 == 0, "x: expected (32k, 3, H, W)");
  igi_tactile_cfg cfg{(int32_t)x.size(0), (int32_t)x.size(2), (int32_t)x.size(3), (int32_t)latent_dim};
  const int64_t n = igi_tactile_param_count(&cfg);
  const size_t nbytes = igi_tactile_workspace_bytes(&cfg);
  TORCH_CHECK(n > 0 && nbytes > 0, "tactile configuration rejected: ", igi_last_error());
  TORCH_CHECK(params.numel() == n && params.device() == x.device(), "params: expected ", n, " floats on x's device");
  c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
  Tensor ws = at::empty({(int64_t)nbytes}, x.options().dtype(at::kByte));
  Tensor y = at::empty({x.size(0), latent_dim}, x.options());
  rc(igi_tactile_forward(&cfg, fp(x), fp(params), fp(y), ws.data_ptr(), nbytes, stream_of(x)), "igi_tactile_forward");
  return {y, ws};
}
Tensor tactile_cnn_bwd(const Tensor& dy, const Tensor& params, Tensor ws, int64_t height, int64_t width) {
  check(dy, "dy"); check(params, "params"); check(ws, "ws", at::kByte);
  TORCH_CHECK(dy.dim() == 2, "dy: expected (B, latent)");
  igi_tactile_cfg cfg{(int32_t)dy.size(0), (int32_t)height, (int32_t)width, (int32_t)dy.size(1)};
  Tensor grads = at::empty_like(params);
  c10::hip::HIPGuardMasqueradingAsCUDA g(dy.device());
  rc(igi_tactile_backward(&cfg, fp(dy), fp(params), fp(grads), ws.data_ptr(), (size_t)ws.numel(), stream_of(dy)),
     "igi_tactile_backward");
  return grads;
}
std::tuple<Tensor, Tensor> spatial_softargmax_fwd(const Tensor& x, bool normalize) {
  check(x, "x");
  TORCH_CHECK(x.dim() == 4, "x: expected (B, C, H, W)");
  const int64_t b = x.size(0), c = x.size(1);
  Tensor out = at::empty({b, 2 * c}, x.options()), stat = at::empty({b * c, 2}, x.options());
  c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
  rc(igi_spatial_softargmax_forward(fp(x), b * c, (int)x.size(2), (int)x.size(3), normalize ? 1 : 0, fp(out), fp(stat),
                                    stream_of(x)), "igi_spatial_softargmax_forward");
  return {out, stat};
}
Tensor spatial_softargmax_bwd(const Tensor& x, const Tensor& out, const Tensor& stat, const Tensor& dout, bool normalize) {
  check(x, "x"); check(out, "out"); check(stat, "stat"); check(dout, "dout");
  TORCH_CHECK(x.dim() == 4 && out.numel() == 2 * x.size(0) * x.size(1) && dout.numel() == out.numel() && stat.numel() == out.numel(),
              "spatial_softargmax_bwd: shapes");
  Tensor dx = at::empty_like(x);
  c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
  rc(igi_spatial_softargmax_backward(fp(x), fp(out), fp(stat), fp(dout), x.size(0) * x.size(1), (int)x.size(2), (int)x.size(3),
                                     normalize ? 1 : 0, fp(dx), stream_of(x)), "igi_spatial_softargmax_backward");
  return dx;
}
std::tuple<Tensor, Tensor> pointnet_max_fwd(const Tensor& x, const Tensor& params) {
  check_rows(x, "x"); check(params, "params");
  TORCH_CHECK(x.dim() == 3 && x.size(2) == 3 && x.size(0) >= 1 && x.size(1) >= 1, "x: expected (B, N, 3)");
  TORCH_CHECK(params.numel() == 64 * 3 + 64 + 256 * 64 + 256 && params.device() == x.device(), "params: 16896 floats on x's device");
  Tensor y = at::empty({x.size(0), 256}, x.options()), idx = at::empty({x.size(0), 256}, x.options().dtype(at::kInt));
  c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
  rc(igi_pointnet_forward(fp(x), pitch_of(x), x.size(0), (int)x.size(1), fp(params), fp(y), idx.data_ptr<int32_t>(), stream_of(x)),
     "igi_pointnet_forward");
  return {y, idx};
}
Tensor pointnet_max_bwd(const Tensor& x, const Tensor& params, const Tensor& dy, const Tensor& idx) {
  check_rows(x, "x"); check(params, "params"); check_rows(dy, "dy"); check(idx, "idx", at::kInt);
  TORCH_CHECK(x.dim() == 3 && dy.dim() == 2 && dy.size(0) == x.size(0) && dy.size(1) == 256 && idx.numel() == dy.numel(),
              "pointnet_max_bwd: expected x (B, points, 3), dy (B, 256), idx (B, 256); got ", x.sizes(), ", ", dy.sizes(), ", ", idx.sizes());
  Tensor grads = at::empty_like(params);
  c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
  Tensor ws = at::empty({(int64_t)igi_pointnet_workspace_bytes(x.size(0))}, x.options().dtype(at::kByte));
  rc(igi_pointnet_backward(fp(x), pitch_of(x), x.size(0), (int)x.size(1), fp(params), fp(dy), pitch_of(dy), idx.data_ptr<int32_t>(), fp(grads),
                           ws.data_ptr(), (size_t)ws.numel(), stream_of(x)), "igi_pointnet_backward");
  return grads;
}

}  // namespace

// Schemas: character for character those of isaacgyminsertion_amd/ops.py (tests/test_torch_library_cpp.py compares them).
TORCH_LIBRARY(mi355ppo, m) {
  m.def("gae_advnorm(Tensor[] rollout, Tensor(a!)[] state, int[] icfg, float[] fcfg, bool normalize_value) -> ()");
  m.def("ppo_minibatch_fwd_bwd(Tensor[] rollout, Tensor(a!)[] state, int[] icfg, float[] fcfg, int mb_index, int step_slot, int phase) -> ()");
  m.def("ppo_clip_adam(Tensor(a!)[] state, int[] icfg, float[] fcfg, int step_slot, int adam_t, float grad_scale) -> ()");
  m.def("ppo_update(Tensor[] rollout, Tensor(a!)[] state, int[] icfg, float[] fcfg, int adam_t0) -> ()");
  m.def("actor_critic_infer(Tensor(a!)[] state, int[] icfg, float[] fcfg, Tensor obs, Tensor priv, bool normalize, bool want_latent) -> (Tensor, Tensor, Tensor)");
  m.def("rollout_policy_step(Tensor(a!)[] state, int[] icfg, float[] fcfg, Tensor obs, Tensor priv, bool normalize, Tensor noise, Tensor? rms_value, Tensor(b!)? obses_t, Tensor(c!)? priv_t, Tensor(d!) actions_t, Tensor(e!) neglogp_t, Tensor(f!) values_t, Tensor(g!) mus_t, Tensor(h!) sigmas_t, Tensor(i!) actions_clamped, Tensor(j!) values_out) -> ()");
  m.def("rms_update_normalize(Tensor x, Tensor(a!) state, float eps, bool train, bool unnorm) -> Tensor");
  m.def("clip_adam_step(Tensor(a!) params, Tensor grads, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, float l2, int t, float grad_scale, Tensor(d!) stats) -> ()");
  m.def("bc_loss_fwd_bwd(Tensor mu, Tensor teacher_actions, Tensor weights, bool want_grad) -> (Tensor, Tensor)");
  m.def("tactile_cnn_fwd(Tensor x, Tensor params, int latent_dim) -> (Tensor, Tensor)");
  m.def("tactile_cnn_bwd(Tensor dy, Tensor params, Tensor(a!) ws, int height, int width) -> Tensor");
  m.def("spatial_softargmax_fwd(Tensor x, bool normalize) -> (Tensor, Tensor)");
  m.def("spatial_softargmax_bwd(Tensor x, Tensor out, Tensor stat, Tensor dout, bool normalize) -> Tensor");
  m.def("pointnet_max_fwd(Tensor x, Tensor params) -> (Tensor, Tensor)");
  m.def("pointnet_max_bwd(Tensor x, Tensor params, Tensor dy, Tensor idx) -> Tensor");
}

TORCH_LIBRARY_IMPL(mi355ppo, CompositeExplicitAutograd, m) {
  m.impl("gae_advnorm", gae_advnorm);
  m.impl("ppo_minibatch_fwd_bwd", ppo_minibatch_fwd_bwd);
  m.impl("ppo_clip_adam", ppo_clip_adam);
  m.impl("ppo_update", ppo_update);
  m.impl("actor_critic_infer", actor_critic_infer);
  m.impl("rollout_policy_step", rollout_policy_step);
  m.impl("rms_update_normalize", rms_update_normalize);
  m.impl("clip_adam_step", clip_adam_step);
  m.impl("bc_loss_fwd_bwd", bc_loss_fwd_bwd);
  m.impl("tactile_cnn_fwd", tactile_cnn_fwd);
  m.impl("tactile_cnn_bwd", tactile_cnn_bwd);
  m.impl("spatial_softargmax_fwd", spatial_softargmax_fwd);
  m.impl("spatial_softargmax_bwd", spatial_softargmax_bwd);
  m.impl("pointnet_max_fwd", pointnet_max_fwd);
  m.impl("pointnet_max_bwd", pointnet_max_bwd);
}
