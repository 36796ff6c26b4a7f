// Rollout-side bookkeeping of PPO.play_steps as two launches per environment step instead of ~25 framework
// kernels (frozen_ppo.py:343-366 sampling / value de-normalisation, :655-683 buffer writes and reward shaping,
// :685-700 episode statistics).  HBM-bound elementwise work over the env batch; a thread owns an environment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace igi {

constexpr float ROLL_LOG_SQRT_2PI = 0.918938533204672741780329736406f;

// Sample a = mu + exp(logstd) * noise, neglogp = sum_q [(a-mu)^2 / (2 sigma^2) + log(sigma) + log(sqrt(2 pi))],
// de-normalise the value (sqrt(var+eps) * clamp(v, +-5) + mean; running_mean_std.py:84-86) and file the step:
// slot pointers address row block t of the time-major arena tensors.  actions_clamped = clamp(a, +-1) for env.step.
__global__ __launch_bounds__(256) void k_rollout_act_store(
    int N, int obs_dim, int priv_dim, int act, const float* __restrict__ obs, const float* __restrict__ priv,
    const float* __restrict__ mu, const float* __restrict__ value_n, const float* __restrict__ logstd,
    const float* __restrict__ noise, const double* __restrict__ rms_value, float eps, float* __restrict__ obses_t,
    float* __restrict__ priv_t, float* __restrict__ actions_t, float* __restrict__ nlp_t, float* __restrict__ values_t,
    float* __restrict__ mus_t, float* __restrict__ sigmas_t, float* __restrict__ actions_clamped,
    float* __restrict__ values_out) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = gid; e < (long long)N * obs_dim; e += stride) obses_t[e] = obs[e];
  for (long long e = gid; e < (long long)N * priv_dim; e += stride) priv_t[e] = priv[e];
  float vm = 0.f, vd = 1.f;
  if (rms_value) { vm = (float)rms_value[0]; vd = sqrtf((float)rms_value[1] + eps); }
  for (long long n = gid; n < N; n += stride) {
    float nlp = 0.f;
    for (int q = 0; q < act; ++q) {
      const float m = mu[n * act + q];
      const float sig = expf(m * 0.f + logstd[q]);
      const float a = m + sig * noise[n * act + q];
      const float x = a - m;
      nlp += ((x * x) / (2.0f * (sig * sig)) + logf(sig)) + ROLL_LOG_SQRT_2PI;
      actions_t[n * act + q] = a;
      mus_t[n * act + q] = m;
      sigmas_t[n * act + q] = sig;
      actions_clamped[n * act + q] = fminf(fmaxf(a, -1.0f), 1.0f);
    }
    nlp_t[n] = nlp;
    float v = value_n[n];
    if (rms_value) v = vd * fminf(fmaxf(v, -5.0f), 5.0f) + vm;
    values_t[n] = v;
    values_out[n] = v;
  }
}

// After env.step: dones / shaped reward into the arena (0.01 r + gamma V timeout when bootstrapping,
// frozen_ppo.py:677-681), running episode accumulators, and the sums the windowed meters need from the
// episodes that just ended: meter[0..3] += {sum reward, sum length, sum success, count} over done envs.
__global__ __launch_bounds__(256) void k_rollout_env_store(
    int N, const float* __restrict__ rewards, const uint8_t* __restrict__ dones, const float* __restrict__ values,
    const uint8_t* __restrict__ time_outs, const float* __restrict__ successes, float gamma, int bootstrap,
    float* __restrict__ rewards_t, uint8_t* __restrict__ dones_t, float* __restrict__ cur_rewards,
    float* __restrict__ cur_lengths, float* __restrict__ cur_success, float* __restrict__ meter) {
  __shared__ float red[4][4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long long)gridDim.x * blockDim.x) {
    const float r = rewards[n];
    const uint8_t d = dones[n];
    dones_t[n] = d;
    float shaped = r;
    if (bootstrap && time_outs) shaped = 0.01f * r + (gamma * values[n]) * (time_outs[n] ? 1.0f : 0.0f);
    rewards_t[n] = shaped;
    const float cr = cur_rewards[n] + r, cl = cur_lengths[n] + 1.0f, cs = cur_success[n] + (successes ? successes[n] : 0.f);
    if (d) { s0 += cr; s1 += cl; s2 += cs; s3 += 1.0f; }
    const float keep = d ? 0.0f : 1.0f;
    cur_rewards[n] = cr * keep;
    cur_lengths[n] = cl * keep;
    cur_success[n] = cs * keep;
  }
  // block sums (fixed tree), then one atomic per block and statistic
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); s3 += __shfl_xor(s3, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; red[wave][3] = s3; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (t != 0.f) atomicAdd(meter + threadIdx.x, t);
  }
}

static int rollout_act_store(int64_t N, int obs_dim, int priv_dim, int act, const float* obs, const float* priv,
                             const float* mu, const float* value_n, const float* logstd, const float* noise,
                             const double* rms_value, float eps, float* obses_t, float* priv_t, float* actions_t,
                             float* nlp_t, float* values_t, float* mus_t, float* sigmas_t, float* actions_clamped,
                             float* values_out, hipStream_t s) {
  if (N < 1 || obs_dim < 1 || priv_dim < 0 || act < 1 || !obs || !mu || !value_n || !logstd || !noise || !obses_t ||
      !actions_t || !nlp_t || !values_t || !mus_t || !sigmas_t || !actions_clamped || !values_out ||
      (priv_dim > 0 && (!priv || !priv_t)))
    return IGI_E_BADARG;
  long long work = N * (long long)(obs_dim > priv_dim ? obs_dim : priv_dim);
  int nb = (int)((work + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_rollout_act_store, dim3(nb), dim3(256), 0, s, (int)N, obs_dim, priv_dim, act, obs, priv, mu,
                     value_n, logstd, noise, rms_value, eps, obses_t, priv_t, actions_t, nlp_t, values_t, mus_t,
                     sigmas_t, actions_clamped, values_out);
  return (int)hipGetLastError();
}

static int rollout_env_store(int64_t N, const float* rewards, const uint8_t* dones, const float* values,
                             const uint8_t* time_outs, const float* successes, float gamma, int bootstrap,
                             float* rewards_t, uint8_t* dones_t, float* cur_rewards, float* cur_lengths,
                             float* cur_success, float* meter, hipStream_t s) {
  if (N < 1 || !rewards || !dones || !values || !rewards_t || !dones_t || !cur_rewards || !cur_lengths ||
      !cur_success || !meter)
    return IGI_E_BADARG;
  int nb = (int)((N + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_rollout_env_store, dim3(nb), dim3(256), 0, s, (int)N, rewards, dones, values, time_outs,
                     successes, gamma, bootstrap, rewards_t, dones_t, cur_rewards, cur_lengths, cur_success, meter);
  return (int)hipGetLastError();
}

}  // namespace igi

// ---------------------------------------------------------------------------------------------
// Distillation loss of the student (ext_adapt.py:812-819): loss = sum_{row,q} w[q] * (clamp(mu,+-1) - clamp(a,+-1))^2
// and d loss / d mu = 2 w[q] (clamp(mu) - clamp(a)) where -1 <= mu <= 1 (clamp's pass-through), else 0.
// Two launches, fixed-order sums (no atomics): per-block partials, then one block adds them.
// ---------------------------------------------------------------------------------------------
namespace igi {
constexpr int BC_BLOCKS = 256;

__global__ __launch_bounds__(256) void k_bc_partial(const float* __restrict__ mu, const float* __restrict__ teacher,
                                                    const float* __restrict__ w, long long total, int act,
                                                    float* __restrict__ dmu, double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const float m = mu[e];
    const float d = fminf(fmaxf(m, -1.0f), 1.0f) - fminf(fmaxf(teacher[e], -1.0f), 1.0f);
    const float wq = w[e % act];
    s += (double)((d * d) * wq);
    if (dmu) dmu[e] = (m >= -1.0f && m <= 1.0f) ? (2.0f * d) * wq : 0.0f;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void k_bc_final(const double* __restrict__ partial, int n, float* __restrict__ loss) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) loss[0] = (float)s;
}

static int bc_loss(const float* mu, const float* teacher, const float* w, int64_t rows, int act, float* loss,
                   float* dmu, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!mu || !teacher || !w || !loss || rows < 1 || act < 1 || !ws) return IGI_E_BADARG;
  if (ws_bytes < sizeof(double) * BC_BLOCKS) return IGI_E_WORKSPACE;
  const long long total = rows * (long long)act;
  int nb = (int)((total + 255) / 256);
  if (nb > BC_BLOCKS) nb = BC_BLOCKS;
  double* partial = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(k_bc_partial, dim3(nb), dim3(256), 0, s, mu, teacher, w, total, act, dmu, partial);
  hipLaunchKernelGGL(k_bc_final, dim3(1), dim3(64), 0, s, partial, nb, loss);
  return (int)hipGetLastError();
}
}  // namespace igi
