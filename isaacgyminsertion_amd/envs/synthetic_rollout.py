"""Seeded synthetic rollout arena of BASELINE.md section 3 / SURVEY.md section 8(d) config 2: what ``play_steps`` leaves in the
``ExperienceBuffer`` (frozen_ppo.py:655-683) for a freshly initialised teacher on Gaussian observations -- old
mus / values / neglogpacs come from the network's own ``act`` (models_split.py:120-134), so PPO ratios start near
1.  Observations / rewards / dones / noise are drawn with a seeded CPU generator (the same streams the oracle-side
generator of the parity tests draws); the policy outputs are computed on the device by
``torch.ops.mi355ppo.actor_critic_infer``.  Nothing here touches ``oracle/``."""
import math
from collections import OrderedDict

import torch

LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def init_teacher_params(units, priv_units, obs_dim=15, priv_dim=64, act_dim=6, seed=42):
    """The reference's initialisation recipe (models_split.py:21-24, 104-117): orthogonal(sqrt 2) Linear weights,
    zero biases, mu gain 0.01, value gain 1, sigma 0."""
    from ..teacher_native import teacher_param_shapes
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    for k, s in teacher_param_shapes(obs_dim, priv_dim, act_dim, list(units), list(priv_units)).items():
        if k.endswith("weight"):
            w = torch.empty(s)
            gain = 0.01 if k.startswith("mu.") else (1.0 if k.startswith("value.") else 2.0 ** 0.5)
            torch.nn.init.orthogonal_(w, gain, generator=g)
            p[k] = w
        else:
            p[k] = torch.zeros(s)
    return p


def teacher_problem(N, T, units, priv_units, obs_dim=15, priv_dim=64, act_dim=6, seed=1234, done_p=0.01,
                    device="cuda:0"):
    """-> (initial parameters (cpu), rollout dict of time-major DEVICE tensors, permutation (cpu int64))."""
    from ..teacher_native import TeacherEngine
    dev = torch.device(device)
    p = init_teacher_params(units, priv_units, obs_dim, priv_dim, act_dim)
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T + 1, N, obs_dim, generator=g)
    priv = torch.randn(T + 1, N, priv_dim, generator=g)
    rewards = 0.1 * torch.randn(T, N, 1, generator=g)
    dones = (torch.rand(T, N, generator=g) < done_p).to(torch.uint8)
    eps = torch.randn(T, N, act_dim, generator=g)
    perm = torch.randperm(N * T, generator=g)
    eng = TeacherEngine(min(N * (T + 1), 16384), 1, 1, units=units, priv_units=priv_units, obs_dim=obs_dim,
                        priv_dim=priv_dim, act_dim=act_dim, device=dev)
    eng.load_params(p)
    # fresh normalisers (mean 0, var 1, count 1): model_act's eval-mode normalisation of the raw observations
    mu, val = eng.infer(obs.reshape(-1, obs_dim).to(dev), priv.reshape(-1, priv_dim).to(dev), normalize=True)
    mu = mu.reshape(T + 1, N, act_dim)
    # de-normalised values as model_act stores them (frozen_ppo.py:365) with the fresh value statistics
    value = (math.sqrt(1.0 + 1e-5) * torch.clamp(val, -5.0, 5.0)).reshape(T + 1, N, 1)
    sigma = torch.ones_like(mu)                                     # exp(logstd = 0)
    actions = mu[:T] + sigma[:T] * eps.to(dev)
    neglogp = (((actions - mu[:T]) ** 2) / (2.0 * sigma[:T] ** 2) + torch.log(sigma[:T]) + LOG_SQRT_2PI).sum(-1)
    ro = dict(obses=obs[:T].to(dev).contiguous(), priv_info=priv[:T].to(dev).contiguous(), rewards=rewards.to(dev),
              values=value[:T].contiguous(), neglogpacs=neglogp.contiguous(), dones=dones.to(dev),
              actions=actions.contiguous(), mus=mu[:T].contiguous(), sigmas=sigma[:T].contiguous(),
              last_values=value[T].contiguous())
    return p, ro, perm
