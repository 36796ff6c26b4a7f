"""ORACLE-side synthetic problem generator (test infrastructure): the seeded synthetic rollout
arena of BASELINE.md section 3 / SURVEY.md section 8(d) config 2, with old mus / values / neglogpacs produced by
the freshly initialised network's own ``act`` (models_split.py:120-134) so PPO ratios start near 1.
CPU tensors; used by tests, smoke() and bench.py (both the GPU leg's inputs and the cpu_baseline)."""
from collections import OrderedDict

import torch

from . import teacher as ot


def init_teacher_params(units, priv_units, obs_dim=15, priv_dim=64, act_dim=6, seed=42):
    """Same initialisation recipe as the reference (models_split.py:21-24, 104-117):
    orthogonal(sqrt 2) Linear weights, zero biases, mu std 0.01, value std 1, sigma 0."""
    g = torch.Generator().manual_seed(seed)
    shapes = ot.teacher_param_shapes(obs_dim, priv_dim, act_dim, units, priv_units)
    p = OrderedDict()
    for k, s in shapes.items():
        if k.endswith("weight"):
            w = torch.empty(s)
            gain = 2.0 ** 0.5
            if k.startswith("mu."):
                gain = 0.01
            elif k.startswith("value."):
                gain = 1.0
            torch.nn.init.orthogonal_(w, gain, generator=g)
            p[k] = w
        else:
            p[k] = torch.zeros(s)
    return p


def teacher_problem(N, T, units, priv_units, obs_dim=15, priv_dim=64, act_dim=6, seed=1234, done_p=0.01):
    """Returns (init params, rollout dict of time-major tensors, permutation)."""
    p = init_teacher_params(units, priv_units, obs_dim, priv_dim, act_dim)
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T + 1, N, obs_dim, generator=g)
    priv = torch.randn(T + 1, N, priv_dim, generator=g)
    rewards = 0.1 * torch.randn(T, N, 1, generator=g)
    dones = (torch.rand(T, N, generator=g) < done_p).to(torch.uint8)
    eps = torch.randn(T, N, act_dim, generator=g)
    rs_o, rs_p, rs_v = ot.RmsState(obs_dim), ot.RmsState(priv_dim), ot.RmsState(1)
    with torch.no_grad():
        flat_o = rs_o.normalize(obs.reshape(-1, obs_dim))
        flat_p = rs_p.normalize(priv.reshape(-1, priv_dim))
        mu, logstd, value, _ = ot.actor_critic(p, flat_o, flat_p, len(priv_units), len(units))
        sigma = torch.exp(logstd)
        mu = mu.reshape(T + 1, N, act_dim)
        sigma = sigma.reshape(T + 1, N, act_dim)
        value = rs_v.unnormalize(value).reshape(T + 1, N, 1)     # model_act stores de-normalised values
        actions = mu[:T] + sigma[:T] * eps
        neglogp = ot.gaussian_neglogp(actions, mu[:T], sigma[:T], torch.log(sigma[:T]))
    ro = dict(obses=obs[:T].contiguous(), priv_info=priv[:T].contiguous(), rewards=rewards,
              values=value[:T].contiguous(), neglogpacs=neglogp.contiguous(), dones=dones,
              actions=actions.contiguous(), mus=mu[:T].contiguous(), sigmas=sigma[:T].contiguous(),
              last_values=value[T].contiguous())
    perm = torch.randperm(N * T, generator=g)
    return p, ro, perm
