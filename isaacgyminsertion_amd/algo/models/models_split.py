"""Teacher network with the reference's interface (algo/models/models_split.py:21-250):
``MLP``, ``layer_init``, ``ActorCriticSplit`` with ``act / full_act / act_inference / act_with_grad /
actor_critic / forward`` and the same state_dict keys, shapes and initialisation.

All parameters are views into ONE flat fp32 vector whose layout the HIP library defines
(igi_teacher_param_offsets: state_dict order, tensors 16-byte aligned); the fused PPO kernels read
and update that vector in place.  Forward evaluation runs through igi_teacher_infer (exact-fp32
MFMA); gradients of this network only exist inside the fused update (igi_teacher_fwd_bwd), so the
methods here return tensors without an autograd graph.
"""
import numpy as np
import torch
import torch.nn as nn

LOG_SQRT_2PI = 0.5 * float(np.log(2.0 * np.pi))


def layer_init(layer, std=np.sqrt(2), bias_const=0.0):
    """models_split.py:21-24"""
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


class MLP(nn.Module):
    """Linear+Tanh stack, tanh after every layer including the last (models_split.py:27-38).
    Container for parameters; evaluated by the fused kernels."""

    def __init__(self, units, input_size):
        super().__init__()
        layers = []
        for output_size in units:
            layers.append(layer_init(nn.Linear(input_size, output_size)))
            layers.append(nn.Tanh())
            input_size = output_size
        self.mlp = nn.Sequential(*layers)


class ActorCriticSplit(nn.Module):
    def __init__(self, kwargs):
        nn.Module.__init__(self)
        actions_num = kwargs['actions_num']
        input_shape = kwargs['input_shape']
        mlp_input_shape = input_shape[0]
        self.units = list(kwargs['actor_units'])
        self.contact_info = kwargs.get('gt_contacts_info', False)
        self.only_contact = kwargs.get('only_contact', False)
        self.priv_mlp_units = list(kwargs['priv_mlp_units'])
        self.priv_info = kwargs['priv_info']
        self.priv_info_dim = kwargs['priv_info_dim']
        self.shared_parameters = kwargs.get('shared_parameters', False)
        self.vt_policy = kwargs.get('vt_policy', False)
        if not self.priv_info or self.contact_info or self.shared_parameters or self.vt_policy:
            # reference defaults: priv_info True, compute_contact_gt False, shared_parameters False,
            # vt_policy hard-wired False (frozen_ppo.py:139) -- the only configuration on the hot path
            raise NotImplementedError("only priv_info=True, no contacts, separate critic is on the hot path")
        self.obs_dim = mlp_input_shape
        self.actions_num = actions_num
        mlp_input_shape += self.priv_mlp_units[-1]
        self.env_mlp = MLP(units=self.priv_mlp_units, input_size=self.priv_info_dim)
        self.actor_mlp = MLP(units=self.units, input_size=mlp_input_shape)
        self.critic_mlp = MLP(units=self.units, input_size=mlp_input_shape)
        self.value = layer_init(torch.nn.Linear(self.units[-1], 1), std=1.0)
        self.mu = layer_init(torch.nn.Linear(self.units[-1], actions_num), std=0.01)
        self.sigma = nn.Parameter(torch.zeros(actions_num, requires_grad=True, dtype=torch.float32),
                                  requires_grad=True)
        for m in self.modules():                      # models_split.py:108-117
            if isinstance(m, nn.Linear) and getattr(m, 'bias', None) is not None:
                torch.nn.init.zeros_(m.bias)
        nn.init.constant_(self.sigma, 0)
        self._flat = None
        self._engine = None
        self._pack(torch.device("cpu"))

    # -- flat parameter vector --------------------------------------------------------------------
    def _layout(self):
        from ...teacher_native import make_cfg, param_layout
        cfg, _ = make_cfg(self.obs_dim, self.priv_info_dim, self.actions_num, self.units, self.priv_mlp_units,
                          2, 1, 1)
        return param_layout(cfg)

    def _pack(self, device, flat=None):
        """(Re)build the flat vector on `device` and re-point every parameter at its slice."""
        total, layout = self._layout()
        params = list(self.parameters())           # registration order == state_dict order, sigma first
        order = [self.sigma] + [p for p in params if p is not self.sigma]
        assert len(order) == len(layout)
        if flat is None:
            flat = torch.zeros(total, dtype=torch.float32, device=device)
        for p, (off, size) in zip(order, layout):
            assert p.numel() == size
            view = flat[off:off + size].view(p.shape)
            view.copy_(p.data.to(flat.device))
            p.data = view
        self._flat = flat

    def bind_flat(self, flat):
        """Adopt the trainer engine's parameter vector (values are copied into it)."""
        self._pack(flat.device, flat)

    @property
    def flat_params(self):
        return self._flat

    def _apply(self, fn, recurse=True):
        new_flat = fn(self._flat)
        if new_flat is self._flat:          # .to(same device) / .float() / .cuda() on a cuda model: nothing moved,
            return self                     # the parameters stay views of the engine's vector
        self._pack(new_flat.device, new_flat)
        self._engine = None                 # a moved model gets a fresh inference engine on first use
        return self

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict)   # copy_ into the views: layout preserved
        return out

    # -- evaluation ---------------------------------------------------------------------------------
    def attach_engine(self, engine):
        self._engine = engine

    def _infer_engine(self, device):
        if self._engine is None:
            from ...teacher_native import TeacherEngine
            eng = TeacherEngine(4096, 1, 1, units=self.units, priv_units=self.priv_mlp_units,
                                obs_dim=self.obs_dim, priv_dim=self.priv_info_dim, act_dim=self.actions_num,
                                device=device)
            self.bind_flat_to(eng)
            self._engine = eng
        return self._engine

    def bind_flat_to(self, engine):
        engine.params.copy_(self._flat.to(engine.params.device))
        self.bind_flat(engine.params)

    def actor_critic(self, obs_dict, display=False):
        """models_split.py:166-232: returns (mu, logstd, value, extrin, extrin_gt)."""
        obs = obs_dict['obs']
        if not obs.is_cuda:
            raise RuntimeError("ActorCriticSplit runs on the HIP device only (no CPU fallback)")
        if 'latent' in obs_dict and obs_dict['latent'] is not None:
            return self._actor_critic_from_latent(obs_dict)
        eng = self._infer_engine(obs.device)
        mu, value, latent = eng.infer(obs, obs_dict['priv_info'], want_latent=True, normalize=False)
        logstd = mu * 0 + self.sigma.detach()
        return mu, logstd, value, None, latent

    def _actor_critic_from_latent(self, obs_dict):
        """models_split.py:187-216 with a student latent: the policy / value heads see ``cat(obs, latent)``
        instead of the privileged embedding.  Built from the native Linear+Tanh op so that autograd can carry a
        distillation loss on ``mu`` back into ``latent`` (ext_adapt.py:799-804); the teacher's weights are
        constants here (the student optimiser never steps them)."""
        from ...hip_linear import linear
        x = torch.cat([obs_dict['obs'], obs_dict['latent']], dim=-1)

        def trunk(mlp):
            h = x
            for layer in mlp.mlp:
                if isinstance(layer, nn.Linear):
                    h = linear(h, layer.weight.detach(), layer.bias.detach(), 'tanh')
            return h

        mu = linear(trunk(self.actor_mlp), self.mu.weight.detach(), self.mu.bias.detach())
        value = linear(trunk(self.critic_mlp), self.value.weight.detach(), self.value.bias.detach())
        extrin_gt = None
        if 'priv_info' in obs_dict and obs_dict['priv_info'] is not None:
            with torch.no_grad():
                _, _, extrin_gt = self._infer_engine(x.device).infer(obs_dict['obs'], obs_dict['priv_info'],
                                                                      want_latent=True, normalize=False)
        logstd = mu * 0 + self.sigma.detach()
        return mu, logstd, value, obs_dict['latent'], extrin_gt

    @torch.no_grad()
    def act(self, obs_dict):
        """models_split.py:120-134 (sampling with the device generator)."""
        mu, logstd, value, _, _ = self.actor_critic(obs_dict)
        sigma = torch.exp(logstd)
        selected_action = mu + sigma * torch.randn_like(mu)
        return {
            'neglogpacs': self.neglogp(selected_action, mu, sigma),
            'values': value, 'actions': selected_action, 'mus': mu, 'sigmas': sigma,
        }

    @torch.no_grad()
    def full_act(self, obs_dict):
        """models_split.py:137-152"""
        mu, logstd, value, _, latent_gt = self.actor_critic(obs_dict)
        sigma = torch.exp(logstd)
        selected_action = mu + sigma * torch.randn_like(mu)
        return {
            'neglogpacs': self.neglogp(selected_action, mu, sigma),
            'values': value, 'actions': selected_action, 'mus': mu, 'sigmas': sigma, 'latent_gt': latent_gt,
        }

    @torch.no_grad()
    def act_inference(self, obs_dict):
        """models_split.py:155-159"""
        mu, logstd, value, latent, latent_gt = self.actor_critic(obs_dict)
        latent = latent_gt if latent is None else latent
        return mu, latent

    def act_with_grad(self, obs_dict):
        """models_split.py:161-164.  With a ``latent`` entry the result carries an autograd graph back to it."""
        mu, logstd, value, latent, _ = self.actor_critic(obs_dict)
        return mu, latent

    @staticmethod
    def neglogp(x, mu, sigma):
        """-Normal(mu, sigma).log_prob(x).sum(1) (models_split.py:128)."""
        return (((x - mu) ** 2) / (2.0 * sigma ** 2) + torch.log(sigma) + LOG_SQRT_2PI).sum(dim=-1)

    @torch.no_grad()
    def forward(self, input_dict):
        """models_split.py:234-250 (evaluation only; training gradients live in the fused update)."""
        prev_actions = input_dict.get('prev_actions', None)
        mu, logstd, value, extrin, extrin_gt = self.actor_critic(input_dict)
        sigma = torch.exp(logstd)
        entropy = (0.5 + LOG_SQRT_2PI + torch.log(sigma)).sum(dim=-1)
        return {
            'prev_neglogp': torch.squeeze(self.neglogp(prev_actions, mu, sigma)),
            'values': value, 'entropy': entropy, 'mus': mu, 'sigmas': sigma,
            'extrin': extrin, 'extrin_gt': extrin_gt,
        }
