"""The student's distillation loss as one native op (ext_adapt.py:812-819):
``bc_loss(mu, teacher_actions, weights) = sum(weights * (clamp(mu, -1, 1) - clamp(teacher_actions, -1, 1)) ** 2)``
with its gradient w.r.t. ``mu`` produced by the same LAUNCH when ``mu`` requires a gradient
(torch.ops.mi355ppo.bc_loss_value_grad -> igi_bc_loss once; the gradient is saved for the backward pass)."""
import torch

from . import ops  # noqa: F401  (registers torch.ops.mi355ppo)


def bc_loss(mu, teacher_actions, weights):
    if not mu.is_cuda:
        raise RuntimeError("bc_loss runs on the HIP device only (no CPU fallback)")
    act = mu.shape[-1]
    m = mu.reshape(-1, act).to(torch.float32).contiguous()
    t = teacher_actions.detach().reshape(-1, act).to(torch.float32).contiguous()
    w = weights.detach().to(torch.float32).contiguous()
    if m.requires_grad and torch.is_grad_enabled():
        # training: value and d/dmu leave the SAME launch, the gradient is kept for the backward pass
        return torch.ops.mi355ppo.bc_loss_value_grad(m, t, w)[0]
    return torch.ops.mi355ppo.bc_loss(m, t, w)
