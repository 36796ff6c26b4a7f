"""The student's distillation loss as one native op (ext_adapt.py:812-819):
``bc_loss(mu, teacher_actions, weights) = sum(weights * (clamp(mu, -1, 1) - clamp(teacher_actions, -1, 1)) ** 2)``
with its gradient w.r.t. ``mu`` produced by the same kernel (torch.ops.mi355ppo.bc_loss -> igi_bc_loss)."""
import torch

from . import ops  # noqa: F401  (registers torch.ops.mi355ppo)


def bc_loss(mu, teacher_actions, weights):
    if not mu.is_cuda:
        raise RuntimeError("bc_loss runs on the HIP device only (no CPU fallback)")
    act = mu.shape[-1]
    m = mu.reshape(-1, act).to(torch.float32).contiguous()
    t = teacher_actions.detach().reshape(-1, act).to(torch.float32).contiguous()
    return torch.ops.mi355ppo.bc_loss(m, t, weights.detach().to(torch.float32).contiguous())
