"""The student's distillation loss as one native op (ext_adapt.py:812-819):
``bc_loss(mu, teacher_actions, weights) = sum(weights * (clamp(mu, -1, 1) - clamp(teacher_actions, -1, 1)) ** 2)``
with its gradient w.r.t. ``mu`` produced in the same launch (igi_bc_loss)."""
import torch

from . import _lib


class _BcLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, teacher, weights):
        if not mu.is_cuda:
            raise RuntimeError("bc_loss runs on the HIP device only (no CPU fallback)")
        L = _lib.lib()
        act = mu.shape[-1]
        m = mu.detach().reshape(-1, act).to(torch.float32).contiguous()
        t = teacher.detach().reshape(-1, act).to(torch.float32).contiguous()
        w = weights.detach().to(torch.float32).contiguous()
        rows = m.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=m.device)
        dmu = torch.empty_like(m) if ctx.needs_input_grad[0] else None
        ws = torch.empty(int(L.igi_bc_loss_workspace_bytes()), dtype=torch.uint8, device=m.device)
        rc = L.igi_bc_loss(_lib.ptr(m), _lib.ptr(t), _lib.ptr(w), rows, act, _lib.ptr(loss), _lib.ptr(dmu),
                           _lib.ptr(ws), ws.numel(), _lib.current_stream(m.device))
        _lib.check(rc, "igi_bc_loss")
        ctx.dmu, ctx.shape = dmu, mu.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        return (ctx.dmu * g).reshape(ctx.shape) if ctx.dmu is not None else None, None, None


def bc_loss(mu, teacher_actions, weights):
    return _BcLossFn.apply(mu, teacher_actions, weights)
