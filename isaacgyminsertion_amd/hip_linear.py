"""``nn.Linear`` with an optional fused activation whose forward / backward run in libigi_hip.so
(igi_linear_forward / igi_linear_backward: exact-fp32 MFMA GEMMs with bias+activation and
activation-gradient epilogues, deterministic split sums for the weight gradient).

``HipLinear`` IS an ``nn.Linear`` (same ``weight`` / ``bias`` parameters and state_dict keys, picked up by
the reference's ``isinstance(m, nn.Linear)`` re-initialisation, tact.py:414-419); ``act`` fuses the
activation that follows it in the reference's ``nn.Sequential`` so that ``Linear -> ReLU`` is one launch —
the Sequential keeps an ``nn.Identity`` in the activation's slot, so parameter indices do not move.
"""
import torch
import torch.nn as nn

from . import _lib

_ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2}


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        if not x.is_cuda:
            raise RuntimeError("HipLinear runs on the HIP device only (no CPU fallback)")
        L = _lib.lib()
        out_f, in_f = weight.shape
        lead = x.shape[:-1]
        x2 = x.reshape(-1, in_f).to(torch.float32)
        if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < in_f):
            x2 = x2.contiguous()
        rows = x2.shape[0]
        ldx = x2.stride(0) if rows > 1 else in_f
        w = weight.detach().contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        y = torch.empty(rows, out_f, dtype=torch.float32, device=x.device)
        rc = L.igi_linear_forward(_lib.ptr(x2), ldx, _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), out_f, rows, in_f, out_f,
                                  act, _lib.current_stream(x.device))
        _lib.check(rc, "igi_linear_forward")
        ctx.save_for_backward(x2, w, y)
        ctx.act, ctx.has_bias, ctx.lead, ctx.ldx = act, bias is not None, lead, ldx
        return y.reshape(*lead, out_f)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        L = _lib.lib()
        out_f, in_f = w.shape
        rows = x2.shape[0]
        d = dy.reshape(rows, out_f).to(torch.float32).contiguous()
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty(rows, in_f, dtype=torch.float32, device=d.device) if need_dx else None
        dw = torch.empty_like(w)
        db = torch.empty(out_f, dtype=torch.float32, device=d.device) if ctx.has_bias else None
        nbytes = L.igi_linear_workspace_bytes(rows, in_f, out_f)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=d.device)
        rc = L.igi_linear_backward(_lib.ptr(x2), ctx.ldx, _lib.ptr(w), _lib.ptr(y), out_f, _lib.ptr(d), out_f,
                                   _lib.ptr(dx), in_f, _lib.ptr(dw), _lib.ptr(db), rows, in_f, out_f, ctx.act,
                                   _lib.ptr(ws), ws.numel(), _lib.current_stream(d.device))
        _lib.check(rc, "igi_linear_backward")
        return (dx.reshape(*ctx.lead, in_f) if need_dx else None), dw, db, None


def linear(x, weight, bias=None, act=None):
    """act(x @ weight.T + bias) on the HIP kernels; act in {None, 'tanh', 'relu'}."""
    return _LinearFn.apply(x, weight, bias, _ACT[act])


class HipLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, act=None):
        super().__init__(in_features, out_features, bias=bias)
        if act not in _ACT:
            raise ValueError(f"unknown activation {act!r}")
        self.act = act

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias, _ACT[self.act])

    def extra_repr(self):
        return super().extra_repr() + (f", act={self.act}" if self.act else "")
