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

from . import ops  # noqa: F401  (registers torch.ops.mi355ppo)

_ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2}


def _apply(x, weight, bias, act):
    """torch.ops.mi355ppo.linear on the flattened rows (autograd registered on the op: the weight / bias products of
    a frozen layer are skipped, see ops.linear_bwd)."""
    if not x.is_cuda:
        raise RuntimeError("HipLinear runs on the HIP device only (no CPU fallback)")
    out_f, in_f = weight.shape
    lead = x.shape[:-1]
    x2 = x.reshape(-1, in_f).to(torch.float32)
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < in_f):
        x2 = x2.contiguous()
    y = torch.ops.mi355ppo.linear(x2, weight if weight.is_contiguous() else weight.contiguous(),
                                  None if bias is None else bias.contiguous(), act)
    return y.reshape(*lead, out_f)


def linear(x, weight, bias=None, act=None):
    """act(x @ weight.T + bias) on the HIP kernels; act in {None, 'tanh', 'relu'}."""
    return _apply(x, weight, bias, _ACT[act])


class HipLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, act=None):
        super().__init__(in_features, out_features, bias=bias)
        if act not in _ACT:
            raise ValueError(f"unknown activation {act!r}")
        self.act = act

    def forward(self, x):
        return _apply(x, self.weight, self.bias, _ACT[self.act])

    def extra_repr(self):
        return super().extra_repr() + (f", act={self.act}" if self.act else "")
