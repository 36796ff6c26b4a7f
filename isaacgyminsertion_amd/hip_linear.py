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


def _chain_enabled():
    """IGI_MLP_CHAIN=0: every layer under its own autograd node (A/B; read once)"""
    global _CHAIN
    if _CHAIN is None:
        import os
        _CHAIN = os.environ.get("IGI_MLP_CHAIN", "1") != "0"
    return _CHAIN


_CHAIN = None
_FWD_FUSED = None


def _fwd_fused():
    """IGI_MLP_FWD_FUSED=0: the chain's forward as one launch per layer (A/B; read once)"""
    global _FWD_FUSED
    if _FWD_FUSED is None:
        import os
        _FWD_FUSED = os.environ.get("IGI_MLP_FWD_FUSED", "1") != "0"
    return _FWD_FUSED


class _MlpChain(torch.autograd.Function):
    """A chain of Linear (+ fused activation) layers: the forward is ONE launch (torch.ops.mi355ppo.mlp_fwd -> igi_mlp_forward:
    32 rows per workgroup through every layer, hidden activations in LDS; layers wider than 256 outputs: one
    torch.ops.mi355ppo.linear launch per layer), the backward ONE native call for the whole chain (torch.ops.mi355ppo.mlp_bwd
    -> igi_mlp_backward: per layer one grid for {weight gradient, data gradient with the lower layer's act'}, one sum of
    all split-row partials) instead of four launches per layer under per-layer autograd.  Bit-identical to the per-layer
    path (tests/test_gpu_linear.py)."""

    @staticmethod
    def forward(ctx, x, acts, *wb):
        n = len(acts)
        ws, bs = wb[:n], wb[n:]
        if _fwd_fused() and x.shape[0] <= 8192 and all(w.shape[0] <= 256 for w in ws) and \
                sum((w.shape[1] + 63) // 64 for w in ws) <= 40:   # (beyond one wave of 32-row workgroups the per-layer launches are faster: 75 vs 54 us at 16384 rows)
            ys = torch.ops.mi355ppo.mlp_fwd(x, list(ws), list(bs), list(acts))    # one launch for the whole chain
            h = ys[-1]
        else:
            ys, h = [], x
            for w, b, a in zip(ws, bs, acts):
                h = torch.ops.mi355ppo.linear(h, w, b, a)
                ys.append(h)
        ctx.acts = acts
        ctx.save_for_backward(x, *ws, *ys)
        return h

    @staticmethod
    def backward(ctx, dy):
        n = len(ctx.acts)
        x, ws, ys = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:]
        need_dx = ctx.needs_input_grad[0]
        need_w = [bool(ctx.needs_input_grad[2 + l] or ctx.needs_input_grad[2 + n + l]) for l in range(n)]
        dx, flat = torch.ops.mi355ppo.mlp_bwd(x, list(ws), list(ys), dy.contiguous(), list(ctx.acts), need_dx, need_w)
        wo, bo, _ = ops.mlp_grad_offsets([x.shape[1]] + [w.shape[0] for w in ws])
        gw = [flat[wo[l]:wo[l] + ws[l].numel()].view_as(ws[l]) if ctx.needs_input_grad[2 + l] else None for l in range(n)]
        gb = [flat[bo[l]:bo[l] + ws[l].shape[0]] if ctx.needs_input_grad[2 + n + l] else None for l in range(n)]
        return (dx if need_dx else None, None, *gw, *gb)


def chain_of(seq):
    """The HipLinear layers of an ``nn.Sequential`` / module list as a chain for ``mlp_chain`` -- or None when the
    container holds anything else than HipLinear layers and the ``nn.Identity`` place-holders that keep the reference's
    state_dict indices (a Dropout or LayerNorm added later must not be skipped silently), or when a layer carries
    forward hooks (``mlp_chain`` calls the op directly, not ``Module.__call__``): the caller then runs the container
    itself."""
    layers = []
    for m in seq:
        if isinstance(m, HipLinear):
            if m._forward_hooks or m._forward_pre_hooks:
                return None
            layers.append(m)
        elif not isinstance(m, torch.nn.Identity):
            return None
    return layers


def run_chain(x, seq, tail=()):
    """``seq`` (+ the HipLinear layers of ``tail``) applied to ``x``: as one native chain when every module is a
    HipLinear / Identity, module by module otherwise."""
    layers, tl = chain_of(seq), chain_of(tail)
    if layers is None or tl is None:
        for m in list(seq) + list(tail):
            x = m(x)
        return x
    return mlp_chain(x, layers + tl)


def mlp_chain(x, layers):
    """``layers``: HipLinear modules applied in sequence to the last dimension of ``x`` (each with its fused
    activation).  One autograd node for the whole chain; every layer needs a bias."""
    layers = list(layers)
    if not layers:
        return x
    if not x.is_cuda:
        raise RuntimeError("mlp_chain runs on the HIP device only (no CPU fallback)")
    if len(layers) == 1 or len(layers) > 8 or any(m.bias is None for m in layers) or not _chain_enabled():
        for m in layers:
            x = m(x)
        return x
    lead = x.shape[:-1]
    in_f = layers[0].in_features
    x2 = x.reshape(-1, in_f).to(torch.float32)
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < in_f):
        x2 = x2.contiguous()
    acts = tuple(_ACT[m.act] for m in layers)
    ws = [m.weight if m.weight.is_contiguous() else m.weight.contiguous() for m in layers]
    bs = [m.bias.contiguous() for m in layers]
    y = _MlpChain.apply(x2, acts, *ws, *bs)
    return y.reshape(*lead, layers[-1].out_features)
