"""Stack of ``nn.TransformerEncoderLayer(d_model=32, nhead=2, dim_feedforward=128, activation="gelu",
batch_first=True, norm_first=True)`` whose forward / backward run in libigi_hip.so
(igi_token_forward / igi_token_backward: MFMA GEMMs for the eight Linears, fused LayerNorm / attention /
GELU / residual-dropout kernels around them) -- the ``sa_decoder`` of the reference's MultiLayerDecoder
(algo/models/transformer/tact.py:137-158).

``HipTransformerEncoder`` keeps ``nn.TransformerEncoder``'s module tree (``layers.{i}.self_attn.in_proj_weight``,
``layers.{i}.linear1.weight``, ``layers.{i}.norm1.weight`` ...): the layer objects are parameter containers, so
state_dicts interchange with the reference; they are never called.
Dropout (0.1 in every layer, active in train mode exactly as in the reference) uses a counter-based mask
seeded from torch's CPU generator, so runs are reproducible under ``torch.manual_seed`` -- the mask stream
itself is this library's, not ATen's (no implementation reproduces another device's dropout stream).
"""
import copy
import ctypes as C

import torch
import torch.nn as nn

from . import _lib


class _TokenEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat_params, cfg_tuple, seed):
        if not x.is_cuda:
            raise RuntimeError("HipTransformerEncoder runs on the HIP device only (no CPU fallback)")
        L = _lib.lib()
        B, S, d = x.shape
        nhead, ff, layers, p, training = cfg_tuple
        cfg = _lib.TokenCfg(B, S, d, nhead, ff, layers, float(p), int(training))
        n_params = L.igi_token_param_count(C.byref(cfg))
        if n_params < 0:
            _lib.check(int(n_params), "igi_token_param_count")      # unsupported shape: raises with the library's message
        if n_params != flat_params.numel():
            raise RuntimeError("parameter vector does not match the token-encoder configuration")
        xx = x.to(torch.float32).contiguous()
        pp = flat_params.detach().to(torch.float32).contiguous()
        y = torch.empty_like(xx)
        nbytes = L.igi_token_workspace_bytes(C.byref(cfg))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = L.igi_token_forward(C.byref(cfg), _lib.ptr(xx), _lib.ptr(pp), _lib.ptr(y), _lib.ptr(ws), nbytes,
                                 C.c_uint64(seed), _lib.current_stream(x.device))
        _lib.check(rc, "igi_token_forward")
        ctx.save_for_backward(pp, ws)
        ctx.cfg_tuple, ctx.seed, ctx.shape = cfg_tuple, seed, (B, S, d)
        return y

    @staticmethod
    def backward(ctx, dy):
        pp, ws = ctx.saved_tensors
        L = _lib.lib()
        B, S, d = ctx.shape
        nhead, ff, layers, p, training = ctx.cfg_tuple
        cfg = _lib.TokenCfg(B, S, d, nhead, ff, layers, float(p), int(training))
        g = dy.to(torch.float32).contiguous()
        dx = torch.empty_like(g)
        grads = torch.empty_like(pp)
        rc = L.igi_token_backward(C.byref(cfg), _lib.ptr(g), _lib.ptr(pp), _lib.ptr(dx), _lib.ptr(grads), _lib.ptr(ws),
                                  ws.numel(), C.c_uint64(ctx.seed), _lib.current_stream(g.device))
        _lib.check(rc, "igi_token_backward")
        return dx, grads, None, None


class HipTransformerEncoder(nn.Module):
    """Drop-in for ``nn.TransformerEncoder(encoder_layer, num_layers, enable_nested_tensor=False)``
    (no final norm, no masks: the reference passes neither)."""

    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        if not (encoder_layer.norm_first and encoder_layer.self_attn.batch_first):
            raise NotImplementedError("built for batch_first, norm_first layers (tact.py:143-146)")
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers

    def flat_parameters(self):
        return torch.cat([p.reshape(-1) for layer in self.layers for p in layer.parameters()])

    def forward(self, src):
        l0 = self.layers[0]
        ps = {float(l0.dropout.p), float(l0.dropout1.p), float(l0.dropout2.p), float(l0.self_attn.dropout)}
        if len(ps) != 1:
            raise NotImplementedError("the four dropout probabilities of a layer are expected to be equal")
        p = ps.pop()
        training = self.training and p > 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if training else 0
        cfg = (l0.self_attn.num_heads, l0.linear1.out_features, self.num_layers, p, training)
        return _TokenEncoderFn.apply(src, self.flat_parameters(), cfg, seed)
