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

import torch
import torch.nn as nn

from . import ops  # noqa: F401  (registers torch.ops.mi355ppo)
from .flat_params import flat_parameters


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
        return flat_parameters(p for layer in self.layers for p in layer.parameters())

    def forward(self, src):
        l0 = self.layers[0]
        ps = {float(l0.dropout.p), float(l0.dropout1.p), float(l0.dropout2.p), float(l0.self_attn.dropout)}
        if len(ps) != 1:
            raise NotImplementedError("the four dropout probabilities of a layer are expected to be equal")
        p = ps.pop()
        training = self.training and p > 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if training else 0
        if not src.is_cuda:
            raise RuntimeError("HipTransformerEncoder runs on the HIP device only (no CPU fallback)")
        y, _ws = torch.ops.mi355ppo.token_encoder_fwd(src.to(torch.float32).contiguous(),
                                                      self.flat_parameters().to(torch.float32).contiguous(),
                                                      l0.self_attn.num_heads, l0.linear1.out_features,
                                                      self.num_layers, float(p), bool(training), seed)
        return y
