"""AllSight tactile encoder with the reference's interface and state_dict
(algo/models/transformer/tactile_cnn.py:7-79): ``SpatialSoftArgmax`` and
``CNNWithSpatialSoftArgmax(latent_dim)`` whose ``cnn.{0,2,4,7}.{weight,bias}`` parameters are ordinary
``nn.Conv2d`` / ``nn.Linear`` parameters (same init, same keys), but whose forward AND backward run in
libigi_hip.so (igi_tactile_forward / igi_tactile_backward: channels-last implicit-GEMM convolutions on
exact-fp32 MFMA with LDS-DMA im2col gathers, fused soft-argmax), reached through the dispatcher ops
torch.ops.mi355ppo.tactile_cnn_fwd / _bwd so the encoder composes with the rest of the student under autograd.
"""
import torch
import torch.nn as nn

from .... import ops  # noqa: F401  (registers torch.ops.mi355ppo)
from ....flat_params import flat_parameters


class SpatialSoftArgmax(nn.Module):
    """tactile_cnn.py:7-58.  Inside ``CNNWithSpatialSoftArgmax`` it is evaluated by the fused encoder op; called on
    its own it runs the standalone kernels (torch.ops.mi355ppo.spatial_softargmax_fwd / _bwd), any channel count,
    with the reference's coordinate grid (``meshgrid(linspace(w), linspace(h))`` flattened against the row-major
    softmax: SURVEY Appendix A12) and its interleaved (x, y) output."""

    def __init__(self, normalize=False):
        super().__init__()
        self.normalize = normalize

    def forward(self, x):
        assert x.ndim == 4, "Expecting a tensor of shape (B, C, H, W)."
        if not x.is_cuda:
            raise RuntimeError("SpatialSoftArgmax runs on the HIP device only (no CPU fallback)")
        out, _stat = torch.ops.mi355ppo.spatial_softargmax_fwd(x.to(torch.float32).contiguous(), bool(self.normalize))
        return out


def tactile_cnn(x, flat_params, latent_dim):
    """(B, 3, H, W) images -> (B, latent): torch.ops.mi355ppo.tactile_cnn_fwd (autograd registered on the op).  The
    native op wants whole 32-image groups: the batch is padded with zero images, which carry zero gradient."""
    if not x.is_cuda:
        raise RuntimeError("tactile CNN runs on the HIP device only (no CPU fallback)")
    b, c, h, w = x.shape
    if c != 3:
        raise RuntimeError("expected (B, 3, H, W): 3 fingers' gray images stacked as channels")
    pad = (-b) % 32
    xx = x.to(torch.float32).contiguous()
    if pad:
        xx = torch.cat([xx, xx.new_zeros(pad, c, h, w)], 0)
    y, _ws = torch.ops.mi355ppo.tactile_cnn_fwd(xx, flat_params.to(torch.float32).contiguous(), latent_dim)
    return y[:b] if pad else y      # (a slice's backward is a zero fill + a copy)


class CNNWithSpatialSoftArgmax(nn.Module):
    def __init__(self, latent_dim):
        super().__init__()
        self.latent_dim = latent_dim
        self.cnn = nn.Sequential(
            nn.Conv2d(in_channels=3, out_channels=32, kernel_size=8, stride=2, padding=0),
            nn.ReLU(),
            nn.Conv2d(in_channels=32, out_channels=64, kernel_size=4, stride=1, padding=0),
            nn.ReLU(),
            nn.Conv2d(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=0),
            nn.ReLU(),
            SpatialSoftArgmax(normalize=True),
            nn.Linear(128, latent_dim),
        )

    def flat_parameters(self):
        """state_dict-order concatenation (differentiable: autograd splits the flat gradient back)."""
        return flat_parameters(self.cnn.parameters())

    def forward(self, x):
        return tactile_cnn(x, self.flat_parameters(), self.latent_dim)
