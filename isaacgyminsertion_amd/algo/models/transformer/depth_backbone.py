"""Depth / segmentation image encoder with the reference's interface and state_dict
(algo/models/transformer/tact.py:81-113): ``DepthOnlyFCBackbone54x96(latent_dim, output_activation=None,
num_channel=1)`` with ``image_compression.{0,3,6,8}.{weight,bias}``; forward (conv 5x5 + max-pool + ELU,
conv 3x3 + ELU, Linear(64768, 128) + ELU, Linear(128, latent)) and backward run in libigi_hip.so
(torch.ops.mi355ppo.depth_backbone_fwd / _bwd -> igi_depth_forward / igi_depth_backward).  The input (an observation) gets no
gradient.  Batches are padded to a multiple of 32 images (zero images, zero output gradient).
"""
import torch
import torch.nn as nn

from .... import ops  # noqa: F401  (registers torch.ops.mi355ppo)
from ....flat_params import flat_parameters


def depth_backbone(x, flat_params, latent_dim):
    """(B, 1, 54, 96) -> (B, latent): torch.ops.mi355ppo.depth_backbone_fwd (autograd registered on the op); batches
    are padded to a multiple of 32 with zero images (zero output gradient)."""
    if not x.is_cuda:
        raise RuntimeError("DepthOnlyFCBackbone54x96 runs on the HIP device only (no CPU fallback)")
    if x.dim() != 4 or tuple(x.shape[1:]) != (1, 54, 96):
        raise RuntimeError(f"expected (B, 1, 54, 96) images, got {tuple(x.shape)}")
    n = x.shape[0]
    b = (n + 31) // 32 * 32
    xx = x.to(torch.float32).contiguous()
    if b != n:
        xx = torch.cat([xx, xx.new_zeros(b - n, 1, 54, 96)])
    y, _ws = torch.ops.mi355ppo.depth_backbone_fwd(xx, flat_params.to(torch.float32).contiguous(), latent_dim)
    return y[:n] if b != n else y


class DepthOnlyFCBackbone54x96(nn.Module):
    def __init__(self, latent_dim, output_activation=None, num_channel=1):
        super().__init__()
        if num_channel != 1:
            raise NotImplementedError("the reference instantiates the depth backbone with one channel (tact.py:305, 323)")
        self.num_channel = num_channel
        self.latent_dim = latent_dim
        activation = nn.ELU()
        # parameter containers with the reference's indices; the Sequential itself is never called
        self.image_compression = nn.Sequential(
            nn.Conv2d(in_channels=num_channel, out_channels=32, kernel_size=5), nn.MaxPool2d(kernel_size=2, stride=2),
            activation, nn.Conv2d(in_channels=32, out_channels=64, kernel_size=3), activation, nn.Flatten(),
            nn.Linear(64 * 23 * 44, 128), activation, nn.Linear(128, latent_dim))
        self.output_activation = nn.Tanh() if output_activation == "tanh" else nn.Identity()

    def flat_parameters(self):
        return flat_parameters(self.image_compression.parameters())

    def forward(self, images):
        return self.output_activation(depth_backbone(images, self.flat_parameters(), self.latent_dim))
