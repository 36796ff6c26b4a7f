"""Segmented point-cloud encoder with the reference's interface and state_dict
(algo/models/transformer/pointnets.py:12-42): ``PointNet(point_channel=3, output_dim=256)`` with
``local_mlp.{0,2}.{weight,bias}``; forward (Linear-GELU-Linear + max over points) and backward run in
libigi_hip.so (torch.ops.mi355ppo.pointnet_max_fwd / _bwd -> igi_pointnet_forward / igi_pointnet_backward).
"""
import torch
import torch.nn as nn

from .... import ops  # noqa: F401  (registers torch.ops.mi355ppo)
from ....flat_params import flat_parameters


def _dense_rows(x):
    """a slice of a wider (B, points, 3) tensor along the point axis runs in place (the kernels take the cloud pitch)"""
    ok = x.dim() == 3 and x.stride(2) == 1 and x.stride(1) == 3 and (x.shape[0] == 1 or x.stride(0) >= 3 * x.shape[1])
    return x if ok else x.contiguous()


class PointNet(nn.Module):
    def __init__(self, point_channel=3, output_dim=256):
        super().__init__()
        if point_channel != 3 or output_dim != 256:
            raise NotImplementedError("the reference instantiates PointNet(3 -> 64 -> 256) only")
        self.local_mlp = nn.Sequential(nn.Linear(point_channel, 64), nn.GELU(), nn.Linear(64, 256))
        self.reset_parameters_()
        self.frame_count = 0

    def reset_parameters_(self):
        """pointnets.py:28-33"""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def flat_parameters(self):
        return flat_parameters(self.local_mlp.parameters())

    def forward(self, x):
        """x: (B, N, 3) -> (B, 256)"""
        if not x.is_cuda:
            raise RuntimeError("PointNet runs on the HIP device only (no CPU fallback)")
        y, _idx = torch.ops.mi355ppo.pointnet_max_fwd(_dense_rows(x.to(torch.float32)),
                                                      self.flat_parameters().to(torch.float32).contiguous())
        return y
