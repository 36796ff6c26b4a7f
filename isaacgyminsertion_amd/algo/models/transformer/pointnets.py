"""Segmented point-cloud encoder with the reference's interface and state_dict
(algo/models/transformer/pointnets.py:12-42): ``PointNet(point_channel=3, output_dim=256)`` with
``local_mlp.{0,2}.{weight,bias}``; forward (Linear-GELU-Linear + max over points) and backward run in
libigi_hip.so (igi_pointnet_forward / igi_pointnet_backward) as one torch.autograd.Function.
"""
import torch
import torch.nn as nn

from .... import _lib


class _PointNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat_params):
        if not x.is_cuda:
            raise RuntimeError("PointNet runs on the HIP device only (no CPU fallback)")
        L = _lib.lib()
        b, n, ch = x.shape
        if ch != 3:
            raise RuntimeError("expected (B, N, 3) points")
        xx = x.to(torch.float32).contiguous()
        p = flat_params.detach().to(torch.float32).contiguous()
        y = torch.empty(b, 256, dtype=torch.float32, device=x.device)
        idx = torch.empty(b, 256, dtype=torch.int32, device=x.device)
        rc = L.igi_pointnet_forward(_lib.ptr(xx), b, n, _lib.ptr(p), _lib.ptr(y), _lib.ptr(idx),
                                    _lib.current_stream(x.device))
        _lib.check(rc, "igi_pointnet_forward")
        ctx.save_for_backward(xx, p, idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        xx, p, idx = ctx.saved_tensors
        L = _lib.lib()
        b, n, _ = xx.shape
        d = dy.to(torch.float32).contiguous()
        grads = torch.empty_like(p)
        nbytes = L.igi_pointnet_workspace_bytes(b)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
        rc = L.igi_pointnet_backward(_lib.ptr(xx), b, n, _lib.ptr(p), _lib.ptr(d), _lib.ptr(idx), _lib.ptr(grads),
                                     _lib.ptr(ws), nbytes, _lib.current_stream(dy.device))
        _lib.check(rc, "igi_pointnet_backward")
        return None, grads


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
        return torch.cat([p.reshape(-1) for p in self.local_mlp.parameters()])

    def forward(self, x):
        """x: (B, N, 3) -> (B, 256)"""
        return _PointNetFn.apply(x, self.flat_parameters())
