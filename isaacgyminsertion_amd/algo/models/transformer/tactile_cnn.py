"""AllSight tactile encoder with the reference's interface and state_dict
(algo/models/transformer/tactile_cnn.py:7-79): ``SpatialSoftArgmax`` and
``CNNWithSpatialSoftArgmax(latent_dim)`` whose ``cnn.{0,2,4,7}.{weight,bias}`` parameters are ordinary
``nn.Conv2d`` / ``nn.Linear`` parameters (same init, same keys), but whose forward AND backward run in
libigi_hip.so (igi_tactile_forward / igi_tactile_backward: channels-last implicit-GEMM convolutions on
exact-fp32 MFMA with LDS-DMA im2col gathers, fused soft-argmax).  The op is a torch.autograd.Function so
the encoder composes with the rest of the student under autograd.
"""
import ctypes as C

import torch
import torch.nn as nn

from .... import _lib


class SpatialSoftArgmax(nn.Module):
    """Kept for state_dict / module-tree parity (it has no parameters); evaluated inside the fused op."""

    def __init__(self, normalize=False):
        super().__init__()
        self.normalize = normalize


class _TactileCNNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat_params, latent_dim):
        if not x.is_cuda:
            raise RuntimeError("tactile CNN runs on the HIP device only (no CPU fallback)")
        L = _lib.lib()
        b, c, h, w = x.shape
        if c != 3:
            raise RuntimeError("expected (B, 3, H, W): 3 fingers' gray images stacked as channels")
        pad = (-b) % 32                         # the native op wants whole 32-image groups
        xx = x.to(torch.float32).contiguous()
        if pad:
            xx = torch.cat([xx, xx.new_zeros(pad, c, h, w)], 0)
        cfg = _lib.TactileCfg(b + pad, h, w, latent_dim)
        nbytes = L.igi_tactile_workspace_bytes(C.byref(cfg))
        if nbytes == 0:
            raise RuntimeError("igi_tactile_workspace_bytes rejected the configuration")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        y = torch.empty(b + pad, latent_dim, dtype=torch.float32, device=x.device)
        p = flat_params.detach().to(torch.float32).contiguous()
        rc = L.igi_tactile_forward(C.byref(cfg), _lib.ptr(xx), _lib.ptr(p), _lib.ptr(y), _lib.ptr(ws), nbytes,
                                   _lib.current_stream(x.device))
        _lib.check(rc, "igi_tactile_forward")
        ctx.cfg, ctx.ws, ctx.nbytes, ctx.b, ctx.pad = cfg, ws, nbytes, b, pad
        ctx.save_for_backward(p)
        return y[:b]

    @staticmethod
    def backward(ctx, dy):
        (p,) = ctx.saved_tensors
        L = _lib.lib()
        d = dy.to(torch.float32).contiguous()
        if ctx.pad:
            d = torch.cat([d, d.new_zeros(ctx.pad, d.shape[1])], 0)   # padded images carry zero gradient
        grads = torch.empty_like(p)
        rc = L.igi_tactile_backward(C.byref(ctx.cfg), _lib.ptr(d), _lib.ptr(p), _lib.ptr(grads), _lib.ptr(ctx.ws),
                                    ctx.nbytes, _lib.current_stream(dy.device))
        _lib.check(rc, "igi_tactile_backward")
        return None, grads, None


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
        return torch.cat([p.reshape(-1) for p in self.cnn.parameters()])

    def forward(self, x):
        return _TactileCNNFn.apply(x, self.flat_parameters(), self.latent_dim)
