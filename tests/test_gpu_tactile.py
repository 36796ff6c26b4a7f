"""Tactile encoder (implicit-GEMM convolutions + spatial soft-argmax + Linear) forward/backward through
the C ABI against golden vectors captured from the reference module (tests/golden/make_golden_encoders.py).
fp32 tolerances: output 2e-5 abs + 1e-4 rel; parameter gradients 3e-4 * max|g| abs + 2e-3 rel (soft-argmax over up to 576 positions: feature errors ~1e-7 enter dW of the Linear)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoders.npz"))


def _load(tag):
    from isaacgyminsertion_amd.algo.models.transformer.tactile_cnn import CNNWithSpatialSoftArgmax
    m = CNNWithSpatialSoftArgmax(32)
    sd = {k[len(tag) + 3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/p/")}
    assert list(sd.keys()) == list(m.state_dict().keys())
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize("tag", ["tac32x64", "tac64x64", "tac_b5"])
def test_tactile_forward_backward_matches_reference(tag):
    m = _load(tag)
    x = torch.from_numpy(G[f"{tag}/x"]).cuda()
    gy = torch.from_numpy(G[f"{tag}/gy"]).cuda()
    y = m(x)
    (y * gy).sum().backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.detach().cpu().numpy(), G[f"{tag}/y"], atol=2e-5, rtol=1e-4)
    for k, p in m.named_parameters():
        ref = G[f"{tag}/g/{k}"]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=3e-4 * np.abs(ref).max(), rtol=2e-3, err_msg=k)


def test_tactile_large_batch_properties():
    """Config-3 scale (2048 images): per-image independence and reproducibility."""
    m = _load("tac32x64")
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2048, 3, 32, 64, generator=g).cuda()
    with torch.no_grad():
        y = m(x)
        y2 = m(x)
        ysub = m(x[512:544])
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    assert torch.isfinite(y).all()
    np.testing.assert_allclose(ysub.cpu().numpy(), y[512:544].cpu().numpy(), atol=1e-6)


def test_standalone_spatial_softargmax_matches_reference():
    """SpatialSoftArgmax.forward on its own (any channel count, non-square maps, normalised and integer grids) and its
    gradient against the reference module (tests/golden/make_golden_softargmax.py): 2e-6 abs on the coordinates
    (in [-1, 1], or up to w for the integer grid: 2e-6 relative to the grid span), gradients 1e-5 of their maximum."""
    from isaacgyminsertion_amd.algo.models.transformer.tactile_cnn import SpatialSoftArgmax
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "softargmax.npz"))
    for tag in ("n_8x24", "n_24x24", "i_10x26", "n_1ch"):
        x = torch.from_numpy(G[f"{tag}/x"]).cuda().requires_grad_()
        m = SpatialSoftArgmax(normalize=bool(G[f"{tag}/normalize"]))
        y = m(x)
        (y * torch.from_numpy(G[f"{tag}/gy"]).cuda()).sum().backward()
        span = 2.0 if m.normalize else float(max(x.shape[2:]))
        np.testing.assert_allclose(y.detach().cpu().numpy(), G[f"{tag}/y"], atol=2e-6 * span, rtol=0, err_msg=tag)
        ref = G[f"{tag}/gx"]
        np.testing.assert_allclose(x.grad.cpu().numpy(), ref, atol=1e-5 * np.abs(ref).max(), rtol=1e-4, err_msg=tag)
    with pytest.raises(RuntimeError):
        SpatialSoftArgmax(True)(torch.zeros(1, 2, 4, 4))


@pytest.mark.parametrize("B,H,W", [(32, 38, 50), (64, 33, 47), (32, 20, 22)])
def test_tactile_odd_image_sizes_match_the_oracle(B, H, W):
    """Image sizes the reference goldens do not hold (odd heights, widths that are not a multiple of 4, the smallest map
    the plan accepts): conv1's forward reads the NCHW tensor in place, its 16-byte LDS-DMA requests starting on any
    4-byte boundary (csrc/gemm_dma.h, ConvDesc::planar), and the soft-argmax grid follows the map.  Against
    oracle/encoders.py (the PyTorch-CPU restatement pinned to the reference goldens by tests/test_oracle_encoders.py), fp32
    and fp64: outputs 2e-5 abs + 1e-4 rel; gradients as close to fp64 as 3e-4 of the largest entry or 3 x the fp32 oracle's
    own error."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import encoders as oe
    m = _load("tac32x64")
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.rand(B, 3, H, W, generator=g)
    gy = torch.randn(B, 32, generator=g)
    y = m(x.cuda())
    (y * gy.cuda()).sum().backward()
    torch.cuda.synchronize()
    y32, g32 = oe.value_and_grads(oe.tactile_cnn, x, sd, gy)
    y64, g64 = oe.value_and_grads(oe.tactile_cnn, x, sd, gy, dtype=torch.float64)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y64.numpy(), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y32.numpy(), atol=2e-5, rtol=1e-4)
    for k, p in m.named_parameters():
        ref = g64[k].numpy()
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        err32 = np.abs(g32[k].numpy() - ref).max()
        assert err <= max(3e-4 * np.abs(ref).max(), 3.0 * err32), (k, err, err32, np.abs(ref).max())
