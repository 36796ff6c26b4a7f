"""PointNet (Linear-GELU-Linear, max over points) forward/backward through the C ABI against golden vectors
captured from the reference module.  Tolerances: output 2e-6 abs + 1e-5 rel (exact-fp32 MFMA vs ATen
fp32); gradients 1e-4 * max|g| abs + 1e-3 rel."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoders.npz"))


def _load(tag):
    from isaacgyminsertion_amd.algo.models.transformer.pointnets import PointNet
    m = PointNet()
    sd = {k[len(tag) + 3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/p/")}
    assert list(sd.keys()) == list(m.state_dict().keys())
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize("tag", ["pn400", "pn37"])
def test_pointnet_matches_reference(tag):
    m = _load(tag)
    x = torch.from_numpy(G[f"{tag}/x"]).cuda()
    gy = torch.from_numpy(G[f"{tag}/gy"]).cuda()
    y = m(x)
    (y * gy).sum().backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.detach().cpu().numpy(), G[f"{tag}/y"], atol=2e-6, rtol=1e-5)
    for k, p in m.named_parameters():
        ref = G[f"{tag}/g/{k}"]
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=1e-4 * np.abs(ref).max(), rtol=1e-3, err_msg=k)


def test_pointnet_properties_at_scale():
    """4096 clouds x 400 points (config 4 scale per object): permutation invariance over the point axis,
    per-sample independence, reproducibility."""
    m = _load("pn400")
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(4096, 400, 3, generator=g) * 0.5).cuda()
    perm = torch.randperm(400, generator=g).cuda()
    with torch.no_grad():
        y = m(x)
        yp = m(x[:, perm])
        ys = m(x[1000:1007])
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert torch.equal(y, yp)                      # max is order independent; each row's chain is unchanged
    assert torch.equal(ys, y[1000:1007])
