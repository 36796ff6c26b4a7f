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


@pytest.mark.parametrize("B,N", [(16, 1), (16, 31), (16, 33), (8, 64), (6, 1000), (3, 8192)])
def test_pointnet_point_counts(B, N):
    """Point counts around the kernel's 32-point tiles (one point, one short of a tile, one over, whole tiles, many tiles, the
    8192-point limit of the packed tile numbers) against oracle/encoders.py in fp64: outputs 4e-6 abs + 1e-5 rel;
    gradients within max(1e-4 of the largest entry, 3 x the fp32 oracle's error).  (cloud, channel) pairs whose two
    best points are within 1e-5 of each other carry no upstream gradient (the pick between equal maxima is undefined in
    the reference as well)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import encoders as oe
    m = _load("pn400")
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, N, 3, generator=g) * 0.5
    gy = torch.randn(B, 256, generator=g)
    if N > 1:
        with torch.no_grad():
            sd64 = {k: v.double() for k, v in sd.items()}
            h = torch.nn.functional.gelu(torch.nn.functional.linear(x.double(), sd64["local_mlp.0.weight"], sd64["local_mlp.0.bias"]))
            top2 = torch.nn.functional.linear(h, sd64["local_mlp.2.weight"], sd64["local_mlp.2.bias"]).topk(2, dim=1)[0]
            gy = torch.where((top2[:, 0] - top2[:, 1]) < 1e-5, torch.zeros_like(gy), gy)
    y = m(x.cuda())
    (y * gy.cuda()).sum().backward()
    torch.cuda.synchronize()
    y32, g32 = oe.value_and_grads(oe.pointnet, x, sd, gy)
    y64, g64 = oe.value_and_grads(oe.pointnet, x, sd, gy, dtype=torch.float64)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y64.numpy(), atol=4e-6, rtol=1e-5)
    for k, p in m.named_parameters():
        ref = g64[k].numpy()
        err, err32 = np.abs(p.grad.cpu().numpy() - ref).max(), np.abs(g32[k].numpy() - ref).max()
        assert err <= max(1e-4 * np.abs(ref).max(), 3.0 * err32), (k, err, err32)


def test_pointnet_rejects_more_points_than_the_tile_numbers_hold():
    m = _load("pn400")
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 8193, 3, device="cuda"))


def _gelu_probe_params(device):
    """Parameters that turn the op into a GELU probe: every hidden unit's pre-activation is the point's x coordinate,
    output channel 0 is +hidden[0], channel 1 is -hidden[0] (layout: W1 (64,3) | b1 (64) | W2 (256,64) | b2 (256))."""
    w1 = torch.zeros(64, 3)
    w1[:, 0] = 1.0
    w2 = torch.zeros(256, 64)
    w2[0, 0], w2[1, 0] = 1.0, -1.0
    return torch.cat([w1.reshape(-1), torch.zeros(64), w2.reshape(-1), torch.zeros(256)]).to(device)


def test_pointnet_gelu_sweep_against_fp64():
    """The kernel's erf-GELU is a fitted erfc polynomial (csrc/pointnet.h), not erff: sweep the pre-activation over
    |x| <= 9 (dense), the far tails, +-0, the smallest normals and denormals THROUGH the op (one point per cloud, identity
    weights) against torch.nn.functional.gelu in fp64: 1e-6 abs + 3e-7 rel (the stated accuracy is 6e-7 abs over |x| <= 9;
    fp32 ATen's own erff form is within the same bound), exact zeros at +-0, finite everywhere."""
    import isaacgyminsertion_amd.ops  # noqa: F401
    dev = "cuda"
    v = torch.cat([
        torch.linspace(-9.0, 9.0, 36001, dtype=torch.float64),
        torch.tensor([-30.0, -12.0, 12.0, 30.0, 1e4, -1e4], dtype=torch.float64),
        torch.tensor([0.0, -0.0, 1.17549435e-38, -1.17549435e-38, 1e-40, -1e-40, 1e-30, -1e-30, 1e-7, -1e-7],
                     dtype=torch.float64),
        torch.logspace(-6, 1, 2000, dtype=torch.float64), -torch.logspace(-6, 1, 2000, dtype=torch.float64),
    ]).float()
    x = torch.zeros(v.numel(), 1, 3)
    x[:, 0, 0] = v
    y, idx = torch.ops.mi355ppo.pointnet_max_fwd(x.to(dev), _gelu_probe_params(dev))
    torch.cuda.synchronize()
    y = y.cpu().double()
    assert torch.isfinite(y).all() and int(idx.abs().max()) == 0
    ref = torch.nn.functional.gelu(v.double())
    for ch, sign in ((0, 1.0), (1, -1.0)):
        err = (y[:, ch] - sign * ref).abs()
        bound = 1e-6 + 3e-7 * ref.abs()
        worst = int((err - bound).argmax())
        assert (err <= bound).all(), (ch, float(v[worst]), float(y[worst, ch]), float(ref[worst]))
    zeros = (v == 0)
    assert (y[zeros][:, :2] == 0).all()
    assert (y[:, 2:] == 0).all()                      # untouched channels: bias 0, weights 0


def test_pointnet_gelu_derivative_sweep_against_fp64():
    """d gelu / dx = Phi(x) + x phi(x) as the backward kernel evaluates it: one cloud of one point per value, the
    gradient of b1[0] IS the derivative (dy = e_0): 2e-6 abs against fp64 over |x| <= 9."""
    import isaacgyminsertion_amd.ops  # noqa: F401
    dev = "cuda"
    params = _gelu_probe_params(dev)
    vals = torch.cat([torch.linspace(-9.0, 9.0, 145, dtype=torch.float64),
                      torch.tensor([0.0, 1e-30, -1e-30, 0.75179, -0.75179], dtype=torch.float64)]).float()
    dy = torch.zeros(1, 256, device=dev)
    dy[0, 0] = 1.0
    for val in vals.tolist():
        x = torch.tensor([[[val, 0.0, 0.0]]], device=dev)
        _y, idx = torch.ops.mi355ppo.pointnet_max_fwd(x, params)
        g = torch.ops.mi355ppo.pointnet_max_bwd(x, params, dy, idx)
        got = float(g[192])                               # b1[0]
        t = torch.tensor(val, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.gelu(t).backward()
        assert abs(got - float(t.grad)) <= 2e-6, (val, got, float(t.grad))
        assert abs(float(g[0]) - float(t.grad) * val) <= 2e-6 * max(1.0, abs(val))   # W1[0][0]: times the input


@pytest.mark.parametrize("B,points", [(2048, (400, 400)), (37, (400, 400)), (600, (64, 400, 33)), (5, (1, 700))])
def test_several_objects_in_one_launch_equal_the_per_object_launches(B, points):
    """pointnet_max_fwd_multi / _bwd_multi (round 6: plug + socket of the student in ONE forward and ONE backward launch,
    tact.py:542-571) against one pointnet_max_fwd / _bwd per slice: encodings and arg-max indices bit-identical; parameter
    gradients to fp32 rounding of the partial sums (the objects share the chip's workgroups: another number of per-workgroup
    records is summed) and bit-identical run to run."""
    from isaacgyminsertion_amd.algo.models.transformer.pointnets import PointNet
    o = torch.ops.mi355ppo
    g = torch.Generator(device="cuda").manual_seed(11)
    nets = [PointNet().cuda() for _ in points]
    for m in nets:
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, device="cuda", generator=g) * (0.5 if p.dim() > 1 else 0.1))
    n = sum(points)
    x = torch.randn(B, n + 3, 3, device="cuda", generator=g)[:, :n]          # rows a pitch apart: a slice of a wider tensor
    params = [m.flat_parameters().detach().contiguous() for m in nets]
    y, idx = o.pointnet_max_fwd_multi(x, params, list(points))
    dy = torch.randn(B, 256 * len(points), device="cuda", generator=g)
    grads = o.pointnet_max_bwd_multi(x, params, list(points), dy, idx)
    grads2 = o.pointnet_max_bwd_multi(x, params, list(points), dy, idx)
    assert torch.equal(grads, grads2)
    off = 0
    for i, (m, q) in enumerate(zip(nets, points)):
        xi = x[:, off:off + q]
        yi, ii = o.pointnet_max_fwd(xi, params[i])
        assert torch.equal(y[:, 256 * i:256 * (i + 1)], yi), i
        assert torch.equal(idx[:, 256 * i:256 * (i + 1)], ii), i
        gi = o.pointnet_max_bwd(xi, params[i], dy[:, 256 * i:256 * (i + 1)], ii)
        ref = gi.cpu().numpy()
        np.testing.assert_allclose(grads[i].cpu().numpy(), ref, atol=2e-5 * np.abs(ref).max(), rtol=1e-4, err_msg=str(i))
        off += q
    # through autograd: the op's backward hands every object its own gradient
    ps = [p.clone().requires_grad_() for p in params]
    yy, _ = o.pointnet_max_fwd_multi(x, ps, list(points))
    (yy * dy).sum().backward()
    for i in range(len(points)):
        assert torch.equal(ps[i].grad, grads[i])
