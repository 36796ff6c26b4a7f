"""The persistent row-block backward level (csrc/rowblock.h) through the C ABI (igi_level_backward) against
(i) an fp64 evaluation of what autograd computes for Linear + Tanh (algo/models/models_split.py:27-38 under
loss.backward(), frozen_ppo.py:583-585):  dx = (dz W) * (1 - x^2),  dW = dz^T x,  db = sum_rows dz;
(ii) the tile kernels it replaces: dx must be BIT-IDENTICAL to igi_gemm_f32 with the tanh' epilogue (same MFMA chain),
dW / db equal to igi_gemm_f32's weight-gradient product up to the fp32 rounding of another summation grouping.
Tolerances: |err| <= 2e-6 * sum |a||b| + 1e-6 against fp64 (the GEMM tests' bound)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(rows, IN, nets, seed):
    g = torch.Generator().manual_seed(seed)
    dz = torch.randn(nets, rows, 128, generator=g) * 0.01
    x = torch.tanh(torch.randn(nets, rows, IN, generator=g) * 1.5)
    w = torch.randn(nets, 128, IN, generator=g) * 0.1
    return dz, x, w


def _run(rows, IN, nets, seed=0):
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    dz, x, w = _problem(rows, IN, nets, seed)
    parts = L.igi_level_backward_parts(rows, IN, nets)
    assert parts > 0
    d_dz, d_x, d_w = dz.cuda(), x.cuda(), w.cuda()
    dx = torch.full((nets, rows, IN), float("nan"), device="cuda")
    dwp = torch.full((parts, nets, 128, IN), float("nan"), device="cuda")
    dbp = torch.full((parts, nets, 128), float("nan"), device="cuda")
    rc = L.igi_level_backward(_lib.ptr(d_dz), _lib.ptr(d_w), _lib.ptr(d_x), _lib.ptr(dx), _lib.ptr(dwp), _lib.ptr(dbp),
                              rows, IN, 128, nets, _lib.current_stream())
    _lib.check(rc, "igi_level_backward")
    torch.cuda.synchronize()
    return (dz, x, w), (dx.cpu(), dwp.cpu(), dbp.cpu()), (d_dz, d_x, d_w)


@pytest.mark.parametrize("rows,IN,nets", [
    (16384, 256, 2),    # trunk layer 3 of BASELINE configs[1]: 32 ranges x 8 blocks, one workgroup per CU
    (16384, 256, 1),    # env_mlp layer 2: 64 ranges x 4 blocks
    (832, 192, 1),      # 13 blocks over 13 ranges, three slices: no XCD grouping (39 workgroups)
    (1600, 64, 2),      # one slice, uneven ranges (25 blocks over ... ranges)
    (256, 128, 2),      # the smallest row count the kernel takes
])
def test_level_backward_vs_fp64(rows, IN, nets):
    (dz, x, w), (dx, dwp, dbp), _ = _run(rows, IN, nets, seed=rows + IN)
    assert bool(torch.isfinite(dx).all() and torch.isfinite(dwp).all() and torch.isfinite(dbp).all())
    dzd, xd, wd = dz.double(), x.double(), w.double()
    ref_dx = torch.einsum("nrk,nki->nri", dzd, wd) * (1.0 - xd * xd)
    mag_dx = torch.einsum("nrk,nki->nri", dzd.abs(), wd.abs())
    assert bool(((dx.double() - ref_dx).abs() <= 2e-6 * mag_dx + 1e-6).all())
    ref_dw = torch.einsum("nrk,nri->nki", dzd, xd)
    mag_dw = torch.einsum("nrk,nri->nki", dzd.abs(), xd.abs())
    assert bool(((dwp.double().sum(0) - ref_dw).abs() <= 2e-6 * mag_dw + 1e-6).all())
    ref_db = dzd.sum(1)
    assert bool(((dbp.double().sum(0) - ref_db).abs() <= 2e-6 * dzd.abs().sum(1) + 1e-6).all())


def test_level_backward_dx_is_bit_identical_to_the_tile_kernel():
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    rows, IN = 4096, 256
    (dz, x, w), (dx, _, _), (d_dz, d_x, d_w) = _run(rows, IN, 1, seed=5)
    ref = torch.empty(rows, IN, device="cuda")
    # dgrad layout: A = dz [rows][128] k-contiguous, B(n, k) = W[k][n] reduction-major, epilogue 2 = acc * (1 - aux^2)
    rc = L.igi_gemm_f32(1, 0, rows, IN, 128, _lib.ptr(d_dz), 128, _lib.ptr(d_w), IN, _lib.ptr(ref), IN, None,
                        _lib.ptr(d_x), IN, 2, 0, _lib.current_stream())
    _lib.check(rc, "igi_gemm_f32")
    torch.cuda.synchronize()
    assert torch.equal(dx[0], ref.cpu())


def test_level_backward_is_bitwise_reproducible():
    a = _run(16384, 256, 2, seed=9)[1]
    b = _run(16384, 256, 2, seed=9)[1]
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_level_backward_refuses_other_shapes():
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    assert L.igi_level_backward_parts(100, 256, 1) == 0      # rows not a multiple of 64
    assert L.igi_level_backward_parts(4096, 100, 1) == 0     # input width not a multiple of 64
    t = torch.zeros(16, device="cuda")
    rc = L.igi_level_backward(_lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), 4096, 256, 64, 1,
                              _lib.current_stream())
    assert rc == -3   # IGI_E_UNSUPPORTED: out_features must be 128


@pytest.mark.parametrize("rows,IN", [(16384, 256), (832, 192), (256, 64)])
def test_level_backward_with_the_layer_below_vs_fp64(rows, IN):
    """igi_level_backward_below (k_rb_level<2, 1>): the weight / bias gradient of the layer below from the data-gradient
    tiles -- sum over parts of below_dW = dx^T x_below, below_db = column sums of dx, with dx = (dz W)(1 - x^2) evaluated in
    fp64 -- and this layer's own dW / db unchanged (bit-identical to igi_level_backward's)."""
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    dz, x, w = _problem(rows, IN, 1, seed=rows)
    xb = torch.randn(rows, 64, generator=torch.Generator().manual_seed(IN))
    parts = L.igi_level_backward_parts(rows, IN, 1)
    d = [t.cuda() for t in (dz, w, x, xb)]
    dwp = torch.full((parts, 128, IN), float("nan"), device="cuda")
    dbp = torch.full((parts, 128), float("nan"), device="cuda")
    bwp = torch.full((parts, IN, 64), float("nan"), device="cuda")
    bbp = torch.full((parts, IN), float("nan"), device="cuda")
    rc = L.igi_level_backward_below(_lib.ptr(d[0]), _lib.ptr(d[1]), _lib.ptr(d[2]), _lib.ptr(d[3]), _lib.ptr(dwp), _lib.ptr(dbp),
                                    _lib.ptr(bwp), _lib.ptr(bbp), rows, IN, 128, _lib.current_stream())
    _lib.check(rc, "igi_level_backward_below")
    torch.cuda.synchronize()
    (_, _, _), (_, dwp0, dbp0), _ = _run(rows, IN, 1, seed=rows)
    assert torch.equal(dwp.cpu(), dwp0[:, 0]) and torch.equal(dbp.cpu(), dbp0[:, 0])
    dzd, xd, wd, xbd = dz[0].double(), x[0].double(), w[0].double(), xb.double()
    dx = (dzd @ wd) * (1.0 - xd * xd)
    ref_w, mag_w = dx.t() @ xbd, dx.abs().t() @ xbd.abs()
    assert bool(((bwp.cpu().double().sum(0) - ref_w).abs() <= 4e-6 * mag_w + 1e-7).all())
    assert bool(((bbp.cpu().double().sum(0) - dx.sum(0)).abs() <= 4e-6 * dx.abs().sum(0) + 1e-7).all())
