"""CPU: oracle/encoders.py (PyTorch-CPU restatement of the reference's CNNWithSpatialSoftArgmax / PointNet) against
golden vectors produced by the reference's own modules (tests/golden/make_golden_encoders.py).  fp32 run: outputs
1e-6 abs + 1e-5 rel, gradients 2e-5 * max|g| (same ATen kernels as the reference, so nearly bit-equal); the fp64
run measures the reference's own fp32 rounding: <= 1.1e-4 of a tensor's largest gradient entry at 32 x 64, up to
4.5e-3 at 64 x 64 (the soft-argmax gradient sums to zero over 576 positions, so the conv gradients under it cancel
heavily) -- the yardstick the HIP tests at scale use (error against fp64 <= 3 x the fp32 oracle's own error)."""
import os

import numpy as np
import pytest
import torch

from oracle import encoders as oe

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "encoders.npz"))


def _case(tag):
    sd = {k[len(tag) + 3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/p/")}
    return sd, torch.from_numpy(G[f"{tag}/x"]), torch.from_numpy(G[f"{tag}/gy"])


@pytest.mark.parametrize("tag,fn", [("tac32x64", oe.tactile_cnn), ("tac64x64", oe.tactile_cnn), ("tac_b5", oe.tactile_cnn),
                                    ("pn400", oe.pointnet), ("pn37", oe.pointnet)])
def test_encoder_oracle_matches_reference_golden(tag, fn):
    torch.set_num_threads(1)
    sd, x, gy = _case(tag)
    y, g = oe.value_and_grads(fn, x, sd, gy)
    np.testing.assert_allclose(y.numpy(), G[f"{tag}/y"], atol=1e-6, rtol=1e-5)
    for k in sd:
        ref = G[f"{tag}/g/{k}"]
        np.testing.assert_allclose(g[k].numpy(), ref, atol=2e-5 * np.abs(ref).max(), rtol=1e-4, err_msg=k)
    # fp64 rerun, evaluated in two chunks (the functional is additive over samples): the reference's fp32 rounding
    y64, g64 = oe.value_and_grads(fn, x, sd, gy, dtype=torch.float64, chunk=(x.shape[0] + 1) // 2)
    np.testing.assert_allclose(y64.numpy(), G[f"{tag}/y"], atol=2e-5, rtol=1e-4)
    for k in sd:
        ref = G[f"{tag}/g/{k}"]
        np.testing.assert_allclose(g64[k].numpy(), ref, atol=1e-2 * np.abs(ref).max(), rtol=1e-2, err_msg=k)


def test_softargmax_oracle_matches_reference_golden():
    S = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "softargmax.npz"))
    for tag in ("n_8x24", "n_24x24", "i_10x26", "n_1ch"):
        x = torch.from_numpy(S[f"{tag}/x"]).requires_grad_()
        y = oe.spatial_softargmax(x, bool(S[f"{tag}/normalize"]))
        (y * torch.from_numpy(S[f"{tag}/gy"])).sum().backward()
        np.testing.assert_allclose(y.detach().numpy(), S[f"{tag}/y"], atol=1e-6, rtol=1e-6, err_msg=tag)
        np.testing.assert_allclose(x.grad.numpy(), S[f"{tag}/gx"], atol=1e-7, rtol=1e-5, err_msg=tag)
