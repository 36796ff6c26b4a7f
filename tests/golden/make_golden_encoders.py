"""Golden vectors for the student encoders from the REFERENCE modules (build container only):
``CNNWithSpatialSoftArgmax`` (tactile_cnn.py:62-79) and ``PointNet`` (pointnets.py:12-42) on CPU:
inputs, parameters (re-scaled so outputs are O(1): the reference's 0.02-std init gives ~1e-6 outputs,
SURVEY Appendix A15), outputs, and parameter gradients of a fixed linear functional of the output.

    python tests/golden/make_golden_encoders.py  ->  tests/golden/encoders.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.install()
from algo.models.transformer.tactile_cnn import CNNWithSpatialSoftArgmax  # noqa: E402  (reference)
from algo.models.transformer.pointnets import PointNet  # noqa: E402  (reference)


def tactile_case(out, tag, B, H, W, seed):
    torch.manual_seed(seed)
    m = CNNWithSpatialSoftArgmax(latent_dim=32)      # default torch init (kaiming-uniform): O(1) features
    x = torch.rand(B, 3, H, W)
    gy = torch.randn(B, 32)
    y = m(x)
    (y * gy).sum().backward()
    out[f"{tag}/x"] = x.numpy()
    out[f"{tag}/gy"] = gy.numpy()
    out[f"{tag}/y"] = y.detach().numpy()
    for k, v in m.state_dict().items():
        out[f"{tag}/p/{k}"] = v.numpy().copy()
    for k, v in m.named_parameters():
        out[f"{tag}/g/{k}"] = v.grad.numpy().copy()


def pointnet_case(out, tag, B, N, seed):
    torch.manual_seed(seed)
    m = PointNet()
    with torch.no_grad():                             # re-scale the 0.02-std init to O(1) activations
        m.local_mlp[0].weight.mul_(40.0)
        m.local_mlp[0].bias.normal_(0, 0.3)
        m.local_mlp[2].weight.mul_(8.0)
        m.local_mlp[2].bias.normal_(0, 0.1)
    x = torch.randn(B, N, 3) * 0.5
    gy = torch.randn(B, 256)
    y = m(x)
    (y * gy).sum().backward()
    out[f"{tag}/x"] = x.numpy()
    out[f"{tag}/gy"] = gy.numpy()
    out[f"{tag}/y"] = y.detach().numpy()
    for k, v in m.state_dict().items():
        out[f"{tag}/p/{k}"] = v.numpy().copy()
    for k, v in m.named_parameters():
        out[f"{tag}/g/{k}"] = v.grad.numpy().copy()


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    tactile_case(out, "tac32x64", 32, 32, 64, 0)      # reference default (3 gray fingers, crop_roi)
    tactile_case(out, "tac64x64", 32, 64, 64, 1)      # BASELINE wording (64x64)
    tactile_case(out, "tac_b5", 5, 32, 64, 2)         # the reference's own __main__ smoke shape (B=5)
    pointnet_case(out, "pn400", 8, 400, 3)
    pointnet_case(out, "pn37", 3, 37, 4)              # ragged point count
    path = os.path.join(HERE, "encoders.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")
