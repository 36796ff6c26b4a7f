"""Standalone SpatialSoftArgmax golden vectors from the REFERENCE module (build container only):
algo/models/transformer/tactile_cnn.py:7-58 forward + autograd backward on non-square feature maps, with normalised
and integer coordinate grids and channel counts other than the tactile encoder's 64.

    python tests/golden/make_golden_softargmax.py  ->  tests/golden/softargmax.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.install()
from algo.models.transformer.tactile_cnn import SpatialSoftArgmax  # noqa: E402  (reference)

if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    g = torch.Generator().manual_seed(0)
    for tag, (b, c, h, w), norm in (("n_8x24", (5, 64, 8, 24), True), ("n_24x24", (3, 7, 24, 24), True),
                                     ("i_10x26", (2, 3, 10, 26), False), ("n_1ch", (4, 1, 5, 3), True)):
        x = (2.0 * torch.randn(b, c, h, w, generator=g)).requires_grad_()
        gy = torch.randn(b, 2 * c, generator=g)
        y = SpatialSoftArgmax(normalize=norm)(x)
        (y * gy).sum().backward()
        out[f"{tag}/x"], out[f"{tag}/gy"] = x.detach().numpy(), gy.numpy()
        out[f"{tag}/y"], out[f"{tag}/gx"] = y.detach().numpy(), x.grad.numpy()
        out[f"{tag}/normalize"] = np.array(int(norm))
    path = os.path.join(HERE, "softargmax.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} KB")
