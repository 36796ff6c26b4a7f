"""Depth-backbone golden vectors from the REFERENCE implementation (build container only): the reference's
``DepthOnlyFCBackbone54x96`` (algo/models/transformer/tact.py:81-113) forward + autograd on CPU for 32 images.

The module has 8.3 M parameters (33 MB), too large for a fixture: weights, inputs and the output gradient are
regenerated from fixed seeds by ``depth_case()`` below (imported by the test as well; torch's CPU generator is
deterministic), and the fixture stores what the reference computed from them: the output, every small gradient
tensor in full, and a strided sample + row/column sums of the 128 x 64768 gradient.

    python tests/golden/make_golden_depth.py  ->  tests/golden/depth.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def depth_case(latent=32, batch=32, seed=0):
    """state_dict (reference key order), images (B,1,54,96) in [0,1] with flat regions (pooling ties), dy."""
    g = torch.Generator().manual_seed(seed)
    sd = {
        "image_compression.0.weight": torch.randn(32, 1, 5, 5, generator=g) * 0.2,
        "image_compression.0.bias": torch.randn(32, generator=g) * 0.1,
        "image_compression.3.weight": torch.randn(64, 32, 3, 3, generator=g) * (1.0 / 288 ** 0.5),
        "image_compression.3.bias": torch.randn(64, generator=g) * 0.1,
        "image_compression.6.weight": torch.randn(128, 64 * 23 * 44, generator=g) * (1.0 / 64768 ** 0.5),
        "image_compression.6.bias": torch.randn(128, generator=g) * 0.1,
        "image_compression.8.weight": torch.randn(latent, 128, generator=g) * (1.0 / 128 ** 0.5),
        "image_compression.8.bias": torch.randn(latent, generator=g) * 0.1,
    }
    x = torch.rand(batch, 1, 54, 96, generator=g)
    x[:, :, :10, :20] = 0.0          # masked-out background, as the segmentation mask produces
    dy = torch.randn(batch, latent, generator=g)
    return sd, x, dy


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    import ref_harness as rh
    rh.install()
    from algo.models.transformer.tact import DepthOnlyFCBackbone54x96 as RefDepth  # reference
    torch.set_num_threads(4)
    sd, x, dy = depth_case()
    m = RefDepth(latent_dim=32, output_activation=None, num_channel=1)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    y = m(x)
    y.backward(dy)
    out = {"y": y.detach().numpy()}
    for k, p in m.named_parameters():
        gr = p.grad.numpy()
        if k == "image_compression.6.weight":
            out["g/" + k + "/sample"] = gr[::8, ::997].copy()
            out["g/" + k + "/rowsum"] = gr.sum(1)
            out["g/" + k + "/colsum"] = gr.sum(0)
        else:
            out["g/" + k] = gr
    np.savez_compressed(os.path.join(HERE, "depth.npz"), **out)
    print("wrote depth.npz", {k: v.shape for k, v in out.items()})
