"""Diagnostic: error of the tactile CNN's parameter gradients against an fp64 CPU run, as a function of how the batch
is evaluated (one tall-tile launch, sums of 64- / 32-image launches) -- beside the fp32 CPU oracle's own error."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoders as oe  # noqa: E402
from isaacgyminsertion_amd.algo.models.transformer.tactile_cnn import CNNWithSpatialSoftArgmax  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "encoders.npz"))
B, H, W, tag = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sd = {k[len(tag) + 3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/p/")}
gen = torch.Generator().manual_seed(B + H)
x = torch.rand(B, 3, H, W, generator=gen)
gy = torch.randn(B, 32, generator=gen)
xc, gyc = x.cuda(), gy.cuda()


def hip(chunk):
    m = CNNWithSpatialSoftArgmax(32)
    m.load_state_dict(sd)
    m = m.cuda()
    for i in range(0, B, chunk):
        (m(xc[i:i + chunk]) * gyc[i:i + chunk]).sum().backward()
    torch.cuda.synchronize()
    return {k: p.grad.cpu().double().numpy() for k, p in m.named_parameters()}


y64, g64 = oe.value_and_grads(oe.tactile_cnn, x, sd, gy, dtype=torch.float64, chunk=128)
y32, g32 = oe.value_and_grads(oe.tactile_cnn, x, sd, gy, chunk=1024)
runs = {"cpu_fp32": {k: v.double().numpy() for k, v in g32.items()}, "hip_whole": hip(B), "hip_sum_of_64": hip(64),
        "hip_sum_of_32": hip(32)}
out = {}
for k in sd:
    ref = g64[k].numpy()
    sc = np.abs(ref).max()
    out[k] = {"max|g|": float(sc), **{n: float(np.abs(r[k] - ref).max() / sc) for n, r in runs.items()}}
print(json.dumps(out, indent=1))
