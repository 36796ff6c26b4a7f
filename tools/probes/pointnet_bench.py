"""(IGI_PN_COLMAX=1: the column-level running-maximum EXPERIMENT -- forward times only are meaningful, see csrc/pointnet.h.)
PointNet forward / backward launch times at the configs[3] per-rank size (2048 clouds x 400 points) and at 8192: the
mean of 50 launches between two stream events, with the algorithmic rate (2 * (3*64 + 64*256) flop per point forward)."""
import json
import os
import sys

if os.environ.get("IGI_PN_COLMAX", "0") != "0":
    os.environ["IGI_PN_COLMAX_TIMING"] = "1"      # the backward is timed on the experiment's (tile-only) arg-max: timing only

import torch

sys.path.insert(0, ".")
import isaacgyminsertion_amd.ops  # noqa: F401,E402

o = torch.ops.mi355ppo
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for B in (512, 1024, 2048, 8192):
    x = torch.randn(B, 400, 3, device=dev, generator=g) * 0.5
    p = torch.randn(16896, device=dev, generator=g) * 0.2
    dy = torch.randn(B, 256, device=dev, generator=g)
    y, idx = o.pointnet_max_fwd(x, p)
    o.pointnet_max_bwd(x, p, dy, idx)
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("fwd", lambda: o.pointnet_max_fwd(x, p)), ("bwd", lambda: o.pointnet_max_bwd(x, p, dy, idx))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            fn()
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / 50, 1)
    # plug + socket of one cloud tensor in ONE launch (round 6): B clouds x 800 points = 2 objects x 400
    x2 = torch.randn(B, 800, 3, device=dev, generator=g) * 0.5
    p2 = [p, torch.randn(16896, device=dev, generator=g) * 0.2]
    dy2 = torch.randn(B, 512, device=dev, generator=g)
    y2, idx2 = o.pointnet_max_fwd_multi(x2, p2, [400, 400])
    for name, fn in (("fwd_2obj", lambda: o.pointnet_max_fwd_multi(x2, p2, [400, 400])),
                     ("bwd_2obj", lambda: o.pointnet_max_bwd_multi(x2, p2, [400, 400], dy2, idx2))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            fn()
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / 50, 1)
    res["fwd_tflops"] = round(2.0 * (3 * 64 + 64 * 256) * B * 400 / res["fwd_us"] / 1e6, 1)
    res["fwd_2obj_tflops"] = round(2.0 * (3 * 64 + 64 * 256) * B * 800 / res["fwd_2obj_us"] / 1e6, 1)
    res["fwd_frac_of_157.3"] = round(res["fwd_tflops"] / 157.3, 3)
    out[f"{B} clouds x 400 points"] = res
print(json.dumps(out, indent=1))
