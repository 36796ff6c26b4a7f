"""Device time of the student's decoder chain (96 -> 32 -> 256 -> 128 -> 64 -> 32 -> 6) forward: igi_mlp_forward (one launch) against
one igi_linear_forward per layer, from the library's own per-launch timestamps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from isaacgyminsertion_amd import ops, _lib  # noqa: F401

for rows in (2048, 8192, 16384):
    dims = [96, 32, 256, 128, 64, 32, 6]; acts = [2] * 5 + [1]
    x = torch.randn(rows, 96, device="cuda")
    ws = [torch.randn(o, i, device="cuda") / i ** 0.5 for i, o in zip(dims[:-1], dims[1:])]
    bs = [torch.zeros(o, device="cuda") for o in dims[1:]]

    def layers():
        h = x
        for w, b, a in zip(ws, bs, acts):
            h = torch.ops.mi355ppo.linear(h, w, b, a)
        return h

    def fused():
        return torch.ops.mi355ppo.mlp_fwd(x, ws, bs, acts)

    for name, f in (("fused", fused), ("layers", layers)):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(50):
            f()
        torch.cuda.synchronize()
        tot = sum(c["total_ms"] for c in _lib.prof_read())
        _lib.prof_enable(False)
        print(rows, name, "device us per chain:", round(tot / 50 * 1e3, 2))
