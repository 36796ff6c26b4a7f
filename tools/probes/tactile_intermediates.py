"""Diagnostic: where does the tactile encoder's gradient lose accuracy?  Runs igi_tactile_forward / backward through the
dispatcher ops, reads a1..a3, feat, dfeat, dz3..dz1 out of the workspace (offsets replicated from make_tactile_plan) and
compares each with an fp64 CPU evaluation."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import encoders as oe  # noqa: E402
from isaacgyminsertion_amd import ops  # noqa: E402,F401

G = np.load(os.path.join(ROOT, "tests", "golden", "encoders.npz"))
B, H, W, tag = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sd = {k[len(tag) + 3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/p/")}
gen = torch.Generator().manual_seed(B + H)
x = torch.rand(B, 3, H, W, generator=gen)
gy = torch.randn(B, 32, generator=gen)


def ru64(n):
    return (n + 63) // 64 * 64


H1, W1 = (H - 8) // 2 + 1, (W - 8) // 2 + 1
H2, W2 = H1 - 3, W1 - 3
H3, W3 = H2 - 2, W2 - 2
M1, M2, M3 = B * H1 * W1, B * H2 * W2, B * H3 * W3
off, o = {}, 0
for name, nbytes in [("zero", 256), ("xin", 4 * B * H * W * 4), ("w1r", 4 * 32 * 256), ("w2r", 4 * 64 * 512), ("w3r", 4 * 64 * 576),
                     ("w2d", 4 * 32 * 1024), ("w3d", 4 * 64 * 576), ("a1", 4 * M1 * 32), ("a2", 4 * M2 * 64), ("a3", 4 * M3 * 64),
                     ("sstat", 4 * B * 64 * 2), ("feat", 4 * B * 128), ("dfeat", 4 * B * 128), ("dz3", 4 * M3 * 64),
                     ("dz2", 4 * M2 * 64), ("dz1", 4 * M1 * 32)]:
    off[name] = (o, nbytes)
    o += ru64(nbytes)

flat = torch.cat([sd[k].reshape(-1) for k in sd]).cuda()
y, ws = torch.ops.mi355ppo.tactile_cnn_fwd(x.cuda(), flat, 32)
grads = torch.ops.mi355ppo.tactile_cnn_bwd(gy.cuda(), flat, ws, H, W)
torch.cuda.synchronize()


def get(name, shape):
    a, n = off[name]
    return ws[a:a + n].view(torch.float32).reshape(shape).cpu().double()


# fp64 reference with retained intermediates (NCHW -> channels-last to match)
p = {k: v.double().requires_grad_(True) for k, v in sd.items()}
xd = x.double()
z1 = F.conv2d(xd, p["cnn.0.weight"], p["cnn.0.bias"], stride=2); a1 = F.relu(z1)
z2 = F.conv2d(a1, p["cnn.2.weight"], p["cnn.2.bias"]); a2 = F.relu(z2)
z3 = F.conv2d(a2, p["cnn.4.weight"], p["cnn.4.bias"]); a3 = F.relu(z3)
feat = oe.spatial_softargmax(a3, True)
for t in (z1, z2, z3, feat):
    t.retain_grad()
yy = F.linear(feat, p["cnn.7.weight"], p["cnn.7.bias"])
(yy * gy.double()).sum().backward()


def cl(t):   # (B, C, H, W) -> (B*H*W, C)
    return t.detach().permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


out = {"y": rel(y.cpu().double(), yy.detach()),
       "a1": rel(get("a1", (M1, 32)), cl(a1)), "a2": rel(get("a2", (M2, 64)), cl(a2)), "a3": rel(get("a3", (M3, 64)), cl(a3)),
       "feat": rel(get("feat", (B, 128)), feat.detach()), "dfeat": rel(get("dfeat", (B, 128)), feat.grad),
       "dz3": rel(get("dz3", (M3, 64)), cl(z3.grad)), "dz2": rel(get("dz2", (M2, 64)), cl(z2.grad)),
       "dz1": rel(get("dz1", (M1, 32)), cl(z1.grad))}
# column sums of the device's dz3 in fp64 vs the exact bias gradient: isolates dz3 from the summation
d3 = get("dz3", (M3, 64))
out["db3_from_device_dz3_summed_fp64"] = rel(d3.sum(0), p["cnn.4.bias"].grad)
out["db3_device"] = rel(grads[32 * 192 + 32 + 64 * 512 + 64 + 64 * 576:][:64].cpu().double(), p["cnn.4.bias"].grad)
err = (d3 - cl(z3.grad))
out["dz3_err_mean_over_all"] = float(err.mean() / cl(z3.grad).abs().mean())
out["dz3_err_where_ref_zero_max"] = float(err[cl(z3.grad) == 0].abs().max()) if (cl(z3.grad) == 0).any() else 0.0
out["relu_mask_mismatch"] = int(((d3 != 0) != (cl(z3.grad) != 0)).sum())
print(json.dumps(out, indent=1))
ref3 = cl(z3.grad)
i = int(err.abs().argmax())
m_, c_ = divmod(i, 64)
b_, k_ = divmod(m_, H3 * W3)
print("worst dz3 element: b", b_, "k", k_, "c", c_, "device", float(d3[m_, c_]), "ref", float(ref3[m_, c_]),
      "max|ref|", float(ref3.abs().max()))
print("  a3 there: device", float(get("a3", (M3, 64))[m_, c_]), "ref", float(cl(a3)[m_, c_]))
fd, fr = get("feat", (B, 128)), feat.detach()
print("  feat (b,c): device", fd[b_, 2 * c_:2 * c_ + 2].tolist(), "ref", fr[b_, 2 * c_:2 * c_ + 2].tolist(), "max|feat|", float(fr.abs().max()))
gd, gr_ = get("dfeat", (B, 128)), feat.grad
print("  dfeat (b,c): device", gd[b_, 2 * c_:2 * c_ + 2].tolist(), "ref", gr_[b_, 2 * c_:2 * c_ + 2].tolist())
col_d = d3.reshape(B, H3 * W3, 64)[b_, :, c_]
col_r = ref3.reshape(B, H3 * W3, 64)[b_, :, c_]
print("  that (b,c) column: device", [round(float(v), 7) for v in col_d[:24]])
print("  that (b,c) column: ref   ", [round(float(v), 7) for v in col_r[:24]])
ss = get("sstat", (B * 64, 2))[b_ * 64 + c_]
a3c = cl(a3).reshape(B, H3 * W3, 64)[b_, :, c_]
print("  sstat device (m, s):", ss.tolist(), " ref m, s:", float(a3c.max()), float(torch.exp(a3c - a3c.max()).sum()))
# how many elements are off by more than 1e-3 of the maximum, and do they cluster in a few (b, c) columns?
bad = (err.abs() > 1e-3 * ref3.abs().max())
print("elements off by > 1e-3 max:", int(bad.sum()), "of", bad.numel(), "; distinct (b,c) columns:",
      int(bad.reshape(B, H3 * W3, 64).any(1).sum()))
