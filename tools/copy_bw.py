"""Calibration: achievable device copy bandwidth on this box (torch copy, read+write)."""
import torch, time
def t(n_mb, iters=20):
    n = n_mb * 1024 * 1024 // 4
    a = torch.empty(n, device='cuda'); b = torch.randn(n, device='cuda')
    for _ in range(3): a.copy_(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): a.copy_(b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"copy {n_mb:5d} MB: {us:8.1f} us  -> {2*n_mb*1.048576/us*1e3/1e3:6.2f} TB/s (read+write)")
for mb in (16, 64, 128, 512, 2048): t(mb)
