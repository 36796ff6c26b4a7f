"""Calibration: achievable pure-write bandwidth (torch fill_) vs buffer size, next to the GEMM epilogue's."""
import torch
for mb in (8, 16, 32, 64, 128, 512):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device='cuda')
    for _ in range(5): a.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 200
    e0.record()
    for _ in range(iters): a.fill_(2.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"fill {mb:4d} MB: {us:7.1f} us -> {mb / us * 1e6 / 1e6:6.2f} TB/s")
