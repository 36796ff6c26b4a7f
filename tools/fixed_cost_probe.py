#!/usr/bin/env python
"""Per-launch fixed cost of the 128-wide LDS-DMA GEMM: time vs K (intercept = fill + epilogue + drain), for the plain
store and the bias+tanh epilogue, fp32 and opt-in bf16-input inner loops.   python tools/fixed_cost_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgyminsertion_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
M, N = 32768, 256
st = torch.cuda.current_stream()


def run(K, epi, iters=200):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
    c = torch.empty(M, N, device=dev); b = torch.zeros(N, device=dev)
    def f():
        L.igi_gemm_f32(1, 1, M, N, K, _lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(c), N, _lib.ptr(b), None, 0, epi, 0, st.cuda_stream)
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for mode in (0, 1):
    L.igi_gemm_set_bf16_inputs(mode)
    for epi in (0, 1):
        ts = [(K, run(K, epi)) for K in (32, 64, 128, 256, 512, 1024)]
        slope = (ts[-1][1] - ts[-2][1]) / (1024 - 512) * 32
        print("bf16" if mode else "fp32", "epi", epi, " ".join(f"K{K}:{t:.1f}" for K, t in ts), f"| us per k-tile {slope:.2f}, intercept {ts[-1][1] - slope * 32:.1f}")
L.igi_gemm_set_bf16_inputs(0)
