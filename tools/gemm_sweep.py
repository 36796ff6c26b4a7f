"""Stand-alone timing of igi_gemm_f32 over K (separates the per-launch fixed cost -- DMA prologue,
epilogue store burst, tail -- from the steady-state k-loop rate).  Run on the GPU box:
    gpurun -- python tools/gemm_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgyminsertion_amd import _lib  # noqa: E402

L = _lib.lib()


def run(akc, bkc, M, N, K, epi=0, iters=20):
    A = torch.randn(M, K, device='cuda') if akc else torch.randn(K, M, device='cuda')
    B = (torch.randn(N, K, device='cuda') if bkc else torch.randn(K, N, device='cuda')) * 0.05
    C = torch.zeros(M, N, device='cuda')
    bias = torch.zeros(N, device='cuda')
    aux = torch.zeros(M, N, device='cuda')
    lda, ldb = (K if akc else M), (K if bkc else N)

    def go():
        rc = L.igi_gemm_f32(akc, bkc, M, N, K, _lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(C), N, _lib.ptr(bias),
                            _lib.ptr(aux), N, epi, 0, _lib.current_stream())
        assert rc == 0
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"akc={akc} bkc={bkc} M={M} N={N} K={K} epi={epi}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF")


if __name__ == "__main__":
    for K in (128, 512, 1024, 2048, 4096):
        run(1, 1, 32768, 256, K, epi=1)
    for K in (256, 1024, 4096):
        run(1, 0, 32768, 512, K, epi=2)
    run(1, 1, 65536, 256, 512, epi=1)
    run(1, 1, 32768, 128, 256, epi=1)
    run(1, 1, 16384, 128, 256, epi=1)
