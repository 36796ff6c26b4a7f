#!/usr/bin/env python
"""Does splitting a dependent GEMM chain into two independent half-batch chains on two HIP streams hide the
per-launch fill/drain?  (experiment behind DESIGN.md "what was tried")
    python tools/overlap_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgyminsertion_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
layers = [(32, 512), (512, 256), (256, 128), (128, 256), (256, 512)]   # fwd 3 layers + 2 dgrad-like products
M = 32768


def make(m):
    xs = [torch.randn(m, layers[0][0], device=dev)] + [torch.empty(m, n, device=dev) for _, n in layers]
    ws = [torch.randn(n, k, device=dev) * 0.05 for k, n in layers]
    bs = [torch.zeros(n, device=dev) for _, n in layers]
    return xs, ws, bs


def chain(buf, stream):
    xs, ws, bs = buf
    for i, (k, n) in enumerate(layers):
        rc = L.igi_gemm_f32(1, 1, xs[i].shape[0], n, k, _lib.ptr(xs[i]), k, _lib.ptr(ws[i]), k, _lib.ptr(xs[i + 1]), n,
                            _lib.ptr(bs[i]), None, 0, 1, 0, stream.cuda_stream)
        assert rc == 0


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


full = make(M)
ha, hb = make(M // 2), make(M // 2)
s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("one stream, full batch      : %.1f us" % timeit(lambda: chain(full, s0)))
print("one stream, two half batches: %.1f us" % timeit(lambda: (chain(ha, s0), chain(hb, s0))))


def two():
    # interleave the launches so both queues are fed
    for i in range(len(layers)):
        for buf, st in ((ha, s1), (hb, s2)):
            xs, ws, bs = buf
            k, n = layers[i]
            L.igi_gemm_f32(1, 1, xs[i].shape[0], n, k, _lib.ptr(xs[i]), k, _lib.ptr(ws[i]), k, _lib.ptr(xs[i + 1]), n,
                           _lib.ptr(bs[i]), None, 0, 1, 0, st.cuda_stream)


print("two streams, half batch each: %.1f us" % timeit(two))

for ways in (4, 8):
    bufs = [make(M // ways) for _ in range(ways)]
    streams = [torch.cuda.Stream() for _ in range(ways)]

    def multi():
        for i in range(len(layers)):
            for buf, st in zip(bufs, streams):
                xs, ws, bs = buf
                k, n = layers[i]
                L.igi_gemm_f32(1, 1, xs[i].shape[0], n, k, _lib.ptr(xs[i]), k, _lib.ptr(ws[i]), k, _lib.ptr(xs[i + 1]), n,
                               _lib.ptr(bs[i]), None, 0, 1, 0, st.cuda_stream)

    print("%d streams, 1/%d batch each    : %.1f us" % (ways, ways, timeit(multi)))
