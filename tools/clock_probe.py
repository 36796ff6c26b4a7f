#!/usr/bin/env python
"""Is the steady-state shortfall of the fp32 GEMM k-loop (135 of 157 TFLOP/s) stall cycles or a lowered clock?
One long dispatch of the 128-wide LDS-DMA kernel (>= 4 ms so the counters are meaningful) is timed here and, under

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d <dir> -- python3 tools/clock_probe.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace ... (separate pass)

the effective clock is GRBM_GUI_ACTIVE / 8 / wall time (MI355X_MICROARCH.md, DVFS give-back) and the MFMA pipe
occupancy is SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CU-cycles)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaacgyminsertion_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("PROBE_M", 262144)), 256, int(os.environ.get("PROBE_K", 4096))
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
c = torch.empty(M, N, device=dev); b = torch.zeros(N, device=dev)
st = torch.cuda.current_stream()
def f():
    L.igi_gemm_f32(1, 1, M, N, K, _lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(c), N, _lib.ptr(b), None, 0, 0, 0, st.cuda_stream)
for _ in range(5): f()
torch.cuda.synchronize()
# >= 2 s of back-to-back launches so the power state is the loaded one
t_end = time.perf_counter() + 2.0
while time.perf_counter() < t_end:
    for _ in range(20): f()
    torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n): f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"M={M} N={N} K={K}: {dt * 1e3:.3f} ms per launch, {2.0 * M * N * K / dt / 1e12:.1f} TFLOP/s")
