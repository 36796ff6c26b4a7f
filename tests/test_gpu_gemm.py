"""Parity of the exact-fp32 MFMA GEMM (csrc/gemm_f32.h) through the C ABI (igi_gemm_f32) against
an fp64 matmul, for the three operand layouts of a Linear layer (forward / dgrad / wgrad), every
tile shape the launcher can pick, ragged edges and all epilogues.
Tolerance: |err| <= 2e-6 * sum_k |a||b| + 1e-6 (fp32 fmaf-chain rounding, K <= 2048)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(akc, bkc, M, N, K, epi=0, accumulate=0, seed=0):
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g) * 0.3
    A = (a if akc else a.t()).contiguous().cuda()
    B = (b if bkc else b.t()).contiguous().cuda()
    lda = K if akc else M
    ldb = K if bkc else N
    bias = torch.randn(N, generator=g).cuda()
    aux = torch.tanh(torch.randn(M, N, generator=g)).cuda()
    c0 = torch.randn(M, N, generator=g)
    Cd = c0.clone().cuda()
    rc = L.igi_gemm_f32(int(akc), int(bkc), M, N, K, _lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(Cd), N,
                        _lib.ptr(bias), _lib.ptr(aux), N, epi, accumulate, _lib.current_stream())
    _lib.check(rc, "igi_gemm_f32")
    torch.cuda.synchronize()
    ref = a.double() @ b.double().t()
    mag = a.abs().double() @ b.abs().double().t()
    if accumulate:
        ref = ref + c0.double()
        mag = mag + c0.abs().double()
    if epi == 1:
        ref = torch.tanh(ref + bias.cpu().double())
    elif epi == 3:
        ref = ref + bias.cpu().double()
    elif epi == 2:
        ref = ref * (1.0 - aux.cpu().double() ** 2)
    err = (Cd.cpu().double() - ref).abs()
    tol = 2e-6 * mag + 1e-6
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} at M{M} N{N} K{K} akc{akc} bkc{bkc} epi{epi}"


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [
    (256, 256, 64),      # 128x128 tiles, exact
    (300, 130, 23),      # ragged everything, K not multiple of 4 (scalar loads)
    (128, 8, 128),       # narrow N  -> 128x32 tile
    (8, 128, 200),       # narrow M  -> 32x128 tile
    (200, 48, 512),      # 128x64 tile
    (33, 33, 1),         # K = 1
])
def test_gemm_layouts(akc, bkc, M, N, K):
    _run(akc, bkc, M, N, K)


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [
    (1000, 256, 128),    # LDS-DMA kernel, BN=256, ragged M (clamped rows)
    (260, 512, 64),      # two n-tiles of 256, 2 k-tiles (prologue only)
    (128, 128, 32),      # single k-tile
    (132, 192, 96),      # BN=128, second n-tile half empty (clamped columns), 3 k-tiles
    (512, 64, 1024),     # N=64 < BN, long k loop (ring wraps many times)
])
def test_gemm_dma_layouts(akc, bkc, M, N, K):
    _run(akc, bkc, M, N, K, seed=M + N + K)


@pytest.mark.parametrize("epi", [1, 2, 3])
def test_gemm_dma_epilogues(epi):
    _run(1, 1, 640, 256, 256, epi=epi)
    _run(1, 0, 384, 128, 128, epi=epi)


@pytest.mark.parametrize("epi", [1, 2, 3])
def test_gemm_epilogues(epi):
    _run(1, 1, 257, 96, 72, epi=epi)
    _run(1, 0, 130, 40, 64, epi=epi, accumulate=1)


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (row<->col swap)."""
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    n = 64
    A = torch.eye(n).cuda()
    Bm = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0)
    B = Bm.cuda()            # B(n,k) k-contiguous: C = A @ B^T = B^T
    Cd = torch.zeros(n, n).cuda()
    rc = L.igi_gemm_f32(1, 1, n, n, n, _lib.ptr(A), n, _lib.ptr(B), n, _lib.ptr(Cd), n, None, None, 0, 0, 0,
                        _lib.current_stream())
    _lib.check(rc)
    torch.cuda.synchronize()
    assert torch.equal(Cd.cpu(), Bm.t().contiguous())


def test_gemm_bitwise_is_fmaf_chain():
    """v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain: bit-identical to the same chain on the host."""
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    M, N, K = 32, 32, 48
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    Cd = torch.zeros(M, N).cuda()
    ad, bd = a.cuda(), b.cuda()
    rc = L.igi_gemm_f32(1, 1, M, N, K, _lib.ptr(ad), K, _lib.ptr(bd), K, _lib.ptr(Cd), N, None, None,
                        0, 0, 0, _lib.current_stream())
    _lib.check(rc)
    torch.cuda.synchronize()
    an, bn = a.numpy(), b.numpy()
    ref = np.zeros((M, N), dtype=np.float32)
    for k in range(K):  # fma in fp64 then round == fmaf for fp32 operands (product is exact in fp64)
        ref = (ref.astype(np.float64) + an[:, k:k + 1].astype(np.float64) * bn[None, :, k].astype(np.float64)).astype(np.float32)
    assert np.array_equal(Cd.cpu().numpy(), ref)


def test_dma_gemm_bitwise_is_fixed_order_fmaf_chain():
    """The LDS-DMA kernel feeds each 32x32x2 MFMA the k pair (k, k+4) of an 8-group (a 16-byte LDS read per lane
    covers four consecutive k): still ONE fmaf chain per output element, in the fixed order
    0,4,1,5,2,6,3,7 inside every group of eight -- bit-identical to that chain on the host."""
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    M, N, K = 256, 128, 64          # aligned, K % 32 == 0 -> gemm_dma_kernel
    g = torch.Generator().manual_seed(4)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    Cd = torch.zeros(M, N).cuda()
    ad, bd = a.cuda(), b.cuda()
    rc = L.igi_gemm_f32(1, 1, M, N, K, _lib.ptr(ad), K, _lib.ptr(bd), K, _lib.ptr(Cd), N, None, None,
                        0, 0, 0, _lib.current_stream())
    _lib.check(rc)
    torch.cuda.synchronize()
    an, bn = a.numpy(), b.numpy()
    ref = np.zeros((M, N), dtype=np.float32)
    for k8 in range(0, K, 8):
        for k in (0, 4, 1, 5, 2, 6, 3, 7):
            kk = k8 + k
            ref = (ref.astype(np.float64) + an[:, kk:kk + 1].astype(np.float64) * bn[None, :, kk].astype(np.float64)).astype(np.float32)
    assert np.array_equal(Cd.cpu().numpy(), ref)
