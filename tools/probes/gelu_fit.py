"""Coefficients and accuracy of the single-range erf-GELU of csrc/pointnet.h (gelu_q / gelu_cdf):
erfc(z) = 2^(z * Q(z)) on [0, ZMAX], Q fitted (Chebyshev basis, Lawson-reweighted least squares) to -log(erfc(z)) / z with
the error weighted by erfc(z) * z, i.e. by its effect on erfc.  Prints, per (ZMAX, degree): the maximum absolute error of
Phi and of x * Phi evaluated in simulated fp32 against fp64 over |x| <= 9, next to the error of the reference's own formula
0.5 * x * (1 + erf(x / sqrt 2)) evaluated in fp32.  `python tools/probes/gelu_fit.py coeffs` prints the (4.0, 9)
coefficients with -log2(e) folded in, as they stand in the kernel."""
import sys

import numpy as np
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as Ch, polynomial as Po
def fit(zmax, deg, iters=60):
    z = (np.cos(np.linspace(0, np.pi, 4001)) + 1) * 0.5 * zmax   # dense Chebyshev-ish grid
    z = np.maximum(z, 1e-9)
    t = -np.log(erfc(z)); T = t / z
    wt = erfc(z) * z          # sensitivity of e = exp(-z T) to T
    wt = np.maximum(wt, 1e-12)
    lw = np.ones_like(z)
    u = 2 * z / zmax - 1
    for _ in range(iters):
        c = Ch.chebfit(u, T, deg, w=wt * lw)
        err = np.abs((Ch.chebval(u, c) - T) * wt)
        lw = lw * (err / err.max() + 1e-3) ** 0.5
        lw /= lw.max()
    # convert to monomial in z
    pu = Ch.cheb2poly(c)                      # poly in u
    # u = a z + b
    a, b = 2 / zmax, -1.0
    pz = np.zeros(1)
    base = np.ones(1)
    for k, ck in enumerate(pu):
        pz = Po.polyadd(pz, ck * base)
        base = Po.polymul(base, np.array([b, a]))
    return pz
def evalf32(pz, x):
    # fp32 Horner simulation: Phi(x) and gelu
    x = x.astype(np.float32)
    z = np.minimum(np.abs(x) * np.float32(0.70710678118654752440), np.float32(ZMAX)).astype(np.float32)
    c = (pz * -1.4426950408889634).astype(np.float32)   # fold -log2(e)
    p = np.full_like(z, c[-1])
    for ck in c[-2::-1]:
        p = (p * z + ck).astype(np.float32)          # fma approximated by double-rounded mul-add (close enough)
    e = np.exp2((z * p).astype(np.float32).astype(np.float64)).astype(np.float32)
    s = (np.float32(0.5) * e).astype(np.float32)
    phi = np.where(x >= 0, (np.float32(1) - s).astype(np.float32), s)
    return phi, (x * phi).astype(np.float32)
if len(sys.argv) > 1 and sys.argv[1] == "coeffs":
    ZMAX = 4.0
    print(", ".join(f"{v:.9e}f" for v in fit(4.0, 9) * -1.4426950408889634))
    sys.exit(0)
for ZMAX, deg in ((4.0, 8), (4.0, 9), (4.0, 10), (4.5, 10), (5.0, 10), (5.0, 11), (5.0, 12)):
    pz = fit(ZMAX, deg)
    x = np.linspace(-9, 9, 2000001)
    phi, g = evalf32(pz, x)
    phit = 0.5 * erfc(-x / np.sqrt(2)); gt = x * phit
    # torch-like fp32 reference: 0.5*x*(1+erf(x/sqrt2)) in fp32
    x32 = x.astype(np.float32)
    gref = (np.float32(0.5) * x32 * (np.float32(1) + erf((x32 * np.float32(0.70710678)).astype(np.float64)).astype(np.float32))).astype(np.float32)
    print(ZMAX, deg, "max|phi err|", np.abs(phi - phit).max(), "max|gelu err|", np.abs(g - gt).max(), " ref fp32 formula gelu err", np.abs(gref - gt).max(),
          " max rel gelu err (|x|<6)", (np.abs(g - gt) / np.maximum(np.abs(gt), 1e-30))[np.abs(x) < 4].max())
