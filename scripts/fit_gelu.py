"""Coefficients and error report of gelu_erf16 (proteingym_amd/csrc/gemm16x_kernel.h).

    gelu(x) = max(x, 0) - a Phi(-a),   a = min(|x|, 6),   Phi(-a) = 2^P(a),   P of degree 6

P is fitted to log2 Phi(-a) by least squares whose weights are re-balanced until the ABSOLUTE error of a 2^P(a) is level over [0, 6]
(the quantity GELU adds to max(x, 0)); the kernel evaluates it in b = -a (odd coefficients change sign).  The report emulates the fp32
evaluation (fma, v_exp_f32 taken as correctly rounded) against fp64 next to torch's own fp32 gelu.      python scripts/fit_gelu.py
"""
import numpy as np
from numpy.polynomial import Polynomial, chebyshev as C
from scipy.special import erf, log_ndtr

A, DEG = 6.0, 6


def fit(deg=DEG):
    a = np.concatenate([np.linspace(0, 3, 150001), np.linspace(3, A, 50001)])
    log2_phi = log_ndtr(-a) / np.log(2)
    target = a * np.exp2(log2_phi)
    w = target * np.log(2) + 1e-12
    best = None
    for _ in range(400):
        c = C.chebfit(a * 2 / A - 1, log2_phi, deg, w=w)
        err = np.abs(a * np.exp2(C.chebval(a * 2 / A - 1, c)) - target)
        if best is None or err.max() < best[0]:
            best = (err.max(), c)
        w = w * (1 + 0.5 * err / err.max())
        w /= w.max()
    return best[0], Polynomial(C.cheb2poly(best[1]))(Polynomial([-1, 2 / A])).coef


def gelu_fp32(coef, x):
    x = x.astype(np.float32)
    b = np.maximum(-np.abs(x), np.float32(-A))
    signed = [np.float32(c if k % 2 == 0 else -c) for k, c in enumerate(coef)]
    q = np.full_like(b, signed[-1])
    for c in signed[-2::-1]:
        q = (q.astype(np.float64) * b + np.float64(c)).astype(np.float32)
    e = np.exp2(q.astype(np.float64)).astype(np.float32)
    return (b.astype(np.float64) * e + np.maximum(x, np.float32(0))).astype(np.float32)


if __name__ == "__main__":
    import torch
    fit_err, coef = fit()
    print("fit error of a 2^P(a) on [0, %g]: %.3e" % (A, fit_err))
    print("coefficients in b = -a, highest degree first:")
    for k in range(DEG, -1, -1):
        print("    %.9ef" % np.float32(coef[k] if k % 2 == 0 else -coef[k]))
    for name, x in (("[-10, 10]", np.linspace(-10, 10, 2000001)), ("N(0, 1.5)", np.random.default_rng(0).normal(0, 1.5, 1000000))):
        r = x.astype(np.float32).astype(np.float64)
        r = 0.5 * r * (1 + erf(r / np.sqrt(2)))
        mine = np.abs(gelu_fp32(coef, x) - r)
        theirs = np.abs(torch.nn.functional.gelu(torch.tensor(x, dtype=torch.float32)).double().numpy() - r)
        print("%-10s fp32 evaluation vs fp64: max %.3e mean %.3e   (torch fp32 gelu: max %.3e mean %.3e)"
              % (name, mine.max(), mine.mean(), theirs.max(), theirs.mean()))
