"""Extended-precision (x87 80-bit ``numpy.longdouble``, 64-bit mantissa) restatement of the collapsed bound in the
PyMC3 ``MarginalSparse(approx="VFE")`` op order -- TEST INFRASTRUCTURE ONLY (same rules as ``vfe_oracle``).

Purpose: an error yardstick.  ``vfe_oracle`` and the HIP path both work in fp64, so when they disagree at 1e-8 on an
ill-conditioned K_uu nothing says which one is off.  This module evaluates the same scalar with ~2000x smaller unit
round-off (its own Cholesky / triangular solves, numpy has no LAPACK for longdouble), so ``|F_fp64 - F_ext|`` IS the
fp64 method's rounding error up to ~1e-3 of itself.  Gradients by central differences in extended precision
(h ~ 1e-6 leaves ~1e-12 relative truncation and ~1e-13 rounding).

Follows the same reference lines as ``vfe_oracle.vfe_pymc3_order`` (models/bayesian_sgpr_hmc.py:60-71) and
``composite_oracle`` (experiments/co2_bayesian_sgpr_hmc.py:107-149).  PARITY UNPINNED like the rest of ``oracle/``.
"""
from __future__ import annotations

import numpy as np

LD = np.longdouble
PI = LD("3.14159265358979323846264338327950288")
LOG2PI = np.log(2 * PI)
SQRT3 = np.sqrt(LD(3))
SQRT5 = np.sqrt(LD(5))


def _ld(a):
    return np.asarray(a, dtype=LD)


def cholesky(A):
    """Lower Cholesky factor, right-looking, vectorised over the trailing block.  Raises on a non-positive pivot."""
    A = _ld(A).copy()
    n = A.shape[0]
    for j in range(n):
        d = A[j, j]
        if not d > 0:
            raise np.linalg.LinAlgError("leading minor of order %d not positive" % (j + 1))
        A[j, j] = np.sqrt(d)
        if j + 1 < n:
            A[j + 1:, j] /= A[j, j]
            A[j + 1:, j + 1:] -= np.outer(A[j + 1:, j], A[j + 1:, j])
    return np.tril(A)


def solve_lower(L, B):
    """L^-1 B by forward substitution (row by row, vectorised over the right-hand sides)."""
    L, B = _ld(L), _ld(B).copy()
    vec = B.ndim == 1
    if vec:
        B = B[:, None]
    for i in range(L.shape[0]):
        if i:
            B[i] -= L[i, :i] @ B[:i]
        B[i] /= L[i, i]
    return B[:, 0] if vec else B


def stationary_k(X, Z, ls, sf2, kernel_id=0):
    X, Z, ls = _ld(X), _ld(Z), _ld(ls)
    diff = X[:, None, :] / ls - Z[None, :, :] / ls
    r2 = (diff * diff).sum(-1)
    if kernel_id == 0:
        return LD(sf2) * np.exp(-r2 / 2)
    r = np.sqrt(r2)
    if kernel_id == 1:
        return LD(sf2) * (1 + SQRT3 * r) * np.exp(-SQRT3 * r)
    return LD(sf2) * (1 + SQRT5 * r + 5 * r2 / 3) * np.exp(-SQRT5 * r)


def composite_k(X, Z, block):
    """Sum of products of isotropic factors; ``block`` as documented in include/sgp.h (SGP_KERNEL_COMPOSITE)."""
    X, Z, b = _ld(X), _ld(Z), _ld(block)
    diff = X[:, None, :] - Z[None, :, :]
    r2 = (diff * diff).sum(-1)
    K = np.zeros(r2.shape, dtype=LD)
    for t in range(int(b[0])):
        base = 1 + 8 * t
        term = b[base] * np.ones_like(r2)
        for f in range(int(b[base + 1])):
            fb = base + 2 + 3 * f
            ty, ls, aux = int(b[fb]), b[fb + 1], b[fb + 2]
            if ty == 0:
                fac = np.exp(-r2 / (2 * ls * ls))
            elif ty == 1:
                a = SQRT3 * np.sqrt(r2) / ls
                fac = (1 + a) * np.exp(-a)
            elif ty == 2:
                a = SQRT5 * np.sqrt(r2) / ls
                fac = (1 + a + a * a / 3) * np.exp(-a)
            elif ty == 3:
                fac = (1 + r2 / (2 * aux * ls * ls)) ** (-aux)
            else:
                s = np.sin(PI * diff / aux)
                fac = np.exp(-(s * s).sum(-1) / (2 * ls * ls))
            term = term * fac
        K = K + term
    return K


def vfe_from_kernels(Kuu, Kuf, kdiag_sum, y, s2):
    """PyMC3 op order on already assembled kernel matrices (Kuu includes the jitter)."""
    y, s2 = _ld(y), LD(s2)
    N = y.shape[0]
    M = Kuu.shape[0]
    Luu = cholesky(Kuu)
    A = solve_lower(Luu, Kuf)
    trace = (LD(kdiag_sum) - (A * A).sum()) / (2 * s2)
    L_B = cholesky(np.eye(M, dtype=LD) + (A / s2) @ A.T)
    c = solve_lower(L_B, A @ (y / s2))
    logdet = N * np.log(s2) / 2 + np.log(np.diagonal(L_B)).sum()
    quad = ((y @ y) / s2 - c @ c) / 2
    return -(N * LOG2PI / 2 + logdet + quad + trace)


def vfe(X, y, Z, ls, sf2, s2, jitter=1e-6, kernel_id=0):
    Z = _ld(Z)
    Kuu = stationary_k(Z, Z, ls, sf2, kernel_id) + LD(jitter) * np.eye(Z.shape[0], dtype=LD)
    return vfe_from_kernels(Kuu, stationary_k(Z, X, ls, sf2, kernel_id), LD(sf2) * len(y), y, s2)


def vfe_composite(X, y, Z, block, s2, jitter=1e-6):
    Z, b = _ld(Z), _ld(block)
    Kuu = composite_k(Z, Z, b) + LD(jitter) * np.eye(Z.shape[0], dtype=LD)
    kd = sum(b[1 + 8 * t] for t in range(int(b[0])))
    return vfe_from_kernels(Kuu, composite_k(Z, X, b), kd * len(y), y, s2)


def central_diff(f, x, h=1e-6):
    """Gradient of a scalar function of a vector by central differences, everything in extended precision."""
    x = _ld(x)
    g = np.zeros(x.shape, dtype=LD)
    for i in range(x.size):
        e = np.zeros(x.shape, dtype=LD)
        e.flat[i] = LD(h) * max(LD(1), abs(x.flat[i]))
        g.flat[i] = (f(x + e) - f(x - e)) / (2 * e.flat[i])
    return g
