"""CPU oracle for sum-of-products ("composite") covariance functions -- TEST INFRASTRUCTURE ONLY (same rules as
vfe_oracle.py: only tests/, smoke() and bench.py's cpu_baseline may import it).

PARITY UNPINNED.  The reference builds its CO2 covariance from PyMC3 covariance classes
(experiments/co2_bayesian_sgpr_hmc.py:107-149):

    n_per**2   * Periodic(1, period=1, ls=l_psmooth) * ExpQuad(1, l_pdecay)
  + n_med**2   * RatQuad(1, l_med, alpha)
  + n_trend**2 * ExpQuad(1, l_trend)
  + n_noise**2 * Matern32(1, l_noise)

PyMC3 is neither vendored nor installed (3.9-3.11, unpinned), so the published definitions of
``pm.gp.cov.{ExpQuad, Matern32, Matern52, RatQuad, Periodic}`` are restated here in torch fp64 (isotropic: one
lengthscale per factor, as the reference uses them with input_dim=1); autograd supplies every gradient the HIP
path has to reproduce.  The bound itself is vfe_oracle's PyMC3 ``MarginalSparse`` op order with this covariance.

Parameter block (shared with include/sgp.h, SGP_COMP_*), 33 doubles:
    [0] nterms (1..4) ; term t at base = 1 + 8 t:
    [base] amp2  [base+1] nfac (1..2) ; factor f at fb = base + 2 + 3 f:  [fb] type  [fb+1] ls  [fb+2] aux
    type: 0 ExpQuad, 1 Matern32, 2 Matern52, 3 RatQuad (aux = alpha), 4 Periodic (aux = period)
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import vfe_oracle as O

DT = torch.float64
COMP_LEN = 33
MAX_TERMS, MAX_FACTORS = 4, 2
EXPQUAD, MATERN32, MATERN52, RATQUAD, PERIODIC = 0, 1, 2, 3, 4
KERNEL_COMPOSITE = 3


def make_block(terms):
    """terms: list of (amp2, [(type, ls, aux), ...]) -> 33-double block."""
    if not 1 <= len(terms) <= MAX_TERMS:
        raise ValueError("1..4 terms")
    b = np.zeros(COMP_LEN)
    b[0] = len(terms)
    for t, (amp2, facs) in enumerate(terms):
        if not 1 <= len(facs) <= MAX_FACTORS:
            raise ValueError("1..2 factors per term")
        base = 1 + 8 * t
        b[base] = amp2
        b[base + 1] = len(facs)
        for f, fac in enumerate(facs):
            ty, ls = fac[0], fac[1]
            aux = fac[2] if len(fac) > 2 else 0.0
            b[base + 2 + 3 * f: base + 5 + 3 * f] = (ty, ls, aux)
    return b


def grad_slots(block):
    """Indices of the block that carry a derivative (amp2, ls, and aux of RatQuad / Periodic factors)."""
    b = np.asarray(block, dtype=np.float64)
    idx = []
    for t in range(int(b[0])):
        base = 1 + 8 * t
        idx.append(base)
        for f in range(int(b[base + 1])):
            fb = base + 2 + 3 * f
            idx.append(fb + 1)
            if int(b[fb]) in (RATQUAD, PERIODIC):
                idx.append(fb + 2)
    return idx


def _factor(ty, diff, r2, ls, aux):
    if ty == EXPQUAD:
        return torch.exp(-0.5 * r2 / (ls * ls))
    if ty == MATERN32:
        a = math.sqrt(3.0) * torch.sqrt(r2 + 1e-300) / ls
        return (1.0 + a) * torch.exp(-a)
    if ty == MATERN52:
        a = math.sqrt(5.0) * torch.sqrt(r2 + 1e-300) / ls
        return (1.0 + a + a * a / 3.0) * torch.exp(-a)
    if ty == RATQUAD:
        return torch.pow(1.0 + r2 / (2.0 * aux * ls * ls), -aux)
    if ty == PERIODIC:
        s = torch.sin(math.pi * diff / aux)
        return torch.exp(-0.5 * (s * s).sum(-1) / (ls * ls))
    raise ValueError("unknown factor type %r" % (ty,))


def composite_k(X, Z, block, structure=None):
    """K[n, m].  ``block`` may be a torch tensor requiring grad; the integer structure (nterms, nfac, types) is read
    from ``structure`` (a plain block) when given, else from ``block`` itself."""
    X, Z = O._t(X), O._t(Z)
    blk = block if isinstance(block, torch.Tensor) else O._t(block)
    st = np.asarray(structure if structure is not None else blk.detach().numpy(), dtype=np.float64)
    diff = X[:, None, :] - Z[None, :, :]
    r2 = (diff * diff).sum(-1)
    K = torch.zeros(X.shape[0], Z.shape[0], dtype=DT)
    for t in range(int(st[0])):
        base = 1 + 8 * t
        term = blk[base] * torch.ones_like(r2)
        for f in range(int(st[base + 1])):
            fb = base + 2 + 3 * f
            term = term * _factor(int(st[fb]), diff, r2, blk[fb + 1], blk[fb + 2])
        K = K + term
    return K


def kdiag(block):
    b = np.asarray(block if not isinstance(block, torch.Tensor) else block.detach().numpy(), dtype=np.float64)
    return float(sum(b[1 + 8 * t] for t in range(int(b[0]))))


def vfe_composite(X, y, Z, block, s2, jitter=1e-6, structure=None):
    """PyMC3 MarginalSparse(VFE) op order (vfe_oracle.vfe_pymc3_order) with the composite covariance; differentiable."""
    X, y, Z = O._t(X), O._t(y), O._t(Z)
    blk = block if isinstance(block, torch.Tensor) else O._t(block)
    st = np.asarray(structure if structure is not None else blk.detach().numpy(), dtype=np.float64)
    s2 = s2 if isinstance(s2, torch.Tensor) else torch.tensor(float(s2), dtype=DT)
    M, N = Z.shape[0], X.shape[0]
    Kuu = composite_k(Z, Z, blk, st) + jitter * torch.eye(M, dtype=DT)
    Kuf = composite_k(Z, X, blk, st)
    Luu = torch.linalg.cholesky(Kuu)
    A = torch.linalg.solve_triangular(Luu, Kuf, upper=False)
    kd = sum(blk[1 + 8 * t] for t in range(int(st[0])))
    trace = (N * kd - (A * A).sum()) / (2.0 * s2)
    L_B = torch.linalg.cholesky(torch.eye(M, dtype=DT) + (A / s2) @ A.T)
    c = torch.linalg.solve_triangular(L_B, (A @ (y / s2))[:, None], upper=False)[:, 0]
    logmarg = -(0.5 * N * O.LOG2PI + 0.5 * N * torch.log(s2) + torch.log(torch.diagonal(L_B)).sum()
                + 0.5 * ((y * y).sum() / s2 - (c * c).sum()))
    return logmarg - trace


def vfe_composite_and_grads(X, y, Z, block, s2, jitter=1e-6):
    st = np.asarray(block, dtype=np.float64)
    with torch.enable_grad():
        blk = O._t(block).clone().requires_grad_(True)
        Zt = O._t(Z).clone().requires_grad_(True)
        s2t = torch.tensor(float(s2), dtype=DT, requires_grad=True)
        F = vfe_composite(X, y, Zt, blk, s2t, jitter, structure=st)
        F.backward()
    g = torch.zeros(COMP_LEN, dtype=DT)
    for i in grad_slots(st):
        g[i] = blk.grad[i]
    return float(F.detach()), {"block": g, "s2": float(s2t.grad), "Z": Zt.grad.clone()}


def predict_composite(Xs, X, y, Z, block, s2, jitter=1e-6, pred_noise=True):
    """Predictive mean / covariance with the optimal q(u) (models/sgpr.py:256-286 algebra, composite covariance)."""
    Xs, X, y, Z = O._t(Xs), O._t(X), O._t(y), O._t(Z)
    st = np.asarray(block, dtype=np.float64)
    blk = O._t(block)
    M = Z.shape[0]
    Kuu = composite_k(Z, Z, blk, st) + jitter * torch.eye(M, dtype=DT)
    Kuf = composite_k(Z, X, blk, st)
    Kus = composite_k(Z, Xs, blk, st)
    Kss = composite_k(Xs, Xs, blk, st)
    Sigma = Kuu + Kuf @ Kuf.T / s2
    mean = Kus.T @ torch.linalg.solve(Sigma, Kuf @ y) / s2
    cov = Kss - Kus.T @ torch.linalg.solve(Kuu, Kus) + Kus.T @ torch.linalg.solve(Sigma, Kus)
    if pred_noise:
        cov = cov + s2 * torch.eye(Xs.shape[0], dtype=DT)
    return mean, cov


def co2_block(n_per=1.0, l_psmooth=1.0, l_pdecay=1.0, n_med=1.0, l_med=1.0, alpha=1.0, n_trend=1.0, l_trend=1.0,
              n_noise=1.0, l_noise=1.0, period=1.0):
    """The reference's CO2 covariance (experiments/co2_bayesian_sgpr_hmc.py:107-149) as a parameter block."""
    return make_block([
        (n_per ** 2, [(PERIODIC, l_psmooth, period), (EXPQUAD, l_pdecay)]),
        (n_med ** 2, [(RATQUAD, l_med, alpha)]),
        (n_trend ** 2, [(EXPQUAD, l_trend)]),
        (n_noise ** 2, [(MATERN32, l_noise)]),
    ])
