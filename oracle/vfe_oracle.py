"""CPU oracle for the collapsed (Titsias / VFE) sparse-GP bound -- TEST INFRASTRUCTURE ONLY.

This module is a CPU fp64 restatement of the arithmetic the reference delegates to
GPyTorch / PyMC3 on its hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product path
(``generalised-gaussian-processes_amd``) never does and fails loudly without its HIP library.

PARITY UNPINNED.  The reference has no tests, golden vectors or fixtures for this path
(SURVEY.md section 4, section 8c), and its numerical engines (gpytorch, pymc3/theano) are
neither vendored under /root/reference nor installed in the build container, and
``models/bayesian_sgpr_hmc.py`` does not parse (merge markers at :271-281).  The oracle is
therefore pinned only against itself: three independently written forms of the same scalar
(dense N x N definition via scipy, the PyMC3 ``MarginalSparse`` op order, and the streaming
sufficient-statistics form) must agree, and the analytic adjoints must agree with torch
autograd.  ``tests/golden/*.npz`` are produced by ``tests/golden/make_golden.py`` from the
dense definition.

What each function follows (paths relative to /root/reference):

* kernel stack ``ScaleKernel(RBFKernel(ard_num_dims=d))``      models/sgpr.py:36, models/bayesian_sgpr_hmc.py:41
  and its PyMC3 twin ``sig_f**2 * ExpQuad(input_dim, ls=ls)``   models/bayesian_sgpr_hmc.py:65
* ``InducingPointKernel`` + ``ExactMarginalLogLikelihood``       models/sgpr.py:37,114,123-125
  (bound = log N(y|0,Qff+s2 I) - tr(Kff-Qff)/(2 s2); intended split   models/sgpr.py:44-62)
* ``pm.gp.MarginalSparse(approx="VFE").marginal_likelihood``     models/bayesian_sgpr_hmc.py:66,71
  (third-party pymc3 3.9-3.11, unpinned; published algorithm restated in ``vfe_pymc3_order``)
* priors ``Gamma(2,1)`` / ``HalfCauchy(1)`` and log transforms  models/bayesian_sgpr_hmc.py:62-68
* theta -> model mapping (noise=sig_n^2, outputscale=sig_f^2)    models/bayesian_sgpr_hmc.py:82-86
* optimal q(u) / predictive algebra                              models/sgpr.py:256-286
* metrics rmse / nlpd / nlpd_marginal / nlpd_mixture             utils/metrics.py:38-67
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

DT = torch.float64
LOG2PI = math.log(2.0 * math.pi)

KERNEL_RBF = 0
KERNEL_MATERN32 = 1
KERNEL_MATERN52 = 2


def _t(a):
    if isinstance(a, torch.Tensor):
        return a.to(DT)
    return torch.as_tensor(np.asarray(a, dtype=np.float64))


# --------------------------------------------------------------------------------------
# kernels  (models/sgpr.py:36 ; models/bayesian_sgpr_hmc.py:65)
# --------------------------------------------------------------------------------------
def sqdist(X, Z, ls):
    """Scaled squared distance r2[n,m] = sum_j ((x_nj - z_mj)/ls_j)^2 by direct differences."""
    X, Z, ls = _t(X), _t(Z), _t(ls)
    Xs = X / ls
    Zs = Z / ls
    diff = Xs[:, None, :] - Zs[None, :, :]
    return (diff * diff).sum(-1)


def kernel_from_r2(r2, sf2, kernel_id=KERNEL_RBF):
    if kernel_id == KERNEL_RBF:
        return sf2 * torch.exp(-0.5 * r2)
    r = torch.sqrt(r2 + 0.0)
    if kernel_id == KERNEL_MATERN32:
        a = math.sqrt(3.0) * r
        return sf2 * (1.0 + a) * torch.exp(-a)
    if kernel_id == KERNEL_MATERN52:
        a = math.sqrt(5.0) * r
        return sf2 * (1.0 + a + a * a / 3.0) * torch.exp(-a)
    raise ValueError("unknown kernel_id %r" % (kernel_id,))


def kern(X, Z, ls, sf2, kernel_id=KERNEL_RBF, chunk=4096):
    """K[n,m] = k(x_n, z_m).  Chunked over rows of X so memory stays O(chunk*M*d)."""
    X, Z, ls = _t(X), _t(Z), _t(ls)
    if kernel_id != KERNEL_RBF and (
        (isinstance(ls, torch.Tensor) and ls.requires_grad) or Z.requires_grad
    ):
        # sqrt at r=0 has no finite derivative under autograd; callers use the analytic path.
        pass
    out = []
    for s in range(0, X.shape[0], chunk):
        out.append(kernel_from_r2(sqdist(X[s:s + chunk], Z, ls), sf2, kernel_id))
    return torch.cat(out, 0) if out else torch.zeros(0, Z.shape[0], dtype=DT)


# --------------------------------------------------------------------------------------
# form 1: dense N x N definition (independent check; N <= ~3000)
# --------------------------------------------------------------------------------------
def vfe_dense(X, y, Z, ls, sf2, s2, jitter=0.0, kernel_id=KERNEL_RBF):
    """F = log N(y | 0, Qff + s2 I) - tr(Kff - Qff)/(2 s2), through scipy's dense MVN logpdf.

    Returns (F, logmarg, trace_term) so either sign convention of the reference's
    InducingPointKernelAddedLossTerm can be reproduced.
    """
    from scipy.stats import multivariate_normal
    import scipy.linalg as sla

    X_, y_, Z_, ls_ = (np.asarray(_t(a)) for a in (X, y, Z, ls))
    Kuf = kern(Z_, X_, ls_, sf2, kernel_id).numpy()
    Kuu = kern(Z_, Z_, ls_, sf2, kernel_id).numpy() + jitter * np.eye(Z_.shape[0])
    Qff = Kuf.T @ sla.solve(Kuu, Kuf, assume_a="pos")
    N = X_.shape[0]
    logmarg = multivariate_normal.logpdf(y_, mean=np.zeros(N), cov=Qff + s2 * np.eye(N), allow_singular=False)
    trace_term = (N * sf2 - np.trace(Qff)) / (2.0 * s2)
    return float(logmarg - trace_term), float(logmarg), float(trace_term)


# --------------------------------------------------------------------------------------
# form 2: PyMC3 MarginalSparse(VFE) op order  (models/bayesian_sgpr_hmc.py:66,71)
#         This is the timed CPU baseline: A = Luu^-1 Kuf materialised.
# --------------------------------------------------------------------------------------
def vfe_pymc3_order(X, y, Z, ls, sf, sn, jitter=1e-6, kernel_id=KERNEL_RBF):
    """logp of ``MarginalSparse(approx='VFE').marginal_likelihood`` -- torch fp64, differentiable.

    ls, sf, sn are *standard deviations / lengthscales* as PyMC3 samples them
    (cov = sf**2 * ExpQuad(ls), noise = sn).  stabilize(Kuu) = Kuu + 1e-6 I.
    """
    X, y, Z, ls = _t(X), _t(y), _t(Z), _t(ls)
    sf = _t(sf)
    sn = _t(sn)
    sf2 = sf * sf
    s2 = sn * sn
    M = Z.shape[0]
    N = X.shape[0]
    Kuu = kernel_from_r2(sqdist(Z, Z, ls), sf2, kernel_id) + jitter * torch.eye(M, dtype=DT)
    Kuf = kernel_from_r2(sqdist(Z, X, ls), sf2, kernel_id)            # M x N materialised
    Luu = torch.linalg.cholesky(Kuu)
    A = torch.linalg.solve_triangular(Luu, Kuf, upper=False)           # M x N
    Lamd = s2 * torch.ones(N, dtype=DT)
    trace = (N * sf2 - (A * A).sum()) / (2.0 * s2)
    A_l = A / Lamd
    L_B = torch.linalg.cholesky(torch.eye(M, dtype=DT) + A_l @ A.T)
    r_l = y / Lamd
    c = torch.linalg.solve_triangular(L_B, (A @ r_l)[:, None], upper=False)[:, 0]
    constant = 0.5 * N * LOG2PI
    logdet = 0.5 * torch.log(Lamd).sum() + torch.log(torch.diagonal(L_B)).sum()
    quadratic = 0.5 * (y @ r_l - c @ c)
    return -(constant + logdet + quadratic + trace)


def vfe_pymc3_order_chunked(X, y, Z, ls, sf, sn, jitter=1e-6, kernel_id=KERNEL_RBF, chunk=65536):
    """Same op order, value only, A processed in column chunks so N=1M fits in host memory.

    Used by bench.py's cpu_baseline at the full C5 size.
    """
    X, y, Z, ls = _t(X), _t(y), _t(Z), _t(ls)
    sf2 = float(sf) ** 2
    s2 = float(sn) ** 2
    M, N = Z.shape[0], X.shape[0]
    with torch.no_grad():
        Kuu = kernel_from_r2(sqdist(Z, Z, ls), sf2, kernel_id) + jitter * torch.eye(M, dtype=DT)
        Luu = torch.linalg.cholesky(Kuu)
        AAt = torch.zeros(M, M, dtype=DT)
        Ay = torch.zeros(M, dtype=DT)
        for s in range(0, N, chunk):
            Kuf = kernel_from_r2(sqdist(Z, X[s:s + chunk], ls), sf2, kernel_id)
            A = torch.linalg.solve_triangular(Luu, Kuf, upper=False)
            AAt += A @ A.T
            Ay += A @ y[s:s + chunk]
        trace = (N * sf2 - torch.diagonal(AAt).sum()) / (2.0 * s2)
        L_B = torch.linalg.cholesky(torch.eye(M, dtype=DT) + AAt / s2)
        c = torch.linalg.solve_triangular(L_B, (Ay / s2)[:, None], upper=False)[:, 0]
        logdet = 0.5 * N * math.log(s2) + torch.log(torch.diagonal(L_B)).sum()
        quadratic = 0.5 * (y @ y / s2 - c @ c)
        return float(-(0.5 * N * LOG2PI + logdet + quadratic + trace))


# --------------------------------------------------------------------------------------
# form 3: streaming sufficient statistics (what the HIP path computes)
# --------------------------------------------------------------------------------------
@dataclass
class SuffStats:
    Phi: torch.Tensor   # M x M   Kuf Kuf^T
    b: torch.Tensor     # M       Kuf y
    yy: float           # y^T y
    kappa: float        # sum_n k(x_n, x_n)
    N: int


def suffstats(X, y, Z, ls, sf2, kernel_id=KERNEL_RBF, chunk=8192):
    X, y, Z, ls = _t(X), _t(y), _t(Z), _t(ls)
    M = Z.shape[0]
    Phi = torch.zeros(M, M, dtype=DT)
    b = torch.zeros(M, dtype=DT)
    for s in range(0, X.shape[0], chunk):
        K = kernel_from_r2(sqdist(Z, X[s:s + chunk], ls), sf2, kernel_id)   # M x c
        Phi = Phi + K @ K.T
        b = b + K @ y[s:s + chunk]
    return SuffStats(Phi, b, float(y @ y), float(sf2) * X.shape[0], int(X.shape[0]))


def suffstats_whitened(X, y, Z, ls, sf2, L, kernel_id=KERNEL_RBF, chunk=8192):
    """W = A A^T, u = A y with A = L^-1 K_uf (App. A.2 op order), accumulated over row chunks."""
    X, y, Z, ls, L = _t(X), _t(y), _t(Z), _t(ls), _t(L)
    M = Z.shape[0]
    N = X.shape[0]
    W = torch.zeros(M, M, dtype=DT)
    u = torch.zeros(M, dtype=DT)
    for s in range(0, N, chunk):
        A = torch.linalg.solve_triangular(L, kern(Z, X[s:s + chunk], ls, sf2, kernel_id), upper=False)
        W += A @ A.T
        u += A @ y[s:s + chunk]
    return SuffStats(W, u, float(y @ y), float(N * sf2), N)


def kuu(Z, ls, sf2, jitter, kernel_id=KERNEL_RBF):
    Z = _t(Z)
    return kernel_from_r2(sqdist(Z, Z, _t(ls)), sf2, kernel_id) + jitter * torch.eye(Z.shape[0], dtype=DT)


def bound_from_stats(Kuu, st: SuffStats, s2, with_adjoints=False, whitened=True, stats_whitened=False):
    """O(M^3) tail.  Returns dict with F, logmarg, trace_term and (optionally) the adjoints of
    F wrt Phi, b, Kuu, s2, kappa.

    ``whitened=True`` (default) evaluates the adjoints between L^-T ... L^-1 from B, B^-1 and g = B^-1 L^-1 b:
        2 s2 Phibar = L^-T (I - B^-1 - g g^T / s2^2) L^-1 ;  -2 Kuubar = L^-T (B + B^-1 - 2 I + g g^T / s2^2) L^-1
    ``stats_whitened=True``: ``st.Phi`` / ``st.b`` already are W = A A^T and u = A y with A = L^-1 K_uf (the PyMC3
    op order of ``vfe_pymc3_order``, App. A.2); the adjoints returned are still those of the unwhitened Phi, b, Kuu.
    ``whitened=False`` is App. A.5 as written (Kuu^-1 - Sigma^-1 - alpha alpha^T ...): identical in exact arithmetic,
    but it cancels O(cond Kuu) entries -- relative gradient errors of 1e-2 at cond 1e8 against 1e-9 for the whitened
    form (tests/studies/logp_noise.py).  Kept for that A/B only."""
    Kuu = _t(Kuu)
    Phi, b = _t(st.Phi), _t(st.b)
    M = Kuu.shape[0]
    N = st.N
    I = torch.eye(M, dtype=DT)
    L = torch.linalg.cholesky(Kuu)
    if stats_whitened:
        W, u = Phi, b
    else:
        V = torch.linalg.solve_triangular(L, Phi, upper=False)                  # L^-1 Phi
        W = torch.linalg.solve_triangular(L, V.T, upper=False).T                # L^-1 Phi L^-T
        u = torch.linalg.solve_triangular(L, b[:, None], upper=False)[:, 0]     # L^-1 b
    W = 0.5 * (W + W.T)
    B = I + W / s2
    LB = torch.linalg.cholesky(B)
    logdetB = 2.0 * torch.log(torch.diagonal(LB)).sum()
    trW = torch.diagonal(W).sum()
    q = torch.linalg.solve_triangular(LB, u[:, None], upper=False)[:, 0]    # LB^-1 L^-1 b  (= s2 * c of App. A.3)
    quad = st.yy / s2 - (q @ q) / (s2 * s2)
    logmarg = -(0.5 * N * LOG2PI + 0.5 * N * math.log(s2) + 0.5 * logdetB + 0.5 * quad)
    trace_term = (st.kappa - trW) / (2.0 * s2)
    F = logmarg - trace_term
    out = {"F": float(F), "logmarg": float(logmarg), "trace_term": float(trace_term),
           "L": L, "LB": LB, "q": q}
    if with_adjoints and whitened:
        Binv = torch.cholesky_inverse(LB)
        g = Binv @ u                                                         # B^-1 L^-1 b
        gg = torch.outer(g, g) / (s2 * s2)

        def sandwich(X):                                                     # L^-T X L^-1 by triangular solves
            Y = torch.linalg.solve_triangular(L.T, X, upper=True)
            return torch.linalg.solve_triangular(L.T, Y.T, upper=True).T

        Phibar = sandwich(I - Binv - gg) / (2.0 * s2)
        Kuubar = -0.5 * sandwich(B + Binv - 2.0 * I + gg)
        Phibar, Kuubar = 0.5 * (Phibar + Phibar.T), 0.5 * (Kuubar + Kuubar.T)
        alpha = torch.linalg.solve_triangular(L.T, g[:, None], upper=True)[:, 0]   # Sigma^-1 b
        bbar = alpha / (s2 * s2)
        kappabar = -1.0 / (2.0 * s2)
        s2bar = -0.5 * (-(Binv * W).sum() / s2 ** 2 + N / s2 - st.yy / s2 ** 2
                        + 2.0 * (u @ g) / s2 ** 3 - (g @ W @ g) / s2 ** 4
                        - st.kappa / s2 ** 2 + trW / s2 ** 2)
        out.update(Phibar=Phibar, bbar=bbar, Kuubar=Kuubar, s2bar=float(s2bar), kappabar=kappabar, alpha=alpha)
    elif with_adjoints:
        Linv = torch.linalg.solve_triangular(L, I, upper=False)
        Kinv = Linv.T @ Linv
        G = torch.linalg.solve_triangular(LB, Linv, upper=False)            # LB^-1 L^-1
        Sinv = G.T @ G                                                       # (Kuu + Phi/s2)^-1
        alpha = Sinv @ b
        aa = torch.outer(alpha, alpha) / (s2 * s2)
        Phibar = (Kinv - Sinv - aa) / (2.0 * s2)
        bbar = alpha / (s2 * s2)
        kappabar = -1.0 / (2.0 * s2)
        KPK = Kinv @ Phi @ Kinv
        Kuubar = -0.5 * (Sinv + aa - Kinv + KPK / s2)
        s2bar = -0.5 * (-(Sinv * Phi).sum() / s2 ** 2 + N / s2 - st.yy / s2 ** 2
                        + 2.0 * (b @ alpha) / s2 ** 3 - (alpha @ Phi @ alpha) / s2 ** 4
                        - st.kappa / s2 ** 2 + (Kinv * Phi).sum() / s2 ** 2)
        out.update(Phibar=Phibar, bbar=bbar, Kuubar=Kuubar, s2bar=float(s2bar), kappabar=kappabar,
                   alpha=alpha, Kinv=Kinv, Sinv=Sinv)
    return out


def vfe_streaming(X, y, Z, ls, sf2, s2, jitter=0.0, kernel_id=KERNEL_RBF):
    st = suffstats(X, y, Z, ls, sf2, kernel_id)
    return bound_from_stats(kuu(Z, ls, sf2, jitter, kernel_id), st, s2)


# --------------------------------------------------------------------------------------
# analytic gradients -- RBF-ARD and Matern, no autograd
# --------------------------------------------------------------------------------------
def _dk_factors(r2, K, sf2, kernel_id):
    """Return H with dK/d(r2) = H (elementwise), so that dK/dl_j = H * d(r2)/dl_j."""
    if kernel_id == KERNEL_RBF:
        return -0.5 * K
    r = torch.sqrt(r2)
    if kernel_id == KERNEL_MATERN32:
        # k = sf2 (1+a) e^-a, a = sqrt3 r ; dk/dr2 = -1.5 sf2 e^-a
        return -1.5 * sf2 * torch.exp(-math.sqrt(3.0) * r)
    if kernel_id == KERNEL_MATERN52:
        a = math.sqrt(5.0) * r
        return -(5.0 / 6.0) * sf2 * (1.0 + a) * torch.exp(-a)
    raise ValueError(kernel_id)


def grads_analytic(X, y, Z, ls, sf2, s2, jitter=0.0, kernel_id=KERNEL_RBF, chunk=4096, whitened=True):
    """dF/d(ls_j), dF/d(sf2), dF/d(s2), dF/dZ by the two-pass scheme the HIP path uses."""
    X, y, Z, ls = _t(X), _t(y), _t(Z), _t(ls)
    M, d = Z.shape
    st = suffstats(X, y, Z, ls, sf2, kernel_id)
    Kuu_ = kuu(Z, ls, sf2, jitter, kernel_id)
    res = bound_from_stats(Kuu_, st, s2, with_adjoints=True, whitened=whitened)
    Phibar, bbar, Kuubar = res["Phibar"], res["bbar"], res["Kuubar"]
    g_ls = torch.zeros(d, dtype=DT)
    g_Z = torch.zeros(M, d, dtype=DT)
    g_sf2 = torch.zeros((), dtype=DT)
    Zs = Z / ls
    # pass 2 over the rows: Kbar_uf = 2 Phibar Kuf + bbar y^T, contracted with dK
    for s in range(0, X.shape[0], chunk):
        Xc = X[s:s + chunk]
        r2 = sqdist(Z, Xc, ls)                                              # M x c
        K = kernel_from_r2(r2, sf2, kernel_id)
        Kbar = 2.0 * Phibar @ K + torch.outer(bbar, y[s:s + chunk])
        g_sf2 = g_sf2 + (Kbar * K).sum() / sf2
        E = Kbar * _dk_factors(r2, K, sf2, kernel_id)                       # dF/d r2
        diff = Zs[:, None, :] - (Xc / ls)[None, :, :]                       # M x c x d  (z~ - x~)
        # d r2 / d ls_j = -2 diff_j^2 / ls_j ;  d r2 / d z_mj = 2 diff_j / ls_j
        g_ls = g_ls + (-2.0 * (E[:, :, None] * diff * diff).sum((0, 1)) / ls)
        g_Z = g_Z + 2.0 * (E[:, :, None] * diff).sum(1) / ls
    # Kuu part (jitter is constant)
    r2u = sqdist(Z, Z, ls)
    Ku = kernel_from_r2(r2u, sf2, kernel_id)
    g_sf2 = g_sf2 + (Kuubar * Ku).sum() / sf2
    Eu = Kuubar * _dk_factors(r2u, Ku, sf2, kernel_id)
    diffu = Zs[:, None, :] - Zs[None, :, :]
    g_ls = g_ls + (-2.0 * (Eu[:, :, None] * diffu * diffu).sum((0, 1)) / ls)
    g_Z = g_Z + 2.0 * ((Eu + Eu.T)[:, :, None] * diffu).sum(1) / ls
    # kappa = N sf2
    g_sf2 = g_sf2 + res["kappabar"] * st.N
    return {"F": res["F"], "g_ls": g_ls, "g_sf2": float(g_sf2), "g_s2": res["s2bar"], "g_Z": g_Z,
            "Phibar": Phibar, "bbar": bbar, "Kuubar": Kuubar}


def grads_autograd(X, y, Z, ls, sf2, s2, jitter=0.0):
    """Same gradients through torch autograd on the PyMC3-order graph (RBF only)."""
    X, y = _t(X), _t(y)
    Zt = _t(Z).clone().requires_grad_(True)
    lst = _t(ls).clone().requires_grad_(True)
    sf = torch.tensor(math.sqrt(sf2), dtype=DT, requires_grad=True)
    sn = torch.tensor(math.sqrt(s2), dtype=DT, requires_grad=True)
    F = vfe_pymc3_order(X, y, Zt, lst, sf, sn, jitter=jitter)
    F.backward()
    return {"F": float(F.detach()), "g_ls": lst.grad, "g_sf": float(sf.grad), "g_sn": float(sn.grad), "g_Z": Zt.grad,
            "g_sf2": float(sf.grad) / (2.0 * math.sqrt(sf2)), "g_s2": float(sn.grad) / (2.0 * math.sqrt(s2))}


# --------------------------------------------------------------------------------------
# HMC target: logp + priors + log-Jacobians  (models/bayesian_sgpr_hmc.py:60-71)
# --------------------------------------------------------------------------------------
def hmc_logp(theta_unc, X, y, Z, jitter=1e-6, with_grad=True):
    """theta_unc = [log ls_1..d, log sig_f, log sig_n] (PyMC3 log transform of positive RVs).

    ls ~ Gamma(alpha=2, beta=1) ; sig_f, sig_n ~ HalfCauchy(beta=1) ; + VFE logp.
    Returns (logp, grad[d+2]).
    """
    th = _t(theta_unc).clone().requires_grad_(with_grad)
    d = th.shape[0] - 2
    ls = torch.exp(th[:d])
    sf = torch.exp(th[d])
    sn = torch.exp(th[d + 1])
    lp = vfe_pymc3_order(X, y, Z, ls, sf, sn, jitter=jitter)
    lp = lp + (torch.log(ls) - ls).sum()                                    # Gamma(2,1): -x + log x (lgamma(2)=0)
    for s in (sf, sn):
        lp = lp + (math.log(2.0) - math.log(math.pi) - torch.log1p(s * s))  # HalfCauchy(1)
    lp = lp + th.sum()                                                      # log|d theta / d theta_unc|
    if not with_grad:
        return float(lp), None
    lp.backward()
    return float(lp.detach()), th.grad.detach().clone()


# --------------------------------------------------------------------------------------
# predictive  (models/sgpr.py:150-160,256-286 ; models/bayesian_sgpr_hmc.py:198-231)
# --------------------------------------------------------------------------------------
def predict(Xs, X, y, Z, ls, sf2, s2, jitter=0.0, kernel_id=KERNEL_RBF, full_cov=False, pred_noise=True):
    """mu* = K*u Sigma^-1 b / s2 ; cov* = K** - K*u Kuu^-1 Ku* + K*u Sigma^-1 Ku* (+ s2 I)."""
    Xs, X, y, Z, ls = _t(Xs), _t(X), _t(y), _t(Z), _t(ls)
    st = suffstats(X, y, Z, ls, sf2, kernel_id)
    res = bound_from_stats(kuu(Z, ls, sf2, jitter, kernel_id), st, s2)
    L, LB, q = res["L"], res["LB"], res["q"]
    Kus = kernel_from_r2(sqdist(Z, Xs, ls), sf2, kernel_id)                 # M x T
    As = torch.linalg.solve_triangular(L, Kus, upper=False)
    C = torch.linalg.solve_triangular(LB, As, upper=False)
    mu = C.T @ q / s2
    if full_cov:
        Kss = kernel_from_r2(sqdist(Xs, Xs, ls), sf2, kernel_id)
        cov = Kss - As.T @ As + C.T @ C
        if pred_noise:
            cov = cov + s2 * torch.eye(Xs.shape[0], dtype=DT)
        return mu, cov
    var = sf2 - (As * As).sum(0) + (C * C).sum(0)
    if pred_noise:
        var = var + s2
    return mu, var


# --------------------------------------------------------------------------------------
# metrics  (utils/metrics.py:38-67)
# --------------------------------------------------------------------------------------
def rmse(pred_mean, y_test, y_std=1.0):
    pred_mean, y_test = _t(pred_mean), _t(y_test)
    return float(y_std * torch.sqrt(torch.mean((pred_mean - y_test) ** 2)))


def nlpd_joint(mu, cov, y_test, y_std=1.0):
    """utils/metrics.py:42-47 -- joint MVN log-prob / n_test - log(y_std), negated."""
    mu, cov, y_test = _t(mu), _t(cov), _t(y_test)
    Lc = torch.linalg.cholesky(cov)
    r = torch.linalg.solve_triangular(Lc, (y_test - mu)[:, None], upper=False)[:, 0]
    T = y_test.shape[0]
    lpd = -0.5 * (r @ r) - torch.log(torch.diagonal(Lc)).sum() - 0.5 * T * LOG2PI
    return float(-(lpd / T - math.log(y_std)))


def nlpd_marginal(mu, var, y_test, y_std=1.0):
    """utils/metrics.py:49-58."""
    mu, var, y_test = _t(mu), _t(var), _t(y_test)
    lp = -0.5 * LOG2PI - 0.5 * torch.log(var) - 0.5 * (y_test - mu) ** 2 / var - math.log(y_std)
    return float(-lp.mean())
