"""CPU oracle for the uncollapsed (SVGP) minibatch bound -- TEST INFRASTRUCTURE ONLY (same rules as vfe_oracle.py).

PARITY UNPINNED: the reference delegates this to GPyTorch's ``VariationalELBO`` over a whitened
``VariationalStrategy`` with a ``CholeskyVariationalDistribution`` (reference models/svgp.py:37,46,88-127;
models/bayesian_svgp.py:144-181), GPyTorch is not vendored / installed, and the reference holds no fixtures.
This is a restatement of the published whitened SVGP bound (Hensman et al. 2013/2015) in torch fp64; autograd
supplies every gradient the HIP path must reproduce.

    L = chol(Kuu + J I) ; A = L^-1 K_ub ; mu = A^T m ; v = k_bb - sum A o A + sum (L_S^T A) o (L_S^T A)
    ELBO / datum = mean_b E_{N(mu_b, v_b)}[log p(y_b | f)] - KL(N(m, L_S L_S^T) || N(0, I)) / N
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import vfe_oracle as O

DT = torch.float64
LIK_GAUSSIAN = 0
LIK_BERNOULLI_PROBIT = 1
GH_POINTS = 20


def gauss_hermite(n=GH_POINTS):
    """Nodes / weights for int f(x) N(x; 0, 1) dx ~= sum w_i f(x_i)."""
    x, w = np.polynomial.hermite.hermgauss(n)
    return torch.as_tensor(x * math.sqrt(2.0), dtype=DT), torch.as_tensor(w / math.sqrt(math.pi), dtype=DT)


def log_ndtr(z):
    return torch.special.log_ndtr(z)


def expected_log_lik(y, mu, v, s2, likelihood):
    if likelihood == LIK_GAUSSIAN:
        return -0.5 * math.log(2.0 * math.pi) - 0.5 * torch.log(s2) - ((y - mu) ** 2 + v) / (2.0 * s2)
    x, w = gauss_hermite()
    f = mu[:, None] + torch.sqrt(v)[:, None] * x[None, :]
    return (log_ndtr(y[:, None] * f) * w[None, :]).sum(1)      # y in {-1, +1}


def svgp_terms(Xb, yb, Z, ls, sf2, s2, m, LS, jitter=1e-6, kernel_id=0, likelihood=LIK_GAUSSIAN):
    Xb, yb, Z, ls, m, LS = (O._t(a) for a in (Xb, yb, Z, ls, m, LS))
    sf2 = O._t(sf2)
    s2 = O._t(s2)
    M = Z.shape[0]
    Kuu = O.kernel_from_r2(O.sqdist(Z, Z, ls), sf2, kernel_id) + jitter * torch.eye(M, dtype=DT)
    Kub = O.kernel_from_r2(O.sqdist(Z, Xb, ls), sf2, kernel_id)
    L = torch.linalg.cholesky(Kuu)
    A = torch.linalg.solve_triangular(L, Kub, upper=False)
    LSl = torch.tril(LS)
    T = LSl.T @ A
    mu = A.T @ m
    v = sf2 - (A * A).sum(0) + (T * T).sum(0)
    ell = expected_log_lik(yb, mu, v, s2, likelihood)
    kl = 0.5 * ((m * m).sum() + (LSl * LSl).sum() - M - 2.0 * torch.log(torch.diagonal(LSl)).sum())
    return ell, kl, mu, v


def svgp_elbo(Xb, yb, Z, ls, sf2, s2, m, LS, N_total, jitter=1e-6, kernel_id=0, likelihood=LIK_GAUSSIAN):
    """GPyTorch VariationalELBO convention: mean over the minibatch of the expected log-lik minus KL / num_data."""
    ell, kl, _, _ = svgp_terms(Xb, yb, Z, ls, sf2, s2, m, LS, jitter, kernel_id, likelihood)
    return ell.mean() - kl / N_total


def svgp_elbo_and_grads(Xb, yb, Z, ls, sf2, s2, m, LS, N_total, jitter=1e-6, kernel_id=0, likelihood=LIK_GAUSSIAN):
    with torch.enable_grad():  # may be called from inside an autograd.Function.forward (grad mode off there)
        return _svgp_elbo_and_grads(Xb, yb, Z, ls, sf2, s2, m, LS, N_total, jitter, kernel_id, likelihood)


def _svgp_elbo_and_grads(Xb, yb, Z, ls, sf2, s2, m, LS, N_total, jitter, kernel_id, likelihood):
    Zt = O._t(Z).clone().requires_grad_(True)
    lst = O._t(ls).clone().requires_grad_(True)
    sf2t = torch.tensor(float(sf2), dtype=DT, requires_grad=True)
    s2t = torch.tensor(float(s2), dtype=DT, requires_grad=True)
    mt = O._t(m).clone().requires_grad_(True)
    LSt = torch.tril(O._t(LS)).clone().requires_grad_(True)
    e = svgp_elbo(Xb, yb, Zt, lst, sf2t, s2t, mt, LSt, N_total, jitter, kernel_id, likelihood)
    e.backward()
    g_s2 = float(s2t.grad) if s2t.grad is not None else 0.0
    return {"elbo": float(e.detach()), "g_Z": Zt.grad, "g_ls": lst.grad, "g_sf2": float(sf2t.grad), "g_s2": g_s2,
            "g_m": mt.grad, "g_LS": torch.tril(LSt.grad)}


def svgp_predict(Xs, Z, ls, sf2, m, LS, jitter=1e-6, kernel_id=0):
    """Latent predictive mean / variance at Xs (models/svgp.py:132-141 goes on through the likelihood)."""
    _, _, mu, v = svgp_terms(Xs, torch.zeros(O._t(Xs).shape[0], dtype=DT), Z, ls, sf2, 1.0, m, LS, jitter, kernel_id, LIK_GAUSSIAN)
    return mu, v
