"""Generates tests/golden/composite/*.npz: inputs + expected outputs of the collapsed bound under sum-of-products
covariances, from oracle/composite_oracle.py (PyMC3 MarginalSparse op order, torch fp64,
autograd gradients), cross-checked against the dense N x N definition.  Run from the repo root:
    python tests/golden/make_golden_composite.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import composite_oracle as CO  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "composite")
DT = torch.float64

CASES = {
    # the reference's CO2 covariance on a C2-like time axis (d = 1); short trend lengthscale keeps cond(Kuu) moderate
    "co2_d1": dict(N=400, M=24, d=1, span=12.0, s2=0.04, jitter=1e-6,
                   block=CO.co2_block(n_per=0.8, l_psmooth=1.3, l_pdecay=3.0, n_med=0.5, l_med=1.1, alpha=0.7, n_trend=1.5,
                                      l_trend=2.0, n_noise=0.3, l_noise=0.4, period=1.0)),
    "m52_plus_per_rq_d2": dict(N=500, M=33, d=2, span=6.0, s2=0.05, jitter=1e-6,
                               block=CO.make_block([(1.2, [(CO.MATERN52, 1.5)]), (0.4, [(CO.PERIODIC, 0.9, 2.5), (CO.RATQUAD, 2.0, 1.3)])])),
    "rbf_times_m32_d3": dict(N=700, M=130, d=3, span=6.0, s2=0.05, jitter=0.0,
                             block=CO.make_block([(0.9, [(CO.EXPQUAD, 1.7), (CO.MATERN32, 3.0)])])),
}


def dense_bound(X, y, Z, blk, s2, jitter):
    from scipy.stats import multivariate_normal
    Kuf = CO.composite_k(Z, X, blk).numpy()
    Kuu = CO.composite_k(Z, Z, blk).numpy() + jitter * np.eye(Z.shape[0])
    Qff = Kuf.T @ np.linalg.solve(Kuu, Kuf)
    N = X.shape[0]
    lm = multivariate_normal.logpdf(y.numpy(), mean=np.zeros(N), cov=Qff + s2 * np.eye(N))
    return float(lm - (N * CO.kdiag(blk) - np.trace(Qff)) / (2.0 * s2))


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, c in CASES.items():
        g = torch.Generator().manual_seed(sum(map(ord, name)))
        N, M, d = c["N"], c["M"], c["d"]
        X = torch.rand(N, d, dtype=DT, generator=g) * c["span"]
        y = torch.sin(X.sum(1)) + 0.3 * torch.cos(2.0 * math_pi() * X[:, 0]) + 0.1 * torch.randn(N, dtype=DT, generator=g)
        Z = X[torch.randperm(N, generator=g)[:M]].clone()
        Xs = torch.rand(9, d, dtype=DT, generator=g) * c["span"]
        blk = c["block"]
        F, gr = CO.vfe_composite_and_grads(X, y, Z, blk, c["s2"], c["jitter"])
        Fd = dense_bound(X, y, Z, blk, c["s2"], c["jitter"])
        assert abs(F - Fd) < 1e-8 * max(1.0, abs(Fd)), (name, F, Fd)
        mu, cov = CO.predict_composite(Xs, X, y, Z, blk, c["s2"], c["jitter"])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), X=X.numpy(), y=y.numpy(), Z=Z.numpy(), Xs=Xs.numpy(),
                            block=np.asarray(blk), s2=c["s2"], jitter=c["jitter"], F=F, g_block=gr["block"].numpy(),
                            g_s2=gr["s2"], g_Z=gr["Z"].numpy(), pred_mean=mu.numpy(), pred_cov=cov.numpy())
        print("%-20s N=%d M=%d d=%d F=%.10f (dense %.10f)" % (name, N, M, d, F, Fd))


def math_pi():
    import math
    return math.pi


if __name__ == "__main__":
    main()
