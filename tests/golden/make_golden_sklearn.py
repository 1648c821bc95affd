"""Third-party pin: kernel values and the exact-GP log marginal likelihood from scikit-learn -- run from the repo
root:  python tests/golden/make_golden_sklearn.py

The reference's engines (GPyTorch, PyMC3) are not installed, so nothing reference-held pins the oracle (parity stays
"unpinned" by the rules).  scikit-learn IS installed and the reference itself calls it
(``GaussianProcessRegressor(...).log_marginal_likelihood`` in experiments/lml_surface.py and
experiments/hyperparameter_identification.py), so these fixtures hold numbers produced by a third party:

* kernel matrices of ``RBF`` (ARD), ``Matern(nu=1.5)``, ``Matern(nu=2.5)``, ``RationalQuadratic`` and
  ``ExpSineSquared`` on fixed inputs.  sklearn's ExpSineSquared is exp(-2 sin^2(pi r / p) / l^2); PyMC3's Periodic
  (the convention of the composite kernels here) is exp(-sin^2(pi r / T) / (2 l^2)):  l_sklearn = 2 l_pymc3.
* with Z = X (every training input an inducing input) Q_ff = K_ff, the trace term vanishes and the collapsed bound IS
  the exact GP log marginal likelihood  log N(y | 0, sf2 K + s2 I) = ``GaussianProcessRegressor.log_marginal_likelihood``
  of ``ConstantKernel(sf2) * RBF(ls) + WhiteKernel(s2)`` -- value and gradient (sklearn differentiates with respect to
  the log-parameters).

Only inputs and sklearn's outputs are stored.
"""
import os

import numpy as np
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, ExpSineSquared, Matern, RationalQuadratic, WhiteKernel

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sklearn")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(11)
    # ---- kernel values
    A = rng.standard_normal((9, 3))
    B = rng.standard_normal((7, 3))
    ls = np.array([0.7, 1.3, 2.1])
    a1 = rng.standard_normal((9, 1)) * 2.0
    b1 = rng.standard_normal((7, 1)) * 2.0
    np.savez(os.path.join(OUT, "kernel_values.npz"), A=A, B=B, ls=ls, a1=a1, b1=b1,
             rbf_ard=RBF(length_scale=ls)(A, B), matern32_ard=Matern(length_scale=ls, nu=1.5)(A, B),
             matern52_ard=Matern(length_scale=ls, nu=2.5)(A, B),
             rbf_iso=RBF(length_scale=0.9)(a1, b1), matern32_iso=Matern(length_scale=0.9, nu=1.5)(a1, b1),
             matern52_iso=Matern(length_scale=0.9, nu=2.5)(a1, b1),
             ratquad_ls=1.7, ratquad_alpha=0.6, ratquad=RationalQuadratic(length_scale=1.7, alpha=0.6)(a1, b1),
             periodic_ls_pymc3=0.8, periodic_period=1.3,
             periodic=ExpSineSquared(length_scale=2 * 0.8, periodicity=1.3)(a1, b1))
    # ---- exact-GP limit of the bound
    for name, N, d in (("lml_d1", 40, 1), ("lml_d3", 60, 3)):
        # inputs no closer than about a lengthscale: K_ff itself (no jitter) must be safely positive definite in fp64
        if d == 1:
            X = (np.linspace(-3, 3, N) + rng.uniform(-0.02, 0.02, N))[:, None]
            lsd = np.array([0.15])
        else:
            X = rng.uniform(-3, 3, (N, d))
            lsd = np.array([0.9, 1.4, 2.0])[:d] * 0.5
        w = rng.standard_normal(d)
        y = np.sin(X @ w) + 0.1 * rng.standard_normal(N)
        sf2, s2 = 1.3, 0.05
        k = ConstantKernel(sf2) * RBF(length_scale=lsd) + WhiteKernel(s2)
        gpr = GaussianProcessRegressor(kernel=k, optimizer=None, alpha=0.0).fit(X, y)
        lml, grad = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
        # theta order of the composite: [log sf2, log ls_1..d, log s2]
        np.savez(os.path.join(OUT, name + ".npz"), X=X, y=y, ls=lsd, sf2=sf2, s2=s2, lml=lml,
                 dlml_dlog_sf2=grad[0], dlml_dlog_ls=grad[1:1 + d], dlml_dlog_s2=grad[1 + d])
        print(name, lml)


if __name__ == "__main__":
    main()
