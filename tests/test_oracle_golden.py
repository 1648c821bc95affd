"""CPU: the oracle's independent forms against the committed golden vectors (which come from the dense
N x N definition), plus internal consistency of the analytic adjoints."""
import math

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from oracle import vfe_oracle as O


def T(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


@pytest.mark.parametrize("name", golden_names())
def test_three_forms_agree_with_golden(name):
    G = load_golden(name)
    X, y, Z, ls = T(G["X"]), T(G["y"]), T(G["Z"]), T(G["ls"])
    sf2, s2, J, kid = float(G["sf2"]), float(G["s2"]), float(G["jitter"]), int(G["kernel_id"])
    tol = 1e-9 * max(1.0, abs(float(G["F"])))
    Fp = float(O.vfe_pymc3_order(X, y, Z, ls, math.sqrt(sf2), math.sqrt(s2), J, kid))
    r = O.vfe_streaming(X, y, Z, ls, sf2, s2, J, kid)
    Fc = O.vfe_pymc3_order_chunked(X, y, Z, ls, math.sqrt(sf2), math.sqrt(s2), J, kid, chunk=97)
    assert abs(Fp - float(G["F"])) < tol
    assert abs(r["F"] - float(G["F"])) < tol
    assert abs(Fc - float(G["F"])) < tol
    ptol = tol * (1000.0 if float(G["grad_rtol"]) > 1e-6 else 1.0)
    assert abs(r["logmarg"] - float(G["logmarg"])) < ptol and abs(r["trace_term"] - float(G["trace_term"])) < ptol
    if N_small(G):
        Fd, _, _ = O.vfe_dense(X, y, Z, ls, sf2, s2, J, kid)
        assert abs(Fd - float(G["F"])) < tol


def N_small(G):
    return G["X"].shape[0] <= 1000


@pytest.mark.parametrize("name", golden_names())
def test_analytic_gradients_match_golden(name):
    G = load_golden(name)
    g = O.grads_analytic(T(G["X"]), T(G["y"]), T(G["Z"]), T(G["ls"]), float(G["sf2"]), float(G["s2"]), float(G["jitter"]),
                         int(G["kernel_id"]))
    rt, rz = float(G["grad_rtol"]), float(G["gz_rtol"])
    assert float((g["g_ls"] - T(G["g_ls"])).abs().max()) < rt * max(1.0, float(T(G["g_ls"]).abs().max()))
    assert abs(g["g_sf2"] - float(G["g_sf2"])) < rt * max(1.0, abs(float(G["g_sf2"])))
    assert abs(g["g_s2"] - float(G["g_s2"])) < rt * max(1.0, abs(float(G["g_s2"])))
    assert float((g["g_Z"] - T(G["g_Z"])).abs().max()) < rz * max(1.0, float(T(G["g_Z"]).abs().max()))


@pytest.mark.parametrize("name", golden_names())
def test_predict_matches_golden(name):
    G = load_golden(name)
    mu, cov = O.predict(T(G["Xs"]), T(G["X"]), T(G["y"]), T(G["Z"]), T(G["ls"]), float(G["sf2"]), float(G["s2"]),
                        float(G["jitter"]), int(G["kernel_id"]), full_cov=True)
    assert float((mu - T(G["pred_mean"])).abs().max()) < 1e-8
    assert float((cov - T(G["pred_cov"])).abs().max()) < 1e-8
    mu2, var = O.predict(T(G["Xs"]), T(G["X"]), T(G["y"]), T(G["Z"]), T(G["ls"]), float(G["sf2"]), float(G["s2"]),
                         float(G["jitter"]), int(G["kernel_id"]))
    assert float((var - T(G["pred_var"])).abs().max()) < 1e-8


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("rbf")])
def test_hmc_logp_matches_golden(name):
    G = load_golden(name)
    for th, lp_ref, g_ref in zip(G["hmc_theta"], G["hmc_logp"], G["hmc_grad"]):
        lp, gr = O.hmc_logp(T(th), T(G["X"]), T(G["y"]), T(G["Z"]))
        assert abs(lp - lp_ref) < 1e-10 * max(1.0, abs(lp_ref))
        assert float((gr - T(g_ref)).abs().max()) < 1e-8 * max(1.0, float(np.abs(g_ref).max()))


def test_metrics_follow_reference_definitions():
    # utils/metrics.py:38-58 -- rmse scales by Y_std; nlpd is the joint log-prob / n - log Y_std
    mu = torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64)
    yt = torch.tensor([0.5, 1.0, 1.0], dtype=torch.float64)
    assert abs(O.rmse(mu, yt, 2.0) - 2.0 * math.sqrt((0.25 + 0 + 1.0) / 3)) < 1e-14
    cov = torch.diag(torch.tensor([1.0, 4.0, 0.25], dtype=torch.float64))
    joint = O.nlpd_joint(mu, cov, yt, 1.0)
    marg = O.nlpd_marginal(mu, torch.diagonal(cov), yt, 1.0)
    assert abs(joint - marg) < 1e-12  # diagonal covariance: joint / n == mean of marginals
