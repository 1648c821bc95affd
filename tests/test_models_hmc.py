"""CPU: model classes, marginal-log-likelihood autograd bridge, NUTS driver and trace surface, through the
oracle-backed test double.  The loss trace of ``SparseGPR.train_model`` is pinned against an independent
torch-autograd loop on the PyMC3-op-order graph (counterparts of the Python callers)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ggp_amd
from fake_engine import OracleEngine
from oracle import vfe_oracle as O

DT = torch.float64


def demo_1d(seed=45, n=1000):
    """experiments/demo_1d_regression.py:55-72 (seeded torch RNG, train on |x| > 2)."""
    torch.manual_seed(seed)
    X = torch.randn(n) * 2 - 1
    Y = torch.sin(X * 3) + 0.3 * torch.cos(X * 3.14) + 0.4 * torch.randn(n)
    idx = (X < -2) | (X > 2)
    return X[idx][:, None].to(DT), Y[idx].to(DT), torch.linspace(-8, 8, 200, dtype=DT), torch.randn(25).to(DT)


def reference_loss_trace(X, y, Z0, steps, lr, jitter=0.0):
    """-F/N under Adam(lr) on GPyTorch's raw parametrisation, all in torch autograd on the oracle graph."""
    d = X.shape[1]
    raw_ls = torch.zeros(1, d, dtype=DT, requires_grad=True)
    raw_os = torch.zeros((), dtype=DT, requires_grad=True)
    raw_n = torch.zeros(1, dtype=DT, requires_grad=True)
    Z = Z0.clone().reshape(-1, d).requires_grad_(True)
    # parameter order must match model.parameters(): likelihood first, then kernel, then Z
    opt = torch.optim.Adam([raw_n, raw_ls, raw_os, Z], lr=lr)
    out = []
    for _ in range(steps):
        opt.zero_grad()
        ls = F.softplus(raw_ls).reshape(-1)
        sf = torch.sqrt(F.softplus(raw_os))
        sn = torch.sqrt(F.softplus(raw_n) + 1e-4)[0]
        loss = -O.vfe_pymc3_order(X, y, Z, ls, sf, sn, jitter=jitter) / X.shape[0]
        out.append(float(loss.detach()))
        loss.backward()
        opt.step()
    return out


def test_sparse_gpr_loss_trace_matches_autograd_reference():
    X, y, Xt, Z0 = demo_1d()
    lik = ggp_amd.GaussianLikelihood()
    model = ggp_amd.SparseGPR(X, y, lik, Z0, engine=OracleEngine(), jitter=1e-6)
    names = [n for n, _ in model.named_parameters()]
    assert names == ['likelihood.noise_covar._n.raw', 'base_covar_module._os.raw', 'base_covar_module.base_kernel._ls.raw',
                     'covar_module.inducing_points'] or len(names) == 4
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    losses = model.train_model(opt, max_steps=12, verbose=False)
    ref = reference_loss_trace(X, y, Z0, 12, 0.01, jitter=1e-6)
    assert len(losses) == 12
    # 25 random 1-D inducing points are nearly collinear (cond(Kuu) ~ 1e6): the two gradient paths agree to ~1e-7
    # and Adam feeds that back into the trace; the first steps agree to 1e-12
    assert np.max(np.abs(np.array(losses[:2]) - np.array(ref[:2]))) < 1e-10
    assert np.max(np.abs(np.array(losses) - np.array(ref))) < 1e-6, (losses[:3], ref[:3])
    assert losses[-1] < losses[0]
    # GPyTorch defaults at raw = 0
    m2 = ggp_amd.SparseGPR(X, y, ggp_amd.GaussianLikelihood(), Z0, engine=OracleEngine())
    assert abs(float(m2.base_covar_module.outputscale.detach()) - math.log(2.0)) < 1e-15
    assert abs(float(m2.likelihood.noise.detach()) - (math.log(2.0) + 1e-4)) < 1e-15
    assert m2.base_covar_module.base_kernel.lengthscale.shape == (1, 1)
    assert m2.num_inducing == 25 and m2.inducing_points.shape == (25, 1)


def test_num_steps_alias_and_inducing_points_track_optimised_Z():
    X, y, Xt, Z0 = demo_1d()
    model = ggp_amd.SparseGPR(X, y, ggp_amd.GaussianLikelihood(), Z0, engine=OracleEngine(), jitter=1e-6)
    opt = torch.optim.Adam(model.parameters(), lr=0.05)
    losses = model.train_model(opt, num_steps=3, verbose=False)
    assert len(losses) == 3
    assert not torch.allclose(model.inducing_points.reshape(-1), Z0)      # moved
    assert model.inducing_points.data_ptr() == model.covar_module.inducing_points.data_ptr()


def test_posterior_predictive_and_metrics():
    X, y, Xt, Z0 = demo_1d()
    model = ggp_amd.SparseGPR(X, y, ggp_amd.GaussianLikelihood(), Z0, engine=OracleEngine(), jitter=1e-6)
    model.base_covar_module.base_kernel.lengthscale = 0.9
    model.base_covar_module.outputscale = 1.4
    model.likelihood.noise = 0.16
    assert abs(float(model.likelihood.noise.detach()) - 0.16) < 1e-12
    pred = model.posterior_predictive(Xt)
    mu, cov = O.predict(Xt[:, None], X, y, Z0[:, None], torch.tensor([0.9], dtype=DT), 1.4, 0.16, 1e-6, full_cov=True)
    assert float((pred.loc - mu).abs().max()) < 1e-8 and float((pred.covariance_matrix - cov).abs().max()) < 1e-8  # the documented predictive tolerance
    assert pred.mean is pred.loc and pred.variance.shape == (200,)
    lo, hi = pred.confidence_region()
    assert torch.all(hi > lo)
    yt = torch.sin(Xt * 3)
    ystd = torch.tensor([1.0])
    assert abs(float(ggp_amd.rmse(pred.loc, yt, ystd)) - O.rmse(pred.loc, yt, 1.0)) < 1e-12
    assert abs(float(ggp_amd.nlpd(pred, yt, ystd)) - O.nlpd_joint(pred.loc, pred.covariance_matrix, yt, 1.0)) < 1e-8
    assert abs(ggp_amd.nlpd_marginal(pred, yt, ystd) - O.nlpd_marginal(pred.loc, torch.diagonal(pred.covariance_matrix), yt, 1.0)) < 1e-10
    q = model.optimal_q_u()
    assert q.loc.shape == (25,)


class GaussTarget:
    """Independent Gaussian in the unconstrained space: checks the sampler itself."""
    def __init__(self, sd):
        self.sd = np.asarray(sd, dtype=float)
        self.ndim = len(sd)

    def logp_and_grad(self, q):
        q = np.asarray(q)
        return float(-0.5 * np.sum((q / self.sd) ** 2)), list(-q / self.sd ** 2)

    def constrain(self, q):
        return {"ls": np.exp(q[:-2]), "sig_f": math.exp(q[-2]), "sig_n": math.exp(q[-1])}


def test_nuts_recovers_gaussian_moments_and_adapts():
    tgt = GaussTarget([0.5, 2.0, 1.0, 0.1])
    tr = ggp_amd.sample_nuts(tgt, 1500, 700, seed=3, start=[0.1, 0.1, 0.1, 0.1])
    th = tr["theta_unc"]
    assert th.shape == (1500, 4)
    assert np.all(np.abs(th.mean(0)) < 0.2 * tgt.sd + 0.02)
    assert np.all(np.abs(th.std(0) / tgt.sd - 1.0) < 0.15)
    acc = tr.get_sampler_stats("mean_tree_accept")
    assert 0.6 < acc.mean() < 0.97
    assert not tr.get_sampler_stats("diverging").any()
    assert np.ptp(tr.get_sampler_stats("step_size")) == 0.0    # frozen after tuning
    assert tr.get_sampler_stats("tree_size").max() <= 1024
    assert tr.n_leapfrog >= int(tr.get_sampler_stats("tree_size").sum())


def test_trace_surface():
    tgt = GaussTarget([1.0, 1.0, 1.0])
    tr = ggp_amd.sample_nuts(tgt, 20, 10, seed=0)
    assert len(tr) == 20 and set(tr[0]) >= {"ls", "sig_f", "sig_n"}
    assert tr["ls"].shape == (20, 1) and tr["sig_f"].shape == (20,)
    assert len(tr[::2]) == 10 and len(tr[::2].get_sampler_stats("step_size")) == 10
    assert tr.get_sampler_stats("perf_counter_diff").sum() > 0
    assert all(isinstance(s["sig_n"], float) for s in tr)


def small_problem():
    g = torch.Generator().manual_seed(1)
    X = torch.randn(60, 1, dtype=DT, generator=g) * 2
    y = torch.sin(X[:, 0]) + 0.2 * torch.randn(60, dtype=DT, generator=g)
    return X, y, torch.linspace(-3, 3, 6, dtype=DT), torch.linspace(-4, 4, 15, dtype=DT)


def test_bayesian_sgpr_hmc_fixed_model_and_mixture():
    X, y, Z0, Xt = small_problem()
    model = ggp_amd.BayesianSparseGPR_HMC(X, y, ggp_amd.GaussianLikelihood(), Z0, engine=OracleEngine(), seed=7)
    trace, steps, perf = model.train_fixed_model(num_tune=40, num_samples=15)
    assert len(trace) == 15 and len(steps) == 1 and len(perf) == 1 and perf[0] > 0
    assert trace["ls"].shape == (15, 1) and np.all(trace["ls"] > 0) and np.all(trace["sig_n"] > 0)
    preds = ggp_amd.mixture_posterior_predictive(model, Xt, trace)
    assert 1 <= len(preds) <= 15
    yt = torch.sin(Xt)
    v = ggp_amd.nlpd_mixture(preds, yt, torch.tensor([1.0]))
    assert math.isfinite(v)
    # the last sample's hypers were written into the model (noise = sig_n^2, outputscale = sig_f^2)
    assert abs(float(model.likelihood.noise) - trace[len(trace) - 1]["sig_n"] ** 2) < 1e-10
    # sampled log-density equals the oracle's HMC target at the same point
    th = trace[3]["theta_unc"]
    lp_ref, _ = O.hmc_logp(torch.tensor(th), X, y, Z0[:, None])
    assert abs(trace.get_sampler_stats("logp")[3] - lp_ref) < 1e-8 * max(1.0, abs(lp_ref))


def test_bayesian_sgpr_hmc_alternating_schedule():
    X, y, Z0, Xt = small_problem()
    model = ggp_amd.BayesianSparseGPR_HMC(X, y, ggp_amd.GaussianLikelihood(), Z0, engine=OracleEngine(), seed=11, jitter=1e-6)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    losses, trace, step_sizes, perf = model.train_model(opt, max_steps=9, hmc_scheduler=[3, 6, 8], verbose=False,
                                                        num_tune_long=15, num_samples_long=4, num_tune_short=8, num_samples_short=3)
    # 3 warm-start losses, nothing at iteration 3 (no trace yet), then one per iteration
    assert len(losses) == 3 + 5
    assert len(step_sizes) == 3 and len(perf) == 3 and len(trace) == 4
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert 'covar_module.inducing_points' not in frozen and len(frozen) == 3
