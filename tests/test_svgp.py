"""SVGP minibatch path.  CPU: model class + autograd bridge through the test double against
an independent torch-autograd loop on the oracle graph.  GPU: sgp_svgp_elbo / sgp_svgp_predict against the oracle."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ggp_amd
from fake_engine import OracleEngine
from oracle import svgp_oracle as S

DT = torch.float64


def problem(N=300, d=2, M=12, seed=0, classify=False):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, d, dtype=DT, generator=g)
    f = torch.sin(X[:, 0] * 1.5) + 0.5 * X[:, 1 % d]
    y = torch.sign(f + 0.3 * torch.randn(N, dtype=DT, generator=g)) if classify else f + 0.2 * torch.randn(N, dtype=DT, generator=g)
    Z = X[torch.randperm(N, generator=g)[:M]].clone()
    m = 0.3 * torch.randn(M, dtype=DT, generator=g)
    LS = torch.tril(0.2 * torch.randn(M, M, dtype=DT, generator=g)) + torch.eye(M, dtype=DT)
    ls = 0.8 + torch.rand(d, dtype=DT, generator=g)
    return X, y, Z, m, LS, ls


def reference_trace(X, y, Z0, batches, lr, lik, N):
    """-ELBO under Adam on GPyTorch's raw parametrisation, torch autograd on the oracle graph."""
    d, M = X.shape[1], Z0.shape[0]
    raw_ls = torch.zeros(1, d, dtype=DT, requires_grad=True)
    raw_os = torch.zeros((), dtype=DT, requires_grad=True)
    raw_n = torch.zeros(1, dtype=DT, requires_grad=True)
    Z = Z0.clone().requires_grad_(True)
    m = torch.zeros(M, dtype=DT, requires_grad=True)
    LS = torch.eye(M, dtype=DT, requires_grad=True)
    params = ([raw_n] if lik == 0 else []) + [raw_os, raw_ls, Z, m, LS]
    opt = torch.optim.Adam(params, lr=lr)
    out = []
    for xb, yb in batches:
        opt.zero_grad()
        s2 = F.softplus(raw_n)[0] + 1e-4 if lik == 0 else torch.tensor(1.0, dtype=DT)
        loss = -S.svgp_elbo(xb, yb, Z, F.softplus(raw_ls).reshape(-1), F.softplus(raw_os), s2, m, LS, N, 1e-6, 0, lik)
        out.append(float(loss.detach()))
        loss.backward()
        opt.step()
    return out


@pytest.mark.parametrize("classify", [False, True])
def test_svgp_model_trace_matches_autograd_reference(classify):
    X, y, Z0, _, _, _ = problem(classify=classify)
    lik = ggp_amd.BernoulliLikelihood() if classify else ggp_amd.GaussianLikelihood()
    model = ggp_amd.StochasticVariationalGP(X, y, lik, Z0, engine=OracleEngine())
    assert model.num_inducing == 12 and model.num_data == 300
    batches = [(X[i:i + 100], y[i:i + 100]) for i in (0, 100, 200)]
    opt = torch.optim.Adam(model.parameters(), lr=0.02)
    losses = model.train_model(opt, batches, minibatch_size=100, num_epochs=2)
    ref = reference_trace(X, y, Z0, batches * 2, 0.02, 1 if classify else 0, 300)
    assert len(losses) == 6
    assert np.max(np.abs(np.array(losses) - np.array(ref))) < 1e-8, (losses, ref)
    pred = model.posterior_predictive(X[:7])
    if classify:
        assert pred.shape == (7,) and torch.all((pred > 0) & (pred < 1))
    else:
        assert pred.loc.shape == (7,) and torch.all(pred.variance > 0)


def test_oracle_svgp_kl_and_gaussian_ell_closed_forms():
    X, y, Z, m, LS, ls = problem()
    ell, kl, mu, v = S.svgp_terms(X, y, Z, ls, 1.3, 0.1, m, LS)
    Smat = LS @ LS.T
    kl_ref = 0.5 * (torch.trace(Smat) + m @ m - 12 - torch.logdet(Smat))
    assert abs(float(kl - kl_ref)) < 1e-10
    # Gaussian expectation by quadrature agrees with the closed form
    x, w = S.gauss_hermite(40)
    fq = mu[:, None] + torch.sqrt(v)[:, None] * x[None, :]
    quad = ((-0.5 * math.log(2 * math.pi * 0.1) - (y[:, None] - fq) ** 2 / 0.2) * w[None, :]).sum(1)
    assert float((quad - ell).abs().max()) < 1e-10


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("N,d,M,lik,kern", [(300, 2, 12, "gaussian", "rbf"), (300, 2, 12, "bernoulli", "rbf"),
                                            (4096, 2, 256, "bernoulli", "rbf"), (1000, 5, 130, "gaussian", "matern52"),
                                            (65, 1, 5, "gaussian", "rbf")])
def test_svgp_elbo_and_grads_vs_oracle(engine, N, d, M, lik, kern):
    X, y, Z, m, LS, ls = problem(N, d, M, seed=N + M, classify=(lik == "bernoulli"))
    kid = {"rbf": 0, "matern32": 1, "matern52": 2}[kern]
    likid = 1 if lik == "bernoulli" else 0
    N_total = 10 * N
    if kid == 0:
        ref = S.svgp_elbo_and_grads(X, y, Z, ls, 1.3, 0.1, m, LS, N_total, 1e-6, kid, likid)
    else:
        ref = None
        ell, kl, _, _ = S.svgp_terms(X, y, Z, ls, 1.3, 0.1, m, LS, 1e-6, kid, likid)
        elbo_ref = float(ell.mean() - kl / N_total)
    D = lambda t: t.to(engine.device).contiguous()  # noqa: E731
    res = engine.svgp_elbo(D(X), D(y), D(Z), ls.tolist(), 1.3, 0.1, D(m), D(LS), N_total, jitter=1e-6, kernel=kern,
                           likelihood=lik, with_grads=True)
    assert int(res["info"].item()) == 0
    out = res["out"].cpu()
    if ref is None:
        assert abs(float(out[0]) - elbo_ref) < 1e-10 * max(1.0, abs(elbo_ref))
        return
    assert abs(float(out[0]) - ref["elbo"]) < 1e-10 * max(1.0, abs(ref["elbo"]))

    def close(a, b, rt=2e-7):
        a, b = a.cpu().reshape(-1), torch.as_tensor(b).reshape(-1)
        return float((a - b).abs().max()) < rt * max(1e-3, float(b.abs().max()))
    assert close(res["g_m"], ref["g_m"])
    assert close(res["g_LS"], ref["g_LS"])
    assert close(res["g_Z"], ref["g_Z"], 1e-6)
    assert close(res["g_ls"], ref["g_ls"], 1e-6)
    assert close(res["g_sf2"], torch.tensor([ref["g_sf2"]]), 1e-6)
    if lik == "gaussian":
        assert close(res["g_s2"], torch.tensor([ref["g_s2"]]))
    # value-only call returns the same bound
    res2 = engine.svgp_elbo(D(X), D(y), D(Z), ls.tolist(), 1.3, 0.1, D(m), D(LS), N_total, jitter=1e-6, kernel=kern, likelihood=lik)
    assert float(res2["out"][0]) == float(res["out"][0])
    mu, v, info = engine.svgp_predict(D(X[:50]), D(Z), ls.tolist(), 1.3, D(m), D(LS), jitter=1e-6, kernel=kern)
    mu_r, v_r = S.svgp_predict(X[:50], Z, ls, 1.3, m, LS, 1e-6, kid)
    # cond(Kuu) ~ 1e6 at M = 256 with jitter 1e-6: two solvers agree to ~1e-9 relative
    assert float((mu.cpu() - mu_r).abs().max()) < 1e-7 and float(((v.cpu() - v_r) / v_r).abs().max()) < 1e-7


@pytest.mark.gpu
def test_svgp_model_on_device_c4_shape(engine):
    """BASELINE config C4 shape at reduced N: 2-D classification, M = 256, minibatch 4096, Bernoulli-probit."""
    g = torch.Generator().manual_seed(4)
    N, M, Bsz = 20000, 256, 4096
    X = torch.randn(N, 2, dtype=DT, generator=g)
    y = torch.sign(torch.sin(2 * X[:, 0]) * torch.cos(X[:, 1]) + 0.1 * torch.randn(N, dtype=DT, generator=g))
    Z0 = X[torch.randperm(N, generator=g)[:M]].clone()
    model = ggp_amd.StochasticVariationalGP(X.to(engine.device), y.to(engine.device), ggp_amd.BernoulliLikelihood(), Z0, engine=engine)
    batches = [(X[i:i + Bsz].to(engine.device), y[i:i + Bsz].to(engine.device)) for i in range(0, N - Bsz + 1, Bsz)]
    opt = torch.optim.Adam(model.parameters(), lr=0.05)
    losses = model.train_model(opt, batches, minibatch_size=Bsz, num_epochs=6)
    ref = reference_trace(X, y, Z0, [(a.cpu(), b.cpu()) for a, b in batches[:2]], 0.05, 1, N)
    assert np.max(np.abs(np.array(losses[:2]) - np.array(ref))) < 1e-8
    assert losses[-1] < losses[0]
    p = model.posterior_predictive(X[:2000].to(engine.device)).cpu()
    acc = float(((p > 0.5) == (y[:2000] > 0)).double().mean())
    assert acc > 0.8, acc


@pytest.mark.gpu
def test_bayesian_svgp_on_device_c4_shape(engine):
    """``BayesianStochasticVariationalGP`` (reference models/bayesian_svgp.py:156-167: five reparametrised hyper-samples
    per minibatch) on the HIP path at the C4 minibatch shape (B = 4096, M = 256): the first minibatch losses against the
    same model on the CPU test double with the same seeds, then a few epochs that must reduce the loss."""
    g = torch.Generator().manual_seed(4)
    N, M, Bsz = 12288, 256, 4096
    X = torch.randn(N, 2, dtype=DT, generator=g)
    y = torch.sin(2 * X[:, 0]) * torch.cos(X[:, 1]) + 0.1 * torch.randn(N, dtype=DT, generator=g)
    Z0 = X[torch.randperm(N, generator=g)[:M]].clone()
    out = []
    for eng in (engine, OracleEngine()):
        torch.manual_seed(0)
        model = ggp_amd.BayesianStochasticVariationalGP(X.to(eng.device), y.to(eng.device), ggp_amd.GaussianLikelihood(), Z0,
                                                        engine=eng, seed=3)
        batches = [(X[i:i + Bsz].to(eng.device), y[i:i + Bsz].to(eng.device)) for i in range(0, N, Bsz)]
        opt = torch.optim.Adam(model.parameters(), lr=0.02)
        ne = 4 if eng is engine else 1
        epoch_losses, batch_losses = model.train_model(opt, batches if eng is engine else batches[:2], num_epochs=ne)
        out.append((epoch_losses, batch_losses, model))
    (ea, ba, ma), (eb, bb, mb) = out
    assert abs(ea[0] / 3 - 0) >= 0 and len(ea) == 4 and all(math.isfinite(v) for v in ea)
    assert ea[-1] < ea[0]
    # first epoch, first two minibatches: same numbers as the CPU double (losses are O(1) per datum)
    ma2 = ggp_amd.BayesianStochasticVariationalGP(X.to(engine.device), y.to(engine.device), ggp_amd.GaussianLikelihood(), Z0,
                                                  engine=engine, seed=3)
    batches = [(X[i:i + Bsz].to(engine.device), y[i:i + Bsz].to(engine.device)) for i in range(0, N, Bsz)]
    _, b2 = ma2.train_model(torch.optim.Adam(ma2.parameters(), lr=0.02), batches[:2], num_epochs=1)
    assert np.max(np.abs(np.array(b2) - np.array(bb))) < 1e-7 * max(1.0, np.max(np.abs(bb))), (b2, bb)
    preds = ma.mixture_posterior_predictive(X[:64].to(engine.device), num_samples=5)
    assert len(preds) == 5 and preds[0].loc.shape == (64,) and bool(torch.all(preds[0].variance > 0))


def test_bayesian_svgp_hyper_distribution_and_training():
    """reference models/bayesian_svgp.py: q(log theta) KL term, 5 reparametrised samples per minibatch."""
    X, y, Z0, _, _, _ = problem(N=200, M=8)
    model = ggp_amd.BayesianStochasticVariationalGP(X, y, ggp_amd.GaussianLikelihood(), Z0, engine=OracleEngine(), seed=3)
    hd = model.log_theta
    assert hd.hyper_dim == 4 and hd.q_sigma_vec.numel() == 10
    q = torch.distributions.MultivariateNormal(hd.q_mu, hd.construct_sigma())
    p = torch.distributions.MultivariateNormal(torch.zeros(4, dtype=DT), 0.01 * torch.eye(4, dtype=DT))
    assert abs(float(hd.kl_per_point()) - float(torch.distributions.kl_divergence(q, p)) / 200) < 1e-10
    s = model.sample_variational_log_hyper(7)
    assert s.shape == (7, 4)
    batches = [(X[i:i + 100], y[i:i + 100]) for i in (0, 100)]
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    epoch_losses, batch_losses = model.train_model(opt, batches, num_epochs=3)
    assert len(epoch_losses) == 3 and len(batch_losses) == 2 and all(math.isfinite(v) for v in epoch_losses)
    assert hd.q_mu.grad is not None and float(hd.q_mu.grad.abs().max()) > 0     # the reparametrisation gradient arrives
    assert model.variational_mean.grad is not None and model.inducing_inputs.grad is not None
    preds = model.mixture_posterior_predictive(X[:9], num_samples=6)
    assert len(preds) == 6 and preds[0].loc.shape == (9,) and torch.all(preds[0].variance > 0)


# ------------------------------------------------------------------------------------------------- batched hyper-samples (C4)
def bayes_reference_trace(X, y, Z0, batches, lr, lik, N, seed, S_hyper=5):
    """Independent statement of the Bayesian SVGP step (reference models/bayesian_svgp.py:144-181): q(log theta) =
    N(q_mu, L L^T + 1e-5 I), S reparametrised draws per minibatch, loss = mean_k(-ELBO(exp(log theta_k)) + KL_theta / n), torch autograd
    on the oracle graph, Adam.  Bernoulli (lik = 1): labels -> +-1, no noise hyper-parameter, q has d + 1 dimensions."""
    d, M = X.shape[1], Z0.shape[0]
    hd = d + (1 if lik == 1 else 2)
    g = torch.Generator().manual_seed(seed)
    q_mu = (torch.randn(hd, dtype=DT, generator=g) * 1e-3).requires_grad_(True)
    q_vec = (torch.randn(hd * (hd + 1) // 2, dtype=DT, generator=g) * 1e-3).requires_grad_(True)
    Z = Z0.clone().requires_grad_(True)
    m = torch.zeros(M, dtype=DT, requires_grad=True)
    LS = torch.eye(M, dtype=DT, requires_grad=True)
    # the model registers: likelihood / kernel raw parameters (unused by the Bayesian loss), Z, m, LS, then q -- Adam treats every
    # parameter independently, so only the ones the loss reaches matter
    opt = torch.optim.Adam([Z, m, LS, q_mu, q_vec], lr=lr)
    rows, cols = torch.tril_indices(hd, hd)
    out = []
    for xb, yb in batches:
        opt.zero_grad()
        lower = torch.zeros(hd, hd, dtype=DT).index_put((rows, cols), q_vec)
        Sg = lower @ lower.T + 1e-5 * torch.eye(hd, dtype=DT)
        kl = 0.5 * (torch.trace(Sg) / 0.01 + (q_mu @ q_mu) / 0.01 - hd + hd * math.log(0.01) - torch.logdet(Sg)) / N
        Lq = torch.linalg.cholesky(Sg)
        yy = torch.where(yb > 0, torch.ones_like(yb), -torch.ones_like(yb)) if lik == 1 else yb
        loss = 0.0
        for _ in range(S_hyper):
            eps = torch.randn(1, hd, dtype=DT, generator=g)
            th = torch.exp((q_mu[None, :] + eps @ Lq.T).flatten())
            s2 = torch.tensor(1.0, dtype=DT) if lik == 1 else th[-1] ** 2
            ls = th[1:] if lik == 1 else th[1:-1]
            loss = loss + (-S.svgp_elbo(xb, yy, Z, ls, th[0], s2, m, LS, N, 1e-6, 0, lik) + kl) / S_hyper
        out.append(float(loss.detach()))
        loss.backward()
        opt.step()
    return out


@pytest.mark.parametrize("classify", [False, True])
def test_bayesian_svgp_trace_matches_autograd_reference_and_batched_equals_chains(classify):
    """Model-level C4 wiring on the CPU double: Bernoulli-probit gets +-1 labels and a (d + 1)-dimensional q(log theta) -- no noise
    hyper-parameter --, the five bounds of a minibatch come from ONE svgp_elbo_batch call, and the loss trace equals an
    independent autograd loop; the per-sample launch chains (batched = False) give the same trace."""
    X, y, Z0, _, _, _ = problem(N=240, M=10, classify=classify)
    if classify:
        y = (y > 0).to(DT)  # {0, 1} labels, as a data set would hold them
    lik = ggp_amd.BernoulliLikelihood() if classify else ggp_amd.GaussianLikelihood()
    batches = [(X[i:i + 80], y[i:i + 80]) for i in (0, 80, 160)]
    ref = bayes_reference_trace(X, y, Z0, batches, 0.02, 1 if classify else 0, 240, seed=5)
    traces = []
    for batched in (True, False):
        eng = OracleEngine()
        calls = {"batch": 0, "single": 0}
        inner_b, inner_s = eng.svgp_elbo_batch, eng.svgp_elbo
        eng.svgp_elbo_batch = lambda *a, **k: (calls.__setitem__("batch", calls["batch"] + 1), inner_b(*a, **k))[1]
        model = ggp_amd.BayesianStochasticVariationalGP(X, y, lik, Z0, engine=eng, seed=5)
        model.batched = batched
        assert model.hyper_dim == (3 if classify else 4) and model.log_theta.hyper_dim == model.hyper_dim
        _, bl = model.train_model(torch.optim.Adam(model.parameters(), lr=0.02), batches, num_epochs=1)
        traces.append(bl)
        assert calls["batch"] == (3 if batched else 0)
        assert model.log_theta.q_mu.grad is not None and float(model.log_theta.q_mu.grad.abs().max()) > 0
    assert np.max(np.abs(np.array(traces[0]) - np.array(ref))) < 1e-9, (traces[0], ref)
    assert np.max(np.abs(np.array(traces[0]) - np.array(traces[1]))) < 1e-12
    preds = model.mixture_posterior_predictive(X[:9], num_samples=4)
    assert len(preds) == 4
    if classify:
        assert preds[0].shape == (9,) and bool(torch.all((preds[0] > 0) & (preds[0] < 1)))
    else:
        assert preds[0].loc.shape == (9,) and bool(torch.all(preds[0].variance > 0))


@pytest.mark.gpu
@pytest.mark.parametrize("lik,kern,B,M,d,S_hyper", [("bernoulli", "rbf", 4096, 256, 2, 5), ("gaussian", "rbf", 1000, 70, 3, 3),
                                                    ("gaussian", "matern32", 333, 40, 2, 8), ("bernoulli", "rbf", 65, 5, 1, 1)])
def test_svgp_elbo_batch_vs_oracle_and_single_chains(engine, lik, kern, B, M, d, S_hyper):
    """sgp_svgp_elbo_batch (S hyper-parameter samples, one launch chain) against oracle/svgp_oracle.py sample by sample -- bound 1e-9,
    gradients 1e-6 -- and against S calls of sgp_svgp_elbo.  First case = BASELINE C4's minibatch: B 4096, M 256, Bernoulli, 5 samples."""
    X, y, Z, m, LS, ls0 = problem(B, d, M, seed=B + M, classify=(lik == "bernoulli"))
    kid = {"rbf": 0, "matern32": 1}[kern]
    likid = 1 if lik == "bernoulli" else 0
    N_total = 100_000
    g = torch.Generator().manual_seed(7)
    ls = ls0[None, :] * torch.exp(0.2 * torch.randn(S_hyper, d, dtype=DT, generator=g))
    sf2 = 1.3 * torch.exp(0.2 * torch.randn(S_hyper, dtype=DT, generator=g))
    s2 = 0.1 * torch.exp(0.3 * torch.randn(S_hyper, dtype=DT, generator=g))
    D = lambda t: t.to(engine.device).contiguous()  # noqa: E731
    res = engine.svgp_elbo_batch(D(X), D(y), D(Z), ls.tolist(), sf2.tolist(), s2.tolist(), D(m), D(LS), N_total, jitter=1e-6, kernel=kern,
                                 likelihood=lik, with_grads=True)
    assert res["info"].cpu().tolist() == [0] * S_hyper
    val = engine.svgp_elbo_batch(D(X), D(y), D(Z), ls.tolist(), sf2.tolist(), s2.tolist(), D(m), D(LS), N_total, jitter=1e-6, kernel=kern,
                                 likelihood=lik)
    assert torch.equal(val["out"], res["out"])

    def close(a, b, rt):
        a, b = a.cpu().reshape(-1), torch.as_tensor(b).reshape(-1)
        return float((a - b).abs().max()) < rt * max(1e-3, float(b.abs().max()))
    for k in range(S_hyper):
        one = engine.svgp_elbo(D(X), D(y), D(Z), ls[k].tolist(), float(sf2[k]), float(s2[k]), D(m), D(LS), N_total, jitter=1e-6, kernel=kern,
                               likelihood=lik, with_grads=True)
        # (1e-10, not rounding level: since round 5 a single factorization runs the chain-workgroup Cholesky and the batch the tile-dataflow
        # one -- two orders of the rank-64 updates, each within 1.1e-15 of LAPACK (tools/potrf_bench.py); cond(K_uu) ~ 1 / jitter does the rest)
        assert abs(float(res["out"][k, 0]) - float(one["out"][0])) < 1e-10 * max(1.0, abs(float(one["out"][0])))
        for key in ("g_m", "g_LS", "g_Z", "g_ls"):
            assert close(res[key][k], one[key].cpu(), 1e-8), (k, key)
        assert close(res["g_sf2"][k], one["g_sf2"].cpu(), 1e-8)
        if kid != 0:
            continue
        ref = S.svgp_elbo_and_grads(X, y, Z, ls[k], float(sf2[k]), float(s2[k]), m, LS, N_total, 1e-6, kid, likid)
        assert abs(float(res["out"][k, 0]) - ref["elbo"]) < 1e-9 * max(1.0, abs(ref["elbo"])), (k, float(res["out"][k, 0]), ref["elbo"])
        assert close(res["g_m"][k], ref["g_m"], 1e-6) and close(res["g_LS"][k], ref["g_LS"], 1e-6)
        assert close(res["g_Z"][k], ref["g_Z"], 1e-6) and close(res["g_ls"][k], ref["g_ls"], 1e-6)
        assert close(res["g_sf2"][k], torch.tensor([ref["g_sf2"]]), 1e-6)
        if lik == "gaussian":
            assert close(res["g_s2"][k], torch.tensor([ref["g_s2"]]), 1e-6)
    assert res["out"][:, 3].cpu().tolist() == [0.0] * S_hyper  # the status words ride along in the result rows
    # q(f*) for all samples in one chain (the mixture predictive of BayesianStochasticVariationalGP) = the per-sample call
    Tn = min(B, 77)
    mu_b, v_b, info_b = engine.svgp_predict_batch(D(X[:Tn]), D(Z), ls.tolist(), sf2.tolist(), D(m), D(LS), jitter=1e-6, kernel=kern)
    assert info_b.cpu().tolist() == [0] * S_hyper and mu_b.shape == (S_hyper, Tn)
    for k in range(S_hyper):
        mu1, v1, _ = engine.svgp_predict(D(X[:Tn]), D(Z), ls[k].tolist(), float(sf2[k]), D(m), D(LS), jitter=1e-6, kernel=kern)
        # (1e-9: batch and single factorizations run different Cholesky kernels since round 5, see above)
        assert float((mu_b[k] - mu1).abs().max()) < 1e-9 and float(((v_b[k] - v1) / v1).abs().max()) < 1e-9
    # the test rows travel in chunks (8192 by default; 32 here: three chunks, the last one short): the same numbers
    try:
        engine.SVGP_PREDICT_CHUNK = 32
        mu_c, v_c, info_c = engine.svgp_predict_batch(D(X[:Tn]), D(Z), ls.tolist(), sf2.tolist(), D(m), D(LS), jitter=1e-6, kernel=kern)
    finally:
        del engine.SVGP_PREDICT_CHUNK
    assert info_c.cpu().tolist() == [0] * S_hyper
    assert float((mu_c - mu_b).abs().max()) < 1e-12 and float(((v_c - v_b) / v_b).abs().max()) < 1e-12
    # the two-halves form (forward, [the caller's copy of the bounds], reverse) gives the same numbers bit for bit
    sp = engine.svgp_elbo_batch(D(X), D(y), D(Z), ls.tolist(), sf2.tolist(), s2.tolist(), D(m), D(LS), N_total, jitter=1e-6, kernel=kern,
                                likelihood=lik, with_grads=True, defer_reverse=True)
    early = sp["out"].clone()
    sp.pop("reverse")()
    assert torch.equal(early, res["out"]) and all(torch.equal(sp[k], res[k]) for k in ("g_m", "g_LS", "g_Z", "g_ls", "g_sf2", "g_s2"))
    # reverse pass of a weighted sum of the S bounds in one launch (sgp_svgp_batch_combine)
    wts = torch.linspace(-0.7, 1.1, S_hyper, dtype=DT)
    gm, gLS, gZ, gth = engine.svgp_batch_combine(res, wts.tolist())
    wd = wts.to(engine.device)
    assert close(gm, torch.einsum("s,sm->m", wd, res["g_m"]).cpu(), 1e-13) and close(gLS, torch.einsum("s,smk->mk", wd, res["g_LS"]).cpu(), 1e-13)
    assert close(gZ, torch.einsum("s,smd->md", wd, res["g_Z"]).cpu(), 1e-13)
    assert close(gth, (torch.cat([res["g_sf2"][:, None], res["g_ls"], res["g_s2"][:, None]], 1) * wd[:, None]).cpu(), 1e-14)
    # every sample has its own status word: K_uu - I is indefinite for each of them (a smooth kernel on M >= 40 points has
    # eigenvalues far below 1), and the next call on the same workspace is clean again
    if S_hyper >= 3:
        bad = engine.svgp_elbo_batch(D(X), D(y), D(Z), ls.tolist(), sf2.tolist(), s2.tolist(), D(m), D(LS), N_total, jitter=-1.0, kernel=kern,
                                     likelihood=lik)
        assert all(1 <= v <= M for v in bad["info"].cpu().tolist()), bad["info"].cpu().tolist()
        again = engine.svgp_elbo_batch(D(X), D(y), D(Z), ls.tolist(), sf2.tolist(), s2.tolist(), D(m), D(LS), N_total, jitter=1e-6, kernel=kern,
                                       likelihood=lik)
        assert again["info"].cpu().tolist() == [0] * S_hyper and torch.equal(again["out"], val["out"])


@pytest.mark.gpu
def test_bayesian_svgp_bernoulli_c4_as_named(engine):
    """BASELINE config C4 AS NAMED: BayesianStochasticVariationalGP x Bernoulli-probit, synthetic 2-D classification, N = 100 000,
    M = 256, SVI minibatch 4096, five hyper-samples per minibatch (reference models/bayesian_svgp.py:144-181; Bernoulli use
    scratch_pymc3.py:78-88).  First minibatches against the same model on the oracle-backed CPU double with the same seeds, then
    two epochs over all 100k rows that must reduce the loss and classify."""
    g = torch.Generator().manual_seed(4)
    N, M, Bsz = 100_000, 256, 4096
    X = torch.randn(N, 2, dtype=DT, generator=g)
    y = (torch.sin(2 * X[:, 0]) * torch.cos(X[:, 1]) + 0.1 * torch.randn(N, dtype=DT, generator=g) > 0).to(DT)  # {0, 1}
    Z0 = X[torch.randperm(N, generator=g)[:M]].clone()
    traces = []
    for eng in (engine, OracleEngine()):
        model = ggp_amd.BayesianStochasticVariationalGP(X.to(eng.device), y.to(eng.device), ggp_amd.BernoulliLikelihood(), Z0, engine=eng, seed=3)
        assert model.hyper_dim == 3
        batches = [(X[i:i + Bsz].to(eng.device), y[i:i + Bsz].to(eng.device)) for i in (0, Bsz)]
        _, bl = model.train_model(torch.optim.Adam(model.parameters(), lr=0.02), batches, num_epochs=1)
        traces.append(bl)
    assert np.max(np.abs(np.array(traces[0]) - np.array(traces[1]))) < 1e-8 * max(1.0, np.max(np.abs(traces[1]))), traces
    Xd, yd = X.to(engine.device), y.to(engine.device)
    model = ggp_amd.BayesianStochasticVariationalGP(Xd, yd, ggp_amd.BernoulliLikelihood(), Z0, engine=engine, seed=3)
    batches = [(Xd[i:i + Bsz], yd[i:i + Bsz]) for i in range(0, N - Bsz + 1, Bsz)]
    ep, bl = model.train_model(torch.optim.Adam(model.parameters(), lr=0.05), batches, num_epochs=2)
    assert len(ep) == 2 and len(bl) == 24 and ep[1] < ep[0] and all(math.isfinite(v) for v in ep)
    probs = torch.stack(model.mixture_posterior_predictive(Xd[:4000], num_samples=5)).mean(0).cpu()
    acc = float(((probs > 0.5) == (y[:4000] > 0)).double().mean())
    assert acc > 0.85, acc
