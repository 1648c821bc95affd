"""The samplers against an EXACT posterior.  tests/golden/posterior_rbf_d1_tiny.npz holds the mean, covariance and fourth
moments of the reference's NUTS target (models/bayesian_sgpr_hmc.py:58-80) on the d = 1 fixture, obtained by quadrature of the
oracle's density over R^3 (tests/golden/make_golden_posterior.py) -- numbers no sampler produced.  Every sampler of this
repository must reproduce them within 4 Monte-Carlo standard errors (MCSE from the chain's own autocorrelation, Geyer's
initial-positive-sequence estimator):

  CPU:  hmc.sample_nuts (the host sampler) over the oracle-backed test double;
        the device sampler's state machine csrc/sgp_nuts.hpp compiled for the host, fed oracle.hmc_logp;
  GPU:  sample_nuts_device (one persistent launch) and sample_nuts over the HIP single launch.
"""
import ctypes as C
import math
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, dev, load_golden

import ggp_amd


def ess(x):
    """Effective sample size of a scalar chain (Geyer 1992: sum of autocorrelation pairs while positive)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    c = x - x.mean()
    f = np.fft.rfft(c, 2 * n)
    acov = np.fft.irfft(f * np.conj(f))[:n] / n
    rho = acov / acov[0]
    s = 0.0
    for k in range(0, n - 1, 2):
        pair = rho[k] + rho[k + 1]
        if pair < 0.0:
            break
        s += pair
    return n / max(2.0 * s - 1.0, 1.0 / n)


def check_moments(theta, P, what, nsig=4.0):
    """theta: draws x 3 in the unconstrained space.  Means and variances against the quadrature, each within nsig MCSE."""
    theta = np.asarray(theta)
    mean, var, m4 = P["mean"], np.diag(P["cov"]), P["m4"]
    for k, name in enumerate(("log ls", "log sig_f", "log sig_n")):
        x = theta[:, k]
        n_eff = ess(x)
        assert n_eff > 100, (what, name, n_eff)
        mcse = math.sqrt(var[k] / n_eff)
        assert abs(x.mean() - mean[k]) < nsig * mcse, "%s: E[%s] = %.4f, exact %.4f, 4 MCSE = %.4f (ESS %.0f)" % (
            what, name, x.mean(), mean[k], nsig * mcse, n_eff)
        sq = (x - mean[k]) ** 2
        mcse_v = math.sqrt(max(m4[k] - var[k] ** 2, 1e-300) / ess(sq))
        assert abs(sq.mean() - var[k]) < nsig * mcse_v, "%s: Var[%s] = %.5f, exact %.5f, 4 MCSE = %.5f" % (
            what, name, sq.mean(), var[k], nsig * mcse_v)
    # the one sizeable correlation of this posterior (ls with sig_f): sign and size
    r_exact = P["cov"][0, 1] / math.sqrt(var[0] * var[1])
    r = np.corrcoef(theta[:, 0], theta[:, 1])[0, 1]
    assert abs(r - r_exact) < 0.15, (what, r, r_exact)


def unconstrained(tr):
    return np.log(np.concatenate([np.asarray(tr["ls"]).reshape(len(tr), -1), np.asarray(tr["sig_f"])[:, None],
                                  np.asarray(tr["sig_n"])[:, None]], 1))


def test_fixture_is_the_quadrature_of_the_oracle_density():
    """Spot check of the stored numbers without redoing the 1.4 M-node rule, and with oracle.hmc_logp ITSELF (not the generator's
    batched restatement): the stored peak is the oracle's value there, and importance sampling from N(mean, 1.5^2 cov) with the
    oracle density reproduces the stored evidence (weights average to 1) and mean."""
    from oracle import vfe_oracle as O
    P = load_golden("posterior_rbf_d1_tiny")
    th = np.array([1.6, 0.2, -1.8])
    assert abs(O.hmc_logp(th, P["X"], P["y"], P["Z"], jitter=float(P["jitter"]), with_grad=False)[0] - float(P["logp_peak"])) < 1e-6
    rng = np.random.default_rng(1)
    Lc = np.linalg.cholesky(2.25 * P["cov"])
    z = rng.standard_normal((4000, 3))
    pts = P["mean"] + z @ Lc.T
    lq = -0.5 * (z * z).sum(1) - np.log(np.diag(Lc)).sum() - 1.5 * math.log(2 * math.pi)
    lp = np.array([O.hmc_logp(p, P["X"], P["y"], P["Z"], jitter=float(P["jitter"]), with_grad=False)[0] for p in pts])
    w = np.exp(lp - lq - float(P["log_evidence"]))
    assert abs(w.mean() - 1.0) < 4.0 * w.std() / math.sqrt(w.size), (w.mean(), w.std())   # evidence
    m = (w[:, None] * pts).sum(0) / w.sum()
    assert np.all(np.abs(m - P["mean"]) < 0.02), (m, P["mean"])


def test_host_sampler_reproduces_the_exact_posterior():
    from fake_engine import OracleEngine
    P = load_golden("posterior_rbf_d1_tiny")
    T = lambda a: torch.as_tensor(a, dtype=torch.float64)
    cb = ggp_amd.CollapsedBound(T(P["X"]), T(P["y"]), jitter=float(P["jitter"]), engine=OracleEngine())
    tr = ggp_amd.sample_nuts(ggp_amd.HmcTarget(cb, T(P["Z"])), 1200, 400, seed=11)
    assert np.asarray(tr.get_sampler_stats("diverging")).mean() <= 0.01
    check_moments(unconstrained(tr), P, "hmc.sample_nuts / oracle double")


def test_device_sampler_state_machine_reproduces_the_exact_posterior(tmp_path):
    """csrc/sgp_nuts.hpp -- the code the persistent kernel executes -- built with g++ and fed the oracle's logp + gradient."""
    from oracle import vfe_oracle as O
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    so = str(tmp_path / "libnuts_host.so")
    subprocess.run([gxx, "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I",
                    os.path.join(ROOT, "generalised-gaussian-processes_amd", "csrc"), "-o", so,
                    os.path.join(ROOT, "tests", "native", "nuts_host.cpp")], check=True, timeout=300)
    lib = C.CDLL(so)
    CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
    lib.nuts_host_run.restype = C.c_long
    lib.nuts_host_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_ulonglong, C.POINTER(C.c_double), CB,
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    P = load_golden("posterior_rbf_d1_tiny")

    def cb(qp, lpp, gp):
        try:  # PyMC3: a failed factorization is a non-finite density (a divergence), not an exception
            lp, g = O.hmc_logp([qp[0], qp[1], qp[2]], P["X"], P["y"], P["Z"], jitter=float(P["jitter"]))
        except RuntimeError:
            lp, g = -math.inf, None
        ok = math.isfinite(lp)
        lpp[0] = lp if ok else -math.inf
        for i in range(3):
            gp[i] = float(g[i]) if ok else 0.0

    draws, tune = 1200, 400
    samples = np.zeros((draws, 3))
    stats = np.zeros((draws, 8))
    q0 = np.array([math.log(2.0), 0.0, 0.0])
    lib.nuts_host_run(3, tune, draws, 10, 0.25, 0.8, 5, q0.ctypes.data_as(C.POINTER(C.c_double)), CB(cb),
                      samples.ctypes.data_as(C.POINTER(C.c_double)), stats.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert stats[:, 4].mean() <= 0.01  # divergent draws: a trajectory that reaches ls ~ 10 x the inducing spacing fails chol(Kuu + 1e-6 I), as in PyMC3
    check_moments(samples, P, "sgp_nuts.hpp (host build) / oracle.hmc_logp")


@pytest.mark.gpu
def test_gpu_samplers_reproduce_the_exact_posterior(engine):
    """2 000 draws from the persistent-kernel sampler and from the host-driven sampler over the HIP single launch."""
    P = load_golden("posterior_rbf_d1_tiny")
    X, y, Z = dev(P["X"], engine), dev(P["y"], engine), dev(P["Z"], engine)
    tgt = ggp_amd.HmcTarget(ggp_amd.CollapsedBound(X, y, jitter=float(P["jitter"]), engine=engine), Z)
    assert tgt.device_sampler_ok()
    tr = ggp_amd.sample_nuts_device(tgt, 2000, 1000, seed=21)
    # (a trajectory that wanders to ls ~ 10 x the inducing spacing fails chol(K_uu + 1e-6 I) and is flagged, as PyMC3 would: rare)
    assert np.asarray(tr.get_sampler_stats("diverging")).mean() <= 0.01
    check_moments(unconstrained(tr), P, "sample_nuts_device (persistent kernel)")
    tr2 = ggp_amd.sample_nuts(tgt, 2000, 1000, seed=22)
    check_moments(unconstrained(tr2), P, "sample_nuts (host tree, HIP single launch)")
    # and the multi-launch streaming path (what M > 128 / several ranks use) as the density
    cb = ggp_amd.CollapsedBound(X, y, jitter=float(P["jitter"]), engine=engine, form="whitened")
    cb.fused = False
    tr3 = ggp_amd.sample_nuts(ggp_amd.HmcTarget(cb, Z), 1000, 500, seed=23)
    check_moments(unconstrained(tr3), P, "sample_nuts (host tree, multi-launch whitened path)")


# ---------------------------------------------------------------------------------------------
# HmcTarget(gradient="sampler") -- VERDICT r4 next-3.  The d = 1 fixture sits in the streaming-order guard's regime all by itself once
# the multi-launch path is forced (estimates over the posterior: 3e-10 .. 1.5e-6, median 1.9e-8, against the tolerance 1e-9): the
# default mode sent ~85 % of the leapfrogs to the whitened order through round 5 (since round 6: to the extended order with both words of a
# double-double Phibar), the sampler mode runs them in the extended order with its explicit-Phibar gradient as it is.  Same pin as every other sampler: means and variances within 4 MCSE of the quadrature.
# ---------------------------------------------------------------------------------------------
def _guarded_target(P, engine, gradient, to_dev=None):
    T = (lambda a: torch.as_tensor(a, dtype=torch.float64)) if to_dev is None else to_dev
    cb = ggp_amd.CollapsedBound(T(P["X"]), T(P["y"]), jitter=float(P["jitter"]), engine=engine)
    cb.fused = False                   # not the single launch: the guarded multi-launch path a 10^6-row shard takes
    cb.whitened_rows_min_work = 0      # ... with its extended order and its streaming-layout whitened order at this size too
    return ggp_amd.HmcTarget(cb, T(P["Z"]), gradient=gradient)


@pytest.fixture()
def multi_launch_path():
    old = ggp_amd.CollapsedBound.WHITENED_MAX_WORK
    ggp_amd.CollapsedBound.WHITENED_MAX_WORK = 0
    yield
    ggp_amd.CollapsedBound.WHITENED_MAX_WORK = old


def test_sampler_gradient_mode_reproduces_the_exact_posterior_on_the_cpu_double(multi_launch_path):
    from fake_engine import FactoredOracleEngine
    P = load_golden("posterior_rbf_d1_tiny")
    tgt = _guarded_target(P, FactoredOracleEngine(), "sampler")
    tr = ggp_amd.sample_nuts(tgt, 1200, 400, seed=31)
    b = tgt.bound
    assert b.n_extended > 0.7 * b.n_grads, (b.n_extended, b.n_grads)       # the mode is what ran
    assert np.asarray(tr.get_sampler_stats("diverging")).mean() <= 0.01
    check_moments(unconstrained(tr), P, "sample_nuts, HmcTarget(gradient='sampler') / oracle double")


@pytest.mark.gpu
def test_sampler_gradient_mode_reproduces_the_exact_posterior_on_the_gpu(engine, multi_launch_path):
    """... on the HIP path (sgp_ctx_suffstats_fwd_extended + sgp_suffstats_bwd with the explicit Phibar), and against the default mode
    on the same problem and seed: acceptance rate within 10 %, adapted step size within 15 %."""
    P = load_golden("posterior_rbf_d1_tiny")
    D = lambda a: dev(a, engine)
    runs = {}
    for mode in ("sampler", "parity"):
        tgt = _guarded_target(P, engine, mode, D)
        tr = ggp_amd.sample_nuts(tgt, 2000, 1000, seed=41)
        b = tgt.bound
        runs[mode] = (tr, b.n_extended / max(1, b.n_grads), float(np.mean(tr.get_sampler_stats("mean_tree_accept"))),
                      float(np.asarray(tr.get_sampler_stats("step_size"))[-1]))
        assert np.asarray(tr.get_sampler_stats("diverging")).mean() <= 0.01
        check_moments(unconstrained(tr), P, "sample_nuts, HmcTarget(gradient=%r) / HIP multi-launch path" % mode)
    # (through round 5 the default mode sent ~85 % of these leapfrogs to the whitened order; since round 6 its gradients take the extended
    # order too -- double-double Phibar, trailing word in pass 2 -- wherever the trailing word's correction stays small: both modes mostly
    # run there now, the default one with the parity-grade gradient)
    assert runs["sampler"][1] > 0.7 and runs["parity"][1] > 0.5, (runs["sampler"][1], runs["parity"][1])
    acc_s, acc_p, eps_s, eps_p = runs["sampler"][2], runs["parity"][2], runs["sampler"][3], runs["parity"][3]
    # (acceptance within 10 %; adapted step sizes within 15 %: round 6 measured 0.594 against 0.537 with the default mode's gradients coming
    # from three routes -- extended with both words, whitened behind a rejected correction, whitened beyond the range)
    assert abs(acc_s - acc_p) < 0.1 * acc_p and abs(eps_s - eps_p) < 0.15 * eps_p, (acc_s, acc_p, eps_s, eps_p)
