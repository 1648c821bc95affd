#!/usr/bin/env python3
"""Minibatch steps per second of BayesianStochasticVariationalGP at the C4 shape (N = 100k, d = 2, M = 256, B = 4096, five
hyper-samples per minibatch, one backward, one Adam step): the five bounds in ONE launch chain (sgp_svgp_elbo_batch, the
default) against five sgp_svgp_elbo chains, Gaussian and Bernoulli-probit (BASELINE C4 as named), with the host-thread cap the
training loops apply (core.few_host_threads) and, for the record, with torch's default pool."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402


def main():
    eng = ggp_amd.HipEngine()
    g = torch.Generator().manual_seed(4)
    N, M, B = 100_000, 256, 4096
    X = torch.randn(N, 2, dtype=torch.float64, generator=g)
    y = torch.sin(2 * X[:, 0]) * torch.cos(X[:, 1]) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z0 = X[torch.randperm(N, generator=g)[:M]].clone()
    Xd, yd = X.to(eng.device), y.to(eng.device)
    batches = [(Xd[i:i + B], yd[i:i + B]) for i in range(0, 16 * B, B)]
    default_threads = torch.get_num_threads()
    yc = (y > 0).to(torch.float64).to(eng.device)
    cases = [("bernoulli", True, True), ("bernoulli", False, True), ("gaussian", True, True), ("gaussian", False, True)]
    uncapped = [("gaussian", True, False)]  # LAST: leaving torch's default pool active slows whatever is measured next (3.4 vs 0.85 ms)

    def bayesian_case(lik, batched, capped):
        torch.manual_seed(0)
        like = ggp_amd.BernoulliLikelihood() if lik == "bernoulli" else ggp_amd.GaussianLikelihood()
        yy = yc if lik == "bernoulli" else yd
        model = ggp_amd.BayesianStochasticVariationalGP(Xd, yy, like, Z0, engine=eng, seed=3)
        model.batched = batched
        bt = [(Xd[i:i + B], yy[i:i + B]) for i in range(0, 16 * B, B)]
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        train = model.train_model if capped else (lambda *a, **k: type(model).train_model.__wrapped__(model, *a, **k))
        train(opt, bt[:4], num_epochs=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, bl = train(opt, bt, num_epochs=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": "C4 BayesianSVGP minibatch step (N 100k, M 256, B 4096, 5 hyper-samples)", "likelihood": lik,
                          "hyper_samples": "one launch chain (sgp_svgp_elbo_batch)" if batched else "five sgp_svgp_elbo chains",
                          "host_threads": "capped (4)" if capped else "torch default (%d)" % default_threads,
                          "steps_per_s": 2 * len(bt) / dt, "ms_per_step": dt / (2 * len(bt)) * 1e3, "last_batch_loss": bl[-1]}), flush=True)
    for c in cases:
        bayesian_case(*c)
    # the plain (non-Bayesian) SVGP of models/svgp.py at the same shape: one bound + gradient per minibatch step
    for lik in ("bernoulli", "gaussian"):
        like = ggp_amd.BernoulliLikelihood() if lik == "bernoulli" else ggp_amd.GaussianLikelihood()
        yy = yc if lik == "bernoulli" else yd
        model = ggp_amd.StochasticVariationalGP(Xd, yy, like, Z0, engine=eng)
        bt = [(Xd[i:i + B], yy[i:i + B]) for i in range(0, 16 * B, B)]
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        model.train_model(opt, bt[:4], num_epochs=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = model.train_model(opt, bt, num_epochs=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": "C4-shaped StochasticVariationalGP minibatch step (one bound + gradient)", "likelihood": lik,
                          "steps_per_s": 2 * len(bt) / dt, "ms_per_step": dt / (2 * len(bt)) * 1e3, "last_loss": losses[-1]}), flush=True)
    # device time of the chain alone: 50 back-to-back bound + gradient calls for 5 samples, no host work in between
    model = ggp_amd.BayesianStochasticVariationalGP(Xd, yc, ggp_amd.BernoulliLikelihood(), Z0, engine=eng, seed=3)
    xb, yb = Xd[:B].contiguous(), torch.where(yc[:B] > 0, 1.0, -1.0).to(torch.float64)
    th = [[1.0, 1.0]] * 5
    args = (xb, yb, model.inducing_inputs.detach(), th, [1.0] * 5, [1.0] * 5, model.variational_mean.detach(), model.chol_variational_covar.detach(), N)
    for _ in range(3):
        eng.svgp_elbo_batch(*args, likelihood="bernoulli", with_grads=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eng.svgp_elbo_batch(*args, likelihood="bernoulli", with_grads=True)
    torch.cuda.synchronize()
    print(json.dumps({"config": "sgp_svgp_elbo_batch alone, S = 5, bound + gradients, back to back", "ms_per_call": (time.perf_counter() - t0) / 50 * 1e3}), flush=True)
    for c in uncapped:
        bayesian_case(*c)


def profile_step():
    """Where the host time of a step goes (cProfile over 32 steps of the batched Bernoulli model) -- stderr."""
    import cProfile
    import pstats
    eng = ggp_amd.HipEngine()
    g = torch.Generator().manual_seed(4)
    N, M, B = 100_000, 256, 4096
    X = torch.randn(N, 2, dtype=torch.float64, generator=g)
    y = (torch.sin(2 * X[:, 0]) * torch.cos(X[:, 1]) > 0).to(torch.float64)
    Z0 = X[torch.randperm(N, generator=g)[:M]].clone()
    Xd, yd = X.to(eng.device), y.to(eng.device)
    model = ggp_amd.BayesianStochasticVariationalGP(Xd, yd, ggp_amd.BernoulliLikelihood(), Z0, engine=eng, seed=3)
    bt = [(Xd[i:i + B], yd[i:i + B]) for i in range(0, 16 * B, B)]
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    model.train_model(opt, bt[:4], num_epochs=1)
    pr = cProfile.Profile()
    pr.enable()
    model.train_model(opt, bt, num_epochs=2)
    pr.disable()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    if "--profile" in sys.argv:
        profile_step()
        sys.exit(0)
    main()
