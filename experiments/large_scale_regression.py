#!/usr/bin/env python3
"""BASELINE config C5 as an end-to-end run (call pattern of experiments/large_scale_regression_SGHMC.py:55-206:
construct -> train_model -> posterior_predictive -> metrics -> JSON), on synthetic N = 1M, d = 8, M = 1024 data
instead of UCI Elevators (absent, no network).  Optionally continues with a short fixed-Z NUTS run."""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402
from ggp_amd import BayesianSparseGPR_HMC, GaussianLikelihood, SparseGPR, nlpd_marginal, rmse  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--num_inducing", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--max_iters", type=int, default=50)
    ap.add_argument("--hmc_samples", type=int, default=0, help="fixed-Z NUTS draws after training (0 = skip)")
    ap.add_argument("--hmc_tune", type=int, default=10)
    ap.add_argument("--hmc_gradient", choices=["parity", "sampler"], default="parity",
                    help="core.HmcTarget: 'sampler' lets the extended evaluation order serve gradients as far as values (guarded regime)")
    args = ap.parse_args()

    g = torch.Generator().manual_seed(0)
    n_test = 10_000
    X = torch.randn(args.n + n_test, args.dim, dtype=torch.float64, generator=g)
    w = torch.randn(args.dim, dtype=torch.float64, generator=g) / math.sqrt(args.dim)
    y = torch.sin(X @ w) + 0.1 * torch.randn(args.n + n_test, dtype=torch.float64, generator=g)
    mu, sd = y[: args.n].mean(), y[: args.n].std()
    y = (y - mu) / sd
    dev = torch.device("cuda", 0)
    Xtr, ytr, Xte, yte = X[: args.n].to(dev), y[: args.n].to(dev), X[args.n:].to(dev), y[args.n:]
    Z0 = X[torch.randperm(args.n, generator=g)[: args.num_inducing]].clone()

    model = SparseGPR(Xtr, ytr, GaussianLikelihood(), Z0, jitter=1e-6)
    model.base_covar_module.base_kernel.lengthscale = 2.0
    opt = torch.optim.Adam(model.parameters(), lr=0.05)
    t0 = time.time()
    losses = model.train_model(opt, max_steps=args.max_iters, verbose=False)
    wall = time.time() - t0
    pred = model.posterior_predictive(Xte)
    ystd = torch.tensor([float(sd)])
    # the reference's result schema (experiments/regression.py:157-179: exp_info + metrics in one flat dict), then extras
    out = ggp_amd.experiment_tools.result_record(
        "Synthetic_N%d_d%d" % (args.n, args.dim), "SGPR", float(rmse(pred.loc, yte, ystd)), nlpd_marginal(pred, yte, ystd), wall,
        num_inducing=args.num_inducing, max_iter=args.max_iters, train_test_split=args.n / float(args.n + n_test),
        N=args.n, d=args.dim, secs_per_step=wall / max(1, args.max_iters), loss_first=losses[0], loss_last=losses[-1],
        nlpd_kind="marginal (utils/metrics.py:49-58; the joint form needs a 10 000 x 10 000 covariance)",
        guard_repeats=int(model._bound().n_guard_reruns))
    if args.hmc_samples > 0:
        hmc = BayesianSparseGPR_HMC(Xtr, ytr, GaussianLikelihood(), model.inducing_points.cpu(), jitter=1e-6, seed=1)
        hmc.hmc_gradient = args.hmc_gradient
        t0 = time.time()
        trace, steps, perf = hmc.train_fixed_model(num_tune=args.hmc_tune, num_samples=args.hmc_samples)
        wall = time.time() - t0
        out["step_sizes"] = [float(v) for v in steps]
        out["perf_times"] = [float(v) for v in perf]
        out["hmc"] = {"gradient_mode": args.hmc_gradient, "draws": len(trace), "tune": args.hmc_tune, "wall_clock_secs": wall, "n_leapfrog": int(trace.n_leapfrog),
                      "leapfrogs_per_sec": trace.n_leapfrog / wall, "step_size": float(steps[0]),
                      "sig_n_mean": float(trace["sig_n"].mean()), "ls_mean": trace["ls"].mean(0).tolist(),
                      # leapfrogs the streaming-order guard sent to the whitened (PyMC3) order
                      "guard_repeats": int(hmc._hmc_bound().n_guard_reruns), "direct_whitened": int(hmc._hmc_bound().n_direct_whitened),
                      "extended_order": int(getattr(hmc._hmc_bound(), "n_extended", 0)), "guard_tolerance_per_datum": hmc._hmc_bound().streaming_tol,
                      "trailing_word_rejections": int(getattr(hmc._hmc_bound(), "n_lo_rejections", 0)),
                      "trailing_word_last_correction": getattr(hmc._hmc_bound(), "last_lo_correction", None),
                      "guard_last_estimate_per_datum": hmc._hmc_bound().last_estimate, "evaluations": int(hmc._hmc_bound().n_evals)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
