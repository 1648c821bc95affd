#!/usr/bin/env python3
"""BASELINE config C1 -- the reference's 1-D demo (experiments/demo_1d_regression.py:55-139) on the HIP core.

Same data recipe (seeded torch RNG, train on |x| > 2, test grid linspace(-8, 8, 1000), Z_init = randn(25)),
same two models (SparseGPR with 2000 Adam steps at lr 0.01; BayesianSparseGPR_HMC with the
[100, 200, 500, 1000, 1500, 1999] HMC schedule), same metrics.  The GPflow "JointHMC" third panel and the
plots are out of scope.  Prints one JSON object (keys follow experiments/regression.py:157-179).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402
from ggp_amd import BayesianSparseGPR_HMC, GaussianLikelihood, SparseGPR, mixture_posterior_predictive  # noqa: E402
from ggp_amd import nlpd, nlpd_mixture, rmse  # noqa: E402


def func(x):
    return torch.sin(x * 3) + 0.3 * torch.cos(x * 3.14)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max_iters", type=int, default=2000)
    ap.add_argument("--num_inducing", type=int, default=25)
    ap.add_argument("--skip_hmc", action="store_true")
    args = ap.parse_args()

    torch.manual_seed(45)
    N = 1000
    X = torch.randn(N) * 2 - 1
    Y = func(X) + 0.4 * torch.randn(N)
    idx = (X < -2) | (X > 2)
    dev = torch.device("cuda", 0)
    X_train, Y_train = X[idx][:, None].double().to(dev), Y[idx].double().to(dev)
    X_test = torch.linspace(-8, 8, 1000).double()
    Y_test = func(X_test)
    Z_init = torch.randn(args.num_inducing).double()
    ystd = torch.tensor([1.0])
    out = {"N_train": int(idx.sum()), "num_inducing": args.num_inducing, "max_iters": args.max_iters}

    model = SparseGPR(X_train, Y_train, GaussianLikelihood(), Z_init, jitter=1e-6)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    t0 = time.time()
    losses = model.train_model(opt, max_steps=args.max_iters, verbose=False)
    wall = time.time() - t0
    pred = model.posterior_predictive(X_test.to(dev))
    # one record per model in the reference's result schema (experiments/regression.py:157-179)
    out["SparseGPR"] = ggp_amd.experiment_tools.result_record(
        "demo_1d", "SGPR", float(rmse(pred.loc, Y_test, ystd)), float(nlpd(pred, Y_test, ystd)), wall, num_inducing=args.num_inducing,
        max_iter=args.max_iters, final_loss=losses[-1],
        lengthscale=model.base_covar_module.base_kernel.lengthscale.detach().cpu().reshape(-1).tolist(),
        outputscale=float(model.base_covar_module.outputscale.detach()), noise=float(model.likelihood.noise.detach()))

    if not args.skip_hmc:
        hmc = BayesianSparseGPR_HMC(X_train, Y_train, GaussianLikelihood(), Z_init, jitter=1e-6, seed=45)
        opt = torch.optim.Adam(hmc.parameters(), lr=0.01)
        sched = [s for s in (100, 200, 500, 1000, 1500, 1999) if s < args.max_iters] or [args.max_iters - 1]
        t0 = time.time()
        losses, trace, step_sizes, perf = hmc.train_model(opt, max_steps=args.max_iters, hmc_scheduler=sched, verbose=False)
        wall = time.time() - t0
        preds = mixture_posterior_predictive(hmc, X_test.to(dev), trace)
        means = torch.stack([p.loc.cpu() for p in preds]).mean(0)
        out["BayesianSGPR_HMC"] = ggp_amd.experiment_tools.result_record(
            "demo_1d", "Bayesian_SGPR_HMC", float(rmse(means, Y_test, ystd)), float(nlpd_mixture(preds, Y_test, ystd)), wall,
            perf_times=perf, step_sizes=step_sizes, num_inducing=args.num_inducing, max_iter=args.max_iters, n_mixture=len(preds),
            ls_mean=float(np.mean(trace["ls"])), sig_n_mean=float(np.mean(trace["sig_n"])),
            leapfrogs_last_phase=int(trace.n_leapfrog), sampler_on_device=bool(getattr(trace, "device_resident", False)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
