#!/usr/bin/env python3
"""BASELINE config C2 end to end -- the reference's experiments/co2_bayesian_sgpr_hmc.py main (:341-399) on the HIP core:
``CompositeBayesianSparseGPR_HMC`` (sum-of-products covariance), Adam warm start on every parameter, then the alternating
schedule (NUTS phases at ``--hmc_scheduler``, Adam on the inducing inputs against the bound averaged over the trace), the
mixture predictive over the final trace, RMSE / NLPD in ppm, and one result record with the reference's JSON keys.
``--mauna PATH`` reads the real series; without it the synthetic Keeling-like series of experiments/co2_composite_hmc.py.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402
from co2_composite_hmc import synthetic_keeling  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mauna", default=None)
    ap.add_argument("--num_inducing", type=int, default=64)
    ap.add_argument("--max_steps", type=int, default=1600)
    ap.add_argument("--hmc_scheduler", type=int, nargs="+", default=[400, 800, 1200, 1500])
    ap.add_argument("--lr", type=float, default=0.02)
    ap.add_argument("--seed", type=int, default=47)
    args = ap.parse_args()
    if args.mauna:
        y_tr, t_tr, y_te, t_te, std = ggp_amd.datasets.load_co2_dataset(args.mauna, 2010)
    else:
        y_tr, t_tr, y_te, t_te, std = synthetic_keeling(seed=args.seed)
    eng = ggp_amd.HipEngine()
    X = torch.as_tensor(t_tr, dtype=torch.float64).to(eng.device)
    y = torch.as_tensor(y_tr, dtype=torch.float64).to(eng.device)
    Xt = torch.as_tensor(t_te, dtype=torch.float64).to(eng.device)
    M = args.num_inducing
    Z0 = X[torch.linspace(0, X.shape[0] - 1, M).round().long()].clone()
    model = ggp_amd.CompositeBayesianSparseGPR_HMC(X, y, ggp_amd.co2_kernel(), Z0, ggp_amd.CO2_LOG_PRIOR_SD, engine=eng, seed=args.seed)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    t0 = time.time()
    losses, trace, step_sizes, perf_times = model.train_model(opt, max_steps=args.max_steps, hmc_scheduler=args.hmc_scheduler)
    wall = time.time() - t0
    preds = model.mixture_posterior_predictive(Xt, trace)
    means = np.stack([m.cpu().numpy() for m, _ in preds])
    variances = np.stack([v.cpu().numpy() for _, v in preds])
    rmse = float(np.sqrt(np.mean((means.mean(0) - y_te) ** 2)) * std)
    variances = np.maximum(variances, 1e-10)  # (a draw with amplitudes ~1e3 loses the 1e-4 noise floor to rounding; GPyTorch clamps too)
    logp = -0.5 * np.log(2 * np.pi * variances) - 0.5 * (y_te[None, :] - means) ** 2 / variances
    nlpd = float(-np.mean(np.log(np.mean(np.exp(logp), 0))) + math.log(std))
    rec = ggp_amd.experiment_tools.result_record(
        "mauna" if args.mauna else "synthetic_keeling", "Bayesian_SGPR_HMC", rmse, nlpd, wall, perf_times=perf_times,
        step_sizes=step_sizes, num_inducing=M, max_iter=args.max_steps, hmc_scheduler=list(args.hmc_scheduler),
        n_train=int(X.shape[0]), warm_start_loss=[losses[0], losses[args.hmc_scheduler[0] - 1]], final_loss=losses[-1],
        diverging_last_trace=int(trace.get_sampler_stats("diverging").sum()), leapfrogs_last_trace=int(trace.n_leapfrog),
        device_resident_sampler=bool(getattr(trace, "device_resident", False)),
        posterior_mean={n: float(v) for (n, _, _), v in zip(model.params, trace["ls"].mean(0))} | {"sigma": float(trace["sig_n"].mean())})
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
