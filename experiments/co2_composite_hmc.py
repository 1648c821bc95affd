#!/usr/bin/env python3
"""BASELINE config C2 -- the NUTS stage of the reference's CO2 experiment (experiments/co2_bayesian_sgpr_hmc.py:99-158,
341-399) on the HIP core, with the reference's own covariance

    n_per^2 Periodic(period=1) * ExpQuad + n_med^2 RatQuad + n_trend^2 ExpQuad + n_noise^2 Matern32,   sigma ~ HalfNormal(1)

and its Normal priors on the log-parameters.  ``--mauna PATH`` reads the real ``mauna.txt`` (not shipped: no network);
without it a synthetic Keeling-like series of the same size (N = 634 monthly points, 60 test months) stands in.
Inducing inputs: every (N / M)-th training time (the reference hands over Z from its optimisation stage).
Prints one JSON object.
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


def synthetic_keeling(n_total=694, seed=47):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n_total, dtype=torch.float64) / 12.0
    co2 = 315.0 + 0.8 * t + 0.012 * t * t + 3.0 * torch.sin(2 * math.pi * t) + 0.8 * torch.sin(4 * math.pi * t + 0.5) \
        + 0.3 * torch.randn(n_total, dtype=torch.float64, generator=g)
    std = float(co2.std(unbiased=False))
    y = (co2 - co2[0]) / std
    return y[:634].numpy(), t[:634, None].numpy(), y[634:694].numpy(), t[634:694, None].numpy(), std


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mauna", default=None, help="path to mauna.txt (year co2, -99.99 = missing)")
    ap.add_argument("--num_inducing", type=int, default=64)
    ap.add_argument("--num_samples", type=int, default=200)
    ap.add_argument("--tune", type=int, default=150)
    ap.add_argument("--max_treedepth", type=int, default=6, help="PyMC3 default is 10; the demo caps the work per draw")
    ap.add_argument("--seed", type=int, default=47)
    ap.add_argument("--sampler", choices=("auto", "device", "host"), default="auto",
                    help="device: the whole NUTS run in one persistent launch (sgp_small_nuts_composite; M <= 128); host: the tree "
                         "logic in Python, one launch per leapfrog; auto: device when the target supports it")
    ap.add_argument("--map_steps", type=int, default=400, help="Adam steps on the log posterior before sampling (0 = start at the prior mean)")
    ap.add_argument("--map_lr", type=float, default=0.05)
    ap.add_argument("--jitter", type=float, default=1e-6,
                    help="added to diag(Kuu): PyMC3's stabilize() value.  (cond(Kuu) reaches 4e8 here; the whitened evaluation\n"
                         "order CollapsedBound picks for this size keeps logp to 1e-8, profiles/r02_logp_noise.json)")
    args = ap.parse_args()

    if args.mauna:
        y_tr, t_tr, y_te, t_te, std = ggp_amd.datasets.load_co2_dataset(args.mauna, 2010)
        data = "mauna.txt"
    else:
        y_tr, t_tr, y_te, t_te, std = synthetic_keeling(seed=args.seed)
        data = "synthetic Keeling-like series"
    eng = ggp_amd.HipEngine()
    dev = eng.device
    X = torch.as_tensor(t_tr, dtype=torch.float64).to(dev)
    y = torch.as_tensor(y_tr, dtype=torch.float64).to(dev)
    Xt = torch.as_tensor(t_te, dtype=torch.float64).to(dev)
    M = args.num_inducing
    Z = X[torch.linspace(0, X.shape[0] - 1, M).round().long()].clone()

    bound = ggp_amd.CollapsedBound(X, y, kernel="composite", jitter=args.jitter, engine=eng)
    target = ggp_amd.CompositeHmcTarget(bound, Z, ggp_amd.co2_kernel(), ggp_amd.CO2_LOG_PRIOR_SD)
    # Warm start, as the reference does before it begins to sample (hmc_scheduler, co2_bayesian_sgpr_hmc.py:205): a
    # few hundred Adam steps on the log posterior from PyMC3's test point; NUTS then starts near the mode instead of
    # adapting its step size in the far tails of an 11-dimensional, badly scaled density.
    theta = list(target.start())
    t_map = time.time()
    m1 = [0.0] * len(theta)
    m2 = [0.0] * len(theta)
    lp_map = float("-inf")
    for it in range(1, args.map_steps + 1):
        lp, gth = target.logp_and_grad(theta)
        if not math.isfinite(lp):
            break
        lp_map = lp
        for k in range(len(theta)):
            m1[k] = 0.9 * m1[k] + 0.1 * gth[k]
            m2[k] = 0.999 * m2[k] + 0.001 * gth[k] * gth[k]
            theta[k] += args.map_lr * (m1[k] / (1 - 0.9 ** it)) / (math.sqrt(m2[k] / (1 - 0.999 ** it)) + 1e-8)
    # The warm start follows gradients; where cond(K_uu + jitter I) reaches 1e15 (an amplitude run to 1e3: lambda_max = 1e9
    # against the 1e-6 jitter) fp64 no longer evaluates the density -- logp moves by O(1e3) for a step of 1e-7 -- and Adam can
    # "climb" that noise (seed 47 at M = 480, profiles/r03_co2_m480_chol_ab.json: every draw then flags an energy error).  Probe the
    # end point; if logp is not smooth there, restart the warm start with every log-parameter kept within 2 prior sd.
    def smooth(th):
        a = target.logp_and_grad(th)[0]
        b = target.logp_and_grad([v + 1e-7 for v in th])[0]
        return math.isfinite(a) and math.isfinite(b) and abs(a - b) < 1.0
    map_guard = "not needed"
    if args.map_steps > 0 and not smooth(theta):
        bound_sd = [2.0 * sd for sd in target.sd] + [3.0]
        theta = list(target.start())
        m1 = [0.0] * len(theta)
        m2 = [0.0] * len(theta)
        for it in range(1, args.map_steps + 1):
            lp, gth = target.logp_and_grad(theta)
            if not math.isfinite(lp):
                break
            lp_map = lp
            for k in range(len(theta)):
                m1[k] = 0.9 * m1[k] + 0.1 * gth[k]
                m2[k] = 0.999 * m2[k] + 0.001 * gth[k] * gth[k]
                theta[k] += args.map_lr * (m1[k] / (1 - 0.9 ** it)) / (math.sqrt(m2[k] / (1 - 0.999 ** it)) + 1e-8)
                theta[k] = max(-bound_sd[k], min(bound_sd[k], theta[k]))
        map_guard = "restarted inside +-2 prior sd: " + ("smooth" if smooth(theta) else "STILL NOT SMOOTH")
    map_secs = time.time() - t_map
    t0 = time.time()
    on_device = args.sampler == "device" or (args.sampler == "auto" and target.device_sampler_ok())
    sample = ggp_amd.sample_nuts_device if on_device else ggp_amd.sample_nuts
    trace = sample(target, n_samples=args.num_samples, tune=args.tune, seed=args.seed, start=theta, max_treedepth=args.max_treedepth)
    wall = time.time() - t0

    # mixture predictive over the draws (models/bayesian_sgpr_hmc.py:198-231): per-draw mean / variance on the test months
    means, variances = [], []
    for row in trace[:: max(1, len(trace) // 20)]:
        c = target.constrain(row["theta_unc"])
        mu, var, _ = bound.predict(Xt, Z, c["kernel"].block(), 1.0, c["sig_n"] ** 2)
        means.append(mu.cpu().numpy())
        variances.append(var.cpu().numpy())
    means, variances = np.stack(means), np.stack(variances)
    mix_mean = means.mean(0)
    rmse = float(np.sqrt(np.mean((mix_mean - y_te) ** 2)) * std)
    variances = np.maximum(variances, 1e-10)  # (a draw with amplitudes ~1e3 loses the 1e-4 noise floor to rounding; GPyTorch clamps too)
    logp = -0.5 * np.log(2 * np.pi * variances) - 0.5 * (y_te[None, :] - means) ** 2 / variances
    nlpd = float(-np.mean(np.log(np.mean(np.exp(logp), 0))) + math.log(std))
    names = [n for n, _, _ in target.params] + ["sigma"]
    post = np.concatenate([trace["ls"], trace["sig_n"][:, None]], 1)
    out = {"config": "C2 CO2, composite covariance, NUTS", "data": data, "N_train": int(X.shape[0]), "num_inducing": M, "jitter": args.jitter,
           "map_steps": args.map_steps, "map_secs": map_secs, "map_guard": map_guard, "logp_after_map": lp_map, "sampler": "device-resident (one persistent launch)" if on_device else "host-driven (one launch per leapfrog)",
           "single_launch_evaluations": bool(bound._small_ok(M)), "num_samples": len(trace), "tune": args.tune, "max_treedepth": args.max_treedepth, "wall_clock_secs": wall, "n_leapfrog": int(trace.n_leapfrog),
           "leapfrogs_per_s": trace.n_leapfrog / wall, "mean_step_size": float(trace.get_sampler_stats("step_size").mean()),
           "diverging": int(trace.get_sampler_stats("diverging").sum()),
           "posterior_mean": {n: float(v) for n, v in zip(names, post.mean(0))},
           "test_rmse_ppm": rmse, "test_nlpd": nlpd}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
