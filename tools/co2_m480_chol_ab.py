#!/usr/bin/env python3
"""Does LAPACK fail where the HIP evaluation fails?  

Answer recorded in profiles/r03_co2_m480_chol_ab.json: NEITHER fails.  The seed whose 200 draws are all flagged never sees a failed
factorization -- its Adam warm start ran the medium-term amplitude to ~1.5e3 (lambda_max(K_uu) = 1e9, cond(K_uu + 1e-6 I) = 1e15),
where fp64 cannot evaluate the density: the bound changes by O(1e3) when theta moves by 1e-7, so the first leapfrog of every
trajectory has an energy error > 1000 and is flagged; the chain never moves.  The other seeds sample at cond ~ 2e10 with the HIP
and LAPACK values agreeing to 1e-5.  (The tool still captures failed factorizations should a build produce them.)

The CO2 NUTS stage at the reference's M = 480 (experiments/co2_bayesian_sgpr_hmc.py:384) flagged 200 of 200 draws as divergent
for one seed of three in round 2; DESIGN section 4a-3 blamed the MODEL at PyMC3's jitter 1e-6 (cond K_uu > 1e10), without a record
of the LAPACK-based oracle failing at the same theta.  This tool makes that record.  It runs experiments/co2_composite_hmc.py's
NUTS stage (host-driven sampler over the multi-launch whitened path -- M = 480 is above the single launch) for a list of seeds
and captures every theta at which the HIP evaluation reported a failed factorization (with its LAPACK-style index: 1..M = K_uu,
M+1..2M = B).  At each captured theta (first --max-points per seed) the CPU oracle then repeats PyMC3's op order with LAPACK:

    K_uu + 1e-6 I -> torch.linalg.cholesky_ex -> A = L^-1 K_uf -> B = I + A A^T / s2 -> cholesky_ex

and records cond(K_uu + 1e-6 I), the smallest eigenvalue of K_uu WITHOUT jitter, both info codes, and logp when it exists.
One JSON object on stdout (-> profiles/r03_co2_m480_chol_ab.json).
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "experiments"))
import ggp_amd  # noqa: E402
from co2_composite_hmc import synthetic_keeling  # noqa: E402
from oracle import composite_oracle as CO  # noqa: E402  (a checker here, never on the product path)


def lapack_at(theta, target, X, y, Z, jitter):
    vals = [math.exp(v) for v in theta[:-1]]
    s2 = math.exp(theta[-1]) ** 2
    blk = torch.as_tensor(np.asarray(target.kernel.with_values(vals).block(), dtype=np.float64))
    st = blk.numpy()
    M = Z.shape[0]
    K0 = CO.composite_k(Z, Z, blk, st)
    ev = torch.linalg.eigvalsh(K0)
    Kuu = K0 + jitter * torch.eye(M, dtype=torch.float64)
    evj = torch.linalg.eigvalsh(Kuu)
    L, info_k = torch.linalg.cholesky_ex(Kuu)
    rec = {"min_eig_Kuu_no_jitter": float(ev[0]), "max_eig_Kuu": float(ev[-1]), "cond_Kuu_plus_jitter": float(evj[-1] / max(float(evj[0]), 1e-300)),
           "min_eig_Kuu_plus_jitter": float(evj[0]), "lapack_info_Kuu": int(info_k), "sigma2": s2}
    if int(info_k) == 0:
        A = torch.linalg.solve_triangular(L, CO.composite_k(Z, X, blk, st), upper=False)
        Bm = torch.eye(M, dtype=torch.float64) + (A / s2) @ A.T
        _, info_b = torch.linalg.cholesky_ex(Bm)
        rec["lapack_info_B"] = int(info_b)
        rec["min_eig_B"] = float(torch.linalg.eigvalsh(Bm)[0])
        if int(info_b) == 0:
            rec["oracle_F"] = float(CO.vfe_composite(X, y, Z, blk, s2, jitter, structure=st))
    return rec


def run_seed(seed, args, eng):
    y_tr, t_tr, _, _, _ = synthetic_keeling(seed=seed)
    X = torch.as_tensor(t_tr, dtype=torch.float64).to(eng.device)
    y = torch.as_tensor(y_tr, dtype=torch.float64).to(eng.device)
    M = args.num_inducing
    Z = X[torch.linspace(0, X.shape[0] - 1, M).round().long()].clone()
    bound = ggp_amd.CollapsedBound(X, y, kernel="composite", jitter=args.jitter, engine=eng)
    target = ggp_amd.CompositeHmcTarget(bound, Z, ggp_amd.co2_kernel(), ggp_amd.CO2_LOG_PRIOR_SD)
    failed, last_info = [], {}
    inner = bound.value_and_grad

    def spy(*a, **k):
        F, g = inner(*a, **k)
        last_info["info"] = int(g.get("info", 0))
        return F, g

    bound.value_and_grad = spy
    raw = target.logp_and_grad
    n_calls = [0]

    def logp_and_grad(theta):
        last_info["info"] = 0
        lp, g = raw(theta)
        n_calls[0] += 1
        if not math.isfinite(lp) and all(abs(float(v)) < 100.0 for v in theta):
            failed.append(([float(v) for v in theta], last_info.get("info", 0)))
        return lp, g

    target.logp_and_grad = logp_and_grad
    theta = list(target.start())
    m1, m2 = [0.0] * len(theta), [0.0] * len(theta)
    lp_map = float("-inf")
    for it in range(1, args.map_steps + 1):  # the experiment's Adam warm start on the log posterior
        lp, gth = target.logp_and_grad(theta)
        if not math.isfinite(lp):
            break
        lp_map = lp
        for k in range(len(theta)):
            m1[k] = 0.9 * m1[k] + 0.1 * gth[k]
            m2[k] = 0.999 * m2[k] + 0.001 * gth[k] * gth[k]
            theta[k] += 0.05 * (m1[k] / (1 - 0.9 ** it)) / (math.sqrt(m2[k] / (1 - 0.999 ** it)) + 1e-8)
    n_map_fail = len(failed)
    trace = ggp_amd.sample_nuts(target, n_samples=args.num_samples, tune=args.tune, seed=seed, start=theta, max_treedepth=6)
    Xc, yc, Zc = X.cpu(), y.cpu(), Z.cpu()
    points = []
    for th, info in failed[: args.max_points]:
        rec = {"theta": th, "hip_info": info, "hip_failed_matrix": "Kuu" if 0 < info <= M else ("B" if info > M else "non-finite F")}
        rec.update(lapack_at(th, target, Xc, yc, Zc, args.jitter))
        points.append(rec)
    # control: the same LAPACK factorizations at thetas the HIP path evaluated fine (the post-tuning draws)
    control = []
    for row in trace[:: max(1, len(trace) // 8)]:
        th = [float(v) for v in row["theta_unc"]]
        rec = lapack_at(th, target, Xc, yc, Zc, args.jitter)
        # the same bound F (no priors) from the HIP path at the same theta, and at theta moved by 1e-7 in every coordinate: the
        # second difference against the analytic gradient's prediction is the evaluation's own noise floor
        vals = [math.exp(v) for v in th[:-1]]
        F, g = inner(Z, target.kernel.with_values(vals).block(), 1.0, math.exp(th[-1]) ** 2, raise_on_fail=False)
        vals2 = [math.exp(v + 1e-7) for v in th[:-1]]
        F2, _ = inner(Z, target.kernel.with_values(vals2).block(), 1.0, math.exp(th[-1] + 1e-7) ** 2, raise_on_fail=False)
        th2 = [v + 1e-7 for v in th]
        rec2 = lapack_at(th2, target, Xc, yc, Zc, args.jitter)
        rec["oracle_F_at_theta_plus_1e-7"] = rec2.get("oracle_F")
        rec.update({"theta": th, "hip_F": F, "hip_F_at_theta_plus_1e-7": F2,
                    "hip_minus_lapack_F": (F - rec["oracle_F"]) if "oracle_F" in rec else None})
        control.append(rec)
    stats = {k: np.asarray(trace.get_sampler_stats(k)) for k in ("step_size", "tree_size", "energy")}
    both = sum(1 for p in points if p["lapack_info_Kuu"] != 0 or p.get("lapack_info_B", 0) != 0)
    return {"seed": seed, "logp_after_map": lp_map, "evaluations": n_calls[0], "hip_failures": len(failed), "hip_failures_during_map": n_map_fail,
            "diverging_draws": int(trace.get_sampler_stats("diverging").sum()), "num_samples": len(trace),
            "mean_step_size": float(trace.get_sampler_stats("step_size").mean()), "n_leapfrog": int(trace.n_leapfrog),
            "points_checked": len(points), "lapack_fails_too": both,
            "lapack_fails_at_Kuu": sum(1 for p in points if p["lapack_info_Kuu"] != 0),
            "lapack_fails_at_B": sum(1 for p in points if p.get("lapack_info_B", 0) != 0),
            "hip_fails_at_Kuu": sum(1 for p in points if p["hip_failed_matrix"] == "Kuu"),
            "hip_fails_at_B": sum(1 for p in points if p["hip_failed_matrix"] == "B"),
            "median_cond_at_failures": float(np.median([p["cond_Kuu_plus_jitter"] for p in points])) if points else None,
            "median_cond_at_accepted_draws": float(np.median([c["cond_Kuu_plus_jitter"] for c in control])),
            "lapack_failures_at_accepted_draws": sum(1 for c in control if c["lapack_info_Kuu"] != 0 or c.get("lapack_info_B", 0) != 0),
            "median_abs_hip_minus_lapack_F_at_accepted_draws": float(np.median([abs(c["hip_minus_lapack_F"]) for c in control if c["hip_minus_lapack_F"] is not None]))
            if any(c["hip_minus_lapack_F"] is not None for c in control) else None,
            "median_abs_F_change_for_1e-7_step": float(np.median([abs(c["hip_F_at_theta_plus_1e-7"] - c["hip_F"]) for c in control])),
            "median_abs_LAPACK_F_change_for_1e-7_step": float(np.median([abs(c["oracle_F_at_theta_plus_1e-7"] - c["oracle_F"]) for c in control
                                                                         if c.get("oracle_F") is not None and c.get("oracle_F_at_theta_plus_1e-7") is not None] or [float("nan")])),
            "mean_tree_size": float(stats["tree_size"].mean()), "energy_sd_over_draws": float(stats["energy"].std()),
            "accepted_draws_checked": control[:3],
            "points": points[: args.keep_points]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="47,48,49")
    ap.add_argument("--num_inducing", type=int, default=480)
    ap.add_argument("--num_samples", type=int, default=200)
    ap.add_argument("--tune", type=int, default=150)
    ap.add_argument("--map_steps", type=int, default=400)
    ap.add_argument("--jitter", type=float, default=1e-6)
    ap.add_argument("--max-points", type=int, default=40)
    ap.add_argument("--keep-points", type=int, default=6, help="per-point records kept in the output (all are counted)")
    args = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    eng = ggp_amd.HipEngine()
    runs = [run_seed(int(s), args, eng) for s in args.seeds.split(",")]
    print(json.dumps({"note": "HIP multi-launch whitened path vs LAPACK (oracle, PyMC3 op order) at the thetas where the HIP evaluation reported a "
                              "failed factorization; CO2 composite kernel, N = 634, M = %d, jitter %g" % (args.num_inducing, args.jitter),
                      "runs": runs}))


if __name__ == "__main__":
    main()
