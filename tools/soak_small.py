#!/usr/bin/env python3
"""Soak test of the cooperative single-launch path and the persistent sampler (GPU): random shapes, repeated evaluations that
must be bit-identical with a clean status word, device-resident NUTS runs that must finish every draw (a lost flag or a missed
fence shows up as SGP_INFO_TIMEOUT or as a differing bit).      python3 tools/soak_small.py [seconds, default 120]"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    eng = ggp_amd.HipEngine()
    dev = eng.device
    rng = np.random.default_rng(7)
    g = torch.Generator().manual_seed(7)
    t0 = time.time()
    stats = {"evals": 0, "eval_mismatch": 0, "eval_bad_info": 0, "nuts_runs": 0, "nuts_leapfrogs": 0, "nuts_bad": 0, "timeouts": 0,
             "shapes": 0, "composite_shapes": 0}
    while time.time() - t0 < budget:
        comp = rng.random() < 0.3
        d = 1 if comp else int(rng.integers(1, 9))
        N = int(rng.choice([1, 17, 64, 65, 382, 634, 1300, 4097, 9000]))
        M = int(min(N, rng.choice([1, 7, 25, 50, 64, 65, 100, 128])))
        X = torch.rand(N, d, dtype=torch.float64, generator=g) * (20.0 if comp else 4.0)
        y = torch.sin(X.sum(1)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
        Z = X[torch.randperm(N, generator=g)[:M]].clone().to(dev)
        Xd, yd = X.to(dev), y.to(dev)
        stats["shapes"] += 1
        stats["composite_shapes"] += int(comp)
        try:
            if comp:
                cb = ggp_amd.CollapsedBound(Xd, yd, kernel="composite", jitter=1e-6, engine=eng)
                tgt = ggp_amd.CompositeHmcTarget(cb, Z, ggp_amd.co2_kernel(), ggp_amd.CO2_LOG_PRIOR_SD)
            else:
                cb = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=eng)
                tgt = ggp_amd.HmcTarget(cb, Z)
            th = (np.asarray(tgt.start()) + 0.3 * rng.standard_normal(tgt.ndim)).tolist()
            ref = None
            for k in range(40):
                lp, gr = tgt.logp_and_grad(th)
                stats["evals"] += 1
                cur = (lp, tuple(gr))
                if ref is None:
                    ref = cur
                elif cur != ref and not (math.isnan(lp) and math.isnan(ref[0])):
                    stats["eval_mismatch"] += 1
                    print("mismatch", (N, d, M, comp), k, lp, ref[0], flush=True)
            if math.isfinite(ref[0]) and tgt.device_sampler_ok():
                tr = ggp_amd.sample_nuts_device(tgt, 20, 30, seed=int(rng.integers(1, 1 << 30)), start=th, max_treedepth=6)
                stats["nuts_runs"] += 1
                stats["nuts_leapfrogs"] += int(tr.n_leapfrog)
                if len(tr) != 20 or not np.all(np.isfinite(tr.get_sampler_stats("logp"))):
                    stats["nuts_bad"] += 1
                    print("bad nuts run", (N, d, M, comp), flush=True)
        except ggp_amd.SgpTimeoutError:
            stats["timeouts"] += 1
            print("TIMEOUT", (N, d, M, comp), flush=True)
    stats["seconds"] = time.time() - t0
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
