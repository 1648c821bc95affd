#!/usr/bin/env python3
"""Leapfrogs per second of the three ways to run NUTS on the small configs (C1 demo-1D, C2 CO2-shaped, the reference's
N = 1300 / M = 100 UCI shape): host-driven sampler over the multi-launch path (round 1), host-driven over the single
launch, and the device-resident sampler.  Prints one JSON line per configuration."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402


def main():
    eng = ggp_amd.HipEngine()
    for name, N, d, M in (("C1 demo-1D", 500, 1, 50), ("C2 CO2-shaped RBF", 634, 1, 128), ("reference UCI shape", 1300, 8, 100)):
        g = torch.Generator().manual_seed(0)
        X = torch.randn(N, d, dtype=torch.float64, generator=g)
        y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
        y = (y - y.mean()) / y.std()
        Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
        X, y = X.to(eng.device), y.to(eng.device)
        res = {"config": name, "N": N, "d": d, "M": M, "tune": 100, "draws": 100}
        for label, fused, device in (("host_sampler_multi_launch", False, False), ("host_sampler_single_launch", True, False),
                                     ("device_resident_sampler", True, True)):
            cb = ggp_amd.CollapsedBound(X, y, jitter=1e-6, engine=eng)
            cb.fused = fused
            tgt = ggp_amd.HmcTarget(cb, Z)
            fn = ggp_amd.sample_nuts_device if device else ggp_amd.sample_nuts
            fn(tgt, 5, 5, seed=1)  # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr = fn(tgt, 100, 100, seed=2)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            res[label] = {"leapfrogs": int(tr.n_leapfrog), "wall_s": round(wall, 4), "leapfrogs_per_s": round(tr.n_leapfrog / wall, 1),
                          "us_per_leapfrog": round(wall / tr.n_leapfrog * 1e6, 1), "mean_step": float(tr.get_sampler_stats("step_size").mean()),
                          "divergent": int(tr.get_sampler_stats("diverging").sum())}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
