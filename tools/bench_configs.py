#!/usr/bin/env python3
"""Evaluations / s of the collapsed bound at the smaller BASELINE.json configs (C1-C3), one GPU.
Diagnostic companion of bench.py (which measures C5); prints one JSON line per config."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

CONFIGS = [("C1 demo-1D", 500, 1, 50), ("C1 demo-1D (reference M)", 382, 1, 25), ("C2 CO2", 634, 1, 128),
           ("C3 elevators", 13279, 18, 512), ("C3 elevators (reference M)", 13279, 18, 100)]


def main():
    eng = ggp_amd.HipEngine()
    for name, N, d, M in CONFIGS:
        g = torch.Generator().manual_seed(0)
        X = torch.randn(N, d, dtype=torch.float64, generator=g)
        y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
        Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
        cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng)
        ls = [2.0 if d > 1 else 0.7] * d
        res = {"config": name, "N": N, "d": d, "M": M}
        for label, fn in (("evals_per_s", lambda: cb.value(Z, ls, 1.0, 0.09)),
                          ("value_and_grad_per_s", lambda: cb.value_and_grad(Z, ls, 1.0, 0.09, want_gz=False)),
                          ("value_and_grad_Z_per_s", lambda: cb.value_and_grad(Z, ls, 1.0, 0.09, want_gz=True))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = 30
            for _ in range(k):
                fn()
            torch.cuda.synchronize()
            res[label] = k / (time.perf_counter() - t0)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
