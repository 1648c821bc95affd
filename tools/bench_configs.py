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
            k, best = 30, float("inf")
            for _ in range(3):  # best of three timed loops: these boxes stall a launch for 10-70 ms about once in a hundred
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(k):
                    fn()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            res[label] = k / best
        print(json.dumps(res))
    # C4: BayesianSVGP-shaped minibatch step (N = 100k, d = 2, M = 256, B = 4096): one bound + full gradient per call
    N, d, M, B = 100_000, 2, 256, 4096
    g = torch.Generator().manual_seed(1)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    f = torch.sin(2.0 * X[:, 0]) * torch.cos(X[:, 1])
    Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
    m = (0.1 * torch.randn(M, dtype=torch.float64, generator=g)).to(eng.device)
    LS = (torch.eye(M, dtype=torch.float64) + 0.01 * torch.tril(torch.randn(M, M, dtype=torch.float64, generator=g))).to(eng.device)
    for lik, yv in (("gaussian", f + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)), ("bernoulli", torch.sign(f))):
        Xd, yd = X.to(eng.device), yv.to(eng.device)
        batches = [torch.randperm(N, generator=g)[:B].to(eng.device) for _ in range(8)]
        Xb = [Xd[b].contiguous() for b in batches]
        yb = [yd[b].contiguous() for b in batches]

        def step(i):
            r = eng.svgp_elbo(Xb[i % 8], yb[i % 8], Z, [1.0, 1.0], 1.0, 0.05, m, LS, N, jitter=1e-6, likelihood=lik, with_grads=True)
            return r["out"]

        for i in range(3):
            step(i)
        k, best = 30, float("inf")
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                o = step(i)
            float(o[0])
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(json.dumps({"config": "C4 SVGP minibatch step (%s)" % lik, "N": N, "d": d, "M": M, "batch": B,
                          "bound_and_grad_per_s": k / best}))


if __name__ == "__main__":
    main()
