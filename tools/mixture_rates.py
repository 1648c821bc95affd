#!/usr/bin/env python3
"""Mixture posterior predictive (reference models/bayesian_sgpr_hmc.py:198-231) over S theta samples: sgp_mixture_predict (eight
samples per chain of launches, PSD gates in one dataflow launch) against the reference's per-sample loop on the HIP engine.
Shapes: the reference's UCI size class (N 1300, d 8, M 100, T 145 test rows = 10 %) and the 1-D demo (N 375, M 25, T 1000)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402
from ggp_amd.hmc import Trace  # noqa: E402


def main():
    eng = ggp_amd.HipEngine()
    for N, d, M, T, S in ((1300, 8, 100, 145, 100), (375, 1, 25, 1000, 100), (13279, 18, 100, 1660, 40)):
        g = torch.Generator().manual_seed(1)
        X = torch.randn(N, d, dtype=torch.float64, generator=g)
        y = torch.sin(X.sum(1) / np.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
        Xt = torch.randn(T, d, dtype=torch.float64, generator=g)
        Z0 = X[torch.randperm(N, generator=g)[:M]].clone()
        model = ggp_amd.BayesianSparseGPR_HMC(X.to(eng.device), y.to(eng.device), ggp_amd.GaussianLikelihood(), Z0, engine=eng, jitter=1e-6)
        rows = [{"ls": np.full(d, 1.5 if d > 1 else 0.7) * (1.0 + 0.01 * i), "sig_f": 1.0 + 0.002 * i, "sig_n": 0.3 + 0.001 * i} for i in range(S)]
        trace = Trace(rows, {"step_size": np.zeros(S)})
        res = {"N": N, "d": d, "M": M, "T": T, "S": S}
        for label, batched in (("batched_ms", True), ("per_sample_loop_ms", False)):
            model.batched_mixture = batched
            import io
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                ggp_amd.mixture_posterior_predictive(model, Xt.to(eng.device), trace)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                preds = ggp_amd.mixture_posterior_predictive(model, Xt.to(eng.device), trace)
                torch.cuda.synchronize()
            res[label] = round((time.perf_counter() - t0) * 1e3, 2)
            res["kept_" + label[:-3]] = len(preds)
        res["speedup"] = round(res["per_sample_loop_ms"] / res["batched_ms"], 2)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
