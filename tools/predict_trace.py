#!/usr/bin/env python3
"""Time of the predictive (sgp_predict) at C3-like sizes: T test points, M inducing points, d dims; with and without
the T x T covariance the reference builds (models/sgpr.py:150-160).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pred -- python3 tools/predict_trace.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

N, d, M, T = (int(v) for v in os.environ.get("SHAPE", "13279,18,512,3320").split(","))
eng = ggp_amd.HipEngine()
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, dtype=torch.float64, generator=g)
y = torch.sin(X.sum(1) / d ** 0.5) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
Xs = torch.randn(T, d, dtype=torch.float64, generator=g).to(eng.device)
Z = X[:M].clone().to(eng.device)
cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng)
ls = [2.0] * d
out = {"N": N, "d": d, "M": M, "T": T}
t0 = time.perf_counter()
for _ in range(3):
    fac = cb.factors(Z, ls, 1.0, 0.09)
torch.cuda.synchronize()
for full in (False, True):
    for _ in range(3):
        cb.predict(Xs, Z, ls, 1.0, 0.09, full_cov=full, factors=fac)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        cb.predict(Xs, Z, ls, 1.0, 0.09, full_cov=full, factors=fac)
    torch.cuda.synchronize()
    out["predict_full_cov_%s_us" % full] = (time.perf_counter() - t0) / K * 1e6
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    cb.factors(Z, ls, 1.0, 0.09)
torch.cuda.synchronize()
out["factors_us"] = (time.perf_counter() - t0) / 10 * 1e6
print(json.dumps(out))
