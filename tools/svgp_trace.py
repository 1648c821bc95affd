#!/usr/bin/env python3
"""Launch profile of one SVGP minibatch step at C4 (N = 100k, d = 2, M = 256, B = 4096, bound + full gradient):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/svgp -- python3 tools/svgp_trace.py
    python3 tools/small_trace.py --analyse <kernel_trace.csv>"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

lik = sys.argv[1] if len(sys.argv) > 1 else "gaussian"
eng = ggp_amd.HipEngine()
N, d, M, B = 100_000, 2, 256, 4096
g = torch.Generator().manual_seed(1)
X = torch.randn(N, d, dtype=torch.float64, generator=g)
f = torch.sin(2.0 * X[:, 0]) * torch.cos(X[:, 1])
y = (f + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)) if lik == "gaussian" else torch.sign(f)
Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
m = (0.1 * torch.randn(M, dtype=torch.float64, generator=g)).to(eng.device)
LS = (torch.eye(M, dtype=torch.float64) + 0.01 * torch.tril(torch.randn(M, M, dtype=torch.float64, generator=g))).to(eng.device)
idx = torch.randperm(N, generator=g)[:B]
Xb, yb = X[idx].contiguous().to(eng.device), y[idx].contiguous().to(eng.device)


def step():
    return eng.svgp_elbo(Xb, yb, Z, [1.0, 1.0], 1.0, 0.05, m, LS, N, jitter=1e-6, likelihood=lik, with_grads=True)["out"]


for _ in range(5):
    step()
torch.cuda.synchronize()
K = 40
ts = []
for _ in range(K):
    t0 = time.perf_counter()
    o = step()
    float(o[0])
    ts.append((time.perf_counter() - t0) * 1e6)
ts.sort()
print(json.dumps({"likelihood": lik, "us_per_step": sum(ts) / K, "min": ts[0], "median": ts[K // 2], "max": ts[-1]}))
