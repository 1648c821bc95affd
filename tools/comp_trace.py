#!/usr/bin/env python3
"""Launch profile of the composite-covariance path at C2 (N = 634, d = 1, M = 64 / 128, CO2 kernel):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/comp -- python3 tools/comp_trace.py
    python3 tools/small_trace.py --analyse <kernel_trace.csv>"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = ggp_amd.HipEngine()
g = torch.Generator().manual_seed(0)
t = torch.linspace(0, 52.8, 634, dtype=torch.float64)
y = 0.05 * t + torch.sin(2 * torch.pi * t) * 0.5 + 0.1 * torch.randn(634, dtype=torch.float64, generator=g)
y = (y - y.mean()) / y.std()
X = t[:, None].to(eng.device)
Z = X[:: 634 // M][:M].clone()
cb = ggp_amd.CollapsedBound(X, y.to(eng.device), kernel="composite", jitter=1e-4, engine=eng)
kern = ggp_amd.co2_kernel(1.0, 1.0, 5.0, 0.5, 1.0, 1.0, 1.0, 10.0, 0.1, 0.5)
blk = kern.block()
for _ in range(5):
    cb.value_and_grad(Z, blk, 1.0, 0.05)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for _ in range(K):
    cb.value_and_grad(Z, blk, 1.0, 0.05)
torch.cuda.synchronize()
print(json.dumps({"M": M, "value_and_grad_us": (time.perf_counter() - t0) / K * 1e6}))
