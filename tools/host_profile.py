#!/usr/bin/env python3
"""cProfile of the host side of one evaluation at a given shape (default C3), sorted by own time: what the interpreter spends between
the launches.  SHAPE=N,d,M  python3 tools/host_profile.py [grad]"""
import cProfile
import math
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
N, d, M = (int(v) for v in os.environ.get("SHAPE", "13279,18,512").split(","))
grad = len(sys.argv) > 1 and sys.argv[1] == "grad"
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, dtype=torch.float64, generator=g)
y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng)
ls = [2.0] * d
fn = (lambda: cb.value_and_grad(Z, ls, 1.0, 0.09)) if grad else (lambda: cb.value(Z, ls, 1.0, 0.09))
for _ in range(10):
    fn()
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    fn()
torch.cuda.synchronize()
print("us per call: %.1f" % ((time.perf_counter() - t0) / K * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    fn()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
