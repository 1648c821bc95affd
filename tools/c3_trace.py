#!/usr/bin/env python3
"""C3 (N 13 279, d 18, M 512) value + gradient evaluations for a rocprofv3 --kernel-trace timeline (tools/last_eval_timeline.py)."""
import sys, math, time, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd
eng = ggp_amd.HipEngine()
N, d, M = 13279, 18, 512
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, dtype=torch.float64, generator=g)
y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng)
ls = [2.0] * d
for _ in range(5): cb.value_and_grad(Z, ls, 1.0, 0.09, want_gz=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): cb.value_and_grad(Z, ls, 1.0, 0.09, want_gz=False)
torch.cuda.synchronize()
print(json.dumps({"ms": (time.perf_counter() - t0) / 20 * 1e3, "form_whitened": cb._whitened(M)}))
