#!/usr/bin/env python3
"""Host time line of one evaluation: entry / exit of every engine call relative to the moment the previous evaluation's read-back
returned (median over evaluations).  SHAPE=N,d,M python3 tools/host_steps.py [grad]"""
import math
import os
import statistics
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
log = []


def wrap(obj, name):
    real = getattr(obj, name)

    def f(*a, **k):
        log.append((name + " >", time.perf_counter()))
        r = real(*a, **k)
        log.append((name + " <", time.perf_counter()))
        return r

    setattr(obj, name, f)


for nm in ("kuu", "kuu_factor", "suffstats", "bound", "streaming_error_report", "suffstats_bwd", "kuu_bwd", "result_buffer"):
    wrap(eng, nm)
for nm in ("_fetch", "_forward", "_prep_Z", "_small_ok", "_guard_on", "_extended_ok", "_kfu_for", "_side_state", "_trace_buf", "_review",
           "_evaluate", "_pass2"):
    wrap(cb, nm)
fn = (lambda: cb.value_and_grad(Z, ls, 1.0, 0.09)) if grad else (lambda: cb.value(Z, ls, 1.0, 0.09))
for _ in range(10):
    fn()
rows = {}
for _ in range(100):
    log.clear()
    t_call = time.perf_counter()
    fn()
    t_ret = time.perf_counter()
    t0 = t_call
    for name, t in log:
        rows.setdefault(name, []).append((t - t0) * 1e6)
    rows.setdefault("returned", []).append((t_ret - t0) * 1e6)
print("us after the call was made (median of 100):")
for name, v in rows.items():
    print("  %-28s %8.1f" % (name, statistics.median(v)))
