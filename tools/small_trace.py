#!/usr/bin/env python3
"""Launch-overhead diagnostic for the small configs: run K leapfrog-shaped evaluations of C2 (N=634, d=1, M=128)
under `rocprofv3 --kernel-trace` and compare wall time per evaluation with the sum of kernel durations
(SHAPE=N,d,M selects another config, e.g. SHAPE=13279,18,512 for C3).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/small -- python3 tools/small_trace.py
    python3 tools/small_trace.py --analyse gpurun_out/small/*/*kernel_trace.csv"""
import csv
import json
import math
import os
import sys
import time

K = 40


def analyse(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]  # second half: steady state
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    names = {}
    for r in rows:
        n = r["Kernel_Name"].split("(")[0][-60:]
        c = names.setdefault(n, [0, 0.0])
        c[0] += 1
        c[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(json.dumps({"launches": len(rows), "busy_us": busy, "span_us": span, "busy_frac": busy / span}))
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:25]:
        print("%6d %9.1f us  %6.2f us/launch  %s" % (c, t, t / c, n))


def main():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ggp_amd
    eng = ggp_amd.HipEngine()
    N, d, M = (int(v) for v in os.environ.get("SHAPE", "634,1,128").split(","))
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
    cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng)
    for _ in range(5):
        cb.value_and_grad(Z, [0.7 if d == 1 else 2.0] * d, 1.0, 0.09, want_gz=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        cb.value_and_grad(Z, [0.7 if d == 1 else 2.0] * d, 1.0, 0.09, want_gz=False)
    torch.cuda.synchronize()
    print(json.dumps({"wall_us_per_eval": (time.perf_counter() - t0) / K * 1e6, "evals": K}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
    else:
        main()
