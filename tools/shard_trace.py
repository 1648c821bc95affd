#!/usr/bin/env python3
"""One rank's share of the 8-GPU C5 job (N = 125 000 rows, d = 8, M = 1024): value-only and value+grad evaluations,
for `rocprofv3 --kernel-trace` timelines of the replicated O(M^3) tail (tools/trace_timeline.py prints them)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
    grad = len(sys.argv) > 2 and sys.argv[2] == "grad"
    eng = ggp_amd.HipEngine()
    X, y, Z = bench.synth(rows, bench.M_IND, bench.DIM)
    cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=bench.JITTER, engine=eng)
    Zd = Z.to(eng.device)
    fn = (lambda: cb.value_and_grad(Zd, [bench.LS] * bench.DIM, bench.SF ** 2, bench.SN ** 2, want_gz=False)) if grad else \
         (lambda: cb.value(Zd, [bench.LS] * bench.DIM, bench.SF ** 2, bench.SN ** 2))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    print(json.dumps({"rows": rows, "grad": grad, "ms_per_eval": (time.perf_counter() - t0) / K * 1e3}))


if __name__ == "__main__":
    main()
