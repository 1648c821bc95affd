#!/usr/bin/env python3
"""Host-side cost of one CollapsedBound.value() call on the multi-launch path: cProfile over K evaluations
    python3 tools/host_overhead.py [rows] [grad]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    grad = len(sys.argv) > 2 and sys.argv[2] == "grad"
    eng = ggp_amd.HipEngine()
    X, y, Z = bench.synth(rows, bench.M_IND, bench.DIM)
    cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=bench.JITTER, engine=eng)
    Zd = Z.to(eng.device)
    ls = [bench.LS] * bench.DIM
    fn = (lambda: cb.value_and_grad(Zd, ls, bench.SF ** 2, bench.SN ** 2)) if grad else (lambda: cb.value(Zd, ls, bench.SF ** 2, bench.SN ** 2))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    print("ms per call: %.3f" % ((time.perf_counter() - t0) / K * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(K):
        fn()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
