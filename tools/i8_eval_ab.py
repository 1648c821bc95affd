#!/usr/bin/env python3
"""One C5 evaluation (CollapsedBound.value) with the contraction on the fp64 cores (0) and under the default rule (1), with and
without the side-stream K_uu chain; then pass 1 alone with its per-kernel HIP-event times.  One JSON object per line."""
import sys, time, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ggp_amd
eng = ggp_amd.HipEngine()
N, M, d = bench.N_TOTAL, bench.M_IND, bench.DIM
X, y, Z = bench.synth(N, M, d)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
for overlap in (True, False):
    for mode in (0, 1):
        eng.lib.sgp_set_contraction(mode)
        cb = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng)
        cb.overlap_tail = overlap
        for _ in range(3):
            cb.value(Zd, [bench.LS] * d, bench.SF ** 2, bench.SN ** 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            F, _ = cb.value(Zd, [bench.LS] * d, bench.SF ** 2, bench.SN ** 2)
        torch.cuda.synchronize()
        print(json.dumps({"overlap_tail": overlap, "contraction": mode, "ms_per_eval": (time.perf_counter() - t0) / 10 * 1e3, "F": F}), flush=True)
import ctypes
eng.lib.sgp_timing_enable(1)
for mode in (0, 1):
    eng.lib.sgp_set_contraction(mode)
    out = eng.suffstats(Xd, yd, Zd, [bench.LS] * d, bench.SF ** 2, "rbf")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        eng.suffstats(Xd, yd, Zd, [bench.LS] * d, bench.SF ** 2, "rbf", out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 8 * 1e3
    ta, ts = ctypes.c_float(), ctypes.c_float()
    eng.lib.sgp_timing_last_ms(0, ctypes.byref(ta)); eng.lib.sgp_timing_last_ms(1, ctypes.byref(ts))
    print(json.dumps({"contraction": mode, "pass1_ms": ms, "assembly_ms": ta.value, "contraction_ms": ts.value}), flush=True)
