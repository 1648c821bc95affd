#!/usr/bin/env python3
"""A/B of SGP_I8_PRIO (s_setprio(1) for the wave at its step top in the integer contraction) alternating inside one process at C5:
per-kernel HIP-event time of the contraction, median of 6."""
import sys, os, time, json, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ggp_amd
eng = ggp_amd.HipEngine()
N, M, d = bench.N_TOTAL, bench.M_IND, bench.DIM
X, y, Z = bench.synth(N, M, d)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
eng.lib.sgp_timing_enable(1)
out = eng.suffstats(Xd, yd, Zd, [bench.LS] * d, bench.SF ** 2, "rbf")
for rep in range(4):
    for prio in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "1")):
        os.environ[sys.argv[1] if len(sys.argv) > 1 else "SGP_I8_PRIO"] = prio
        ms = []
        for _ in range(6):
            eng.suffstats(Xd, yd, Zd, [bench.LS] * d, bench.SF ** 2, "rbf", out=out)
            t = ctypes.c_float(); eng.lib.sgp_timing_last_ms(1, ctypes.byref(t)); ms.append(t.value)
        print(json.dumps({(sys.argv[1] if len(sys.argv) > 1 else "SGP_I8_PRIO"): prio, "contraction_ms_median": sorted(ms)[3]}), flush=True)
