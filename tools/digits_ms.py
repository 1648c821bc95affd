#!/usr/bin/env python3
"""Median device time of the value-only kernel assembly (kfu_digits_kernel<.., false>) and the integer contraction at C5, via the
library's HIP-event hooks -- A/B companion of tools/kernel_ms.py for the evaluation path (no K'_fu kept)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
N = int(os.environ.get("ROWS", bench.N_TOTAL))
X, y, Z = bench.synth(N, bench.M_IND, bench.DIM)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
eng.lib.sgp_timing_enable(1)
ms = {0: [], 1: []}
for _ in range(int(os.environ.get("REPS", 12))):
    eng.suffstats(Xd, yd, Zd, [bench.LS] * bench.DIM, bench.SF ** 2, "rbf")
    for k in ms:
        t = ctypes.c_float(0.0)
        eng.lib.sgp_timing_last_ms(k, ctypes.byref(t))
        ms[k].append(t.value)
med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
print(json.dumps({"rows": N, "digits_ms": round(med[0], 3), "i8_syrk_ms": round(med[1], 3)}))
