#!/usr/bin/env python3
"""Measured error of the streaming evaluation order against the guard's estimate, near the tolerance: F in the streaming order (integer and
fp64 contraction, guard off) minus F in the whitened order (which the oracle confirms to 1e-9 per datum, tests/test_int8_theta_sweep.py),
per datum, next to `sgp_streaming_error_estimate`.  One JSON line per theta.  ROWS / M override the C5 shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

N = int(os.environ.get("ROWS", bench.N_TOTAL))
M = int(os.environ.get("M", bench.M_IND))
eng = ggp_amd.HipEngine()
X, y, Z = bench.synth(N, M, bench.DIM)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
cs = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng)   # form="auto": the estimate is computed ...
cs.streaming_tol = float("inf")                                        # ... and never acted upon
cw = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng, form="whitened")
for ls in (2.0, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0, 8.0):
    for sn in (0.3, 0.145, 0.05):
        row = {"N": N, "M": M, "ls": ls, "sig_n": sn}
        Fw, pw = cw.value(Zd, [ls] * bench.DIM, 1.0, sn * sn, raise_on_fail=False)
        row["F_whitened_per_datum"] = Fw / N
        for mode, key in ((1, "i8"), (0, "f64")):
            eng.lib.sgp_set_contraction(mode)
            F, parts = cs.value(Zd, [ls] * bench.DIM, 1.0, sn * sn, raise_on_fail=False)
            row["err_%s_per_datum" % key] = abs(F - Fw) / N if parts.get("info", 0) == 0 else None
            row["estimate_%s" % key] = cs.last_estimate
        eng.lib.sgp_set_contraction(1)
        e = row["estimate_i8"]
        row["err_over_estimate_i8"] = (row["err_i8_per_datum"] / e) if e and row["err_i8_per_datum"] is not None else None
        row["err_over_estimate_f64"] = (row["err_f64_per_datum"] / e) if e and row["err_f64_per_datum"] is not None else None
        print(json.dumps(row), flush=True)
