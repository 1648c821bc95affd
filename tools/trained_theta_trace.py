#!/usr/bin/env python3
"""Six value + gradient evaluations of the NUTS target at C5's trained ARD theta in the default (parity) mode -- the extended order with both
words of Phibar -- for a rocprofv3 kernel trace (tools/last_eval_timeline.py <trace.csv> kuu_kernel prints the last one)."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
X, y, Z = bench.synth(bench.N_TOTAL, bench.M_IND, bench.DIM)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
ls_tr = [4.870895252562722, 2.274348615181124, 7.035384773166531, 6.388168428424034, 7.176420862837876, 3.3523772450641136,
         2.314383327914714, 6.492694463809999]
th = [math.log(v) for v in ls_tr] + [0.0, math.log(0.14415221312756948)]
tb = ggp_amd.CollapsedBound(Xd, yd, kernel="rbf", jitter=bench.JITTER, engine=eng)
tgt = ggp_amd.HmcTarget(tb, Zd, gradient=os.environ.get("MODE", "parity"))
for _ in range(2):
    tgt.logp_and_grad(th)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    lp, gr = tgt.logp_and_grad(th)
torch.cuda.synchronize()
print({"ms_per_leapfrog": (time.perf_counter() - t0) / 8 * 1e3, "logp": lp, "grad0": float(gr[0]), "correction": tb.last_lo_correction, "tier": tb.last_tier, "rejections": tb.n_lo_rejections})
