#!/usr/bin/env python3
"""End to end for d > 8 (round 6: the trailing-word product runs eight dimensions per pass): value + gradient of the bound in the default mode
-- the extended order with both words of Phibar where the estimate allows -- against the whitened order's factored pass 2, N 200 000, d 18, M 512."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
N, d, M = 200000, 18, 512
g = torch.Generator().manual_seed(3)
X = torch.randn(N, d, dtype=torch.float64, generator=g)
y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
Xd, yd = X.to(eng.device), y.to(eng.device)
for lsv, sn in ((6.0, 0.1), (9.0, 0.1), (12.0, 0.05), (16.0, 0.05), (24.0, 0.03)):
    ls = [lsv * (1.0 + 0.03 * j) for j in range(d)]
    cb = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=eng)
    cw = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=eng, form="whitened")
    F, gr = cb.value_and_grad(Z, ls, 1.0, sn * sn, want_gz=False)
    Fw, gw = cw.value_and_grad(Z, ls, 1.0, sn * sn, want_gz=False)
    scale = max(1.0, float(gw["ls"].abs().max()))
    print(json.dumps({"ls": lsv, "sig_n": sn, "tier": cb.last_tier, "estimate_per_datum": cb.last_estimate, "trailing_word_correction": cb.last_lo_correction,
                      "rejections": cb.n_lo_rejections, "F_diff_per_datum": (F - Fw) / N,
                      "g_ls_diff": float((gr["ls"] - gw["ls"]).abs().max()) / scale,
                      "g_sf2_diff": abs(float(gr["sf2"]) - float(gw["sf2"])) / max(1.0, abs(float(gw["sf2"])))}), flush=True)
