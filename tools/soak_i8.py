#!/usr/bin/env python3
"""Soak of the integer-core contraction: repeated launches at C5 and at ragged shapes must reproduce their first result bit for bit
(exact integer sums, fixed fold order) -- a race in the LDS stage ring would flip digits."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
eng.lib.sgp_set_contraction(2)
g = torch.Generator().manual_seed(0)
bad = 0
for (N, M, d, reps) in ((1_000_000, 1024, 8, 40), (125_000, 1024, 8, 150), (333_333, 513, 5, 80), (70_001, 129, 2, 200), (9_000, 1500, 3, 200)):
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(eng.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(eng.device)
    Z = torch.randn(M, d, dtype=torch.float64, generator=g).to(eng.device)
    ls = [1.5] * d
    ref = eng.suffstats(X, y, Z, ls, 1.0, "rbf").clone()
    out = torch.empty_like(ref)
    diffs = 0
    for _ in range(reps):
        eng.suffstats(X, y, Z, ls, 1.0, "rbf", out=out)
        diffs += 0 if torch.equal(out, ref) else 1
    bad += diffs
    print(json.dumps({"N": N, "M": M, "d": d, "launches": reps, "launches_that_differ": diffs}), flush=True)
sys.exit(1 if bad else 0)
