#!/usr/bin/env python3
"""Device time of the single-launch tile-dataflow Cholesky (sgp_chol_lower) at the tail's sizes, and its accuracy
against torch.linalg.cholesky on the host.  The launch sequence per call is zero flags, clear L^-1, dataflow kernel,
time-out check; the number printed is the whole call."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
for M in (128, 256, 512, 1024, 2048):
    g = torch.Generator().manual_seed(M)
    R = torch.randn(M, M + 64, dtype=torch.float64, generator=g)
    A = (R @ R.T / M + torch.eye(M, dtype=torch.float64)).to(eng.device)
    L, info = eng.chol_lower(A)
    ref = torch.linalg.cholesky(A.cpu())
    err = float((torch.tril(L).cpu() - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            eng.chol_lower(A)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(json.dumps({"M": M, "us_per_call_incl_clone": sorted(ts)[2], "us_per_64_block": sorted(ts)[2] / (M / 64),
                      "rel_err_vs_lapack": err, "info": int(info.item())}))
