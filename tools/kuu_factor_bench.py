#!/usr/bin/env python3
"""Device time of sgp_kuu_factor_ex (prep + factorization + inverse + gate + trace) and of sgp_chol_lower at small block counts."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
for M in (130, 192, 256, 320, 384, 512):
    g = torch.Generator().manual_seed(M)
    R = torch.randn(M, M + 64, dtype=torch.float64, generator=g)
    A = (R @ R.T / M + torch.eye(M, dtype=torch.float64)).to(eng.device)
    tr = torch.empty(eng.lib.sgp_kuu_inverse_trace_len(), dtype=torch.float64, device=eng.device)
    res = {"M": M}
    for name, fn in (("kuu_factor_us", lambda: eng.kuu_factor(A, trace_out=tr)), ("chol_lower_us", lambda: eng.chol_lower(A))):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rep in range(5):
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[name] = round(sorted(ts)[2], 1)
    print(json.dumps(res))
