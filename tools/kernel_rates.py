#!/usr/bin/env python3
"""Per-kernel device time of the three streaming kernels at several N (M=1024, d=8), via the library's timing hooks.
Tells memory-bound from MFMA-bound: K'_fu fits the 256 MB Infinity Cache up to N = 32k."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

M, D = int(os.environ.get("M", 1024)), 8
eng = ggp_amd.HipEngine()
lib = eng.lib
for N in (16384, 32768, 131072, 1_000_000):
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, D, dtype=torch.float64, generator=g).to(eng.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(eng.device)
    Z = X[:M].clone()
    Pb = torch.randn(M, M, dtype=torch.float64, generator=g)
    Pb = (Pb + Pb.T).to(eng.device).contiguous()
    bb = torch.randn(M, dtype=torch.float64, generator=g).to(eng.device)
    kfu = eng.kfu_buffer(N, M)
    lib.sgp_timing_enable(1)
    ts = {0: [], 1: [], 2: []}
    for rep in range(4):
        eng.suffstats(X, y, Z, [2.0] * D, 1.0, "rbf", kfu=kfu)
        eng.suffstats_bwd(X, y, Z, [2.0] * D, 1.0, Pb, bb, -0.5, "rbf", want_gz=False, kfu=kfu)
        for s in ts:
            t = ctypes.c_float()
            assert lib.sgp_timing_last_ms(s, ctypes.byref(t)) == 0
            ts[s].append(t.value)
    lib.sgp_timing_enable(0)
    a, s_, k = (sorted(ts[i])[1] for i in (0, 1, 2))
    mp = (M + 127) // 128 * 128
    print(json.dumps({"N": N, "M": M, "assemble_ms": a, "assemble_GBps": 8.0 * N * mp / a / 1e6,
                      "syrk_ms": s_, "syrk_TF_alg": N * M * (M + 1) / s_ / 1e9,
                      "kbar_ms": k, "kbar_TF": 2.0 * N * mp * mp / k / 1e9}))
    del kfu, X
