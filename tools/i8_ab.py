#!/usr/bin/env python3
"""Pass-1 contraction on the integer matrix cores against the fp64 one: the same packed statistics from both
(sgp_set_contraction 0 / 2), elementwise difference relative to max |Phi|, and the time per value-only evaluation at the
shapes where the library's auto rule (mode 1) has to decide.  One JSON object per line."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402


def stats_pair(eng, N, M, d, kernel, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(eng.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(eng.device)
    Z = X[torch.randperm(N, generator=g)[:M].to(eng.device)].clone()
    ls = [0.7 + 0.1 * j for j in range(d)]
    out = {}
    for mode in (0, 2):
        eng.lib.sgp_set_contraction(mode)
        out[mode] = eng.suffstats(X, y, Z, ls, 1.3, kernel).clone()
        assert eng.lib.sgp_contraction_last() == (1 if mode == 2 else 0)
    eng.lib.sgp_set_contraction(-1)
    a, b = out[0], out[2]
    phi_a, phi_b = a[:M * M], b[:M * M]
    scale = float(phi_a.abs().max())
    return {"N": N, "M": M, "d": d, "kernel": kernel, "max_abs_phi": scale,
            "phi_diff_over_max": float((phi_a - phi_b).abs().max()) / scale,
            "phi_int8_asymmetry": float((phi_b.view(M, M) - phi_b.view(M, M).T).abs().max()),
            "b_diff": float((a[M * M:M * M + M] - b[M * M:M * M + M]).abs().max()),
            "tail_equal": bool(torch.equal(a[M * M + M:], b[M * M + M:]))}


def time_eval(eng, N, M, d, reps=8):
    g = torch.Generator().manual_seed(1)
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(eng.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(eng.device)
    Z = X[:M].clone()
    ls = [1.0] * d
    rec = {"N": N, "M": M, "d": d}
    eng.lib.sgp_timing_enable(1)
    for mode in (0, 2):
        eng.lib.sgp_set_contraction(mode)
        out = eng.suffstats(X, y, Z, ls, 1.0, "rbf")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.suffstats(X, y, Z, ls, 1.0, "rbf", out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        ta, ts = ctypes.c_float(), ctypes.c_float()
        eng.lib.sgp_timing_last_ms(0, ctypes.byref(ta))
        eng.lib.sgp_timing_last_ms(1, ctypes.byref(ts))
        rec["fp64" if mode == 0 else "int8"] = {"pass1_ms": ms, "assembly_ms": ta.value, "contraction_ms": ts.value}
    eng.lib.sgp_set_contraction(-1)
    eng.lib.sgp_timing_enable(0)
    return rec


def main():
    eng = ggp_amd.HipEngine()
    if "--boundary" in sys.argv:  # shapes around the default rule's thresholds (rows >= 65536, M > 128) and the multi-GPU shards of C5
        for (N, M, d) in ((65536, 256, 8), (65536, 384, 2), (100000, 256, 2), (131072, 256, 8), (65536, 2048, 8), (500000, 1024, 8),
                          (250000, 1024, 8), (40000, 1024, 8), (30000, 512, 18), (65536, 200, 8),
                          (13279, 512, 18), (16384, 256, 8), (20000, 384, 4), (1000000, 128, 8), (200000, 100, 8), (8192, 1024, 8), (4096, 512, 8)):
            print(json.dumps(time_eval(eng, N, M, d, reps=20)), flush=True)
        return
    for (N, M, d, k) in ((500, 50, 1, "rbf"), (5000, 300, 8, "rbf"), (5000, 300, 3, "matern32"), (20000, 129, 2, "matern52"),
                         (70000, 1024, 8, "rbf"), (33000, 513, 18, "rbf")):
        print(json.dumps(stats_pair(eng, N, M, d, k)), flush=True)
    for (N, M, d) in ((1 << 20, 1024, 8), (131072, 1024, 8), (1 << 20, 512, 8), (1 << 20, 256, 8), (65536, 1024, 8), (262144, 2048, 8)):
        print(json.dumps(time_eval(eng, N, M, d)), flush=True)


if __name__ == "__main__":
    main()
