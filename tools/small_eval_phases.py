#!/usr/bin/env python3
"""Where the time of one sgp_small_eval launch goes: s_memrealtime stamps (100 MHz) of every workgroup's phases.
    python3 tools/small_eval_phases.py            (SHAPE=N,d,M ; KERNEL=co2 for the composite CO2 covariance, d = 1)"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

CHAIN = ["start", "Kuu evaluated + factored", "L published", "B complete (waited)", "chol(B) done", "LB published", "c0, g, h solved",
         "g, h published", "gradient partials in (waited)", "done"]
ROW = ["start", "z staged", "L seen", "forward slabs done", "all partials in", "slices done", "LB seen", "reverse slabs done"]
KUU = ["start", "LB seen", "Q columns done", "R + contraction done"]


def main():
    eng = ggp_amd.HipEngine()
    N, d, M = (int(v) for v in os.environ.get("SHAPE", "634,1,128").split(","))
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
    X, y = X.to(eng.device), y.to(eng.device)
    th = torch.tensor([0.7 if d == 1 else 2.0] * d + [1.0, 0.09], dtype=torch.float64).to(eng.device)
    kern, kw = "rbf", {}
    if os.environ.get("KERNEL", "rbf") == "co2":
        blk = ggp_amd.co2_kernel(0.5, 1.0, 5.0, 1.0, 3.0, 1.0, 0.5, 2.0, 0.1, 0.5).block()
        th = torch.tensor(blk + [0.09], dtype=torch.float64).to(eng.device)
        kern, kw = "composite", {"composite": {"structure": blk}}
    stamps = torch.zeros(211 * 16, dtype=torch.int64, device=eng.device)
    eng.lib.sgp_small_debug_stamps(C.c_void_p(stamps.data_ptr()))
    for _ in range(20):
        eng.small_eval(X, y, Z, th, 1e-6, kern, mode=0, want_grad=True, **kw)
    torch.cuda.synchronize()
    eng.lib.sgp_small_debug_stamps(C.c_void_p(0))
    s = stamps.cpu().reshape(211, 16)
    t0 = int(s[0, 0])
    grow = min((N + 63) // 64, 64)

    def show(name, row, labels):
        print(name)
        prev = None
        for k, lab in enumerate(labels):
            t = (int(row[k]) - t0) / 100.0
            print("   %-32s %8.1f us%s" % (lab, t, "" if prev is None else "   (+%.1f)" % (t - prev)))
            prev = t

    nv = 1 if M <= 64 else 2
    show("chain workgroup", s[0], CHAIN)
    show("Kuu-adjoint workgroup 0 of %d" % nv, s[1], KUU)
    show("row workgroup 0 of %d" % grow, s[1 + nv], ROW)
    show("row workgroup %d" % (grow - 1), s[nv + grow], ROW)
    r = s[1 + nv]
    names = ["first slab staged (factor blocks, x, y)", "K_uf assembled in registers", "forward solve done", "slab stored, column sums",
             "(A^T A partial done)"]
    print("row workgroup 0, first forward slab")
    for k, nm in zip(range(8, 13), names):
        print("   %-42s %8.1f us" % (nm, (int(r[k]) - t0) / 100.0))
    print("row workgroup 0, first reverse slab (LB seen at %.1f us)" % ((int(r[6]) - t0) / 100.0))
    for k, nm in ((13, "LB blocks staged"), (14, "c0, g solved"), (15, "B^-1 a (two solves), Abar formed"), (10, "L blocks staged, Kbar solved"),
                  (7, "contraction done")):
        print("   %-42s %8.1f us" % (nm, (int(r[k]) - t0) / 100.0))


if __name__ == "__main__":
    main()
