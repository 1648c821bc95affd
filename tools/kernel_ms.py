#!/usr/bin/env python3
"""Median device time of the three dominant kernels at C5 (assembly, pass-1 contraction, pass-2 contraction), via the
library's HIP-event hooks -- the quick A/B companion of bench.py (no CPU baseline, no end-to-end loop)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402


def main():
    eng = ggp_amd.HipEngine()
    N = int(os.environ.get("ROWS", bench.N_TOTAL))
    X, y, Z = bench.synth(N, bench.M_IND, bench.DIM)
    Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
    ls, sf2, s2 = [bench.LS] * bench.DIM, bench.SF ** 2, bench.SN ** 2
    kfu = eng.kfu_buffer(N, bench.M_IND)
    eng.lib.sgp_timing_enable(1)
    packed = eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", kfu=kfu)
    adj = eng.bound(eng.kuu(Zd, ls, sf2, bench.JITTER, "rbf"), packed, s2, N, with_adjoints=True)
    ms = {0: [], 1: [], 2: []}
    for _ in range(int(os.environ.get("REPS", 8))):
        eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", out=packed, kfu=kfu)
        eng.suffstats_bwd(Xd, yd, Zd, ls, sf2, adj["Phibar"], adj["bbar"], -0.5 / s2, "rbf", kfu=kfu)
        for k in ms:
            t = ctypes.c_float(0.0)
            eng.lib.sgp_timing_last_ms(k, ctypes.byref(t))
            ms[k].append(t.value)
    med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
    print(json.dumps({"rows": N, "assemble_ms": round(med[0], 3), "syrk_ms": round(med[1], 3), "kbar_ms": round(med[2], 3)}))


if __name__ == "__main__":
    main()
