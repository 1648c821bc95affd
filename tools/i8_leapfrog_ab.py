#!/usr/bin/env python3
"""Why pass 1 on the integer cores does not speed a leapfrog up: pass 1 + pass 2 back to back at C5 with the contraction on the
fp64 cores (mode 0) and on the integer cores (mode 2, the assembly writing the fp64 block and the digit planes), per-kernel HIP-event
times of both passes and the wall time of the pair."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
N, M, d = bench.N_TOTAL, bench.M_IND, bench.DIM
X, y, Z = bench.synth(N, M, d)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
ls, sf2, s2 = [bench.LS] * d, bench.SF ** 2, bench.SN ** 2
eng.lib.sgp_timing_enable(1)
for rep in range(2):
    for mode in (0, 2):
        eng.lib.sgp_set_contraction(mode)
        kfu = eng.kfu_buffer(N, M)
        packed = eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", kfu=kfu)
        Kuu = eng.kuu(Zd, ls, sf2, bench.JITTER, "rbf")
        adj = eng.bound(Kuu, packed, s2, N, with_adjoints=True)
        torch.cuda.synchronize()
        ms = {0: [], 1: [], 2: []}
        t0 = time.perf_counter()
        for _ in range(6):
            eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", out=packed, kfu=kfu)
            eng.suffstats_bwd(Xd, yd, Zd, ls, sf2, adj["Phibar"], adj["bbar"], -1.0 / (2.0 * s2), "rbf", kfu=kfu)
            for slot in ms:
                t = ctypes.c_float(0.0)
                eng.lib.sgp_timing_last_ms(slot, ctypes.byref(t))
                ms[slot].append(t.value)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 6 * 1e3
        med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
        print(json.dumps({"contraction": "int8" if mode == 2 else "fp64", "assembly_ms": med[0], "contraction_ms": med[1], "pass2_ms": med[2],
                          "pass1_plus_pass2_wall_ms": wall}), flush=True)
        del kfu
