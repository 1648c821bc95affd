#!/usr/bin/env python3
"""Probe: can kernel assembly into digit planes (VALU-bound, 86 registers) run BESIDE the integer contraction (two 188-register waves
per SIMD, 129 KB of LDS) now that both fit on a CU?  Two engines (own workspaces) on two streams: a small head shard whose contraction
should overlap the assembly of the big tail shard.  Wall time of the pair against the two calls back to back on one stream."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

N, M, d = bench.N_TOTAL, bench.M_IND, bench.DIM
X, y, Z = bench.synth(N, M, d)
ea, eb = ggp_amd.HipEngine(), ggp_amd.HipEngine()
dev = ea.device
Xd, yd, Zd = X.to(dev), y.to(dev), Z.to(dev)
ls, sf2 = [bench.LS] * d, bench.SF ** 2
s1 = torch.cuda.Stream(device=dev, priority=-1)
s2 = torch.cuda.Stream(device=dev)
for head in (125_000, 250_000, 500_000):
    Xa, ya, Xb, yb = Xd[:head].contiguous(), yd[:head].contiguous(), Xd[head:].contiguous(), yd[head:].contiguous()
    oa, ob = ea.suffstats(Xa, ya, Zd, ls, sf2, "rbf"), eb.suffstats(Xb, yb, Zd, ls, sf2, "rbf")
    torch.cuda.synchronize()

    def serial():
        ea.suffstats(Xa, ya, Zd, ls, sf2, "rbf", out=oa)
        eb.suffstats(Xb, yb, Zd, ls, sf2, "rbf", out=ob)

    def overlapped():
        cur = torch.cuda.current_stream(dev)
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ea.suffstats(Xa, ya, Zd, ls, sf2, "rbf", out=oa)
        with torch.cuda.stream(s2):
            eb.suffstats(Xb, yb, Zd, ls, sf2, "rbf", out=ob)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    res = {"head_rows": head}
    for name, fn in (("serial_ms", serial), ("two_streams_ms", overlapped), ("serial_again_ms", serial)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 6 * 1e3
    print(json.dumps(res), flush=True)
