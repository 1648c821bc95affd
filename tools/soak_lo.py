#!/usr/bin/env python3
"""Race screen of the trailing-word product (kphi_lo3_kernel: LDS-DMA stages behind raw barriers, wave-private DMA'd blocks in its contraction):
the same inputs must give the same BITS on every call -- partial sums and reductions have a fixed order -- over many calls, several shapes,
with and without the assembly kernel's fp16 image, and while another stream keeps the chip busy (the timing between workgroups changes).
    python tools/soak_lo.py [calls per shape]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = ggp_amd.HipEngine()
side = torch.cuda.Stream(device=eng.device)
noise = torch.randn(4096, 4096, device=eng.device)
bad_total = 0
for (N, M, d) in ((1000000, 1024, 8), (200000, 1024, 8), (60000, 512, 4), (30000, 384, 3), (9000, 200, 2), (5000, 896, 6), (20000, 512, 18), (8000, 256, 32)):
    g = torch.Generator().manual_seed(N + M)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[:M].clone()
    ls, sf2 = [2.0 + 0.1 * j for j in range(d)], 1.1
    Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
    Kuu = eng.kuu(Zd, ls, sf2, 1e-6, "rbf")
    linv, _ = eng.kuu_factor(Kuu)
    kfu, kh = eng.kfu_buffer(N, M), eng.kfu_f16_buffer(N, M)
    eng.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, "rbf", kfu=kfu, level=2, kfu_f16=kh)
    P = torch.randn(M, M, dtype=torch.float64, generator=g)
    sc = torch.logspace(-14, -10, M, dtype=torch.float64)
    P = ((P + P.T) * sc[:, None] * sc[None, :]).to(eng.device)
    first, bad = None, 0
    n = reps if N * M < 3e8 else max(20, reps // 10)
    for it in range(n):
        if it % 3 == 1:   # a busy neighbour on another stream
            with torch.cuda.stream(side):
                (noise @ noise).sum()
        acc = torch.zeros(d + 1, dtype=torch.float64, device=eng.device)
        mode = it % 3   # 0: fp64 block + image, 1: image only, 2: fp64 block only (the product converts it itself: the same image, the same bits)
        eng.suffstats_bwd_lo(Xd, yd, Zd, ls, sf2, P, kfu if mode != 1 else None, acc, "rbf", **({"kfu_f16": kh} if mode != 2 else {}))
        a = acc.cpu()
        if first is None:
            first = a
        elif not torch.equal(a, first):
            bad += 1
    torch.cuda.synchronize()
    bad_total += bad
    print(json.dumps({"N": N, "M": M, "d": d, "calls": n, "calls_with_other_bits": bad}), flush=True)
print(json.dumps({"different_bits_total": bad_total}))
sys.exit(1 if bad_total else 0)
