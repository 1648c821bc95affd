#!/usr/bin/env python3
"""sgp_suffstats_bwd_lo against the fp64 pass 2 run on the same matrix, over shapes / offsets / lengthscales that stress the third version's
fp16 contraction (centred hi + lo inputs, expansion of (z - x)^2): prints the relative error per case.  SGP_LO_KERNEL=2 gives the fp64 contraction."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
worst = 0.0
for (N, M, d, shift, lsv, spread) in ((3000, 200, 3, 0.0, 1.5, 1.0), (5000, 1024, 8, 0.0, 2.0, 1.0), (5000, 256, 8, 40.0, 2.0, 1.0), (4000, 256, 2, -300.0, 0.3, 1.0),
                                      (6000, 512, 4, 5.0, 0.5, 20.0), (777, 130, 1, 0.0, 1.0, 1.0), (4000, 300, 3, 0.0, 1.5, 1.0), (900, 100, 2, 0.0, 1.0, 1.0), (5000, 640, 8, 0.0, 2.5, 1.0), (3000, 896, 6, 0.0, 2.0, 1.0), (6000, 512, 18, 0.0, 3.0, 1.0), (4000, 256, 32, 2.0, 4.0, 1.0), (5000, 1024, 9, 0.0, 2.5, 1.0), (3000, 256, 16, -5.0, 3.0, 1.0), (20000, 1024, 8, 0.0, 3.0, 1.0), (2500, 256, 5, 1000.0, 0.05, 3.0)):
    g = torch.Generator().manual_seed(N + M)
    X = spread * torch.randn(N, d, dtype=torch.float64, generator=g) + shift
    y = torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[:M].clone()
    ls, sf2 = [lsv * (1.0 + 0.1 * j) for j in range(d)], 1.3
    Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
    kfu = eng.kfu_buffer(N, M)
    eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", kfu=kfu)
    zero = torch.zeros(M, dtype=torch.float64, device=eng.device)
    Pr = torch.randn(M, M, dtype=torch.float64, generator=g)
    sc = torch.logspace(-14, -9, M, dtype=torch.float64)
    Pr = ((Pr + Pr.T) * sc[:, None] * sc[None, :]).to(eng.device)
    exact = eng.suffstats_bwd(Xd, yd, Zd, ls, sf2, Pr, zero, 0.0, "rbf", want_gz=False, kfu=kfu).cpu()
    acc = torch.zeros(d + 1, dtype=torch.float64, device=eng.device)
    delta = torch.zeros(d + 1, dtype=torch.float64, device=eng.device)
    eng.suffstats_bwd_lo(Xd, yd, Zd, ls, sf2, Pr, kfu, acc, "rbf", delta=delta)
    acc = acc.cpu()
    err = float((acc - exact).abs().max()) / float(exact.abs().max())
    worst = max(worst, err)
    print(json.dumps({"N": N, "M": M, "d": d, "shift": shift, "ls": lsv, "spread": spread, "rel_err": err,
                      "delta_equals_added": bool(torch.equal(delta.cpu(), acc))}), flush=True)
print(json.dumps({"worst": worst}))
