#!/usr/bin/env python3
"""Which 64 x 64 blocks of L^-1 differ from the default launch's under a workgroup budget (debugging aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
dev = eng.device
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(M)
R = torch.randn(M, M + 3, dtype=torch.float64, generator=g)
K = (R @ R.T / M + torch.eye(M, dtype=torch.float64)).to(dev)
I0, _ = eng.kuu_factor(K)
Mp = int(round(I0.numel() ** 0.5))
torch.cuda.synchronize()
L0 = eng._ws["kuu_factor"][:Mp * Mp * 8].view(torch.float64).view(Mp, Mp).clone()  # (the factor sits at the start of the call's workspace)
ref = torch.linalg.inv(torch.linalg.cholesky(K.cpu()))
print("default vs LAPACK: %.3g" % float((I0.view(Mp, Mp)[:M, :M].cpu() - ref).abs().max()))
nb = Mp // 64
for budget in (3, 5, 17):
    e = ggp_amd.HipEngine(own_context=True)
    e.set_option("cu_budget", budget)
    Ii, info = e.kuu_factor(K)
    torch.cuda.synchronize()
    Lb = e._ws["kuu_factor"][:Mp * Mp * 8].view(torch.float64).view(Mp, Mp).clone()
    DL = (Lb - L0).abs().cpu()
    badL = [(i, j, float(DL[64 * i:64 * i + 64, 64 * j:64 * j + 64].max())) for i in range(Mp // 64) for j in range(i + 1)
            if float(DL[64 * i:64 * i + 64, 64 * j:64 * j + 64].max()) > 0]
    print("budget %d: %d tiles of the FACTOR differ; first: %s" % (budget, len(badL), badL[:10]))
    D = (Ii.view(Mp, Mp) - I0.view(Mp, Mp)).abs().cpu()
    bad = []
    for i in range(nb):
        for j in range(i + 1):
            m = float(D[64 * i:64 * i + 64, 64 * j:64 * j + 64].max())
            if m > 0:
                bad.append((i, j, m))
    print("budget %d info %d: %d blocks differ; first: %s; vs LAPACK %.3g" % (budget, int(info.item()), len(bad), bad[:12],
          float((Ii.view(Mp, Mp)[:M, :M].cpu() - ref).abs().max())))
    if bad:
        rows = sorted(set(b[0] for b in bad)); cols = sorted(set(b[1] for b in bad))
        print("   rows", rows[:20], "cols", cols[:20])
