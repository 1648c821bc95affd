#!/usr/bin/env python3
"""Which workgroup budgets reproduce the default launch's bits: sgp_chol_lower (factor only) and sgp_kuu_factor_ex (factor + inverse)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
dev = eng.device
for M in (192, 512, 1024, 2048):
    g = torch.Generator().manual_seed(M)
    R = torch.randn(M, M + 3, dtype=torch.float64, generator=g)
    K = (R @ R.T / M + torch.eye(M, dtype=torch.float64)).to(dev)
    L0, _ = eng.chol_lower(K)
    I0, _ = eng.kuu_factor(K)
    for budget in (2, 3, 4, 5, 6, 7, 8, 9, 10, 16, 17, 33, 64):
        e = ggp_amd.HipEngine(own_context=True)
        e.set_option("cu_budget", budget)
        for shared in (0, 1):   # the static deal, and the ticketed claim of SGP_OPT_SHARED_DEVICE
            e.set_option("shared_device", shared)
            badL = badI = 0
            dL = dI = 0.0
            for rep in range(10):
                L, info = e.chol_lower(K)   # (no context argument: the engine binds its own to the thread, include/sgp.h sgp_ctx_bind_thread)
                Ii, info2 = e.kuu_factor(K)
                torch.cuda.synchronize()
                if not torch.equal(L, L0) or int(info.item()) != 0:
                    badL += 1
                    dL = max(dL, float((L - L0).abs().max()))
                if not torch.equal(Ii, I0) or int(info2.item()) != 0:
                    badI += 1
                    dI = max(dI, float((Ii - I0).abs().max()))
            if badL or badI:
                print("M %d budget %d shared_device %d: factor differs %d/10 (max %.3g), inverse differs %d/10 (max %.3g)" % (M, budget, shared, badL, dL, badI, dI), flush=True)
print("done")
