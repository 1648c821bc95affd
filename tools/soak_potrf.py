#!/usr/bin/env python3
"""Soak of the chain-workgroup Cholesky with the inverse formed inside the launch (csrc/sgp_potrf_chain.hpp): thousands of
sgp_kuu_factor_ex calls at the tail's sizes -- alone, under several workgroup budgets, and beside a contraction-sized kernel on another
stream -- must all report status 0 and return the bits of the first call (L^-1 and tr(K_uu^-1)).  python3 tools/soak_potrf.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
eng = ggp_amd.HipEngine()
dev = eng.device
n_tr = eng.lib.sgp_kuu_inverse_trace_len()
other = torch.cuda.Stream(device=dev)
A = torch.randn(4096, 4096, device=dev)  # (fp32 matmul on another stream: keeps the chip busy beside the factorization)
t_end = time.time() + budget_s
calls = bad = 0
report = {}
outer = 0
while time.time() < t_end:
    outer += 1
    for M in (130, 512, 1000, 1024, 2048):
        # a NEW matrix every round through the same workspaces: a read of anything left in a cache by the previous launch shows
        g = torch.Generator().manual_seed(M + 7919 * outer)
        R = torch.randn(M, M + 3, dtype=torch.float64, generator=g)
        K = (R @ R.T / M + torch.eye(M, dtype=torch.float64)).to(dev)
        ref = None
        for budget in (0, 0, 3, 9, 64, 0):
            e = eng if budget == 0 else ggp_amd.HipEngine(own_context=True)
            if budget:
                e.set_option("cu_budget", budget)
            for rep in range(6):
                busy = rep % 2 == 1
                if busy:
                    with torch.cuda.stream(other):
                        for _ in range(3):
                            A @ A
                tr = torch.empty(n_tr, dtype=torch.float64, device=dev)
                linv, info = e.kuu_factor(K, trace_out=tr)
                torch.cuda.synchronize()
                calls += 1
                st = int(info.item())
                cur = (linv.clone(), tr[:1 + M].clone())
                if ref is None:
                    ref = cur
                ok = st == 0 and torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])
                if not ok:
                    bad += 1
                    print("BAD M", M, "budget", budget, "rep", rep, "busy", busy, "info", st, flush=True)
                report[M] = report.get(M, 0) + 1
print("potrf soak: %d calls (%s), bad %d" % (calls, ", ".join("M %d: %d" % kv for kv in sorted(report.items())), bad))
sys.exit(1 if bad else 0)
