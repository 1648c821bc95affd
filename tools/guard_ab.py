"""Same-process A/B of round 4's additions on the critical path: the streaming-order guard (tr(Kuu^-1) + estimate kernels) and the
conditioning gate (three small launches at the end of the K_uu chain), at C3 and at C5.  Alternating, three rounds."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()


def problem(N, d, M):
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
    return ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng), Z


for name, (N, d, M), reps in (("C3", (13279, 18, 512), 60), ("C5", (bench.N_TOTAL, bench.DIM, bench.M_IND), 12)):
    cb, Z = problem(N, d, M)
    ls = [2.0] * d
    for rnd in range(3):
        for guard, cond in ((1, 1), (0, 1), (1, 0), (0, 0)):
            cb.streaming_tol = 1e-9 if guard else 0.0
            eng.set_option("cond_limit", 1e13 if cond else 0.0)
            out = {"config": name, "round": rnd, "guard": guard, "cond_gate": cond}
            for label, fn in (("value_us", lambda: cb.value(Z, ls, 1.0, 0.09)), ("value_grad_us", lambda: cb.value_and_grad(Z, ls, 1.0, 0.09))):
                for _ in range(4):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                out[label] = round((time.perf_counter() - t0) / reps * 1e6, 1)
            print(json.dumps(out), flush=True)
    eng.set_option("cond_limit", 1e13)
    del cb
