#!/usr/bin/env python3
"""Wall time of one value / one value+gradient evaluation at C5 in the two evaluation orders (form="streaming" / "whitened"), and the
relative difference of their gradients.  The whitened order is what the streaming guard falls back to: at C5's trained
hyper-parameters NUTS spends most of its leapfrogs there, so its cost is a number of its own.  ROWS / REPS / LS / SIGN override."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

N = int(os.environ.get("ROWS", bench.N_TOTAL))
reps = int(os.environ.get("REPS", 5))
ls = float(os.environ.get("LS", bench.LS))
sn = float(os.environ.get("SIGN", bench.SN))
eng = ggp_amd.HipEngine()
X, y, Z = bench.synth(N, bench.M_IND, bench.DIM)
Zd = Z.to(eng.device)
lsv = torch.full((bench.DIM,), ls, dtype=torch.float64)
out = {"rows": N, "ls": ls, "sig_n": sn, "fully_factored": os.environ.get("SGP_BWD_FULLY_FACTORED", "0")}
grads = {}
forms = os.environ.get("FORMS", "streaming,whitened").split(",")
for form in forms:
    b = ggp_amd.CollapsedBound(X, y, "rbf", engine=eng, form=form)
    b.streaming_tol = float("inf")
    for mode in ("value", "value_and_grad"):
        fn = (lambda: b.value(Zd, lsv, bench.SF ** 2, sn ** 2)) if mode == "value" else (lambda: b.value_and_grad(Zd, lsv, bench.SF ** 2, sn ** 2))
        r = fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        out["%s_%s_ms" % (form, mode)] = round(sorted(ts)[len(ts) // 2], 2)
    grads[form] = (float(r[0]), [torch.as_tensor(r[1][k], dtype=torch.float64).detach().cpu().reshape(-1) for k in ("ls", "sf2", "s2")])
    del b
if len(forms) < 2:
    print(json.dumps(out))
    sys.exit(0)
out["value_rel_diff"] = abs(grads["streaming"][0] - grads["whitened"][0]) / abs(grads["whitened"][0])
gs, gw = torch.cat(grads["streaming"][1]), torch.cat(grads["whitened"][1])
out["grad_rel_diff"] = float((gs - gw).norm() / gw.norm())
print(json.dumps(out))
