"""GPU study: gradients of the whitened order on ill-conditioned 1-D problems (chunked routine and streaming layout) against torch autograd
through the PyMC3-order graph, for the two product orders of the factored pass 2 (SGP_BWD_FULLY_FACTORED=0 / 1, one process each):
    SGP_BWD_FULLY_FACTORED=0 python tests/studies/factored_pass2_orders.py ; SGP_BWD_FULLY_FACTORED=1 python tests/studies/factored_pass2_orders.py"""
import json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ggp_amd
from oracle import vfe_oracle as O
eng = ggp_amd.HipEngine()
out = {"fully": os.environ.get("SGP_BWD_FULLY_FACTORED", "0"), "cases": []}
for N, M, ls, sf2, s2 in ((6340, 128, 3.0, 1.3, 0.01), (6340, 200, 3.0, 1.3, 0.01), (6340, 200, 6.0, 1.0, 0.0025), (20000, 256, 4.0, 1.0, 0.01),
                          (6340, 200, 3.0, 300.0, 0.01), (6340, 200, 3.0, 3000.0, 0.01)):
    g = torch.Generator().manual_seed(3)
    X = torch.linspace(0, 52.8, N, dtype=torch.float64)[:, None]
    y = torch.sin(X[:, 0] * 2 * math.pi) * 0.3 + 0.04 * X[:, 0] + 0.05 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[torch.linspace(0, N - 1, M).round().long()].clone()
    lst = torch.tensor([ls], dtype=torch.float64)
    ref = O.grads_autograd(X, y, Z, lst, sf2, s2, 1e-6)
    cond = float(torch.linalg.cond(O.kuu(Z, lst, sf2, 1e-6)))
    row = {"N": N, "M": M, "ls": ls, "cond": cond}
    for work, key in ((1 << 40, "chunked"), (0, "rows")):
        cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng, form="whitened")
        cb.whitened_rows_min_work = work
        cb.fused = False
        F, gr = cb.value_and_grad(Z.to(eng.device), [ls], sf2, s2, want_gz=True)
        row[key] = {"F_rel": abs(F - float(ref["F"])) / abs(float(ref["F"])),
                    "g_ls_rel": float((gr["ls"] - ref["g_ls"]).abs().max() / ref["g_ls"].abs().max()),
                    "g_sf2_rel": abs(gr["sf2"] - ref["g_sf2"]) / abs(ref["g_sf2"]), "g_s2_rel": abs(gr["s2"] - ref["g_s2"]) / abs(ref["g_s2"]),
                    "g_Z_rel": float((gr["Z"].cpu() - ref["g_Z"]).abs().max() / ref["g_Z"].abs().max())}
    out["cases"].append(row)
print(json.dumps(out))
