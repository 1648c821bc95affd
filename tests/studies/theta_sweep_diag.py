"""Diagnostic behind tests/test_int8_theta_sweep.py: F from the integer-core contraction, from the fp64 contraction and from the
PyMC3-order CPU oracle over the theta range, with the status word of every cell (nothing raises).  One JSON line per cell."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ggp_amd  # noqa: E402
from oracle import vfe_oracle as O  # noqa: E402

N = int(os.environ.get("N", 200000))
D = 8
eng = ggp_amd.HipEngine()
torch.set_num_threads(os.cpu_count() or 1)
for M, dup in ((512, False), (512, True), (1024, True)):
    X, y, Z = bench.synth(N, M, D)
    if dup:
        Z[1::16] = Z[0::16][: Z[1::16].shape[0]]
    Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
    cb = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=eng)
    if os.environ.get("GUARD", "1") == "0":
        cb.streaming_tol = 0.0
    for ls in (0.2, 0.5, 1.0, 2.0, 5.0, 20.0):
        for sn in (0.01, 0.3, 3.0):
            if M == 1024 and sn == 3.0:
                continue
            row = {"M": M, "dup": dup, "ls": ls, "sn": sn}
            for mode, key in ((1, "i8"), (0, "f64")):
                eng.lib.sgp_set_contraction(mode)
                F, parts = cb.value(Zd, [ls] * D, 1.0, sn * sn, raise_on_fail=False)
                row["F_" + key] = F
                row["info_" + key] = parts.get("info", 0)
                row["used_" + key] = eng.lib.sgp_contraction_last()
                # the guard of DESIGN.md 4f (on unless GUARD=0): the estimate this evaluation carried, and how many evaluations of this
                # bound have been repeated in / sent directly to the whitened order so far
                row["estimate_" + key] = cb.last_estimate
                row["whitened_so_far_" + key] = cb.n_guard_reruns + cb.n_direct_whitened
            eng.lib.sgp_set_contraction(1)
            try:
                row["F_ref"] = float(O.vfe_pymc3_order_chunked(X, y, Z, torch.full((D,), ls, dtype=torch.float64), 1.0, sn, 1e-6))
            except Exception as e:  # noqa: BLE001
                row["F_ref"] = None
                row["ref_error"] = str(e)[:80]
            if row["F_ref"] is not None:
                row["err_i8_perN"] = abs(row["F_i8"] - row["F_ref"]) / N
                row["err_f64_perN"] = abs(row["F_f64"] - row["F_ref"]) / N
            print(json.dumps(row), flush=True)
