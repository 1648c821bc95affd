#!/usr/bin/env python3
"""One-off parity check at the full headline size (C5: N = 1 000 000, d = 8, M = 1024): the HIP bound against the CPU
oracle (PyMC3 op order, chunked) on ALL rows -- bench.py's cpu_baseline only times a 100k-row sample.  Takes ~2 minutes of
host time on the GPU box.  Prints one JSON object (kept under profiles/).  Test infrastructure: imports oracle/."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (same synthetic generator as the benchmark)
import ggp_amd  # noqa: E402
from oracle import vfe_oracle as O  # noqa: E402


def main():
    N, M, d = bench.N_TOTAL, bench.M_IND, bench.DIM
    X, y, Z = bench.synth(N, M, d)
    eng = ggp_amd.HipEngine()
    cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=bench.JITTER, engine=eng)
    F_hip, parts = cb.value(Z.to(eng.device), [bench.LS] * d, bench.SF ** 2, bench.SN ** 2)
    torch.set_num_threads(os.cpu_count())
    t0 = time.time()
    F_cpu = O.vfe_pymc3_order_chunked(X, y, Z, [bench.LS] * d, bench.SF, bench.SN, bench.JITTER)
    secs = time.time() - t0
    out = {"workload": "C5 N=%d d=%d M=%d, all rows" % (N, d, M), "F_hip": F_hip, "F_cpu_oracle": F_cpu,
           "abs_diff": abs(F_hip - F_cpu), "rel_diff": abs(F_hip - F_cpu) / abs(F_cpu),
           "diff_per_datum": abs(F_hip - F_cpu) / N, "tolerance_north_star_rel": 1e-8,
           "cpu_oracle_seconds": secs, "cpu_threads": os.cpu_count()}
    # gradients (the leapfrog path) against torch autograd on the PyMC3-order graph, on the first GR rows
    GR = 100_000
    Xg, yg = X[:GR], y[:GR]
    cbg = ggp_amd.CollapsedBound(Xg.to(eng.device), yg.to(eng.device), jitter=bench.JITTER, engine=eng)
    Fg, g = cbg.value_and_grad(Z.to(eng.device), [bench.LS] * d, bench.SF ** 2, bench.SN ** 2, want_gz=True)
    t0 = time.time()
    ref = O.grads_autograd(Xg, yg, Z, [bench.LS] * d, bench.SF ** 2, bench.SN ** 2, bench.JITTER)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    out["gradients"] = {"rows": GR, "rel_err_ls": rel(g["ls"], ref["g_ls"]), "rel_err_Z": rel(g["Z"].cpu(), ref["g_Z"]),
                        "rel_err_sf2": abs(g["sf2"] - ref["g_sf2"]) / abs(ref["g_sf2"]),
                        "rel_err_s2": abs(g["s2"] - ref["g_s2"]) / abs(ref["g_s2"]),
                        "rel_err_F": abs(Fg - ref["F"]) / abs(ref["F"]), "autograd_seconds": time.time() - t0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
