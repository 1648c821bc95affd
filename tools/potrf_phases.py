#!/usr/bin/env python3
"""Critical path of the dataflow Cholesky, block column by block column (needs a library built with -DSGP_POTRF_STAMPS:
    SGP_EXTRA_HIPCC_FLAGS=-DSGP_POTRF_STAMPS python3 -c "import sys; sys.path.insert(0, 'generalised-gaussian-processes_amd'); import build; build.build_library(force=True)"
    python3 tools/potrf_phases.py [M])"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

LAB = ["X of the previous step seen", "updates done, waiting for L(j,j)", "L(j,j) seen", "solve done", "X X^T applied", "X published",
       "factor start", "factor done", "tile published"]
eng = ggp_amd.HipEngine()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(M)
R = torch.randn(M, M + 64, dtype=torch.float64, generator=g)
A = (R @ R.T / M + torch.eye(M, dtype=torch.float64)).to(eng.device)
for _ in range(5):
    eng.chol_lower(A)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 16))()
eng.lib.sgp_debug_potrf_stamps.argtypes = [C.c_void_p]
rc = eng.lib.sgp_debug_potrf_stamps(C.cast(buf, C.c_void_p))
assert rc == 0
nb = M // 64
t0 = buf[0 * 16 + 6]
prev_pub = None
for j in range(nb):
    row = [buf[j * 16 + k] for k in range(9)]
    rel = [(v - t0) / 100.0 for v in row]
    if j == 0:
        print("step %2d: factor start %.1f done %.1f published %.1f" % (j, rel[6], rel[7], rel[8]))
    else:
        print("step %2d: " % j + "  ".join("%s %.1f" % (LAB[k], rel[k]) for k in range(9))
              + "   | previous tile published -> this factor start: %.1f us" % (rel[6] - prev_pub))
    prev_pub = rel[8]
print("total %.1f us for %d block columns = %.2f us per column" % ((buf[(nb - 1) * 16 + 8] - t0) / 100.0, nb, (buf[(nb - 1) * 16 + 8] - t0) / 100.0 / nb))
