#!/usr/bin/env python3
"""Critical path of the chain-workgroup Cholesky (csrc/sgp_potrf_chain.hpp), block column by block column.  Needs a library built
with -DSGP_POTRF_STAMPS:
    SGP_EXTRA_HIPCC_FLAGS=-DSGP_POTRF_STAMPS python3 -c "import sys; sys.path.insert(0, 'generalised-gaussian-processes_amd'); import build; build.build_library(force=True)"
    python3 tools/potrf_chain_phases.py [M]
Stamps (s_memrealtime, 100 MHz) per step: 0 chain start | 1-4 end of the pivot chain of panel 0-3 | 5 S-wave 0 has S = A - US (the prep
item's flag seen, loaded) | 9 S-wave 0 has UD(j+1) | 6 last panel step of the solve done | 7 last rank-16 update done | 8 step barrier."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

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
nb = (M + 127) // 128 * 2
t0 = buf[0]
for j in range(nb):
    v = [(buf[j * 16 + k] - t0) / 100.0 for k in range(10)]
    s = "step %2d: start %7.1f | panels +%.1f +%.1f +%.1f +%.1f" % (j, v[0], v[1] - v[0], v[2] - v[0], v[3] - v[0], v[4] - v[0])
    if j + 1 < nb:
        s += " | S ready +%.1f  solve done +%.1f  update done +%.1f" % (v[5] - v[0], v[6] - v[0], v[7] - v[0])
    s += " | panel 3 written back +%.1f | barrier +%.1f" % (v[9] - v[0], v[8] - v[0])
    print(s)
    if 1 <= j < nb - 1:  # the fused item of column j - 1 (tile (j+1, j-1), then US(j)), relative to this step's start
        w = [(buf[(j - 1) * 16 + k] - t0) / 100.0 - v[0] for k in range(10, 16)]
        print("         US(%d) item: start %+.1f  updates done %+.1f  panel 3 seen %+.1f  X solved %+.1f  X(j-1) flag seen %+.1f  product done %+.1f" % ((j,) + tuple(w)))
end = (buf[(nb - 1) * 16 + 8] - t0) / 100.0
print("total %.1f us for %d block columns = %.2f us per column" % (end, nb, end / nb))
