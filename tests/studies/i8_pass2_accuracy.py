#!/usr/bin/env python3
"""CPU study (numpy, no GPU): what pass 2 on the integer matrix cores would cost in accuracy.

    Kbar[n, m] = sum_m' K'[n, m'] Pb[m', m]          Pb = the symmetrised Phibar of the streaming evaluation order

with K' in the digit planes pass 1 already forms (q = rint(K' 2^54), balanced base-256 digits) and every ROW of the symmetric Pb
scaled by a power of two and split into balanced digits the same way; kept: the digit pairs of depth da + db < S (depth 0 = the
most significant digit), S (S + 1) / 2 products.  Reports, against the fp64 product, the largest element error of Kbar relative to the
row scale and the relative error of the lengthscale / amplitude gradients of the K_fu path, for S = 3..7.

    python tests/studies/i8_pass2_accuracy.py [N] [M]         (default 16384 x 1024, bench.py's synthetic problem and theta)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from oracle import i8_digits_oracle as D  # noqa: E402
from oracle import vfe_oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else bench.M_IND
ls_v = float(os.environ.get("LS", bench.LS))
sn = float(os.environ.get("SIGN", bench.SN))
X, y, Z = bench.synth(N, M, bench.DIM)
ls = torch.full((bench.DIM,), ls_v, dtype=torch.float64)
sf2, s2 = bench.SF ** 2, sn ** 2
st = O.suffstats(X, y, Z, ls, sf2, 0)
Kuu = O.kuu(Z, ls, sf2, 1e-6, 0)
res = O.bound_from_stats(Kuu, st, s2, with_adjoints=True)
Pb = res["Phibar"].numpy() if isinstance(res, dict) else res.Phibar.numpy()
bbar = (res["bbar"] if isinstance(res, dict) else res.bbar).numpy()
Pb = 0.5 * (Pb + Pb.T)
Kp = O.kern(X, Z, ls, 1.0, 0).numpy()                       # K' (unit amplitude)
yv = y.numpy()

# digits of K' (pass 1's planes) and of the rows of Pb
a = D.digits(D.quantise(Kp))                                # a[p][N, M], p = 0 least significant .. 6
NB = 7
ex = np.ceil(np.log2(np.abs(Pb).max(axis=1) + 1e-300)).astype(np.int64)   # |Pb[m, :]| <= 2^ex[m]
scale = np.ldexp(1.0, (8 * NB - 2) - ex)                    # q_b = rint(Pb 2^(54 - ex)) : |q_b| <= 2^54
qb = np.rint(Pb * scale[:, None]).astype(np.int64)
sign = np.sign(qb)
bd = D.digits(np.abs(qb))                                   # digits of |q_b| ...
bd = [(d.astype(np.int64) * sign).astype(np.int16) for d in bd]  # ... carried with the sign (|digit| <= 128)

C_ref = Kp @ Pb                                             # fp64 product [n, m]: Kbar^T without the b-term
Xs, Zs = (X / ls).numpy(), (Z / ls).numpy()
r2 = ((Xs[:, None, :] - Zs[None, :, :]) ** 2).sum(-1) if N * M * bench.DIM < 3e8 else None


def grads(C):
    """lengthscale / amplitude gradients of the K_fu path from C = K' Pb (the kbar epilogue in numpy), RBF profile"""
    kbar = 2.0 * sf2 * C + yv[:, None] * bbar[None, :]
    g_sf2 = float((kbar * Kp).sum())
    dr2 = kbar * sf2 * (-0.5 * Kp)                           # dF/dr2
    g_ls = np.empty(bench.DIM)
    for j in range(bench.DIM):
        diff2 = (Xs[:, j:j + 1] - Zs[None, :, j]) ** 2
        g_ls[j] = float((dr2 * diff2).sum()) * (-2.0 / ls_v)
    return g_ls, g_sf2


g_ref = grads(C_ref)
rows = []
for S in (3, 4, 5, 6, 7):
    C = np.zeros_like(C_ref)
    for da in range(S):
        for db in range(S - da):
            pa, pb = 6 - da, NB - 1 - db
            prod = a[pa].astype(np.float64) @ bd[pb].astype(np.float64).T     # exact: |sum| < 2^53
            C += prod * np.ldexp(1.0, 8 * pa - 54) * (np.ldexp(1.0, 8 * pb) / scale)[None, :]
    g = grads(C)
    err = np.abs(C - C_ref)
    rows.append({"S": S, "pairs": S * (S + 1) // 2,
                 "max_elem_err_rel_rowmax": float((err / np.ldexp(1.0, ex)[None, :]).max()),
                 "max_elem_err_rel_Cmax": float(err.max() / np.abs(C_ref).max()),
                 "g_ls_rel_err": float(np.abs(g[0] - g_ref[0]).max() / np.abs(g_ref[0]).max()),
                 "g_sf2_rel_err": abs(g[1] - g_ref[1]) / abs(g_ref[1])})
    print(json.dumps(rows[-1]), flush=True)
# Round 6: ASYMMETRIC digit budgets.  The rounding of K' is a fresh random error in every data row (the gradient sums over n: it grows like
# sqrt(N)); the rounding of Pb is ONE perturbation that every row sees (it grows like N -- tests/studies/explicit_phibar_pass2.py).  So K' may
# keep fewer digits than Pb: SA digits of K', all seven of Pb, pairs of depth da + db < S.
if os.environ.get("ASYM", "1") == "1":
    for SA, S in ((2, 7), (3, 7), (3, 8), (3, 9), (4, 7), (4, 8), (4, 9), (4, 10), (5, 9)):
        C = np.zeros_like(C_ref)
        npairs = 0
        for da in range(SA):
            for db in range(NB):
                if da + db >= S:
                    continue
                pa, pb = 6 - da, NB - 1 - db
                prod = a[pa].astype(np.float64) @ bd[pb].astype(np.float64).T
                C += prod * np.ldexp(1.0, 8 * pa - 54) * (np.ldexp(1.0, 8 * pb) / scale)[None, :]
                npairs += 1
        g = grads(C)
        err = np.abs(C - C_ref)
        print(json.dumps({"digits_of_Kprime": SA, "depth_cut": S, "pairs": npairs,
                          "max_elem_err_rel_Cmax": float(err.max() / np.abs(C_ref).max()),
                          "g_ls_rel_err": float(np.abs(g[0] - g_ref[0]).max() / np.abs(g_ref[0]).max()),
                          "g_sf2_rel_err": abs(g[1] - g_ref[1]) / abs(g_ref[1])}), flush=True)
print(json.dumps({"N": N, "M": M, "ls": ls_v, "sig_n": sn, "Pb_absmax": float(np.abs(Pb).max()), "C_absmax": float(np.abs(C_ref).max()),
                  "cancellation_rowmax_over_C": float(np.ldexp(1.0, ex).max() / np.abs(C_ref).max())}))
