#!/usr/bin/env python3
"""CPU study (numpy, no GPU) for DESIGN.md section 8's next lever: how much of the streaming order's error goes away when Phi = K_uf K_fu
and the triple product W = L^-1 Phi L^-T carry 11 more bits (x87 long double, 64-bit significand, as a stand-in for the double-double
Phi the integer contraction could deliver and a double-double tail), everything else -- chol(K_uu), its explicit inverse, B, chol(B), the
solves -- staying fp64 exactly as the library has them.

    F_ref   the PyMC3 op order (oracle.vfe_pymc3_order)            F_stream  fp64 Phi, fp64 triple product (today's streaming order)
    F_ext   long-double Phi and triple product, W rounded to fp64, the same fp64 tail behind it

    python tests/studies/extended_streaming_order.py [N] [M]        (default 20000 x 256, bench.py's synthetic problem)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from oracle import vfe_oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
LD = np.longdouble
X, y, Z = bench.synth(N, M, bench.DIM)
yv = y.numpy()


def tail(W, u, yy, kappa, s2):
    """F from the whitened statistics in fp64 (the library's sgp_bound_from_whitened_stats, restated)"""
    B = np.eye(M) + W / s2
    LB = np.linalg.cholesky(B)
    c = np.linalg.solve(LB, u) / s2
    logdet = 2.0 * np.log(np.diag(LB)).sum()
    return -0.5 * (N * np.log(2 * np.pi) + N * np.log(s2) + logdet + yy / s2 - c @ c) - 0.5 * (kappa - np.trace(W)) / s2


for ls_v, sn in ((2.0, 0.3), (3.0, 0.3), (3.5, 0.145), (5.0, 0.145), (8.0, 0.145), (20.0, 0.3)):
    ls = torch.full((bench.DIM,), ls_v, dtype=torch.float64)
    s2 = sn * sn
    F_ref = float(O.vfe_pymc3_order(X, y, Z, ls, 1.0, sn, 1e-6))
    K = O.kern(X, Z, ls, 1.0, 0).numpy()                       # N x M, fp64 kernel values (the data both orders start from)
    Kuu = O.kuu(Z, ls, 1.0, 1e-6, 0).numpy()
    L = np.linalg.cholesky(Kuu)
    Linv = np.linalg.solve(L, np.eye(M))                       # the explicit inverse the library forms (fp64)
    b = K.T @ yv
    yy, kappa = float(yv @ yv), float(N)
    # today's streaming order
    Phi = K.T @ K
    W = Linv @ Phi @ Linv.T
    u = Linv @ b
    row = {"N": N, "M": M, "ls": ls_v, "sig_n": sn, "tr_Kuu_inv": float((Linv ** 2).sum())}
    try:
        row["err_stream_per_datum"] = abs(tail(0.5 * (W + W.T), u, yy, kappa, s2) - F_ref) / N
    except np.linalg.LinAlgError:
        row["err_stream_per_datum"] = None                      # B not positive definite
    # 11 more bits in Phi and in the triple product (and in b, u)
    Kl, Ll = K.astype(LD), Linv.astype(LD)
    Phil = Kl.T @ Kl
    Wl = (Ll @ Phil @ Ll.T)
    ul = Ll @ (Kl.T @ yv.astype(LD))
    try:
        row["err_ext_per_datum"] = abs(tail(np.asarray(0.5 * (Wl + Wl.T), dtype=np.float64), np.asarray(ul, dtype=np.float64), yy, kappa, s2) - F_ref) / N
        # ... and with u = L^-1 b left in fp64 (is the double-double b needed?)
        row["err_ext_fp64_u_per_datum"] = abs(tail(np.asarray(0.5 * (Wl + Wl.T), dtype=np.float64), u, yy, kappa, s2) - F_ref) / N
    except np.linalg.LinAlgError:
        row["err_ext_per_datum"] = None
    row["estimate_per_datum"] = 2.0 ** -53 * float(np.diag(Phi).max()) * row["tr_Kuu_inv"] / (s2 * N)
    print(json.dumps(row), flush=True)
