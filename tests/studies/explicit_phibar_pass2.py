#!/usr/bin/env python3
"""CPU study (numpy, no GPU) for VERDICT r5 next-1: WHICH quantity of the extended order's explicit-Phibar pass 2 needs more bits?

The extended order (tier 1) delivers the VALUE of the bound at 20 ms where the whitened order needs 40, but its gradient -- pass 2 from the
explicit Phibar = L^-T C L^-1 / (2 s2), whose cond(K_uu)-sized entries cancel in Kbar = 2 K Phibar -- holds 1e-6 only up to 3 x the guard's
tolerance, so a parity-grade leapfrog in the guarded regime pays T = K' L^-T and T^T T of the whitened order (74 instead of ~55 ms at C5).
Three candidate error sources, separated here against an x87 80-bit yardstick (everything from X, Z, theta in long double):

    formation   Phibar formed by two fp64 products from the fp64 L^-1 and C          (today)
    rounding    Phibar correctly formed (long double here, double-double on the GPU), then rounded to ONE fp64 word
    accumulation the fp64 accumulation of K' Phibar itself

Variants of the K_fu path (the K_uu path, the tail and the b-term are the same fp64 numbers for all of them):

    v0_today            fp64-formed Phibar, fp64 product                              (today's tier 1)
    v1_dd_formed_hi     long-double-formed Phibar rounded to fp64, fp64 product       (double-double formation, nothing else)
    v2_hi_plus_lo_f32   v1 + an fp32 product with the low word Phibar_lo              (VERDICT's proposal)
    v2_hi_plus_lo_bf16  v1 + a bf16-operand / fp32-accumulate product with Phibar_lo
    v3_ld_product       long-double-formed Phibar, long-double product               (exact accumulation: what upstream fp64 errors leave)
    v4_fp64_formed_ld   fp64-formed Phibar, long-double product                       (formation error alone)
    v5_factored_fp64    ((K' L^-T)(C / s2)) L^-1 / 2 in fp64                          (today's tier 2 = what passes 1e-6 on the GPU)

Metric = the GPU suite's: max_j |g_ls[j] - truth| / max(1, max_j |truth|), and |g_sf2 - truth| / max(1, |truth|), on the TOTAL gradient.

    python tests/studies/explicit_phibar_pass2.py [N] [M] [cells]      cells: comma list of  iso<ls>:<sig_n>  |  trained  |  trained2
    (default 16384 x 512; M = 1024 and N up to 262144 were run for DESIGN.md: profiles/r06_explicit_phibar_pass2_study.jsonl)
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from oracle import vfe_extended as E  # noqa: E402
from oracle import vfe_oracle as O  # noqa: E402

LD = np.longdouble
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
CELLS = (sys.argv[3] if len(sys.argv) > 3 else "iso3.5:0.145,iso5:0.145,trained").split(",")
CHUNK = int(os.environ.get("CHUNK", 2048))
PROCS = int(os.environ.get("PROCS", os.cpu_count() or 1))
TRAINED = {"trained": ([3.75, 2.61, 3.38, 5.50, 3.49, 3.17, 2.64, 3.65], 0.145),       # tools/extended_grad_check.py: C5's trained ARD theta
           "trained2": ([4.87, 2.27, 7.04, 6.39, 7.18, 3.35, 2.31, 6.49], 0.144)}
D = bench.DIM
G = {}   # arrays shared with the forked workers


def bf16(a):
    """round-to-nearest-even to 8 significant bits, returned as float32"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def contract(C, Kp, Xs, Zs, ls, yb, dtype):
    """gradient sums of the K_fu path from C = K' Phibar (sf2 = 1): kbar = 2 C + y bbar^T ; g_sf2 = sum kbar k ; g_ls_j = sum kbar k diff_j^2 / ls_j"""
    kb = (2 * C.astype(dtype) + yb.astype(dtype)) * Kp.astype(dtype)
    out = np.zeros(D + 1, dtype=LD)
    for j in range(D):
        diff = Xs[:, j:j + 1].astype(dtype) - Zs[None, :, j].astype(dtype)
        out[j] = LD((kb * diff * diff).sum(dtype=dtype)) / LD(ls[j])
    out[D] = LD(kb.sum(dtype=dtype))
    return out


def pass_W(rng):
    """truth, pass A: partial W = sum_n a_n a_n^T, u = sum a_n y_n in long double (a_n = L^-1 k_n, long-double kernels and factor)"""
    lo, hi = rng
    Kl = E.stationary_k(G["X"][lo:hi], G["Z"], G["ls"], 1.0, 0)            # c x M long double
    T = Kl @ G["Linv_t"].T
    return T.T @ T, T.T @ G["y"][lo:hi].astype(LD)


def pass_g(rng):
    lo, hi = rng
    X, Z, ls, y = G["X"][lo:hi], G["Z"], G["ls"], G["y"][lo:hi]
    Xs, Zs = X / ls, Z / ls
    res = {}
    # truth: long double kernels, long-double Phibar / bbar of the long-double tail
    Kl = E.stationary_k(X, Z, ls, 1.0, 0)
    Xl, Zl = X.astype(LD) / ls.astype(LD), Z.astype(LD) / ls.astype(LD)
    res["truth"] = contract(Kl @ G["Pb_t"], Kl, Xl, Zl, ls, np.outer(y.astype(LD), G["bbar_t"]), LD)
    # the library's inputs: fp64 kernel values
    Kp = O.kern(torch.from_numpy(X), torch.from_numpy(Z), torch.from_numpy(ls), 1.0, 0).numpy()
    yb = np.outer(y, G["bbar"])
    Kpl = Kp.astype(LD)
    f32 = np.float32
    prods = {
        "v0_today": Kp @ G["Pb64"],
        "v1_dd_formed_hi": Kp @ G["Pb_hi"],
        "v3_ld_product": Kpl @ G["Pb_l"],
        "v4_fp64_formed_ld": Kpl @ G["Pb64"].astype(LD),
        "v5_factored_fp64": 0.5 * (((Kp @ G["Linv"].T) @ (G["C"] / G["s2"])) @ G["Linv"]),
    }
    prods["v2_hi_plus_lo_f32"] = prods["v1_dd_formed_hi"] + (Kp.astype(f32) @ G["Pb_lo"].astype(f32)).astype(np.float64)
    prods["v2_hi_plus_lo_bf16"] = prods["v1_dd_formed_hi"] + (bf16(Kp) @ bf16(G["Pb_lo"])).astype(np.float64)
    for k, C in prods.items():
        res[k] = contract(C, Kp, Xs, Zs, ls, yb, LD)
    # the candidate with the epilogue's own fp64 sums as well
    res["v2_hi_plus_lo_f32_fp64_epilogue"] = contract(prods["v2_hi_plus_lo_f32"], Kp, Xs, Zs, ls, yb, np.float64)
    res["v0_today_fp64_epilogue"] = contract(prods["v0_today"], Kp, Xs, Zs, ls, yb, np.float64)
    res["_cancel"] = np.array([np.abs(G["Pb64"]).max(), np.abs(prods["v0_today"]).max()], dtype=LD)
    return res


def kuu_path(Kuubar, Z, ls, dtype):
    """the K_uu path of the gradient (sf2 = 1): g_sf2 += sum Kuubar Ku ; g_ls_j += sum Kuubar Ku diff_j^2 / ls_j"""
    Zs = Z.astype(dtype) / ls.astype(dtype)
    Ku = E.stationary_k(Z, Z, ls, 1.0, 0).astype(dtype) if dtype is LD else O.kuu(torch.from_numpy(Z), torch.from_numpy(ls), 1.0, 0.0, 0).numpy()
    kb = Kuubar.astype(dtype) * Ku
    out = np.zeros(D + 1, dtype=LD)
    for j in range(D):
        diff = Zs[:, j:j + 1] - Zs[None, :, j]
        out[j] = LD((kb * diff * diff).sum()) / LD(ls[j])
    out[D] = LD(kb.sum())
    return out


def ld_inverse_lower(L):
    """explicit inverse of a lower-triangular long-double matrix (forward substitution on the identity)"""
    return E.solve_lower(L, np.eye(L.shape[0], dtype=LD))


def run_cell(name, pool):
    if name in TRAINED:
        lsv, sn = TRAINED[name]
    else:
        a, b = name[3:].split(":")
        lsv, sn = [float(a)] * D, float(b)
    ls = np.array(lsv, dtype=np.float64)
    s2 = sn * sn
    X, y, Z = G["X"], G["y"], G["Z"]
    t0 = time.time()
    rngs = [(lo, min(lo + CHUNK, N)) for lo in range(0, N, CHUNK)]
    G["ls"] = ls
    # ---------------- truth: the whole whitened pipeline in long double (explicit long-double inverses: error 2^-64 cond(L))
    Kuu_t = E.stationary_k(Z, Z, ls, 1.0, 0) + LD(bench.JITTER) * np.eye(M, dtype=LD)
    L_t = E.cholesky(Kuu_t)
    G["Linv_t"] = ld_inverse_lower(L_t)
    # ---------------- the library's fp64 factor and explicit inverse
    Kuu = O.kuu(torch.from_numpy(Z), torch.from_numpy(ls), 1.0, bench.JITTER, 0).numpy()
    L = np.linalg.cholesky(Kuu)
    import scipy.linalg as sl
    Linv = sl.solve_triangular(L, np.eye(M), lower=True)
    G["Linv"], G["s2"] = Linv, s2
    return_parts = pool.map(pass_W, rngs)
    W_t = sum(p[0] for p in return_parts)
    u_t = sum(p[1] for p in return_parts)
    I = np.eye(M, dtype=LD)
    B_t = I + 0.5 * (W_t + W_t.T) / LD(s2)
    LB_t = E.cholesky(B_t)
    LBinv_t = ld_inverse_lower(LB_t)
    Binv_t = LBinv_t.T @ LBinv_t
    g_t = Binv_t @ u_t
    C_t = I - Binv_t - np.outer(g_t, g_t) / LD(s2) ** 2
    G["Pb_t"] = G["Linv_t"].T @ (C_t / (2 * LD(s2))) @ G["Linv_t"]
    Kuubar_t = -0.5 * (G["Linv_t"].T @ (B_t + Binv_t - 2 * I + np.outer(g_t, g_t) / LD(s2) ** 2) @ G["Linv_t"])
    G["bbar_t"] = (G["Linv_t"].T @ g_t) / LD(s2) ** 2
    # ---------------- the extended order's tail in fp64: W, u to (nearly) full precision -- tier 1 gets them from exact sums and a
    # double-double sandwich -- rounded to fp64, then B, chol(B), B^-1, g, C, Kuubar, bbar as the library forms them (fp64, explicit inverses)
    Linv_l = Linv.astype(LD)
    # (the statistics of the fp64 kernel values through the fp64 L^-1 in long double = what the double-double sandwich returns)
    # cheap route: W64 = Linv Phi Linv^T with Phi accumulated in long double would need another N M^2 long-double pass; the truth's W
    # differs from it by the rounding of K' and L^-1 only (relative 1e-16 kappa(L) ~ 1e-12): use the truth's W rounded to fp64
    W = np.asarray(0.5 * (W_t + W_t.T), dtype=np.float64)
    u = np.asarray(u_t, dtype=np.float64)
    B = np.eye(M) + W / s2
    LB = np.linalg.cholesky(B)
    LBinv = sl.solve_triangular(LB, np.eye(M), lower=True)
    Binv = LBinv.T @ LBinv
    g = Binv @ u
    C = np.eye(M) - Binv - np.outer(g, g) / s2 ** 2
    C = 0.5 * (C + C.T)
    G["C"] = C
    Pb64 = (Linv.T @ (C / (2 * s2))) @ Linv
    G["Pb64"] = 0.5 * (Pb64 + Pb64.T)
    Pb_l = Linv_l.T @ (C.astype(LD) / (2 * LD(s2))) @ Linv_l
    G["Pb_l"] = 0.5 * (Pb_l + Pb_l.T)
    G["Pb_hi"] = np.asarray(G["Pb_l"], dtype=np.float64)
    G["Pb_lo"] = np.asarray(G["Pb_l"] - G["Pb_hi"].astype(LD), dtype=np.float64)
    Kuubar = -0.5 * ((Linv.T @ (B + Binv - 2 * np.eye(M) + np.outer(g, g) / s2 ** 2)) @ Linv)
    Kuubar = 0.5 * (Kuubar + Kuubar.T)
    G["bbar"] = (Linv.T @ g) / s2 ** 2
    t1 = time.time()
    # pool workers were forked before G was filled for this cell: a fresh pool per pass keeps it simple
    with mp.get_context("fork").Pool(PROCS) as p2:
        parts = p2.map(pass_g, rngs)
    keys = [k for k in parts[0] if not k.startswith("_")]
    sums = {k: sum(p[k] for p in parts) for k in keys}
    gu_t = kuu_path(Kuubar_t, Z, ls, LD)
    gu = kuu_path(Kuubar, Z, ls, LD)
    kappabar_N = LD(-1.0 / (2 * s2)) * N
    truth = sums["truth"] + gu_t
    truth[D] += kappabar_N
    row = {"cell": name, "N": N, "M": M, "ls": lsv, "sig_n": sn,
           "estimate": float(2.0 ** -53 * float(np.diag(np.asarray(W_t, dtype=np.float64)).max()) * 0 + 0),  # filled below
           "Phibar_absmax": float(max(p["_cancel"][0] for p in parts)), "KPhibar_absmax": float(max(p["_cancel"][1] for p in parts)),
           "g_ls_truth": [float(v) for v in truth[:D]], "g_sf2_truth": float(truth[D]),
           "g_ls_fu_over_total": float(np.abs(sums["truth"][:D]).max() / max(1.0, float(np.abs(truth[:D]).max()))),
           "seconds": None}
    # the guard's estimate at this theta: 2^-53 max Phi_ii tr(Kuu^-1) / (s2 N)
    Kp_diag = None
    phi_max = 0.0
    for lo, hi in rngs:
        Kp = O.kern(torch.from_numpy(X[lo:hi]), torch.from_numpy(Z), torch.from_numpy(ls), 1.0, 0).numpy()
        Kp_diag = (Kp * Kp).sum(0) if Kp_diag is None else Kp_diag + (Kp * Kp).sum(0)
    phi_max = float(Kp_diag.max())
    row["estimate"] = 2.0 ** -53 * phi_max * float((Linv ** 2).sum()) / (s2 * N)
    sl_, sf_ = max(1.0, float(np.abs(truth[:D]).max())), max(1.0, abs(float(truth[D])))
    for k in keys:
        if k == "truth":
            continue
        tot = sums[k] + gu
        tot[D] += kappabar_N
        row[k] = {"g_ls": float(np.abs(tot[:D] - truth[:D]).max() / sl_), "g_sf2": float(abs(tot[D] - truth[D]) / sf_)}
    # how much of the floor is the K_uu path's own fp64 formation: the long-double K_fu path with the fp64 K_uu path
    tot = sums["truth"] + gu
    tot[D] += kappabar_N
    row["truth_fu_with_fp64_kuu_path"] = {"g_ls": float(np.abs(tot[:D] - truth[:D]).max() / sl_), "g_sf2": float(abs(tot[D] - truth[D]) / sf_)}
    row["seconds"] = [round(t1 - t0, 1), round(time.time() - t1, 1)]
    print(json.dumps(row), flush=True)


def main():
    X, y, Z = bench.synth(N, M, D)
    G["X"], G["y"], G["Z"] = X.numpy(), y.numpy(), Z.numpy()
    torch.set_num_threads(1)
    for name in CELLS:
        # (pass_W needs G["ls"], G["Linv_t"] of THIS cell: the pool is forked inside run_cell's first map through a lazy wrapper)
        run_cell(name, _LazyPool())


class _LazyPool:
    """forks its workers at the first map(), i.e. after the caller has filled G for this pass"""

    def map(self, fn, it):
        with mp.get_context("fork").Pool(PROCS) as p:
            return p.map(fn, it)


if __name__ == "__main__":
    main()
