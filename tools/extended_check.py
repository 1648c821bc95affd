#!/usr/bin/env python3
"""The extended streaming order (sgp_suffstats_fwd_extended) against the whitened order over theta at C5: F per datum through both, the
streaming order's own error and estimate next to them, and the time of one extended evaluation.  One JSON line per theta."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

N = int(os.environ.get("ROWS", bench.N_TOTAL))
M = int(os.environ.get("M", bench.M_IND))
LEVEL = int(os.environ.get("LEVEL", 1))  # 1: 34 digit pairs, 2: 39
eng = ggp_amd.HipEngine()
X, y, Z = bench.synth(N, M, bench.DIM)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
cs = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng)
cs.streaming_tol = float("inf")
cw = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng, form="whitened")


def extended(ls, sf2, s2):
    Kuu = eng.kuu(Zd, ls, sf2, bench.JITTER, "rbf")
    linv, info = eng.kuu_factor(Kuu)
    packed = eng.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, "rbf", level=LEVEL)
    res = eng.bound(Kuu, packed, s2, N, kuu_linv=linv, kuu_info=info, whitened=True)
    return float(res["out"].cpu()[0]), int(res["info"].cpu()[0])


for ls_v in (2.0, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0, 8.0, 12.0):
    for sn in (0.3, 0.145, 0.05):
        ls = [ls_v] * bench.DIM
        Fw, _ = cw.value(Zd, ls, 1.0, sn * sn, raise_on_fail=False)
        Fs, ps = cs.value(Zd, ls, 1.0, sn * sn, raise_on_fail=False)
        Fe, ie = extended(ls, 1.0, sn * sn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            extended(ls, 1.0, sn * sn)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        grad_diff = grad_suite = grad_suite_fp64_formed = grad_suite_hi_only = None
        if os.environ.get("GRADS", "0") == "1":  # gradients: the explicit Phibar of the extended order against the factored pass 2 of the whitened one
            _, gw = cw.value_and_grad(Zd, ls, 1.0, sn * sn, want_gz=False, raise_on_fail=False)
            for dd, lo in ((True, True), (True, False), (False, False)):   # Phibar in double-double with / without its trailing word in pass 2 (round 6); by two fp64 products (rounds 4-5)
                cx = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng, form="extended")
                cx.extended_level = LEVEL
                cx.extended_dd_phibar, cx.extended_lo = dd, lo
                _, gx = cx.value_and_grad(Zd, ls, 1.0, sn * sn, want_gz=False, raise_on_fail=False)
                if gx.get("info", 0) == 0 and gw.get("info", 0) == 0:
                    a = torch.cat([gx["ls"], torch.tensor([gx["sf2"], gx["s2"]], dtype=torch.float64)])
                    b = torch.cat([gw["ls"], torch.tensor([gw["sf2"], gw["s2"]], dtype=torch.float64)])
                    # the parity suite's metric: lengthscale gradients against their largest component, sf2 / s2 against max(1, |ref|)
                    suite = max(float((gx["ls"] - gw["ls"]).abs().max() / gw["ls"].abs().max()),
                                abs(gx["sf2"] - gw["sf2"]) / max(1.0, abs(gw["sf2"])), abs(gx["s2"] - gw["s2"]) / max(1.0, abs(gw["s2"])))
                    if dd and lo:
                        grad_diff = float(((a - b).abs() / b.abs().clamp_min(1e-300)).max())
                        grad_suite = suite
                    elif dd:
                        grad_suite_hi_only = suite
                    else:
                        grad_suite_fp64_formed = suite
                del cx
        print(json.dumps({"N": N, "M": M, "ls": ls_v, "sig_n": sn, "estimate": cs.last_estimate, "grad_max_rel_diff": grad_diff, "grad_suite_metric": grad_suite, "grad_suite_metric_dd_leading_word_only": grad_suite_hi_only, "grad_suite_metric_fp64_formed_phibar": grad_suite_fp64_formed,
                          "err_streaming": abs(Fs - Fw) / N if ps.get("info", 0) == 0 else None,
                          "err_extended": abs(Fe - Fw) / N if ie == 0 else None, "info_extended": ie, "extended_ms": round(ms, 2), "level": LEVEL}), flush=True)
