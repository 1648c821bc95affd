"""The guarded regime anchored to the CPU oracle at the FULL size of C5.

NUTS at C5's trained hyper-parameters (reference models/bayesian_sgpr_hmc.py:60-78,160-180: `train_fixed_model` samples at the
trained Z / theta) spends 93-98 % of its leapfrogs where the streaming order's error estimate is above the tolerance
(profiles/r04_experiment_large_scale.json: ARD lengthscales 2.3 ... 7.2, sig_n = 0.144).  Until this file the evidence there was the
library's own whitened order (tools/extended_check.py, tools/guard_calibration.py); here it is the oracle:

 * N = 1 000 000, M = 1024, the trained ARD theta: form="auto" must leave the streaming order, land in the EXTENDED order (asserted)
   and agree with oracle.vfe_pymc3_order_chunked on all rows to 1e-8 per datum (north_star's tolerance);
 * the first 100 000 rows at M = 1024: value + gradient in the whitened order's rows layout with the T = K' L^-T hand-over to
   sgp_suffstats_bwd_factored_ex (asserted to be what ran) against torch autograd through the PyMC3-order graph to 1e-6, and the
   extended order's explicit-Phibar gradient inside its range against the same.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

# mean of the 30 draws of profiles/r04_experiment_large_scale.json (hmc.ls_mean, hmc.sig_n_mean); sig_f is not recorded there: 1.0
LS_TRAINED = [4.870895252562722, 2.274348615181124, 7.035384773166531, 6.388168428424034, 7.176420862837876, 3.3523772450641136,
              2.314383327914714, 6.492694463809999]
SN_TRAINED = 0.14415221312756948
SF_TRAINED = 1.0


def test_c5_trained_theta_value_on_all_rows_lands_in_the_extended_order(engine):
    import bench
    import ggp_amd
    from oracle import vfe_oracle as O
    N, M, d = bench.N_TOTAL, bench.M_IND, bench.DIM
    X, y, Z = bench.synth(N, M, d)
    cb = ggp_amd.CollapsedBound(X.to(engine.device), y.to(engine.device), jitter=bench.JITTER, engine=engine)
    F_hip, parts = cb.value(Z.to(engine.device), LS_TRAINED, SF_TRAINED ** 2, SN_TRAINED ** 2)
    est = cb.last_estimate
    assert est is not None and est > cb.streaming_tol, ("the trained theta must trip the streaming guard", est)
    assert cb.n_guard_reruns == 1 and cb.n_extended == 1, (cb.n_guard_reruns, cb.n_extended, est)
    # a second evaluation goes there directly (no wasted streaming attempt) and returns the same bits
    F_again, _ = cb.value(Z.to(engine.device), LS_TRAINED, SF_TRAINED ** 2, SN_TRAINED ** 2)
    assert cb.n_extended == 2 and F_again == F_hip
    del cb
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    F_cpu = O.vfe_pymc3_order_chunked(X, y, Z, LS_TRAINED, SF_TRAINED, SN_TRAINED, bench.JITTER)
    assert abs(F_hip - F_cpu) / N < 1e-8, (F_hip, F_cpu, abs(F_hip - F_cpu) / N, est)


def test_c5_trained_theta_gradients_rows_layout_with_t_handover_and_extended_order(engine, monkeypatch):
    import bench
    import ggp_amd
    from oracle import vfe_oracle as O
    M, d, GR = bench.M_IND, bench.DIM, 100_000
    X, y, Z = bench.synth(bench.N_TOTAL, M, d)
    Xg, yg = X[:GR].contiguous(), y[:GR].contiguous()
    Xd, yd, Zd = Xg.to(engine.device), yg.to(engine.device), Z.to(engine.device)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ref = O.grads_autograd(Xg, yg, Z, LS_TRAINED, SF_TRAINED ** 2, SN_TRAINED ** 2, bench.JITTER)

    def check(tag, F, g):
        assert abs(F - ref["F"]) / GR < 1e-8, (tag, F, ref["F"])
        scale = max(1.0, float(ref["g_ls"].abs().max()))
        assert float((g["ls"] - ref["g_ls"]).abs().max()) < 1e-6 * scale, (tag, g["ls"], ref["g_ls"])
        assert abs(g["sf2"] - ref["g_sf2"]) < 1e-6 * max(1.0, abs(ref["g_sf2"])), (tag, g["sf2"], ref["g_sf2"])
        assert abs(g["s2"] - ref["g_s2"]) < 1e-6 * max(1.0, abs(ref["g_s2"])), (tag, g["s2"], ref["g_s2"])

    # (1) the whitened order in the rows layout, T handed over to the factored pass 2
    calls = {"rows_t_out": 0, "bwd_t_in": 0}
    real_rows, real_bwd = engine.suffstats_whitened_rows, engine.suffstats_bwd_factored

    def spy_rows(*a, **k):
        calls["rows_t_out"] += k.get("t_out") is not None
        return real_rows(*a, **k)

    def spy_bwd(*a, **k):
        calls["bwd_t_in"] += k.get("t_in") is not None
        return real_bwd(*a, **k)

    monkeypatch.setattr(engine, "suffstats_whitened_rows", spy_rows)
    monkeypatch.setattr(engine, "suffstats_bwd_factored", spy_bwd)
    cw = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=engine, form="whitened")
    assert GR * M >= cw.whitened_rows_min_work
    Fw, gw = cw.value_and_grad(Zd, LS_TRAINED, SF_TRAINED ** 2, SN_TRAINED ** 2, want_gz=False)
    assert calls == {"rows_t_out": 1, "bwd_t_in": 1}, calls
    check("whitened rows layout + T hand-over", Fw, gw)
    del cw

    # (2) form="auto": at this theta the estimate is far beyond the old gradient range of the extended order (3 x the tolerance) -- since
    # round 6 a value + gradient evaluation takes that order all the same (asserted), with its Phibar formed in double-double and the
    # trailing word applied in pass 2 (sgp_phibar_dd, sgp_suffstats_bwd_lo), accepted on the size of the trailing word's correction
    ca = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=engine)
    Fa, ga = ca.value_and_grad(Zd, LS_TRAINED, SF_TRAINED ** 2, SN_TRAINED ** 2, want_gz=False)
    est = ca.last_estimate
    check("auto (estimate %.3g, reruns %d, extended %d)" % (est or -1.0, ca.n_guard_reruns, ca.n_extended), Fa, ga)
    assert est > 10.0 * ca.extended_grad_range * ca.streaming_tol, est
    assert ca.last_tier == 1 and ca.n_lo_rejections == 0 and ca.last_lo_correction <= ca.extended_lo_max_correction, (ca.last_tier, ca.last_lo_correction)
    del ca

    # (3) the three generations of the extended order's explicit-Phibar gradient against autograd: both words of the double-double matrix
    # (held to 1e-6: the product's route), its leading word alone, the fp64-formed matrix of rounds 4-5 (reported)
    errs = {}
    for tag, dd, lo in (("both words", True, True), ("leading word", True, False), ("fp64-formed", False, False)):
        ce = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=engine, form="extended")
        ce.extended_dd_phibar, ce.extended_lo = dd, lo
        Fe, ge = ce.value_and_grad(Zd, LS_TRAINED, SF_TRAINED ** 2, SN_TRAINED ** 2, want_gz=False)
        assert abs(Fe - ref["F"]) / GR < 1e-8, (Fe, ref["F"])
        errs[tag] = max(float((ge["ls"] - ref["g_ls"]).abs().max()) / max(1.0, float(ref["g_ls"].abs().max())),
                        abs(ge["sf2"] - ref["g_sf2"]) / max(1.0, abs(ref["g_sf2"])))
        if dd and lo:
            check("extended, both words of the double-double Phibar", Fe, ge)
        del ce
    print("extended-order gradient against autograd at an estimate of %.3g: %s" % (est, ", ".join("%s %.3g" % kv for kv in errs.items())))
    assert errs["both words"] < 0.2 * errs["fp64-formed"]
