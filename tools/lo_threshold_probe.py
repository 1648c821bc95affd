#!/usr/bin/env python3
"""Calibration of `CollapsedBound.extended_lo_max_correction` where the reference's sampler lives (train_fixed_model, models/bayesian_sgpr_hmc.py:160-180):
theta drawn around C5's trained ARD theta (log-normal perturbations of 0, 0.3 %, 1 %, 3 %, 10 % -- NUTS there moves by ~1e-3 per step), and scaled
along the lengthscale axis.  Per theta: the streaming-order estimate, the trailing word's correction over the gradient (what the check sees) and
what is left of the extended order's gradient against the whitened order's factored pass 2 (the suite's metric)."""
import json
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
N, M = bench.N_TOTAL, bench.M_IND
X, y, Z = bench.synth(N, M, bench.DIM)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
cw = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng, form="whitened")
ca = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng)
ca.extended_lo_max_correction = float("inf")      # accept whatever the trailing word says: this tool measures what is left
ca.extended_grad_range_lo = 16384.0
LS = [4.870895252562722, 2.274348615181124, 7.035384773166531, 6.388168428424034, 7.176420862837876, 3.3523772450641136, 2.314383327914714, 6.492694463809999]
SN = 0.14415221312756948
rng = np.random.default_rng(6)
thetas = [(LS, SN, "trained")]
for sd in (0.003, 0.01, 0.03, 0.1):
    for k in range(4):
        f = np.exp(sd * rng.standard_normal(9))
        thetas.append(([LS[j] * f[j] for j in range(8)], SN * f[8], "perturbed %g" % sd))
for scale in (0.8, 0.9, 1.1, 1.2, 1.35):
    thetas.append(([v * scale for v in LS], SN, "lengthscales x %g" % scale))
for sn in (0.1, 0.2, 0.3):
    thetas.append((LS, sn, "sig_n %g" % sn))
for ls, sn, tag in thetas:
    Fw, gw = cw.value_and_grad(Zd, ls, 1.0, sn * sn, want_gz=False)
    ca.guard = type(ca.guard)()
    ca._lo_skip = ca._lo_pause = 0
    Fa, ga = ca.value_and_grad(Zd, ls, 1.0, sn * sn, want_gz=False)
    left = max(float((ga["ls"] - gw["ls"]).abs().max() / max(1.0, float(gw["ls"].abs().max()))), abs(ga["sf2"] - gw["sf2"]) / max(1.0, abs(gw["sf2"])),
               abs(ga["s2"] - gw["s2"]) / max(1.0, abs(gw["s2"])))
    print(json.dumps({"theta": tag, "tier": ca.last_tier, "estimate": ca.last_estimate, "correction_over_gradient": ca.last_lo_correction,
                      "left_against_whitened": left, "left_over_correction": left / ca.last_lo_correction if ca.last_lo_correction else None}), flush=True)
