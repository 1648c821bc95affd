import json, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ggp_amd
eng = ggp_amd.HipEngine()
N, M = bench.N_TOTAL, bench.M_IND
X, y, Z = bench.synth(N, M, bench.DIM)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
cs = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng); cs.streaming_tol = float("inf")
cw = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng, form="whitened")
cx = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=eng, form="extended")
thetas = [([3.75, 2.61, 3.38, 5.50, 3.49, 3.17, 2.64, 3.65], 1.0, 0.145), ([4.87, 2.27, 7.04, 6.39, 7.18, 3.35, 2.31, 6.49], 1.0, 0.144),
          ([2.5] * 8, 1.0, 0.05), ([2.8] * 8, 1.0, 0.1), ([3.0] * 8, 1.0, 0.145), ([2.0, 6.0, 2.0, 6.0, 2.0, 6.0, 2.0, 6.0], 1.0, 0.145), ([3.2] * 8, 1.0, 0.2)]
for ls, sf, sn in thetas:
    cs.value(Zd, ls, sf * sf, sn * sn, raise_on_fail=False)
    est = cs.last_estimate
    Fw, gw = cw.value_and_grad(Zd, ls, sf * sf, sn * sn, want_gz=False)
    b = torch.cat([gw["ls"], torch.tensor([gw["sf2"], gw["s2"]], dtype=torch.float64)])
    row = {"ls": ls, "sig_n": sn, "estimate": est, "est_times_s2": est * sn * sn}
    # the explicit Phibar of the extended order: formed in double-double with / without its trailing word in pass 2 (round 6), by two fp64 products (rounds 4-5)
    for tag, dd, lo in (("dd_both_words", True, True), ("dd_leading_word", True, False), ("fp64_formed", False, False)):
        cx.extended_dd_phibar, cx.extended_lo = dd, lo
        Fx, gx = cx.value_and_grad(Zd, ls, sf * sf, sn * sn, want_gz=False)
        a = torch.cat([gx["ls"], torch.tensor([gx["sf2"], gx["s2"]], dtype=torch.float64)])
        row[tag] = {"suite_metric": max(float((gx["ls"] - gw["ls"]).abs().max() / gw["ls"].abs().max()), abs(gx["sf2"] - gw["sf2"]) / max(1.0, abs(gw["sf2"])),
                                        abs(gx["s2"] - gw["s2"]) / max(1.0, abs(gw["s2"]))),
                    "grad_max_rel_component": float(((a - b).abs() / b.abs()).max()), "grad_rel_norm": float((a - b).norm() / b.norm())}
    row["dF_per_datum"] = abs(Fx - Fw) / N
    print(json.dumps(row), flush=True)
