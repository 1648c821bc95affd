#!/usr/bin/env python3
"""Device time of the pieces of the extended order's pass 2 at C5 (HIP events): the double-double formation of Phibar (sgp_phibar_dd), the
fp64 contraction with its leading word (sgp_suffstats_bwd) and the fp16 product + contraction with its trailing word (sgp_suffstats_bwd_lo)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402

eng = ggp_amd.HipEngine()
N, M, d = int(os.environ.get("ROWS", bench.N_TOTAL)), bench.M_IND, bench.DIM
X, y, Z = bench.synth(N, M, d)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
ls, sf2, s2 = [3.0] * d, 1.0, 0.145 ** 2
Kuu = eng.kuu(Zd, ls, sf2, bench.JITTER, "rbf")
linv, _info = eng.kuu_factor(Kuu)
kfu = eng.kfu_buffer(N, M)
kh = eng.kfu_f16_buffer(N, M) if os.environ.get("LO_F16_IMAGE", "1") == "1" else None
packed = eng.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, "rbf", kfu=kfu, level=2, **({"kfu_f16": kh} if kh is not None else {}))
res = eng.bound(Kuu, packed, s2, N, with_adjoints=True, kuu_linv=linv, kuu_info=_info, whitened=True, want_cw=True)
g = eng.empty(d + 1)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


hi, lo = eng.phibar_dd(res["Cw"], linv, s2, want_lo=True)
out = {"N": N, "M": M, "d": d,
       "phibar_dd_ms": timed(lambda: eng.phibar_dd(res["Cw"], linv, s2, want_lo=True)),
       "suffstats_bwd_leading_word_ms": timed(lambda: eng.suffstats_bwd(Xd, yd, Zd, ls, sf2, hi, res["bbar"], -0.5 / s2, "rbf", out=g, kfu=kfu)),
       "suffstats_bwd_lo_ms": timed(lambda: eng.suffstats_bwd_lo(Xd, yd, Zd, ls, sf2, lo, kfu, g, "rbf", **({"kfu_f16": kh} if kh is not None else {}))),
       "fp16_image_from_assembly": kh is not None,
       "extended_forward_ms": timed(lambda: eng.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, "rbf", kfu=kfu, level=2, **({"kfu_f16": kh} if kh is not None else {})))}
out["lo_product_tflops"] = 2.0 * N * M * M / (out["suffstats_bwd_lo_ms"] * 1e-3) / 1e12
print(json.dumps(out))
