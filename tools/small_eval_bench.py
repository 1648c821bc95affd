#!/usr/bin/env python3
"""Device time of the single-launch small-problem kernel (sgp_small_eval): K back-to-back launches between two events,
and the host-visible time of one evaluation including the device-to-host copy of the result.
    python3 tools/small_eval_bench.py            (SHAPES="500,1,50;634,1,128" to override; KERNEL=co2: the composite CO2
    covariance on the d = 1 shapes)"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402

K = 200


def main():
    eng = ggp_amd.HipEngine()
    shapes = os.environ.get("SHAPES", "382,1,25;500,1,50;634,1,64;634,1,128;1300,8,100;1300,13,100;5000,8,100;13279,16,100;13279,18,100;13279,18,128")
    for spec in shapes.split(";"):
        N, d, M = (int(v) for v in spec.split(","))
        g = torch.Generator().manual_seed(0)
        X = torch.randn(N, d, dtype=torch.float64, generator=g)
        y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
        Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
        X, y = X.to(eng.device), y.to(eng.device)
        th = torch.tensor([0.7 if d == 1 else 2.0] * d + [1.0, 0.09], dtype=torch.float64).to(eng.device)
        res = {"N": N, "d": d, "M": M}
        cases = (("value_us", dict(want_grad=False)), ("value_grad_us", dict(want_grad=True)),
                 ("value_grad_Z_us", dict(want_grad=True, want_gz=True)))
        kern, ckw, nres = "rbf", {}, d
        if os.environ.get("KERNEL", "rbf") == "co2":
            if d != 1:
                continue
            blk = ggp_amd.co2_kernel(0.5, 1.0, 5.0, 1.0, 3.0, 1.0, 0.5, 2.0, 0.1, 0.5).block()
            th = torch.tensor(blk + [0.09], dtype=torch.float64).to(eng.device)
            kern, ckw, nres, cases = "composite", {"composite": {"structure": blk}}, len(blk) - 1, cases[:2]
            res["kernel"] = "co2 composite"
        for label, kw in cases:
            kw = dict(kw, **ckw)
            out, info = eng.small_result(nres)
            for _ in range(5):
                eng.small_eval(X, y, Z, th, 1e-6, kern, mode=0, out=out, **kw)
            torch.cuda.synchronize()
            best = float("inf")
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(K):
                    eng.small_eval(X, y, Z, th, 1e-6, kern, mode=0, out=out, **kw)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / K)
            res[label] = round(best, 2)
        # host-visible: launch + one D2H copy of [out | status]
        out, info = eng.small_result(nres)
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(K):
                eng.small_eval(X, y, Z, th, 1e-6, kern, mode=0, want_grad=True, out=out, **ckw)
                host = out.to("cpu")
            best = min(best, (time.perf_counter() - t0) / K * 1e6)
        res["value_grad_with_readback_us"] = round(best, 2)
        res["info"] = int(info.item())
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
