#!/usr/bin/env python3
"""VERDICT r5 next-5, the measurement before the build: what would ONE captured graph per evaluation save at C3 (N 13 279, d 18, M 512; ~30
launches in 0.41 / 0.74 ms)?  The multi-launch evaluation is enqueued without a host round trip (CollapsedBound._forward / _pass2), so it can
be captured as it is -- with theta baked into the kernel arguments, i.e. valid for THIS theta only: a timing instrument, not a product path
(the product would need theta in device memory).  Compares, alternating: plain launches vs replay of the captured graph, both ending in the
same single device-to-host copy of the result buffer; single stream in both (the two-stream arrangement of the product is timed beside them)."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402
from ggp_amd import core as C  # noqa: E402

eng = ggp_amd.HipEngine()
N, d, M = 13279, 18, int(os.environ.get("M_IND", 512))
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, dtype=torch.float64, generator=g)
y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
Xd, yd = X.to(eng.device), y.to(eng.device)
ls, sf2, s2 = [2.0] * d, 1.0, 0.09
out = {}
for with_grad in (False, True):
    cb = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=eng)
    prod = (lambda: cb.value_and_grad(Z, ls, sf2, s2, want_gz=False)) if with_grad else (lambda: cb.value(Z, ls, sf2, s2))
    for _ in range(5):
        prod()
    cb1 = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=eng)
    cb1.overlap_tail = False
    nh = d
    extra = (nh + 1) if with_grad else 0

    def enqueue():
        res = cb1._forward(Z, ls, sf2, s2, with_adjoints=with_grad, extra=extra, tier=C.TIER_STREAMING, report=True)
        if with_grad:
            head = res["buf"].numel() - extra
            cb1._pass2(res, Z, ls, sf2, s2, False, res["buf"][head:])
        return res

    for _ in range(5):
        r = enqueue()
        r["buf"].to("cpu")
    ref = enqueue()["buf"].to("cpu").clone()
    side = torch.cuda.Stream(device=eng.device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for _ in range(2):
            enqueue()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            gres = enqueue()
    torch.cuda.synchronize()
    graph.replay()
    got = gres["buf"].to("cpu")
    keep = list(range(ggp_amd.engine.OUT_LEN)) + list(range(got.numel() - extra, got.numel()))   # (the padding, and the half of the status word's double behind it, are never written)
    same = bool(torch.equal(got[keep].view(torch.int64), ref[keep].view(torch.int64)))
    if not same and os.environ.get("VERBOSE"):
        print("differing slots:", [(i, float(got[i]), float(ref[i])) for i in keep if got[i:i + 1].view(torch.int64) != ref[i:i + 1].view(torch.int64)], file=sys.stderr)

    def t_plain():
        enqueue()["buf"].to("cpu")

    def t_graph():
        graph.replay()
        gres["buf"].to("cpu")

    def timed(fn, n=200):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    rows = {"product_two_streams_us": [], "plain_one_stream_us": [], "graph_replay_us": []}
    for _ in range(3):
        rows["product_two_streams_us"].append(round(timed(prod), 1))
        rows["plain_one_stream_us"].append(round(timed(t_plain), 1))
        rows["graph_replay_us"].append(round(timed(t_graph), 1))
    rows["graph_result_equals_plain_bits"] = same
    out["value_and_grad" if with_grad else "value"] = rows
print(json.dumps({"N": N, "d": d, "M": M, **out}))
