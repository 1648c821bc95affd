#!/usr/bin/env python3
"""Where the host turn-around between two evaluations goes: time from the end of the read-back (`_fetch` returns) to the entry of the
next evaluation's first library call (pass 1) and to its return, on the multi-launch path.
    python3 tools/host_gap.py [rows] [grad]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ggp_amd  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
    grad = len(sys.argv) > 2 and sys.argv[2] == "grad"
    eng = ggp_amd.HipEngine()
    X, y, Z = bench.synth(rows, bench.M_IND, bench.DIM)
    cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=bench.JITTER, engine=eng)
    Zd = Z.to(eng.device)
    ls = [bench.LS] * bench.DIM
    marks = {"fetch_in": [], "fetch_out": [], "p1_in": [], "p1_out": []}
    real_fetch, real_ss = cb._fetch, eng.suffstats

    def fetch(*a, **k):
        marks["fetch_in"].append(time.perf_counter())
        r = real_fetch(*a, **k)
        marks["fetch_out"].append(time.perf_counter())
        return r

    def ss(*a, **k):
        marks["p1_in"].append(time.perf_counter())
        r = real_ss(*a, **k)
        marks["p1_out"].append(time.perf_counter())
        return r

    cb._fetch, eng.suffstats = fetch, ss
    fn = (lambda: cb.value_and_grad(Zd, ls, bench.SF ** 2, bench.SN ** 2, want_gz=False)) if grad else (lambda: cb.value(Zd, ls, bench.SF ** 2, bench.SN ** 2))
    for _ in range(40):
        fn()
    fo, pi, po, fi = marks["fetch_out"], marks["p1_in"], marks["p1_out"], marks["fetch_in"]
    last_fetch_out = [fo[i] for i in range(len(fo))]
    # evaluation k: p1_in[k] comes after the last fetch_out before it
    a, b, c = [], [], []
    for k in range(10, len(pi)):
        prev = max(t for t in last_fetch_out if t < pi[k])
        a.append((pi[k] - prev) * 1e6)
        b.append((po[k] - pi[k]) * 1e6)
        nxt = min(t for t in fi if t > po[k])
        c.append((nxt - po[k]) * 1e6)
    med = statistics.median
    print("rows %d grad %s: read-back returns -> pass-1 call %.1f us; pass-1 call %.1f us; pass-1 returned -> (last) fetch entered %.1f us" % (rows, grad, med(a), med(b), med(c)))


if __name__ == "__main__":
    main()
