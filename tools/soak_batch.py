#!/usr/bin/env python3
"""Soak of the batched-over-theta launch chains (sgp_svgp_elbo_batch, sgp_mixture_predict): random shapes for SECONDS seconds;
every batch must equal the per-sample calls, repeat bit for bit, report clean status words and never time out (the S
factorizations of a batch share one dataflow launch and its flags)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402


def main():
    eng = ggp_amd.HipEngine()
    secs = float(os.environ.get("SECONDS_TO_RUN", "120"))
    rng = np.random.default_rng(7)
    t0 = time.time()
    n_svgp = n_mix = bad = mism = 0
    worst = 0.0
    D = lambda t: t.to(eng.device).contiguous()  # noqa: E731
    while time.time() - t0 < secs:
        B, M, d, S = int(rng.integers(1, 5000)), int(rng.integers(1, 300)), int(rng.integers(1, 7)), int(rng.integers(1, 9))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        X = torch.randn(B, d, dtype=torch.float64, generator=g)
        lik = "bernoulli" if rng.random() < 0.5 else "gaussian"
        f = torch.sin(X.sum(1))
        y = torch.sign(f + 0.3 * torch.randn(B, dtype=torch.float64, generator=g)) if lik == "bernoulli" else f + 0.2 * torch.randn(B, dtype=torch.float64, generator=g)
        y[y == 0] = 1.0
        Z = torch.randn(M, d, dtype=torch.float64, generator=g)
        m = 0.3 * torch.randn(M, dtype=torch.float64, generator=g)
        LS = torch.tril(0.2 * torch.randn(M, M, dtype=torch.float64, generator=g)) + torch.eye(M, dtype=torch.float64)
        ls = (0.4 + torch.rand(S, d, dtype=torch.float64, generator=g)).tolist()
        sf2 = (0.7 + torch.rand(S, dtype=torch.float64, generator=g)).tolist()
        s2 = (0.05 + 0.2 * torch.rand(S, dtype=torch.float64, generator=g)).tolist()
        args = (D(X), D(y), D(Z), ls, sf2, s2, D(m), D(LS), 50000)
        r1 = eng.svgp_elbo_batch(*args, likelihood=lik, with_grads=True)
        r1 = {k: v.clone() for k, v in r1.items()}
        r2 = eng.svgp_elbo_batch(*args, likelihood=lik, with_grads=True)
        if not all(torch.equal(r1[k], r2[k]) for k in r1):
            mism += 1
        if r1["info"].cpu().tolist() != [0] * S:
            bad += 1
        k = int(rng.integers(S))
        one = eng.svgp_elbo(D(X), D(y), D(Z), ls[k], sf2[k], s2[k], D(m), D(LS), 50000, likelihood=lik, with_grads=True)
        for key in ("g_m", "g_LS", "g_Z", "g_ls"):
            a, b = r1[key][k].reshape(-1), one[key].reshape(-1)
            worst = max(worst, float((a - b).abs().max() / max(1e-3, float(b.abs().max()))))
        worst = max(worst, abs(float(r1["out"][k, 0]) - float(one["out"][0])) / max(1.0, abs(float(one["out"][0]))))
        n_svgp += 1
        # mixture predictive on the same data (regression targets)
        T = int(rng.integers(1, 400))
        Xs = torch.randn(T, d, dtype=torch.float64, generator=g)
        yr = f + 0.2 * torch.randn(B, dtype=torch.float64, generator=g)
        q1 = eng.mixture_predict(D(X), D(yr), D(Xs), D(Z), ls, sf2, s2, jitter=1e-6, full_cov=True, gate_jitter=1e-4)
        q1 = {kk: (v.clone() if v is not None else None) for kk, v in q1.items()}
        q2 = eng.mixture_predict(D(X), D(yr), D(Xs), D(Z), ls, sf2, s2, jitter=1e-6, full_cov=True, gate_jitter=1e-4)
        if not all(torch.equal(q1[kk], q2[kk]) for kk in ("mean", "var", "cov", "info", "gate")):
            mism += 1
        if any(v < 0 for v in q1["info"].cpu().tolist() + q1["gate"].cpu().tolist()):
            bad += 1
        n_mix += 1
    print(json.dumps({"seconds": time.time() - t0, "svgp_batches": n_svgp, "mixture_batches": n_mix, "repeat_mismatch": mism,
                      "bad_status_or_timeout": bad, "worst_rel_diff_batch_vs_single": worst}))


if __name__ == "__main__":
    main()
