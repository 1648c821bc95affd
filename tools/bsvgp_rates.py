#!/usr/bin/env python3
"""Minibatch steps per second of BayesianStochasticVariationalGP at the C4 shape (N = 100k, d = 2, M = 256, B = 4096, five
hyper-samples per minibatch = five bound + gradient chains, one backward, one Adam step), with torch's default host thread
pool and with the cap the training loops apply (core.few_host_threads)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd  # noqa: E402


def main():
    eng = ggp_amd.HipEngine()
    g = torch.Generator().manual_seed(4)
    N, M, B = 100_000, 256, 4096
    X = torch.randn(N, 2, dtype=torch.float64, generator=g)
    y = torch.sin(2 * X[:, 0]) * torch.cos(X[:, 1]) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z0 = X[torch.randperm(N, generator=g)[:M]].clone()
    Xd, yd = X.to(eng.device), y.to(eng.device)
    batches = [(Xd[i:i + B], yd[i:i + B]) for i in range(0, 16 * B, B)]
    default_threads = torch.get_num_threads()
    for label, capped in (("capped (4)", True), ("torch default (%d)" % default_threads, False)):
        torch.manual_seed(0)
        model = ggp_amd.BayesianStochasticVariationalGP(Xd, yd, ggp_amd.GaussianLikelihood(), Z0, engine=eng, seed=3)
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        train = model.train_model if capped else (lambda *a, **k: type(model).train_model.__wrapped__(model, *a, **k))
        train(opt, batches[:4], num_epochs=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, bl = train(opt, batches, num_epochs=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": "C4 BayesianSVGP minibatch step", "host_threads": label, "steps_per_s": 2 * len(batches) / dt,
                          "ms_per_step": dt / (2 * len(batches)) * 1e3, "last_batch_loss": bl[-1]}), flush=True)


if __name__ == "__main__":
    main()
