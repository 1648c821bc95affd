import json, math, os, sys, time
import torch
sys.path.insert(0, "/root/repo")
import ggp_amd
eng = ggp_amd.HipEngine()
for name, N, d, M in (("C3 elevators", 13279, 18, 512), ("C3 reference M", 13279, 18, 100), ("8-GPU shard of C5", 125000, 8, 1024)):
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    y = (y - y.mean()) / y.std()
    Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
    X, y = X.to(eng.device), y.to(eng.device)
    cb = ggp_amd.CollapsedBound(X, y, jitter=1e-6, engine=eng)
    tgt = ggp_amd.HmcTarget(cb, Z)
    ggp_amd.sample_nuts(tgt, 3, 3, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr = ggp_amd.sample_nuts(tgt, 30, 30, seed=2)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ls = [2.0] * d
    t1 = time.perf_counter()
    for _ in range(50):
        cb.value_and_grad(Z, ls, 1.0, 0.09, want_gz=False)
    torch.cuda.synchronize()
    vg = (time.perf_counter() - t1) / 50
    th = [0.1] * (d + 2)
    t1 = time.perf_counter()
    for _ in range(50):
        tgt.logp_and_grad(th)
    torch.cuda.synchronize()
    lg = (time.perf_counter() - t1) / 50
    print(json.dumps({"config": name, "leapfrogs": int(tr.n_leapfrog), "us_per_leapfrog": round(wall / tr.n_leapfrog * 1e6, 1),
                      "us_per_value_and_grad": round(vg * 1e6, 1), "us_per_logp_and_grad": round(lg * 1e6, 1), "guard_repeats": cb.n_guard_reruns, "direct_whitened": cb.n_direct_whitened}))
