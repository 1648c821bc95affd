"""Soak test (GPU): 300 evaluations at C5 must be bit-identical with a clean status word, then ~1000 value+gradient\nevaluations at mixed mid sizes with the two-stream path forced.  Run on an MI355X: python tools/soak.py"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ggp_amd
eng = ggp_amd.HipEngine(); dev = eng.device
g = torch.Generator().manual_seed(0)
N, M, d = 1_000_000, 1024, 8
X = torch.randn(N, d, dtype=torch.float64, generator=g)
w = torch.randn(d, dtype=torch.float64, generator=g) / math.sqrt(d)
y = torch.sin(X @ w) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
Z = X[torch.randperm(N, generator=g)[:M]].clone().to(dev)
cb = ggp_amd.CollapsedBound(X.to(dev), y.to(dev), jitter=1e-6, engine=eng)
t0 = time.time()
F0, _ = cb.value(Z, [2.0] * d, 1.0, 0.09)
bad = 0
tmax = 0.0
for i in range(300):
    t1 = time.perf_counter()
    if i % 3 == 2:
        F, gr = cb.value_and_grad(Z, [2.0] * d, 1.0, 0.09, raise_on_fail=False)
        info = gr["info"]
    else:
        F, parts = cb.value(Z, [2.0] * d, 1.0, 0.09, raise_on_fail=False)
        info = parts["info"]
    dt = time.perf_counter() - t1
    tmax = max(tmax, dt)
    if info != 0 or F != F0:
        bad += 1
        print("C5 iter", i, "info", info, "F", F, F0)
print("C5 soak: 300 evaluations, bad", bad, "slowest %.1f ms" % (tmax * 1e3), "total %.1f s" % (time.time() - t0))
del cb
# small / mid sizes with the two-stream path forced
bad2 = 0
for rep in range(40):
    for (n, dd, m) in [(3000, 2, 64), (20000, 4, 300), (50000, 8, 512), (8000, 3, 1000)]:
        Xs = torch.randn(n, dd, dtype=torch.float64, generator=g).to(dev)
        ys = torch.randn(n, dtype=torch.float64, generator=g).to(dev)
        Zs = Xs[:m].clone()
        c2 = ggp_amd.CollapsedBound(Xs, ys, jitter=1e-6, engine=eng)
        c2.overlap_min_work = 0
        ref = None
        for k in range(6):
            F, gr = c2.value_and_grad(Zs, [1.5] * dd, 1.0, 0.1, raise_on_fail=False)
            if gr["info"] != 0 or (ref is not None and F != ref):
                bad2 += 1
                print("mid", rep, (n, dd, m), k, gr["info"], F, ref)
            ref = F
print("mid soak: bad", bad2, "total %.1f s" % (time.time() - t0))
