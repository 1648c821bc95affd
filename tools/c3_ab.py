import sys, os, math, time, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ggp_amd
eng = ggp_amd.HipEngine()
if os.environ.get("CU_BUDGET"):
    eng.set_option("cu_budget", int(os.environ["CU_BUDGET"]))
N, d, M = (int(v) for v in os.environ.get("SHAPE", "13279,18,512").split(","))
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, dtype=torch.float64, generator=g); y = torch.sin(X.sum(1)/math.sqrt(d)) + 0.1*torch.randn(N, dtype=torch.float64, generator=g)
Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng)
ls=[2.0]*d
res={}
for label, fn in (("value", lambda: cb.value(Z, ls, 1.0, 0.09)), ("value_grad", lambda: cb.value_and_grad(Z, ls, 1.0, 0.09))):
    for _ in range(5): fn()
    best=1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize(); best=min(best,(time.perf_counter()-t0)/30)
    res[label+"_us"]=round(best*1e6,1)
print(json.dumps(res))
