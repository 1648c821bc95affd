"""C3 (N 13 279, d 18, M 512) value / value + gradient rates for every library snapshot under _bisect/<sha>/ (built by hand from
`git worktree`s; VERDICT r3 next-4a: where did the value-only rate go between rounds 2 and 3).  One subprocess per snapshot, each
importing ITS OWN package copy; three alternations so box drift shows."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, math, time, json, torch
sys.path.insert(0, os.getcwd())
import ggp_amd
eng = ggp_amd.HipEngine()
N, d, M = 13279, 18, 512
g = torch.Generator().manual_seed(0)
X = torch.randn(N, d, dtype=torch.float64, generator=g); y = torch.sin(X.sum(1)/math.sqrt(d)) + 0.1*torch.randn(N, dtype=torch.float64, generator=g)
Z = X[torch.randperm(N, generator=g)[:M]].clone().to(eng.device)
cb = ggp_amd.CollapsedBound(X.to(eng.device), y.to(eng.device), jitter=1e-6, engine=eng)
ls=[2.0]*d
res={}
for label, fn in (("value", lambda: cb.value(Z, ls, 1.0, 0.09)), ("value_grad", lambda: cb.value_and_grad(Z, ls, 1.0, 0.09))):
    for _ in range(8): fn()
    best=1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(40): fn()
        torch.cuda.synchronize(); best=min(best,(time.perf_counter()-t0)/40)
    res[label+"_us"]=round(best*1e6,1)
print(json.dumps(res))
'''
order = sys.argv[1:] or sorted(os.listdir(os.path.join(ROOT, "_bisect")))
for rep in range(3):
    for sha in order + ["HEAD"]:
        cwd = ROOT if sha == "HEAD" else os.path.join(ROOT, "_bisect", sha)
        r = subprocess.run([sys.executable, "-c", CHILD], cwd=cwd, capture_output=True, text=True, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"error": r.stderr[-300:]})
        print(json.dumps({"sha": sha, "rep": rep, **json.loads(line)}), flush=True)
