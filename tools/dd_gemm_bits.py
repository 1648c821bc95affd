import os, sys, torch, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ggp_amd
eng = ggp_amd.HipEngine()
out = []
for (N, M, d) in ((3000, 200, 3), (20000, 1024, 8), (5000, 384, 5)):
    g = torch.Generator().manual_seed(N + M)
    X = torch.randn(N, d, dtype=torch.float64, generator=g); y = torch.randn(N, dtype=torch.float64, generator=g); Z = X[:M].clone()
    ls, sf2, s2 = [1.5 + 0.2 * j for j in range(d)], 1.3, 0.05
    Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
    Kuu = eng.kuu(Zd, ls, sf2, 1e-6, "rbf"); linv, info = eng.kuu_factor(Kuu)
    Cw = torch.randn(M, M, dtype=torch.float64, generator=g); Cw = ((Cw + Cw.T) / 2).to(eng.device)
    hi, lo = eng.phibar_dd(Cw, linv, s2, want_lo=True)
    packed = eng.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, "rbf", level=2)
    h = hashlib.sha256()
    for t in (hi, lo, packed):
        h.update(t.cpu().numpy().tobytes())
    out.append(h.hexdigest()[:16])
print(" ".join(out))
