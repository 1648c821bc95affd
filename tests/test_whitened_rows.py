"""The whitened (PyMC3) evaluation order in the streaming layout -- sgp_suffstats_fwd_whitened_rows + sgp_suffstats_bwd_factored_ex, what a
large shard takes when the streaming-order guard sends an evaluation to the whitened order -- against the chunked routine it
replaces there, the CPU oracle, and the golden fixtures.  Tolerances: the two routines sum the same fp64 products in different orders
(1e-12 of the largest entry where K_uu is well conditioned, 1e-10 with the explicit inverse's |L^-1| |K| rounding at cond 1e6); T = K' L^-T
against torch's triangular solve and W, u against the oracle 1e-8 (as test_whitened_stats_match_oracle_pymc3_order); gradients with T
handed over against gradients recomputing it: the same launches on the same numbers (1e-13); the bound and its gradients against the fixtures: the tolerances of test_gpu_parity.py::test_bound_and_grads_golden."""
import math

import numpy as np
import pytest
import torch

from conftest import dev, golden_names, load_golden

pytestmark = pytest.mark.gpu

KNAME = {0: "rbf", 1: "matern32", 2: "matern52"}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def unpack(packed, M):
    p = packed.cpu().numpy()
    return p[: M * M].reshape(M, M), p[M * M: M * M + M], p[M * M + M], p[M * M + M + 1]


def problem(N, M, d, seed, ls=1.4):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X[:, 0]) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[torch.randperm(N, generator=g)[:M]].clone() + 0.05 * torch.randn(M, d, dtype=torch.float64, generator=g)
    return X, y, Z, [ls * (1.0 + 0.1 * j) for j in range(d)]


@pytest.mark.parametrize("kernel", ["rbf", "matern32", "matern52"])
@pytest.mark.parametrize("N,M,d", [(5000, 200, 3), (777, 130, 1), (256, 64, 8), (20000, 384, 5), (270000, 128, 2)])
def test_rows_layout_matches_the_chunked_routine_and_the_oracle(engine, kernel, N, M, d):
    from oracle import vfe_oracle as O
    X, y, Z, ls = problem(N, M, d, N + M)
    sf2 = 1.7
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    Kuu = engine.kuu(Zd, ls, sf2, 1e-6, kernel)
    linv, info = engine.kuu_factor(Kuu)
    assert int(info.item()) == 0
    old = engine.suffstats_whitened(Xd, yd, Zd, ls, sf2, linv, kernel)
    new = engine.suffstats_whitened_rows(Xd, yd, Zd, ls, sf2, linv, kernel)
    t = engine.kfu_buffer(N, M)
    t.fill_(float("nan"))
    kept = engine.suffstats_whitened_rows(Xd, yd, Zd, ls, sf2, linv, kernel, t_out=t)
    assert torch.equal(new, kept)                                      # one super-chunk either way: the same launches
    W0, u0, yy0, ka0 = unpack(old, M)
    W1, u1, yy1, ka1 = unpack(new, M)
    assert relerr(W1, W0) < 1e-10 and relerr(u1, u0) < 1e-10 and abs(yy1 - yy0) <= 1e-13 * abs(yy0) and ka1 == ka0
    assert np.array_equal(W1, W1.T)
    kid = {"rbf": 0, "matern32": 1, "matern52": 2}[kernel]
    lst = torch.tensor(ls, dtype=torch.float64)
    Lr = torch.linalg.cholesky(O.kuu(Z, lst, sf2, 1e-6, kid))
    Kfu1 = O.kern(X, Z, lst, 1.0, kid)                                 # unit amplitude: T carries no sf2
    Tr = torch.linalg.solve_triangular(Lr, Kfu1.T, upper=False).T
    Mp = (M + 127) // 128 * 128
    Tg = t[: ((N + 255) // 256 * 256) * Mp].reshape(-1, Mp).cpu()
    assert relerr(Tg[:N, :M], Tr) < 1e-8                              # cond(L) ~ 1e3 .. 1e5 on both sides
    assert float(Tg[N:].abs().max() if Tg.shape[0] > N else 0.0) == 0.0 and float(Tg[:, M:].abs().max() if Mp > M else 0.0) == 0.0
    # the product itself, against the same factors (the library's own L^-1): componentwise inside the rounding of an M-term dot product
    # -- Npad >= 16384 takes the 128 x 128-tile kernel (gemm_tall_kernel; 270 000 rows: two row blocks per workgroup, the last
    # workgroup short), smaller shards the 64 x 64 one
    Li = linv.cpu().reshape(Mp, Mp)[:M, :M]
    Tp = Kfu1 @ Li.T
    bound = 2.0 * M * np.finfo(np.float64).eps * (Kfu1.abs() @ Li.abs().T) + 1e-300
    assert bool(((Tg[:N, :M] - Tp).abs() <= bound).all()), float(((Tg[:N, :M] - Tp).abs() / bound).max())
    A = sf2 * Tr.T
    assert relerr(W1, (A @ A.T).numpy()) < 1e-8 and relerr(u1, (A @ y).numpy()) < 1e-8


def test_rows_layout_accumulates_over_super_chunks(engine):
    """With a K'_fu budget below the shard the library walks several super-chunks (assembly, product, partials, contraction with
    accumulate = 1): the statistics agree with the one-block evaluation to rounding of the different split sums."""
    X, y, Z, ls = problem(9000, 256, 4, 5)
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    Kuu = engine.kuu(Zd, ls, 1.0, 1e-6, "rbf")
    linv, _ = engine.kuu_factor(Kuu)
    one = engine.suffstats_whitened_rows(Xd, yd, Zd, ls, 1.0, linv, "rbf")
    try:
        engine.lib.sgp_set_kfu_budget_bytes(2048 * 256 * 8)            # 2048 rows at a time: 5 super-chunks, the last one short
        engine._ws.pop("fwd_whitened_rows", None)
        many = engine.suffstats_whitened_rows(Xd, yd, Zd, ls, 1.0, linv, "rbf")
    finally:
        engine.lib.sgp_set_kfu_budget_bytes(0)
        engine._ws.pop("fwd_whitened_rows", None)
    M = 256
    Wa, ua, _, _ = unpack(one, M)
    Wb, ub, _, _ = unpack(many, M)
    assert relerr(Wb, Wa) < 1e-13 and relerr(ub, ua) < 1e-13


@pytest.mark.parametrize("kernel", ["rbf", "matern52"])
@pytest.mark.parametrize("want_gz", [False, True])
def test_factored_pass_2_takes_the_kept_product(engine, kernel, want_gz):
    X, y, Z, ls = problem(6000, 200, 3, 9)
    sf2, s2, M = 1.3, 0.04, 200
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    Kuu = engine.kuu(Zd, ls, sf2, 1e-6, kernel)
    linv, kinfo = engine.kuu_factor(Kuu)
    t = engine.kfu_buffer(6000, M)
    packed = engine.suffstats_whitened_rows(Xd, yd, Zd, ls, sf2, linv, kernel, t_out=t)
    res = engine.bound(Kuu, packed, s2, 6000, with_adjoints=True, kuu_linv=linv, kuu_info=kinfo, whitened=True, want_cw=True)
    a = engine.suffstats_bwd_factored(Xd, yd, Zd, ls, sf2, linv, res["Cw"], s2, res["bbar"], -0.5 / s2, kernel, want_gz=want_gz)
    b = engine.suffstats_bwd_factored(Xd, yd, Zd, ls, sf2, linv, res["Cw"], s2, res["bbar"], -0.5 / s2, kernel, want_gz=want_gz, t_in=t)
    assert relerr(b.cpu().numpy(), a.cpu().numpy()) < 1e-13
    # and against the explicit adjoint on this well-conditioned problem (cond ~ 1e5: the explicit Phibar still holds 1e-8)
    c = engine.suffstats_bwd(Xd, yd, Zd, ls, sf2, res["Phibar"], res["bbar"], -0.5 / s2, kernel, want_gz=want_gz)
    assert relerr(b.cpu().numpy(), c.cpu().numpy()) < 1e-7


@pytest.mark.parametrize("name", golden_names())
def test_bound_and_grads_golden_through_the_rows_layout(engine, name):
    import ggp_amd
    G = load_golden(name)
    kern = KNAME[int(G["kernel_id"])]
    cb = ggp_amd.CollapsedBound(dev(G["X"], engine), dev(G["y"], engine), kernel=kern, jitter=float(G["jitter"]), engine=engine,
                                form="whitened")
    cb.whitened_rows_min_work = 0
    cb.fused = False                                                   # not the single-launch path: the two-pass pipeline
    Z = dev(G["Z"], engine)
    before = dict(engine._ws)
    F, parts = cb.value(Z, G["ls"], float(G["sf2"]), float(G["s2"]))
    assert "fwd_whitened_rows" in engine._ws or "fwd_whitened_rows" in before
    ill = float(G["grad_rtol"]) > 1e-6
    tolF = 1e-9 * max(1.0, abs(float(G["F"])))
    assert abs(F - float(G["F"])) < tolF, (F, float(G["F"]))
    F2, g = cb.value_and_grad(Z, G["ls"], float(G["sf2"]), float(G["s2"]), want_gz=True)
    assert abs(F2 - float(G["F"])) < tolF and "bwd_factored_t" in engine._ws
    rt, rz = 1e-6, (1e-4 if ill else 1e-6)
    assert relerr(g["ls"].numpy(), G["g_ls"]) < rt, (g["ls"].numpy(), G["g_ls"])
    assert abs(g["sf2"] - float(G["g_sf2"])) < rt * max(1.0, abs(float(G["g_sf2"])))
    assert abs(g["s2"] - float(G["g_s2"])) < rt * max(1.0, abs(float(G["g_s2"])))
    assert relerr(g["Z"].cpu().numpy(), G["g_Z"]) < rz


def test_ill_conditioned_inducing_set_rows_layout_agrees_with_the_chunked_order(engine):
    """1-D inputs, inducing spacing 0.42 against a lengthscale of 3 (cond(K_uu) ~ 1e8, the CO2 regime): F and the gradients of the two
    whitened routines agree to 1e-9 / 1e-6 -- both are the PyMC3 order; the streaming order is off by 1e-4 here."""
    import ggp_amd
    g = torch.Generator().manual_seed(3)
    N, M = 6340, 128
    X = torch.linspace(0, 52.8, N, dtype=torch.float64)[:, None]
    y = torch.sin(X[:, 0] * 2 * math.pi) * 0.3 + 0.04 * X[:, 0] + 0.05 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[torch.linspace(0, N - 1, M).round().long()].clone()
    out = []
    for work in (1 << 40, 0):
        cb = ggp_amd.CollapsedBound(X.to(engine.device), y.to(engine.device), jitter=1e-6, engine=engine, form="whitened")
        cb.whitened_rows_min_work = work
        cb.fused = False
        out.append(cb.value_and_grad(Z.to(engine.device), [3.0], 1.3, 0.01, want_gz=True))
    (Fa, ga), (Fb, gb) = out
    assert abs(Fa - Fb) < 1e-9 * abs(Fa)
    assert relerr(gb["ls"].numpy(), ga["ls"].numpy()) < 1e-6 and abs(gb["s2"] - ga["s2"]) < 1e-6 * abs(ga["s2"])
    assert relerr(gb["Z"].cpu().numpy(), ga["Z"].cpu().numpy()) < 1e-5


_CHAIN_SCRIPT = r"""
import json, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import ggp_amd
from test_whitened_rows import problem
eng = ggp_amd.HipEngine()
X, y, Z, ls = problem(20000, 256, 4, 11)
Xd, yd, Zd = X.to(eng.device), y.to(eng.device), Z.to(eng.device)
Kuu = eng.kuu(Zd, ls, 1.3, 1e-6, "rbf")
linv, kinfo = eng.kuu_factor(Kuu)
t = eng.kfu_buffer(20000, 256)
packed = eng.suffstats_whitened_rows(Xd, yd, Zd, ls, 1.3, linv, "rbf", t_out=t)
res = eng.bound(Kuu, packed, 0.04, 20000, with_adjoints=True, kuu_linv=linv, kuu_info=kinfo, whitened=True, want_cw=True)
g = eng.suffstats_bwd_factored(Xd, yd, Zd, ls, 1.3, linv, res["Cw"], 0.04, res["bbar"], -12.5, "rbf", want_gz=True, t_in=t)
print(json.dumps(g.cpu().tolist()))
"""


def test_three_product_chain_of_the_factored_pass_2_agrees_with_the_two_product_one():
    """SGP_BWD_FULLY_FACTORED=1 (read once per process, hence the child processes) keeps T2 = T (Cw / s2) as a product of its own -- a
    full-range tall GEMM -- in front of pass 2's; the default multiplies (Cw / s2)(L^-1 / 2) first.  Same gradients to 1e-10."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ("0", "1"):
        env = dict(os.environ, SGP_BWD_FULLY_FACTORED=flag)
        r = subprocess.run([sys.executable, "-c", _CHAIN_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.asarray(json.loads(r.stdout.strip().splitlines()[-1])))
    assert relerr(outs[1], outs[0]) < 1e-10 and not np.array_equal(outs[0], outs[1])   # (two different chains did run)


@pytest.mark.parametrize("N", [0, 1, 255])
def test_rows_layout_handles_empty_and_tiny_shards(engine, N):
    """An empty shard (a rank of a sharded job can hold no rows) leaves zeros; one row and 255 rows (one padded 256-row block) agree
    with the chunked routine."""
    X, y, Z, ls = problem(max(N, 1), 96, 3, 21)
    X, y = X[:N], y[:N]
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    Kuu = engine.kuu(Zd, ls, 1.0, 1e-6, "rbf")
    linv, _ = engine.kuu_factor(Kuu)
    t = engine.kfu_buffer(N, 96)
    new = engine.suffstats_whitened_rows(Xd, yd, Zd, ls, 1.0, linv, "rbf", t_out=t)
    old = engine.suffstats_whitened(Xd, yd, Zd, ls, 1.0, linv, "rbf")
    W1, u1, yy1, ka1 = unpack(new, 96)
    W0, u0, yy0, ka0 = unpack(old, 96)
    if N == 0:
        assert not W1.any() and not u1.any() and yy1 == 0.0 and ka1 == 0.0
    else:
        assert relerr(W1, W0) < 1e-10 and relerr(u1, u0) < 1e-10 and abs(yy1 - yy0) <= 1e-13 * abs(yy0) and ka1 == ka0
