"""The EXTENDED streaming order (sgp_suffstats_fwd_extended: Phi on the integer matrix cores with 34 (level 1) or 39 (level 2, what the product runs: reach 2^14 x the tolerance) digit pairs and a double-double fold,
W = L^-1 Phi L^-T by two double-double products) against the whitened order it stands in for, the oracle, and autograd.  Tolerances: W, u
against the whitened routine 1e-10 of the largest entry on well-conditioned problems (both are accurate there; the explicit inverse's
rounding is common to both); F against the PyMC3-order oracle 1e-9 per datum where the streaming order itself is off by 1e-7 .. 1e-6;
gradients (explicit Phibar on the kept K'_fu) against autograd 1e-6 (inside the narrow range where a value + gradient evaluation takes this order: beyond an estimate
of ~3e-9 the explicit Phibar cancels and CollapsedBound sends gradients to the whitened order, profiles/r04_extended_order_c5.jsonl)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
@pytest.mark.parametrize("N,M,d", [(5000, 200, 3), (777, 130, 1), (20000, 384, 5), (0, 96, 2), (1, 64, 2)])
def test_extended_statistics_match_the_whitened_order(engine, kernel, N, M, d):
    X, y, Z, ls = problem(max(N, 1), M, d, N + M)
    X, y = X[:N], y[:N]
    sf2 = 1.7
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    linv, info = engine.kuu_factor(engine.kuu(Zd, ls, sf2, 1e-6, kernel))
    assert int(info.item()) == 0
    ref = engine.suffstats_whitened_rows(Xd, yd, Zd, ls, sf2, linv, kernel)
    kfu = engine.kfu_buffer(N, M)
    kfu.fill_(float("nan"))
    ext = engine.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, kernel, kfu=kfu)
    assert torch.equal(ext, engine.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, kernel))       # with or without the kept block; reproducible
    W0, u0, yy0, ka0 = unpack(ref, M)
    W1, u1, yy1, ka1 = unpack(ext, M)
    if N == 0:
        assert not W1.any() and not u1.any() and yy1 == 0.0 and ka1 == 0.0
        return
    assert relerr(W1, W0) < 1e-10 and relerr(u1, u0) < 1e-10 and abs(yy1 - yy0) <= 1e-13 * abs(yy0) and ka1 == ka0
    assert np.array_equal(W1, W1.T)
    # the kept block is the K'_fu the streaming order keeps, bit for bit
    k2 = engine.kfu_buffer(N, M)
    engine.suffstats(Xd, yd, Zd, ls, sf2, kernel, kfu=k2)
    n = engine.lib.sgp_kfu_len(N, M)
    assert torch.equal(kfu[:n], k2[:n])


def test_extended_statistics_accumulate_over_super_chunks(engine):
    X, y, Z, ls = problem(9000, 256, 4, 5)
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    linv, _ = engine.kuu_factor(engine.kuu(Zd, ls, 1.0, 1e-6, "rbf"))
    import ggp_amd
    one = engine.suffstats_extended(Xd, yd, Zd, ls, 1.0, linv, "rbf")
    # the budget is an option of the CONTEXT the call runs in (ABI version 3: sgp_ctx_suffstats_fwd_extended): an engine of its own
    small = ggp_amd.HipEngine(own_context=True)
    small.set_option("kfu_budget_bytes", 2048 * 256 * 8)               # 2048 rows at a time: 5 super-chunks, the last one short
    many = small.suffstats_extended(Xd, yd, Zd, ls, 1.0, linv, "rbf")
    assert engine.get_option("kfu_budget_bytes") != small.get_option("kfu_budget_bytes")
    Wa, ua, _, _ = unpack(one, 256)
    Wb, ub, _, _ = unpack(many, 256)
    assert relerr(Wb, Wa) < 1e-14 and relerr(ub, ua) < 1e-13          # the double-double sums differ in their last bits only


def test_extended_order_where_the_streaming_order_fails(engine):
    """N = 200 000, M = 512, long lengthscales: the streaming order is off by 1e-7 .. 1e-4 per datum, the extended order agrees with the
    PyMC3-order oracle to 1e-9 per datum inside its range (estimate <= 2^14 x the tolerance), and the bound built with form="auto" picks
    the tier by itself."""
    import bench
    import ggp_amd
    from oracle import vfe_oracle as O
    N, M, D = 200000, 512, 8
    X, y, Z = bench.synth(N, M, D)
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    ce = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine, form="extended")
    cs = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine)
    cs.streaming_tol = float("inf")
    ca = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine)
    torch.set_num_threads(max(1, (torch.get_num_threads())))
    seen_tier1 = 0
    for ls, sn in ((3.0, 0.3), (4.0, 0.3), (5.0, 0.3), (8.0, 0.145)):
        F_ref = float(O.vfe_pymc3_order_chunked(X, y, Z, torch.full((D,), ls, dtype=torch.float64), 1.0, sn, 1e-6))
        Fs, _ = cs.value(Zd, [ls] * D, 1.0, sn * sn)
        est = cs.last_estimate
        Fe, _ = ce.value(Zd, [ls] * D, 1.0, sn * sn)
        if est <= 16384e-9:
            assert abs(Fe - F_ref) / N < 1e-9, (ls, sn, est, abs(Fe - F_ref) / N, abs(Fs - F_ref) / N)
        before = ca.n_extended
        Fa, _ = ca.value(Zd, [ls] * D, 1.0, sn * sn)
        assert abs(Fa - F_ref) / N < 1e-8, (ls, sn, est)
        if 1e-9 < est <= 16384e-9 and ca.n_extended > before:
            seen_tier1 += 1
    assert seen_tier1 >= 1, "no cell of this sweep ran in the extended tier"
    # Gradients at such a theta: through round 5 the same bound sent them to the whitened order (the explicit Phibar of this order cancels).
    # Since round 6 they take this order too -- Phibar formed in double-double, its trailing word applied in pass 2 -- as long as the
    # trailing word's correction stays small against the gradient, and agree with the whitened order's factored pass 2 to 1e-6; with the
    # trailing word switched off (what another kernel or d > 8 gets) the old rule holds.
    before = ca.n_extended
    Fg, g = ca.value_and_grad(Zd, [4.0] * D, 1.0, 0.09, want_gz=False)
    cw = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine, form="whitened")
    Fw, gw = cw.value_and_grad(Zd, [4.0] * D, 1.0, 0.09, want_gz=False)
    assert ca.n_extended == before + 1 and ca.last_tier == 1 and ca.last_lo_correction <= ca.extended_lo_max_correction
    assert abs(Fg - Fw) / N < 1e-9 and float((g["ls"] - gw["ls"]).abs().max()) < 1e-6 * max(1.0, float(gw["ls"].abs().max()))
    assert abs(g["sf2"] - gw["sf2"]) < 1e-6 * max(1.0, abs(gw["sf2"])) and abs(g["s2"] - gw["s2"]) < 1e-6 * max(1.0, abs(gw["s2"]))
    ca.extended_lo = False
    before = ca.n_extended
    Fg, g = ca.value_and_grad(Zd, [4.0] * D, 1.0, 0.09, want_gz=False)
    assert ca.n_extended == before and ca.last_tier == 2 and Fg == Fw and torch.equal(g["ls"], gw["ls"])


def test_extended_order_gradients_against_autograd(engine):
    """50 000 rows x 512 inducing points at a trained-like theta: value + gradient through the extended order (explicit Phibar on the
    kept K'_fu) against torch autograd through the PyMC3-order graph."""
    import bench
    import ggp_amd
    from oracle import vfe_oracle as O
    N, M, D = 50000, 512, 8
    X, y, Z = bench.synth(N, M, D)
    ls = torch.tensor([3.7, 2.6, 3.4, 5.5, 3.5, 3.2, 2.6, 3.7], dtype=torch.float64)
    sf2, s2 = 1.1, 0.145 ** 2
    ref = O.grads_autograd(X, y, Z, ls, sf2, s2, 1e-6)
    cb = ggp_amd.CollapsedBound(X.to(engine.device), y.to(engine.device), jitter=1e-6, engine=engine, form="extended")
    F, g = cb.value_and_grad(Z.to(engine.device), ls.tolist(), sf2, s2, want_gz=True)
    assert abs(F - float(ref["F"])) / N < 1e-9
    assert float((g["ls"] - ref["g_ls"]).abs().max()) < 1e-6 * float(ref["g_ls"].abs().max())
    assert abs(g["sf2"] - ref["g_sf2"]) < 1e-6 * max(1.0, abs(ref["g_sf2"])) and abs(g["s2"] - ref["g_s2"]) < 1e-6 * max(1.0, abs(ref["g_s2"]))
    assert relerr(g["Z"].cpu().numpy(), ref["g_Z"].numpy()) < 1e-5


@pytest.mark.gpu
def test_double_double_phibar_and_its_trailing_word_in_pass_2(engine):
    """Round 6.  sgp_phibar_dd forms Phibar = L^-T (C / 2 s2) L^-1 in double-double: (i) its two words against an x87
    long-double product of the same inputs; (ii) sgp_suffstats_bwd_lo -- dC = K' Phibar_lo on the fp16 matrix cores (rows scaled by powers of two), contracted with dK
    in fp64 -- against the fp64 pass 2 run on the trailing word itself (what it approximates: three digits are asked for, 3e-3 is held);
    (iii) the same for a symmetric matrix whose rows span ten decades (the per-row power-of-two scaling; ragged shapes, padded rows and columns
    adding nothing); (iv) the fp16 image of K'_fu from the assembly kernel; (v) inputs beyond the contraction's fp16 format.  The kernel is a three-digit product: it is NOT asked to resolve cancellation, which the trailing word -- rounding
    residuals -- does not have."""
    import numpy as np
    import ggp_amd
    for (N, M, d) in ((3000, 200, 3), (20000, 384, 8), (777, 130, 1), (6000, 1024, 8), (4000, 256, 18), (3000, 300, 32), (2500, 200, 9)):
        g = torch.Generator().manual_seed(N + M)
        X = torch.randn(N, d, dtype=torch.float64, generator=g)
        y = torch.randn(N, dtype=torch.float64, generator=g)
        Z = X[:M].clone()
        ls, sf2, s2 = [1.5 + 0.2 * j for j in range(d)], 1.3, 0.05
        Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
        Kuu = engine.kuu(Zd, ls, sf2, 1e-6, "rbf")
        linv, info = engine.kuu_factor(Kuu)
        Mp = (M + 127) // 128 * 128
        Cw = torch.randn(M, M, dtype=torch.float64, generator=g)
        Cw = (Cw + Cw.T) / 2
        hi, lo = engine.phibar_dd(Cw.to(engine.device), linv, s2, want_lo=True)
        Li = linv.view(Mp, Mp)[:M, :M].cpu().numpy().astype(np.longdouble)
        ref = Li.T @ (Cw.numpy().astype(np.longdouble) / (2 * np.longdouble(s2))) @ Li
        got = hi.cpu().numpy().astype(np.longdouble) + lo.cpu().numpy().astype(np.longdouble)
        scale = float(np.abs(ref).max())
        assert float(np.abs(got - ref).max()) < 1e-17 * scale * M, (N, M, float(np.abs(got - ref).max()) / scale)   # (long double itself: 5e-20 per operation)
        assert float(np.abs(hi.cpu().numpy() - np.asarray(ref, dtype=np.float64)).max()) <= 2.3e-16 * scale          # the leading word IS the rounded matrix
        assert float(lo.abs().max()) <= 1.2e-16 * float(hi.abs().max())
        kfu = engine.kfu_buffer(N, M)
        engine.suffstats(Xd, yd, Zd, ls, sf2, "rbf", kfu=kfu)
        zero = torch.zeros(M, dtype=torch.float64, device=engine.device)
        Pr = torch.randn(M, M, dtype=torch.float64, generator=g)
        Pr = ((Pr + Pr.T) * torch.logspace(-14, -9, M, dtype=torch.float64)[:, None] * torch.logspace(-14, -9, M, dtype=torch.float64)[None, :]).to(engine.device)
        for P in (lo, Pr):   # (the trailing word; a symmetric matrix whose rows span ten decades -- no cancellation to resolve in either)
            exact = engine.suffstats_bwd(Xd, yd, Zd, ls, sf2, P, zero, 0.0, "rbf", want_gz=False, kfu=kfu).cpu()
            acc = torch.zeros(d + 1, dtype=torch.float64, device=engine.device)
            engine.suffstats_bwd_lo(Xd, yd, Zd, ls, sf2, P, kfu, acc, "rbf")
            acc = acc.cpu()
            assert float((acc - exact).abs().max()) <= 3e-3 * float(exact.abs().max()), (N, M, d, acc, exact)
        # (iv) the fp16 image of K'_fu written by the assembly kernel of the extended order is the image the product would have made itself:
        # the same correction bit for bit with and without it, and without the fp64 block at all
        kfu2, kh = engine.kfu_buffer(N, M), engine.kfu_f16_buffer(N, M)
        engine.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, "rbf", kfu=kfu2, level=2, kfu_f16=kh)
        n_el = ((N + 255) // 256 * 256) * Mp
        assert torch.equal(kh[:n_el], kfu2[:n_el].to(torch.float32).to(torch.float16))
        outs = []
        for kw in (dict(kfu=kfu2), dict(kfu=kfu2, kfu_f16=kh), dict(kfu=None, kfu_f16=kh)):
            acc = torch.zeros(d + 1, dtype=torch.float64, device=engine.device)
            engine.suffstats_bwd_lo(Xd, yd, Zd, ls, sf2, lo, kw.pop("kfu"), acc, "rbf", **kw)
            outs.append(acc.cpu())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (N, M, d, outs)
    with pytest.raises(ValueError):   # the image is written beside the fp64 block, never alone
        engine.suffstats_extended(Xd, yd, Zd, ls, sf2, linv, "rbf", kfu=None, level=2, kfu_f16=kh)
    with pytest.raises(ValueError):   # the product needs one of the two
        engine.suffstats_bwd_lo(Xd, yd, Zd, ls, sf2, lo, None, torch.zeros(d + 1, dtype=torch.float64, device=engine.device), "rbf")
    # (v) inputs beyond the fp16 format of the contraction (an inducing point more than 128 lengthscales from the mean inducing point): nothing is
    # added and the correction is reported as NaN -- the caller (core.py) then repeats the evaluation in the whitened order
    N, M, d = 2000, 256, 2
    g = torch.Generator().manual_seed(5)
    X = 30.0 * torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[:M].clone()
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    ls = [0.1, 0.1]
    kfu = engine.kfu_buffer(N, M)
    engine.suffstats(Xd, yd, Zd, ls, 1.0, "rbf", kfu=kfu)
    P = torch.randn(M, M, dtype=torch.float64, generator=g)
    P = (1e-12 * (P + P.T)).to(engine.device)
    acc = torch.ones(d + 1, dtype=torch.float64, device=engine.device)
    delta = torch.zeros(d + 1, dtype=torch.float64, device=engine.device)
    engine.suffstats_bwd_lo(Xd, yd, Zd, ls, 1.0, P, kfu, acc, "rbf", delta=delta)
    assert bool(torch.isnan(delta).all()) and bool((acc == 1.0).all())
    assert engine.bwd_lo_supported(1000, 64, 9) and not engine.bwd_lo_supported(1000, 64, 33) and not engine.bwd_lo_supported(1000, 64, 3, "matern32")
