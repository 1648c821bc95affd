"""Single-launch path for small problems (sgp_small_eval): parity with the fixtures and the oracle, through the C ABI."""
import math

import numpy as np
import pytest
import torch

from conftest import dev, golden_names, load_golden

KNAME = {0: "rbf", 1: "matern32", 2: "matern52"}
SMALL = [n for n in golden_names() if load_golden(n)["X"].shape[1] <= 24]  # incl. rbf_d18_mid: Elevator's d = 18 at M = 128


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def _theta(G, engine):
    return dev(np.concatenate([G["ls"], [float(G["sf2"]), float(G["s2"])]]), engine)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL)
def test_small_value_and_grad_golden(engine, name):
    """PyMC3 op order with substitution solves: 1e-9 on F and 1e-6 on every gradient, duplicate-Z fixture included."""
    G = load_golden(name)
    d = G["X"].shape[1]
    X, y, Z = dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine)
    kern = KNAME[int(G["kernel_id"])]
    assert engine.small_supported(X.shape[0], Z.shape[0], d, kern)
    out, gz, info = engine.small_eval(X, y, Z, _theta(G, engine), float(G["jitter"]), kern, mode=0, want_grad=True, want_gz=True)
    o = out.cpu().numpy()
    assert int(info.item()) == 0
    F = float(G["F"])
    assert abs(o[0] - F) < 1e-9 * max(1.0, abs(F)), (o[0], F)
    assert abs(o[d + 3] - float(G["logmarg"])) < 1e-8 * max(1.0, abs(F)) and abs(o[d + 4] - float(G["trace_term"])) < 1e-8 * max(1.0, abs(F))
    ill = float(G["grad_rtol"]) > 1e-6
    assert relerr(o[1:1 + d], G["g_ls"]) < 1e-6, (o[1:1 + d], G["g_ls"])
    assert abs(o[1 + d] - float(G["g_sf2"])) < 1e-6 * max(1.0, abs(float(G["g_sf2"])))
    assert abs(o[2 + d] - float(G["g_s2"])) < 1e-6 * max(1.0, abs(float(G["g_s2"])))
    assert relerr(gz.cpu().numpy(), G["g_Z"]) < (1e-4 if ill else 1e-6)
    # value-only launch: same F, bit for bit
    out2, _, info2 = engine.small_eval(X, y, Z, _theta(G, engine), float(G["jitter"]), kern, mode=0, want_grad=False)
    assert int(info2.item()) == 0 and float(out2[0]) == o[0]


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in SMALL if n.startswith("rbf")])
def test_small_hmc_target_golden(engine, name):
    G = load_golden(name)
    d = G["X"].shape[1]
    X, y, Z = dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine)
    for th, lp_ref, g_ref in zip(G["hmc_theta"], G["hmc_logp"], G["hmc_grad"]):
        out, _, info = engine.small_eval(X, y, Z, dev(th, engine), 1e-6, "rbf", mode=1, want_grad=True)
        o = out.cpu().numpy()
        assert int(info.item()) == 0
        assert abs(o[0] - lp_ref) < 1e-9 * max(1.0, abs(lp_ref))
        assert np.max(np.abs(o[1:d + 3] - g_ref)) < 1e-6 * max(1.0, np.max(np.abs(g_ref)))
    # outside exp()'s range: zero density, not an exception
    bad = np.array(list(G["hmc_theta"][0]))
    bad[0] = 400.0
    out, _, info = engine.small_eval(X, y, Z, dev(bad, engine), 1e-6, "rbf", mode=1, want_grad=True)
    assert float(out[0]) == -math.inf and int(info.item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,d,kern", [(1, 1, 1, "rbf"), (63, 7, 2, "rbf"), (64, 64, 3, "matern32"), (65, 65, 1, "rbf"),
                                         (634, 128, 1, "rbf"), (500, 50, 1, "matern52"), (1300, 100, 13, "rbf"),
                                         (5000, 100, 8, "rbf"), (1300, 100, 18, "rbf"), (700, 60, 24, "matern32"),
                                         (13279, 100, 18, "rbf")])
def test_small_shapes_vs_oracle(engine, N, M, d, kern):
    """Edge shapes: one row, slab boundaries, both padded sizes, several slabs per workgroup (N = 5000 -> 79 slabs on 64
    workgroups), d up to 13 (the reference's small UCI sets), d = 18 (Elevator, also at its full 13 279 training rows with the
    reference's default M = 100: large_scale_regression_SGHMC.py:39-51,244) and the limit d = 24."""
    from oracle import vfe_oracle as O
    kid = {"rbf": 0, "matern32": 1, "matern52": 2}[kern]
    g = torch.Generator().manual_seed(N + M)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X.sum(1) / math.sqrt(d)) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = (X[torch.randperm(N, generator=g)[:M]] if M <= N else torch.randn(M, d, dtype=torch.float64, generator=g)).clone()
    ls = torch.rand(d, dtype=torch.float64, generator=g) * 0.8 + (0.6 if d == 1 else 1.5)
    sf2, s2, jit = 1.3, 0.07, 1e-6
    # reference gradients: autograd through the PyMC3-order graph where it exists (RBF), the closed form otherwise
    ref = O.grads_autograd(X, y, Z, ls, sf2, s2, jit) if kid == 0 else O.grads_analytic(X, y, Z, ls, sf2, s2, jit, kid)
    Fr = float(O.vfe_pymc3_order(X, y, Z, ls, math.sqrt(sf2), math.sqrt(s2), jit, kid))
    th = torch.cat([ls, torch.tensor([sf2, s2], dtype=torch.float64)]).to(engine.device)
    out, gz, info = engine.small_eval(X.to(engine.device), y.to(engine.device), Z.to(engine.device), th, jit, kern, mode=0,
                                      want_grad=True, want_gz=True)
    o = out.cpu().numpy()
    assert int(info.item()) == 0
    assert abs(o[0] - Fr) < 1e-9 * max(1.0, abs(Fr)), (o[0], Fr)
    assert relerr(o[1:1 + d], ref["g_ls"].numpy()) < 1e-6
    assert abs(o[1 + d] - ref["g_sf2"]) < 1e-6 * max(1.0, abs(ref["g_sf2"])) and abs(o[2 + d] - ref["g_s2"]) < 1e-6 * max(1.0, abs(ref["g_s2"]))
    assert relerr(gz.cpu().numpy(), ref["g_Z"].numpy()) < 1e-6
    # bit-reproducible
    out2, gz2, _ = engine.small_eval(X.to(engine.device), y.to(engine.device), Z.to(engine.device), th, jit, kern, mode=0,
                                     want_grad=True, want_gz=True)
    assert torch.equal(out2[:d + 5], out[:d + 5]) and torch.equal(gz2, gz)


@pytest.mark.gpu
def test_small_reports_non_pd(engine):
    X = torch.randn(40, 2, dtype=torch.float64)
    y = torch.randn(40, dtype=torch.float64)
    Z = torch.zeros(6, 2, dtype=torch.float64)  # identical inducing rows, no jitter
    th = torch.tensor([1.0, 1.0, 1.0, 0.1], dtype=torch.float64).to(engine.device)
    out, _, info = engine.small_eval(X.to(engine.device), y.to(engine.device), Z.to(engine.device), th, 0.0, "rbf", mode=0, want_grad=True)
    assert 1 <= int(info.item()) <= 6


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL)
def test_fused_launch_agrees_with_the_multi_launch_whitened_path(engine, name):
    """CollapsedBound routes M <= 128 through the single launch; ``fused = False`` keeps the multi-launch whitened path
    (sgp_kuu_factor -> sgp_suffstats_fwd_whitened -> sgp_bound_from_whitened_stats -> pass 2).  Same mathematics, different
    association order (substitution solves vs explicit L^-1, per-slab partial sums vs GEMM tiles), so the two agree to
    rounding, not bit for bit: 1e-10 relative on F, 1e-7 on the gradients (1e-5 on the duplicate-Z fixture)."""
    import ggp_amd
    G = load_golden(name)
    kern = KNAME[int(G["kernel_id"])]
    X, y, Z = dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine)
    a = ggp_amd.CollapsedBound(X, y, kernel=kern, jitter=float(G["jitter"]), engine=engine)
    b = ggp_amd.CollapsedBound(X, y, kernel=kern, jitter=float(G["jitter"]), engine=engine, form="whitened")
    b.fused = False
    assert a._small_ok(Z.shape[0]) and not b._small_ok(Z.shape[0])
    Fa, ga = a.value_and_grad(Z, G["ls"], float(G["sf2"]), float(G["s2"]), want_gz=True)
    Fb, gb = b.value_and_grad(Z, G["ls"], float(G["sf2"]), float(G["s2"]), want_gz=True)
    ill = float(G["grad_rtol"]) > 1e-6
    assert abs(Fa - Fb) < 1e-10 * max(1.0, abs(Fb))
    rt = 1e-5 if ill else 1e-7
    assert relerr(ga["ls"].numpy(), gb["ls"].numpy()) < rt and abs(ga["sf2"] - gb["sf2"]) < rt * max(1.0, abs(gb["sf2"]))
    assert abs(ga["s2"] - gb["s2"]) < rt * max(1.0, abs(gb["s2"]))
    assert relerr(ga["Z"].cpu().numpy(), gb["Z"].cpu().numpy()) < (1e-3 if ill else 1e-6)
    Fv, parts = a.value(Z, G["ls"], float(G["sf2"]), float(G["s2"]))
    assert Fv == Fa and abs(parts["trace_term"] - gb["trace_term"]) < 1e-8 * max(1.0, abs(Fb))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rbf_d3_small", "rbf_d1_tiny", "rbf_d18_mid"])
def test_device_resident_nuts_matches_the_host_driven_sampler(engine, name):
    """sgp_small_nuts (one persistent launch) against hmc.NUTS driven from the host with the same splitmix stream and the
    same evaluations (the single-launch NUTS target): identical trees, draws equal to rounding (the sampler's own exp / log
    / sin / cos come from the device math library there, from libm here)."""
    import ggp_amd
    from ggp_amd.hmc import NUTS, DiagMassAdapter, SplitMix
    G = load_golden(name)
    X, y, Z = dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine)
    cb = ggp_amd.CollapsedBound(X, y, jitter=1e-6, engine=engine)
    tgt = ggp_amd.HmcTarget(cb, Z)
    assert tgt.device_sampler_ok()
    q0 = np.array(tgt.start()) + 0.05
    tune, draws, seed = 20, 15, 1234
    r = engine.small_nuts(X, y, Z, q0, tune, draws, seed, jitter=1e-6, max_treedepth=8)
    assert r["info"] == 0 and r["draws"] == tune + draws
    nuts = NUTS(tgt.logp_and_grad, tgt.ndim, max_treedepth=8, rng=SplitMix(seed))
    q = q0.copy()
    lp, g = nuts._eval(q)
    nuts.mass = DiagMassAdapter(tgt.ndim, initial_mean=q)
    rows, sizes, steps = [], [], []
    for it in range(tune + draws):
        q, lp, g, st = nuts.draw(q, lp, g, it < tune)
        if it >= tune:
            rows.append(q.copy())
            sizes.append(st["tree_size"])
            steps.append(st["step_size"])
    assert r["evaluations"] == nuts.n_leapfrog
    assert np.array_equal(r["stats"][:, 1].numpy(), np.array(sizes, dtype=np.float64))
    assert np.allclose(r["stats"][:, 0].numpy(), steps, rtol=1e-9)
    assert np.allclose(r["samples"].numpy(), np.array(rows), rtol=1e-7, atol=1e-9)
    assert np.all(r["seconds"].numpy() > 0.0)


@pytest.mark.gpu
def test_sample_nuts_device_trace_surface_and_posterior(engine):
    """The Trace the reference reads (models/bayesian_sgpr_hmc.py:144-157, demo_1d_regression.py:199-206) from a device run,
    and agreement of the posterior means with a host-driven run of the same length (statistical)."""
    import ggp_amd
    G = load_golden("rbf_d3_small")
    X, y, Z = dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine)
    tgt = ggp_amd.HmcTarget(ggp_amd.CollapsedBound(X, y, jitter=1e-6, engine=engine), Z)
    tr = ggp_amd.sample_nuts_device(tgt, 150, 150, seed=3)
    assert len(tr) == 150 and tr["ls"].shape == (150, 3) and tr[0]["sig_n"] > 0
    assert tr.get_sampler_stats("step_size").shape == (150,) and float(tr.get_sampler_stats("perf_counter_diff").sum()) > 0
    assert len(tr[::2]) == 75 and not tr.get_sampler_stats("diverging").any()
    tr2 = ggp_amd.sample_nuts(tgt, 150, 150, seed=4)
    m1 = np.log(np.concatenate([tr["ls"], tr["sig_f"][:, None], tr["sig_n"][:, None]], 1)).mean(0)
    m2 = np.log(np.concatenate([tr2["ls"], tr2["sig_f"][:, None], tr2["sig_n"][:, None]], 1)).mean(0)
    s2 = np.log(np.concatenate([tr2["ls"], tr2["sig_f"][:, None], tr2["sig_n"][:, None]], 1)).std(0)
    assert np.all(np.abs(m1 - m2) < 0.6 * s2 + 0.02), (m1, m2, s2)
    lf_per_s = tr.n_leapfrog / tr.wall_clock_secs
    print("device NUTS: %d leapfrogs in %.3f s = %.0f / s" % (tr.n_leapfrog, tr.wall_clock_secs, lf_per_s))


@pytest.mark.gpu
def test_integration_md_small_stub_runs(engine):
    """The single-launch / persistent-sampler binding shown in INTEGRATION.md is executed verbatim and checked against the
    oracle's NUTS target."""
    import os
    import re
    from conftest import ROOT
    from oracle import vfe_oracle as O
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "sgp_small_nuts(" in b)
    stub = stub.replace('C.CDLL("generalised-gaussian-processes_amd/csrc/libsgp_hip.so")',
                        "C.CDLL(%r)" % os.path.join(ROOT, "generalised-gaussian-processes_amd", "csrc", "libsgp_hip.so"))
    ns = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    lp_ref, g_ref = O.hmc_logp(torch.tensor([0.3, 0.0, -1.0], dtype=torch.float64), ns["X"].cpu(), ns["y"].cpu(), ns["Z"].cpu())
    assert abs(ns["logp"] - lp_ref) < 1e-8 * max(1.0, abs(lp_ref))
    assert np.max(np.abs(np.array(ns["grad"]) - g_ref.numpy())) < 1e-5 * max(1.0, float(g_ref.abs().max()))
    assert ns["draws_done"] == 200 and ns["leapfrogs"] > 200 and bool(torch.all(ns["ls_draws"] > 0))


@pytest.mark.gpu
def test_small_eval_batch_equals_single_evaluations(engine):
    """S hyper-parameter sets in one launch (sgp_small_eval_batch, the theta-averaged loss of row a7): every slot equals
    the stand-alone evaluation of the same theta bit for bit, a non-representable theta gives -inf in its slot only."""
    G = load_golden("rbf_d3_small")
    d = 3
    X, y, Z = dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine)
    base = np.concatenate([G["ls"], [float(G["sf2"]), float(G["s2"])]])
    rows = np.stack([base * (1.0 + 0.1 * k) for k in range(5)])
    outs, gz, infos = engine.small_eval_batch(X, y, Z, dev(rows, engine), 1e-6, "rbf", mode=0, want_grad=True, want_gz=True)
    assert infos.cpu().tolist() == [0] * 5
    for k in range(5):
        o, g, info = engine.small_eval(X, y, Z, dev(rows[k], engine), 1e-6, "rbf", mode=0, want_grad=True, want_gz=True)
        assert torch.equal(outs[k], o[:d + 5]) and torch.equal(gz[k], g) and int(info.item()) == 0
    th = np.stack([G["hmc_theta"][0], [500.0, 0.0, 0.0, 0.0, 0.0], G["hmc_theta"][1]])
    outs, _, infos = engine.small_eval_batch(X, y, Z, dev(th, engine), 1e-6, "rbf", mode=1, want_grad=True)
    o = outs.cpu().numpy()
    assert abs(o[0, 0] - G["hmc_logp"][0]) < 1e-9 * abs(G["hmc_logp"][0]) and abs(o[2, 0] - G["hmc_logp"][1]) < 1e-9 * abs(G["hmc_logp"][1])
    assert o[1, 0] == -np.inf and infos.cpu().tolist() == [0, 0, 0]


@pytest.mark.gpu
def test_batch_skip_path_keeps_every_workgroup_on_its_own_request(engine):
    """Out-of-range thetas (the chain workgroup skips the evaluation and moves on) interleaved with valid ones, value only (no
    gradient hand-shake that would make the chain workgroup wait): every slot must still equal its stand-alone evaluation.  Before
    the read acknowledgement (SY_ACK) a lagging workgroup could pick up the NEXT request's theta -- ADVICE r2, sgp_small.hip:1559."""
    G = load_golden("rbf_d3_small")
    X, y, Z = dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine)
    good = [G["hmc_theta"][0] + 0.05 * k for k in range(12)]
    bad = np.array([400.0, 0.0, 0.0, 0.0, 0.0])
    rows, expect_inf = [], []
    for k, t in enumerate(good):
        for _ in range(k % 4):  # runs of 0-3 skipped requests between evaluations
            rows.append(bad)
            expect_inf.append(True)
        rows.append(t)
        expect_inf.append(False)
    rows += [bad, bad]  # ... and right before DONE
    expect_inf += [True, True]
    th = np.stack(rows)
    singles = {}
    for want_grad in (False, True):
        for rep in range(20):
            outs, _, infos = engine.small_eval_batch(X, y, Z, dev(th, engine), 1e-6, "rbf", mode=1, want_grad=want_grad)
            o = outs.cpu().numpy()
            assert infos.cpu().tolist() == [0] * len(rows)
            for i, t in enumerate(rows):
                if expect_inf[i]:
                    assert o[i, 0] == -np.inf
                    continue
                key = (want_grad, tuple(t))
                if key not in singles:
                    singles[key] = engine.small_eval(X, y, Z, dev(t, engine), 1e-6, "rbf", mode=1, want_grad=want_grad)[0].cpu().numpy()
                assert o[i, 0] == singles[key][0], (rep, i)
                if want_grad:
                    assert np.array_equal(o[i, 1:6], singles[key][1:6]), (rep, i)


@pytest.mark.gpu
def test_small_launch_adapts_to_the_cu_budget_and_refuses_below_four(engine):
    """The single launch's workgroups wait for each other, so all of them must be resident: its grid is cut to the CUs the calling
    thread may use (sgp_set_cu_budget, for CU-masked streams) -- 7 slabs walked by ONE row workgroup at a budget of 4, same bound
    and gradient to rounding -- and below 4 CUs (chain + two K_uu-adjoint + one row workgroup) the library says so up front:
    sgp_small_supported() = 0, the entry point returns an error instead of spinning into a time-out."""
    import ctypes as C
    G = load_golden("rbf_d3_small")
    X, y, Z = dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine)
    N, d = G["X"].shape
    M = G["Z"].shape[0]
    th = _theta(G, engine)
    kn = KNAME[int(G["kernel_id"])]
    o0, _, i0 = engine.small_eval(X, y, Z, th, float(G["jitter"]), kn, mode=0, want_grad=True)
    assert int(i0.item()) == 0 and abs(float(o0[0]) - float(G["F"])) < 1e-9 * abs(float(G["F"]))
    ws = engine._small_ws(N, M, d)
    out = engine.empty(d + 5)
    info = torch.zeros(1, dtype=torch.int32, device=engine.device)
    try:
        engine.lib.sgp_set_cu_budget(4)
        assert engine.small_supported(N, M, d, "rbf")
        o4, _, i4 = engine.small_eval(X, y, Z, th, float(G["jitter"]), kn, mode=0, want_grad=True)
        assert int(i4.item()) == 0
        assert float((o4[:d + 3] - o0[:d + 3]).abs().max()) < 1e-10 * float(o0[:d + 3].abs().max())
        engine.lib.sgp_set_cu_budget(3)
        assert not engine.small_supported(N, M, d, "rbf")
        st = engine.lib.sgp_small_eval(engine._ptr(X), d, engine._ptr(y), engine._ptr(Z), d, engine._ptr(th), N, M, d, 0, 1e-6, 0, 1,
                                       engine._ptr(out), None, C.c_void_p(info.data_ptr()), engine._ptr(ws), ws.numel(), engine._stream())
        assert st in (-2, -4)  # SGP_ERR_DIM from the shape gate or SGP_ERR_LAUNCH from the launch check
    finally:
        engine.lib.sgp_set_cu_budget(0)
    o, _, i = engine.small_eval(X, y, Z, th, float(G["jitter"]), kn, mode=0, want_grad=True)
    assert int(i.item()) == 0 and torch.equal(o[:d + 5], o0[:d + 5])


@pytest.mark.gpu
def test_small_dimension_limit(engine):
    """d = 24 is the last dimension the single launch takes (LDS of the M <= 128 class is full there); d = 25 is answered with 0 by
    sgp_small_supported and the Python layer falls back to the multi-launch path with the same numbers."""
    import ggp_amd
    from oracle import vfe_oracle as O
    assert engine.small_supported(500, 100, 24, "rbf") and not engine.small_supported(500, 100, 25, "rbf")
    g = torch.Generator().manual_seed(3)
    N, M, d = 500, 40, 25
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X.sum(1) / 5.0) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[:M].clone()
    ls = torch.full((d,), 4.0, dtype=torch.float64)
    cb = ggp_amd.CollapsedBound(X.to(engine.device), y.to(engine.device), jitter=1e-6, engine=engine)
    assert not cb._small_ok(M)
    F, gr = cb.value_and_grad(Z.to(engine.device), ls.tolist(), 1.2, 0.05, want_gz=False)
    ref = O.grads_autograd(X, y, Z, ls, 1.2, 0.05, 1e-6)
    assert abs(F - ref["F"]) < 1e-9 * abs(ref["F"]) and relerr(gr["ls"].numpy(), ref["g_ls"].numpy()) < 1e-6


@pytest.mark.gpu
def test_small_evaluations_in_flight_on_two_streams(engine):
    """The single launch's sync words live in its workspace: the engine keeps one workspace per HIP stream, so evaluations enqueued
    on two streams at once (each a different problem) neither corrupt each other's flags nor time out -- 200 alternating launches,
    every result equal to the one obtained alone."""
    Ga, Gb = load_golden("rbf_d3_small"), load_golden("rbf_d1_tiny")
    prob = []
    for G in (Ga, Gb):
        prob.append((dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine), _theta(G, engine), float(G["jitter"])))
    alone = [engine.small_eval(X, y, Z, th, j, "rbf", mode=0, want_grad=True)[0].clone() for X, y, Z, th, j in prob]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=engine.device), torch.cuda.Stream(device=engine.device)]
    outs = [[], []]
    for it in range(100):
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                X, y, Z, th, j = prob[k]
                o, _, info = engine.small_eval(X, y, Z, th, j, "rbf", mode=0, want_grad=True)
                outs[k].append((o, info))
    torch.cuda.synchronize()
    for k, G in ((0, Ga), (1, Gb)):
        n = G["X"].shape[1] + 5  # [value | d + 2 gradients | logmarg | trace]; the buffer's tail (status word, padding) is not compared
        for o, info in outs[k][-5:] + outs[k][:5]:
            assert int(info.item()) == 0 and torch.equal(o[:n], alone[k][:n])
