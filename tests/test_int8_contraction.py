"""Pass-1 contraction on the integer matrix cores (csrc/sgp_suffstats_i8.hip, include/sgp.h: sgp_set_contraction).

K'_fu in [0, 1] is split into seven balanced 8-bit digit planes, the 28 digit-pair GEMMs are exact int32 sums, the fold to
fp64 happens once per split -- so the statistics have to meet the SAME tolerances as the fp64 contraction: against the golden
fixtures (generated from the reference's stack, tests/golden/), against the CPU oracle on ragged shapes, through the bound,
and at BASELINE's full size against the fp64 contraction of the same rows.  Every test forces the integer path (mode 2; the
default mode 1 takes it for value-only calls with rows x M_p^2 >= 2^32) and checks that it actually ran.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import dev, golden_names, load_golden  # noqa: E402

pytestmark = pytest.mark.gpu
KNAME = {0: "rbf", 1: "matern32", 2: "matern52"}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def unpack(packed, M):
    h = packed.cpu().numpy()
    return h[:M * M].reshape(M, M), h[M * M:M * M + M], float(h[M * M + M]), float(h[M * M + M + 1])


@pytest.fixture()
def int8(engine):
    prev = engine.lib.sgp_set_contraction(2)
    yield engine
    engine.lib.sgp_set_contraction(prev)


@pytest.mark.parametrize("name", [n for n in golden_names() if int(load_golden(n)["kernel_id"]) in KNAME])
def test_int8_suffstats_golden(int8, name):
    engine = int8
    G = load_golden(name)
    M = G["Z"].shape[0]
    packed = engine.suffstats(dev(G["X"], engine), dev(G["y"], engine), dev(G["Z"], engine), G["ls"], float(G["sf2"]),
                              KNAME[int(G["kernel_id"])])
    assert engine.lib.sgp_contraction_last() == 1
    Phi, b, yy, kappa = unpack(packed, M)
    assert relerr(Phi, G["Phi"]) < 1e-12
    assert relerr(b, G["b"]) < 1e-12
    assert abs(yy - float(G["yy"])) < 1e-12 * abs(float(G["yy"]))
    assert abs(kappa - float(G["kappa"])) < 1e-12 * abs(float(G["kappa"]))
    assert np.array_equal(Phi, Phi.T)


@pytest.mark.parametrize("kid", [0, 1, 2])
@pytest.mark.parametrize("N,M,d", [(1, 1, 1), (15, 3, 2), (17, 130, 4), (1000, 129, 5), (4097, 257, 8), (333, 64, 18), (2500, 40, 32),
                                   (700, 384, 3), (33000, 640, 2)])
def test_int8_suffstats_vs_oracle_shapes(int8, N, M, d, kid):
    from oracle import vfe_oracle as O
    engine = int8
    g = torch.Generator().manual_seed(N * 7 + M)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.randn(N, dtype=torch.float64, generator=g)
    Z = torch.randn(M, d, dtype=torch.float64, generator=g)
    Z[0] = X[0]  # k' = 1 exactly: the top of the digit range
    ls = 0.7 + torch.rand(d, dtype=torch.float64, generator=g) * math.sqrt(d)
    st = O.suffstats(X, y, Z, ls, 1.7, kid)
    packed = engine.suffstats(X.to(engine.device), y.to(engine.device), Z.to(engine.device), ls.tolist(), 1.7, KNAME[kid])
    assert engine.lib.sgp_contraction_last() == 1
    Phi, b, yy, kappa = unpack(packed, M)
    assert relerr(Phi, st.Phi.numpy()) < 1e-12
    assert relerr(b, st.b.numpy()) < 1e-11
    assert abs(yy - st.yy) <= 1e-12 * abs(st.yy)
    assert abs(kappa - st.kappa) <= 1e-12 * abs(st.kappa)


@pytest.mark.parametrize("data", ["random", "lattice", "near_one"])
def test_int8_phi_is_the_exact_sum_of_digitised_products(int8, data):
    """The property the design rests on: Phi from the integer cores equals the EXACT sum of products of the digitised values
    q = rint(K' 2^54) -- up to the dropped digit pairs (< 6 x 2^-54 per product, zero-mean) and one fp64 fold.  Host side: q from
    the fp64 K'_fu the same assembly launch writes beside the digit planes (Kfu_out), the products in Python integers."""
    engine = int8
    g = torch.Generator().manual_seed(5)
    N, M, d = 777, 9, 2
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    Z = torch.randn(M, d, dtype=torch.float64, generator=g)
    lsv = [1.1, 0.9]
    if data == "lattice":   # the same few kernel values over and over: truncation errors that repeat instead of averaging out
        gx = torch.arange(N, dtype=torch.float64)
        X = torch.stack([(gx % 37) * 0.125, (gx // 37) * 0.25], 1)
        Z = X[::97][:M].clone()
    elif data == "near_one":  # lengthscale >> the data: every k' within 1e-3 of 1, the top of the digit range
        lsv = [60.0, 45.0]
    X, Z = X.to(engine.device), Z.to(engine.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(engine.device)
    kfu = engine.kfu_buffer(N, M)
    packed = engine.suffstats(X, y, Z, lsv, 1.0, "rbf", kfu=kfu)  # digit planes AND the fp64 block from one assembly
    assert engine.lib.sgp_contraction_last() == 1
    Mp = 128
    K = kfu.view(-1, Mp)[:N, :M].cpu().numpy()
    q = np.rint(K * 2.0 ** 54).astype(np.int64)
    exact = [[sum(int(a) * int(b) for a, b in zip(q[:, i], q[:, j])) for j in range(M)] for i in range(M)]
    Phi = packed[:M * M].view(M, M).cpu().numpy()
    worst = 0.0
    for i in range(M):
        for j in range(M):
            ref = exact[i][j] / 2.0 ** 108  # one rounding of an exact rational
            worst = max(worst, abs(Phi[i, j] - ref))
    # dropped digit pairs: < 6 x 2^-54 per product in the worst case, zero-mean; N products
    # (+ the rounding of the fp64 result itself: "near_one" sums 777 products of ~1, one ulp of Phi is 1.1e-13)
    assert worst < 6 * 2.0 ** -54 * math.sqrt(N) * 4 + float(np.spacing(np.abs(Phi).max())), worst
    assert worst < 5e-16 * float(np.abs(Phi).max())


def test_int8_phi_bit_identical_to_the_cpu_digit_oracle(int8):
    """oracle/i8_digits_oracle.py restates the kernel pair in numpy integers.  Fed with the fp64 K'_fu the assembly itself wrote, and
    folded split by split in the kernel's order (8 splits of 128 rows here, summed 0 .. 7 by reduce_phi_kernel), it must reproduce the
    HIP Phi BIT FOR BIT: every digit, every int32 group sum and every fp64 fold is determined."""
    from oracle import i8_digits_oracle as D
    engine = int8
    g = torch.Generator().manual_seed(8)
    N, M, d = 1000, 11, 2
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(engine.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(engine.device)
    Z = torch.randn(M, d, dtype=torch.float64, generator=g).to(engine.device)
    kfu = engine.kfu_buffer(N, M)
    packed = engine.suffstats(X, y, Z, [0.9, 1.2], 1.0, "matern32", kfu=kfu)
    assert engine.lib.sgp_contraction_last() == 1
    K = kfu.view(-1, 128)[:, :M].cpu().numpy()            # all 1024 padded rows (zeros beyond N)
    a = D.digits(D.quantise(K))
    Phi = np.zeros((M, M))
    for s0 in range(0, K.shape[0], 128):                   # nsplit = 8, 4 steps of 32 rows each
        Phi = Phi + D.phi_from_digits([x[s0:s0 + 128] for x in a])
    got = packed[:M * M].view(M, M).cpu().numpy()
    assert np.array_equal(np.tril(got), np.tril(Phi)), float(np.abs(got - Phi).max())
    assert np.array_equal(got, got.T)


@pytest.mark.parametrize("name", [n for n in golden_names() if int(load_golden(n)["kernel_id"]) in KNAME])
def test_int8_bound_golden(int8, name):
    """The bound through the streaming order with the statistics from the integer cores: the tolerances of
    test_bound_and_grads_golden (1e-9 on F; 1e-8 on the duplicate-inducing-row fixture)."""
    import ggp_amd
    engine = int8
    G = load_golden(name)
    cb = ggp_amd.CollapsedBound(dev(G["X"], engine), dev(G["y"], engine), kernel=KNAME[int(G["kernel_id"])], jitter=float(G["jitter"]),
                                engine=engine, form="streaming")
    F, parts = cb.value(dev(G["Z"], engine), G["ls"], float(G["sf2"]), float(G["s2"]))
    assert engine.lib.sgp_contraction_last() == 1
    ill = float(G["grad_rtol"]) > 1e-6
    tolF = (1e-8 if ill else 1e-9) * max(1.0, abs(float(G["F"])))
    assert abs(F - float(G["F"])) < tolF, (F, float(G["F"]))


@pytest.mark.parametrize("name", [n for n in golden_names() if int(load_golden(n)["kernel_id"]) in KNAME])
def test_int8_value_and_grad_golden(int8, name):
    """Value + gradient with pass 1 on the integer cores (the assembly writes the fp64 block pass 2 reads AND the digit planes):
    the tolerances of test_bound_and_grads_golden."""
    import ggp_amd
    engine = int8
    G = load_golden(name)
    cb = ggp_amd.CollapsedBound(dev(G["X"], engine), dev(G["y"], engine), kernel=KNAME[int(G["kernel_id"])], jitter=float(G["jitter"]),
                                engine=engine, form="streaming")
    F, g = cb.value_and_grad(dev(G["Z"], engine), G["ls"], float(G["sf2"]), float(G["s2"]), want_gz=True)
    assert engine.lib.sgp_contraction_last() == 1
    ill = float(G["grad_rtol"]) > 1e-6
    assert abs(F - float(G["F"])) < (1e-8 if ill else 1e-9) * max(1.0, abs(float(G["F"])))
    rt, rz = 1e-6, (1e-4 if ill else 1e-6)
    assert relerr(g["ls"].numpy(), G["g_ls"]) < rt
    assert abs(g["sf2"] - float(G["g_sf2"])) < rt * max(1.0, abs(float(G["g_sf2"])))
    assert abs(g["s2"] - float(G["g_s2"])) < rt * max(1.0, abs(float(G["g_s2"])))
    assert relerr(g["Z"].cpu().numpy(), G["g_Z"]) < rz


def test_int8_kept_block_equals_the_fp64_assembly(engine):
    """The fp64 K'_fu written beside the digit planes is bit-identical to the one kfu_assemble_kernel writes."""
    g = torch.Generator().manual_seed(3)
    N, M, d = 1500, 200, 5
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(engine.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(engine.device)
    Z = torch.randn(M, d, dtype=torch.float64, generator=g).to(engine.device)
    blocks = []
    prev = engine.lib.sgp_set_contraction(0)
    try:
        for mode in (0, 2):
            engine.lib.sgp_set_contraction(mode)
            kfu = engine.kfu_buffer(N, M)
            kfu.fill_(float("nan"))
            engine.suffstats(X, y, Z, [0.9] * d, 1.0, "matern52", kfu=kfu)
            assert engine.lib.sgp_contraction_last() == (1 if mode == 2 else 0)
            blocks.append(kfu.clone())
    finally:
        engine.lib.sgp_set_contraction(prev)
    assert torch.equal(blocks[0], blocks[1])


def test_int8_super_chunks_accumulate(int8):
    """K'_fu budget below the shard: the digit planes are built and contracted super-chunk by super-chunk into the same slabs."""
    engine = int8
    g = torch.Generator().manual_seed(11)
    N, M, d = 3000, 140, 3
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(engine.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(engine.device)
    Z = torch.randn(M, d, dtype=torch.float64, generator=g).to(engine.device)
    ref = engine.suffstats(X, y, Z, [1.0] * d, 1.2, "rbf").clone()
    engine.lib.sgp_set_kfu_budget_bytes(1024 * 256 * 8)  # 1024 rows of the padded 256 columns at a time: 3 super-chunks
    try:
        got = engine.suffstats(X, y, Z, [1.0] * d, 1.2, "rbf").clone()
        assert engine.lib.sgp_contraction_last() == 1
    finally:
        engine.lib.sgp_set_kfu_budget_bytes(0)
    assert relerr(got[:M * M].cpu().numpy(), ref[:M * M].cpu().numpy()) < 1e-14
    assert relerr(got[M * M:].cpu().numpy(), ref[M * M:].cpu().numpy()) < 1e-13


def test_int8_bit_reproducible_under_repetition(int8):
    """The digit-pair sums are exact integers and every fp64 fold / slab sum has a fixed order: repeated launches must agree bit for
    bit.  A race in the staged LDS ring (a stage read before it has landed, a slot restaged before its reads retired) would show up
    here as a flipped digit somewhere among 60 launches of 1 000+ workgroups each."""
    engine = int8
    g = torch.Generator().manual_seed(21)
    N, M, d = 150_000, 640, 3
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(engine.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(engine.device)
    Z = torch.randn(M, d, dtype=torch.float64, generator=g).to(engine.device)
    ref = engine.suffstats(X, y, Z, [1.0, 0.8, 1.3], 1.1, "rbf").clone()
    assert engine.lib.sgp_contraction_last() == 1
    out = torch.empty_like(ref)
    for _ in range(60):
        engine.suffstats(X, y, Z, [1.0, 0.8, 1.3], 1.1, "rbf", out=out)
        assert torch.equal(out, ref)


def test_int8_nan_input_reaches_the_bound(int8):
    engine = int8
    X = torch.randn(300, 2, dtype=torch.float64)
    X[17, 1] = float("nan")
    y = torch.randn(300, dtype=torch.float64)
    Z = torch.randn(20, 2, dtype=torch.float64)
    packed = engine.suffstats(X.to(engine.device), y.to(engine.device), Z.to(engine.device), [1.0, 1.0], 1.0, "rbf")
    assert engine.lib.sgp_contraction_last() == 1
    assert bool(torch.isnan(packed).any())


def test_int8_full_size_against_fp64_contraction(engine):
    """BASELINE C5 (N = 2^20 padded rows of the 1M, M = 1024, d = 8): the default mode takes the integer cores for the value-only
    evaluation; its statistics against the fp64 contraction of the same rows, and the bound from both."""
    import bench
    import ggp_amd
    N, M, d = bench.N_TOTAL, bench.M_IND, bench.DIM
    X, y, Z = bench.synth(N, M, d)
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    prev = engine.lib.sgp_set_contraction(1)
    try:
        a = engine.suffstats(Xd, yd, Zd, [bench.LS] * d, bench.SF ** 2, "rbf").clone()
        assert engine.lib.sgp_contraction_last() == 1, "the default rule must take the integer cores at the headline shape"
        engine.lib.sgp_set_contraction(0)
        b = engine.suffstats(Xd, yd, Zd, [bench.LS] * d, bench.SF ** 2, "rbf").clone()
        assert engine.lib.sgp_contraction_last() == 0
        pa, pb = a[:M * M], b[:M * M]
        # the fp64 contraction rounds each of its 2^20 / splits accumulation steps: ~1e-14 of max |Phi| between the two
        assert float((pa - pb).abs().max()) < 5e-14 * float(pb.abs().max())
        assert torch.equal(a[M * M:], b[M * M:])  # b, yy, kappa never touch the digits
        cb = ggp_amd.CollapsedBound(Xd, yd, jitter=bench.JITTER, engine=engine)
        engine.lib.sgp_set_contraction(1)
        F1, _ = cb.value(Zd, [bench.LS] * d, bench.SF ** 2, bench.SN ** 2)
        assert engine.lib.sgp_contraction_last() == 1
        engine.lib.sgp_set_contraction(0)
        F0, _ = cb.value(Zd, [bench.LS] * d, bench.SF ** 2, bench.SN ** 2)
        assert abs(F1 - F0) < 1e-9 * abs(F0), (F1, F0)
    finally:
        engine.lib.sgp_set_contraction(prev)


def test_two_contexts_in_one_process_keep_their_own_contraction_mode(engine):
    """ABI version 2: two engines with a library context each (include/sgp.h: sgp_ctx_create) -- one pinned to the fp64 matrix
    cores, one to the integer cores -- run the same pass 1 on two streams of one process; each context reports what IT ran, the
    default context (the session engine, the deprecated setters) is untouched, and the statistics agree to rounding."""
    import ggp_amd
    g = torch.Generator().manual_seed(12)
    N, M, d = 70_000, 256, 4          # rows x Mp^2 = 2^32.1: the default rule takes the integer cores here
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(engine.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(engine.device)
    Z = X[:M].clone()
    ea, eb = ggp_amd.HipEngine(own_context=True), ggp_amd.HipEngine(own_context=True)
    assert ea._ctx and eb._ctx and ea._ctx != eb._ctx
    assert ea.set_option("contraction", 0) == 1.0     # previous value: the default rule
    eb.set_option("contraction", 2)
    assert ea.would_use_i8(N, M) is False and eb.would_use_i8(N, M) is True and engine.would_use_i8(N, M) is True
    sa, sb = torch.cuda.Stream(device=engine.device), torch.cuda.Stream(device=engine.device)
    ready = torch.cuda.current_stream(engine.device).record_event()
    with torch.cuda.stream(sa):
        sa.wait_event(ready)
        pa = ea.suffstats(X, y, Z, [1.3] * d, 1.0, "rbf")
    with torch.cuda.stream(sb):
        sb.wait_event(ready)
        pb = eb.suffstats(X, y, Z, [1.3] * d, 1.0, "rbf")
    before = engine.contraction_last()
    torch.cuda.synchronize()
    assert ea.contraction_last() == 0 and eb.contraction_last() == 1
    assert engine.contraction_last() == before == engine.lib.sgp_contraction_last()   # the default context did not run anything
    assert float((pa - pb).abs().max()) < 1e-13 * float(pa.abs().max())
    assert torch.equal(pa[M * M:], pb[M * M:])
    # a small shard in the rule-following default context still takes fp64, and says so in ITS context only
    engine.suffstats(X[:3000].contiguous(), y[:3000].contiguous(), Z, [1.3] * d, 1.0, "rbf")
    assert engine.contraction_last() == 0 and eb.contraction_last() == 1
    # options of one context never leak: the conditioning gate of `ea` switched off, `eb` keeps refusing the same matrix
    z = torch.linspace(0.0, 52.0, 300, dtype=torch.float64)[:, None].to(engine.device)
    bad = engine.kuu(z, [3.0], 2.0e6, 1e-6, "rbf")
    ea.set_option("cond_limit", 0.0)
    assert int(ea.kuu_factor(bad)[1].cpu()[0]) == 0 and int(eb.kuu_factor(bad)[1].cpu()[0]) > 0
    # ABI version 3: the guard's fallback orders run in the engine's context too.  `ea` gets a K'_fu budget of 4096 rows (its extended
    # order then accumulates 18 super-chunks), `eb` keeps the default and streams the same shard in one piece -- on two streams at once;
    # the default context's budget is untouched, and both give the same statistics (double-double sums: last bits only).
    dflt_budget = engine.get_option("kfu_budget_bytes")
    ea.set_option("kfu_budget_bytes", 4096 * 256 * 8)
    linv, _ = engine.kuu_factor(engine.kuu(Z, [1.3] * d, 1.0, 1e-6, "rbf"))
    torch.cuda.synchronize()
    ready = torch.cuda.current_stream(engine.device).record_event()
    with torch.cuda.stream(sa):
        sa.wait_event(ready)
        xa = ea.suffstats_extended(X, y, Z, [1.3] * d, 1.0, linv, "rbf", level=2)
        wa = ea.suffstats_whitened_rows(X, y, Z, [1.3] * d, 1.0, linv, "rbf")
    with torch.cuda.stream(sb):
        sb.wait_event(ready)
        xb = eb.suffstats_extended(X, y, Z, [1.3] * d, 1.0, linv, "rbf", level=2)
        pb2 = eb.suffstats(X, y, Z, [1.3] * d, 1.0, "rbf")
    torch.cuda.synchronize()
    assert engine.get_option("kfu_budget_bytes") == dflt_budget and eb.get_option("kfu_budget_bytes") == dflt_budget
    assert ea.get_option("kfu_budget_bytes") == 4096 * 256 * 8
    assert float((xa - xb).abs().max()) < 1e-13 * float(xb.abs().max())
    assert float((wa - xb)[: M * M].abs().max()) < 1e-9 * float(xb[: M * M].abs().max())   # whitened rows (fp64 T^T T) against the exact-Phi route
    assert torch.equal(pb2, pb)                                                            # `eb` streamed as before, undisturbed
    del ea, eb


def test_two_host_threads_with_a_context_each_run_concurrently(engine):
    """The process-wide switches of round 3 were not thread-safe.  Two host threads, each with its own engine
    (= its own library context, workspace and HIP stream), evaluate pass 1 + the tail concurrently, one pinned to the fp64 and one to
    the integer contraction, 12 times each: every result equals the thread's own serial evaluation bit for bit, and each context
    keeps reporting its own contraction."""
    import threading
    import ggp_amd
    g = torch.Generator().manual_seed(21)
    N, M, d = 70_000, 256, 4
    X = torch.randn(N, d, dtype=torch.float64, generator=g).to(engine.device)
    y = torch.randn(N, dtype=torch.float64, generator=g).to(engine.device)
    Z = X[:M].clone()
    modes = (0, 2)
    engines = [ggp_amd.HipEngine(own_context=True) for _ in modes]
    for e, m in zip(engines, modes):
        e.set_option("contraction", m)
    serial = []
    for e in engines:
        cb = ggp_amd.CollapsedBound(X, y, jitter=1e-6, engine=e)
        serial.append(cb.value(Z, [1.3] * d, 1.0, 0.09)[0])
    torch.cuda.synchronize()
    out, errs = [[], []], []

    def work(k):
        try:
            torch.cuda.set_device(engine.device)
            with torch.cuda.stream(torch.cuda.Stream(device=engine.device)):
                cb = ggp_amd.CollapsedBound(X, y, jitter=1e-6, engine=engines[k])
                for _ in range(12):
                    out[k].append(cb.value(Z, [1.3] * d, 1.0, 0.09)[0])
                    assert engines[k].contraction_last() == (1 if modes[k] else 0)
        except Exception as exc:  # noqa: BLE001 - reported below, in the main thread
            errs.append((k, repr(exc)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert out[0] == [serial[0]] * 12 and out[1] == [serial[1]] * 12
    assert abs(serial[0] - serial[1]) / N < 1e-8   # two contraction modes of the streaming order: 1.3e-9 per datum observed here
