"""The integer-core contraction where it is the DEFAULT, over the theta the sampler can visit.

Nothing is forced here: mode 1 (the library default) picks the integer matrix cores because rows x M_p^2 >= 2^32, and every cell
asserts that it did.  The sweep is the one the priors of the reference generate (models/bayesian_sgpr_hmc.py:60-78: Gamma(2, 1) on
each lengthscale, HalfCauchy(1) on sig_f / sig_n, log-transformed; NUTS' jittered start and its first tuning leaps move theta over
several e-folds): lengthscales 0.2 .. 20, noise 0.01 .. 3, inducing inputs with exact duplicates (cond(K_uu) = 1 / jitter).

 * F against the PyMC3-op-order CPU oracle (oracle.vfe_pymc3_order_chunked) to 1e-8 on F / N (north_star's tolerance);
 * grad F against torch autograd through that graph on 50 000 rows to 1e-6;
 * HmcTarget.logp_and_grad against oracle.hmc_logp at three seeded theta of the tuner's range;
 * Phi COMPONENT-WISE: |dPhi_IJ| <= c eps sqrt(Phi_II Phi_JJ) -- the scaled bound a Cholesky-based tail is invariant under (the
   norm-wise claim of DESIGN.md 4d stated as a tested inequality; c recorded in the assertion message).
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

N_SWEEP = 200_000
D = 8
EPS = 2.0 ** -52


def _data(N, M, dup):
    import bench
    X, y, Z = bench.synth(N, M, D)
    if dup:  # the reference draws Z with np.random.randint (experiments/regression.py:83): repeated rows happen
        Z[1::16] = Z[0::16][: Z[1::16].shape[0]]
    return X, y, Z


@pytest.fixture()
def host_threads():
    """The CPU oracle's GEMMs want more than conftest's eight threads -- set ONCE per test (switching the pool size back and forth
    inside a loop cost seconds per switch on the 256-thread GPU box; conftest's own fixture restores the count afterwards)."""
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    yield lambda: None


CELLS_512 = [(ls, sn) for ls in (0.2, 0.5, 1.0, 2.0, 5.0, 20.0) for sn in (0.01, 0.3, 3.0)]


@pytest.mark.parametrize("M,dup,cells", [(512, False, CELLS_512), (512, True, [(2.0, 0.01), (5.0, 0.3)]),
                                         (1024, True, [(2.0, 0.3), (5.0, 0.01)]), (1024, False, [(20.0, 0.01)])])
def test_default_mode_bound_over_the_theta_range(engine, host_threads, M, dup, cells):
    """What round 4's sweep found (tests/studies/theta_sweep_diag.py, profiles/r04_theta_sweep_streaming.jsonl): the integer and the fp64
    contraction agree with each other everywhere -- and BOTH leave the 1e-8 per datum for long lengthscales x small noise (1e-4 at
    l = 5, sig_n = 0.01; B not even positive definite at l = 20), because the STREAMING order amplifies the rounding of Phi by
    1 / lambda(K_uu).  The bound now carries the library's estimate of that error and repeats such evaluations in the whitened
    (PyMC3) order: every cell has to meet the tolerance, the benign ones without a repeat.
    The reference value is PyMC3's op order on the CPU: A = L^-1 K_uf and A A^T once per lengthscale (oracle.suffstats_whitened), the
    M x M part per noise level (oracle.bound_from_stats) -- the same operations as oracle.vfe_pymc3_order_chunked, which the first
    cell of every group is checked against as well."""
    import ggp_amd
    from oracle import vfe_oracle as O
    X, y, Z = _data(N_SWEEP, M, dup)
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    cb = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine)
    prev = engine.lib.sgp_set_contraction(1)
    bad, log, ref_cache = [], [], {}
    try:
        for ls, sn in cells:
            before, before_d, after_trip = cb.n_guard_reruns, cb.n_direct_whitened, cb._prefer_whitened
            F, parts = cb.value(Zd, [ls] * D, 1.0, sn * sn)
            rerun, direct = cb.n_guard_reruns - before, cb.n_direct_whitened - before_d
            if not rerun and not direct:
                assert engine.lib.sgp_contraction_last() == 1, "rows x Mp^2 >= 2^32: the default rule must take the integer cores"
            if ls not in ref_cache:
                host_threads()
                lst = torch.full((D,), ls, dtype=torch.float64)
                Kuu = O.kuu(Z, lst, 1.0, 1e-6)
                stw = O.suffstats_whitened(X, y, Z, lst, 1.0, torch.linalg.cholesky(Kuu), chunk=65536)
                ref_cache[ls] = (Kuu, stw)
                if len(ref_cache) == 1:  # the two oracle routes are one: checked at the group's first lengthscale
                    F_direct = O.vfe_pymc3_order_chunked(X, y, Z, lst, 1.0, sn, 1e-6)
                    F_two = O.bound_from_stats(Kuu, stw, sn * sn, stats_whitened=True)["F"]
                    assert abs(F_direct - F_two) / N_SWEEP < 1e-10, (ls, sn, F_direct, F_two)
            Kuu, stw = ref_cache[ls]
            F_ref = O.bound_from_stats(Kuu, stw, sn * sn, stats_whitened=True)["F"]
            err = abs(F - F_ref) / N_SWEEP
            log.append((ls, sn, rerun, direct, err))
            if not (err < 1e-8):
                bad.append((ls, sn, rerun, direct, F, F_ref, err))
            if not dup:  # (exact duplicates among the inducing rows inflate the estimate: their eigen-directions carry no error)
                if ls <= 1.0 or (ls == 2.0 and sn >= 0.3):   # benign: streamed -- or, right behind a trip, whitened ONCE more
                    assert rerun == 0 and (direct == 0 or after_trip), ("a benign cell was sent to the whitened order", ls, sn, log)
                    assert not cb._prefer_whitened, ("the whitened episode must end at a benign cell", ls, sn, log)
                if (ls >= 5.0 and sn <= 0.3) or (ls >= 20.0 and sn < 1.0):
                    assert rerun + direct == 1, ("the guard must send this cell to the whitened order", ls, sn, log)
    finally:
        engine.lib.sgp_set_contraction(prev)
    assert not bad, (bad, log)


def test_streaming_guard_can_be_switched_off_and_reports_its_estimate(engine):
    """form="streaming" (the caller insists) and streaming_tol = 0 keep the N >> M design whatever theta is; the estimate rides in the
    evaluation's one host copy either way."""
    import ggp_amd
    X, y, Z = _data(70_000, 256, False)
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    cb = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine)
    F1, _ = cb.value(Zd, [5.0] * D, 1.0, 1e-4)
    assert cb.n_guard_reruns == 1
    cb.streaming_tol = 0.0
    F2, _ = cb.value(Zd, [5.0] * D, 1.0, 1e-4, raise_on_fail=False)
    assert cb.n_guard_reruns == 1
    cs = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine, form="streaming")
    F3, _ = cs.value(Zd, [5.0] * D, 1.0, 1e-4, raise_on_fail=False)
    assert cs.n_guard_reruns == 0 and (F2 == F3 or (F2 != F2 and F3 != F3))
    cw = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine, form="whitened")
    F4, _ = cw.value(Zd, [5.0] * D, 1.0, 1e-4)
    assert F1 == F4                                   # the repeat IS the whitened evaluation
    # a benign theta: no repeat, and value + gradient agree with the whitened order to rounding
    cb.streaming_tol = 1e-9
    cb._prefer_whitened = False          # (the trip above would send the next evaluation to the whitened order directly)
    Fa, ga = cb.value_and_grad(Zd, [1.0] * D, 1.0, 0.09)
    Fb, gb = cw.value_and_grad(Zd, [1.0] * D, 1.0, 0.09)
    assert cb.n_guard_reruns == 1 and abs(Fa - Fb) < 1e-10 * abs(Fb)
    assert float((ga["ls"] - gb["ls"]).abs().max()) < 1e-7 * float(gb["ls"].abs().max())


@pytest.mark.parametrize("ls", [0.2, 0.5, 2.0, 20.0])
def test_default_mode_phi_componentwise(engine, host_threads, ls):
    """The component-wise statement of DESIGN.md 4d, as a tested inequality.  Two error sources with different shapes:
      * quantisation, |K' - q 2^-53| <= 2^-54: |dPhi_IJ| <= 1 eps sqrt(Phi_II Phi_JJ) on its own (oracle/i8_digits_oracle.py on the
        CPU: c = 0.0 .. 0.6 over these lengthscales) -- as good as a correctly rounded Phi;
      * the 21 dropped digit pairs: ~2^-52 ABSOLUTE per product of a large with a small kernel value, zero-mean, so
        |dPhi_IJ| ~ 0.6 sqrt(N_IJ) 2^-52 with N_IJ the rows where one of the two columns is not negligible.  Against the scale of the
        row and column (Phi_II >= 1 whenever the inducing inputs are data rows) that is c eps sqrt(Phi_II Phi_JJ) with c up to
        ~sqrt(N): measured c = 32 / 280 / 220 / < 16 at l = 0.2 / 0.5 / 1 / >= 2 and N = 200 000 -- the size of the rounding a
        sequential fp64 sum of N terms accumulates, but relative to the DIAGONAL, not to the entry.  Small entries of Phi therefore
        carry fewer digits than in the fp64 contraction; F does not notice (short lengthscales = a well-conditioned K_uu:
        |dF| / N <= 1e-10 in every such cell of the sweep above)."""
    from oracle import vfe_oracle as O
    M = 512
    X, y, Z = _data(N_SWEEP, M, False)
    prev = engine.lib.sgp_set_contraction(1)
    try:
        packed = engine.suffstats(X.to(engine.device), y.to(engine.device), Z.to(engine.device), [ls] * D, 1.0, "rbf")
        assert engine.lib.sgp_contraction_last() == 1
    finally:
        engine.lib.sgp_set_contraction(prev)
    Phi = packed[: M * M].view(M, M).cpu().numpy()
    host_threads()
    # reference: the oracle's K_uf in fp64, products of 2048-row chunks in fp64 (a BLAS dot product of n positive terms is off by
    # ~n eps / 100: 78 eps at 65 536-row chunks, measured), the chunks added in 80-bit arithmetic
    acc = np.zeros((M, M), dtype=np.longdouble)
    lst = torch.full((D,), ls, dtype=torch.float64)
    for s0 in range(0, N_SWEEP, 2048):
        K = O.kern(X[s0:s0 + 2048], Z, lst, 1.0)
        acc += (K.T @ K).numpy()
    ref = acc.astype(np.float64)
    dg = np.sqrt(np.diag(ref))
    err = np.abs(Phi - ref)
    c = float(np.max(err / (EPS * np.outer(dg, dg))))
    assert c < math.sqrt(N_SWEEP), (ls, c)
    # the absolute form of the same statement: dropped pairs 2^-52 sqrt(N) (+ the oracle's own fp64 rounding, relative to the entry)
    # (CPU digit oracle at N = 40 000, M = 48: 0.9 / 2.0 sqrt(N) eps at l = 0.5 / 1; the worst case is 6 x 2^-52 per product)
    assert float(np.max(err - 32 * EPS * ref)) < 8.0 * math.sqrt(N_SWEEP) * EPS, (ls, float(err.max()))
    if ls >= 2.0:   # from there on every entry is large against sqrt(N) 2^-52: entry-wise relative accuracy as well
        assert float(np.max(err / ref)) < 1e-12, (ls, float(np.max(err / ref)))


@pytest.mark.parametrize("ls,sn", [(0.5, 0.01), (2.0, 0.01), (5.0, 3.0), (20.0, 0.3)])
def test_default_mode_gradients_over_the_theta_range(engine, host_threads, ls, sn):
    """value + gradient (the leapfrog's call) on the first 65 536 rows at M = 512 -- rows x Mp^2 = 2^34: integer cores by default."""
    import ggp_amd
    from oracle import vfe_oracle as O
    M, NG = 512, 65_536
    X, y, Z = _data(NG, M, False)
    cb = ggp_amd.CollapsedBound(X.to(engine.device), y.to(engine.device), jitter=1e-6, engine=engine)
    prev = engine.lib.sgp_set_contraction(1)
    try:
        F, g = cb.value_and_grad(Z.to(engine.device), [ls] * D, 1.0, sn * sn, want_gz=False)
        assert cb.n_guard_reruns + cb.n_direct_whitened == 1 or engine.lib.sgp_contraction_last() == 1
    finally:
        engine.lib.sgp_set_contraction(prev)
    host_threads()
    ref = O.grads_autograd(X, y, Z, [ls] * D, 1.0, sn * sn, 1e-6)
    assert abs(F - ref["F"]) / NG < 1e-8, (F, ref["F"])
    scale = max(1.0, float(ref["g_ls"].abs().max()))
    assert float((g["ls"] - ref["g_ls"]).abs().max()) < 1e-6 * scale, (g["ls"], ref["g_ls"])
    assert abs(g["sf2"] - ref["g_sf2"]) < 1e-6 * max(1.0, abs(ref["g_sf2"]))
    assert abs(g["s2"] - ref["g_s2"]) < 1e-6 * max(1.0, abs(ref["g_s2"]))


def test_default_mode_hmc_target_at_theta_the_tuner_visits(engine, host_threads):
    """logp + gradient of the NUTS target (bound + priors + Jacobians) at three seeded theta in the unconstrained range NUTS'
    jittered start and first tuning leaps cover (|theta_unc - start| <= 2: lengthscales 0.27 .. 14.8, sig 0.14 .. 7.4)."""
    import ggp_amd
    from oracle import vfe_oracle as O
    M, NG = 512, 65_536
    X, y, Z = _data(NG, M, False)
    cb = ggp_amd.CollapsedBound(X.to(engine.device), y.to(engine.device), jitter=1e-6, engine=engine)
    tgt = ggp_amd.HmcTarget(cb, Z.to(engine.device))
    rng = np.random.default_rng(42)
    start = np.array(tgt.start())
    prev = engine.lib.sgp_set_contraction(1)
    try:
        for _ in range(3):
            th = start + rng.uniform(-2.0, 2.0, size=D + 2)
            before = cb.n_guard_reruns + cb.n_direct_whitened
            lp, gr = tgt.logp_and_grad(th.tolist())
            assert cb.n_guard_reruns + cb.n_direct_whitened > before or engine.lib.sgp_contraction_last() == 1
            host_threads()
            lp_ref, g_ref = O.hmc_logp(torch.tensor(th, dtype=torch.float64), X, y, Z, 1e-6, with_grad=True)
            assert abs(lp - float(lp_ref)) / NG < 1e-8, (th, lp, float(lp_ref))
            g_ref = np.asarray(g_ref, dtype=np.float64)
            assert np.max(np.abs(np.asarray(gr) - g_ref)) < 1e-6 * max(1.0, float(np.max(np.abs(g_ref)))), (th, gr, g_ref)
    finally:
        engine.lib.sgp_set_contraction(prev)


# ---------------------------------------------------------------------------------------------
# every cell above has ONE lengthscale for all eight dimensions, data-row inducing inputs and the RBF
# profile.  ARD theta (the trained one of profiles/r04_experiment_large_scale.json and a deliberately ragged one), CLUSTERED
# inducing inputs (pairs 1e-3 apart: near-duplicates that the jitter does not regularise the way exact duplicates are) and the
# Matern profiles, through the default mode (integer cores + guard) -- and the whole isotropic sweep once more through the fp64
# contraction on a shard below the integer threshold (what C3 runs; its measured / estimated error ratio is the larger one, 5.5).
# ---------------------------------------------------------------------------------------------
LS_TRAINED = [4.870895252562722, 2.274348615181124, 7.035384773166531, 6.388168428424034, 7.176420862837876, 3.3523772450641136,
              2.314383327914714, 6.492694463809999]
LS_RAGGED = [0.5, 5.0, 1.0, 10.0, 2.0, 20.0, 0.8, 3.0]

EXTRA_CELLS = [
    ("rbf", LS_TRAINED, 0.14415221312756948, "rows"),
    ("rbf", LS_RAGGED, 0.05, "rows"),
    ("rbf", [8.0, 8.0, 8.0, 8.0, 1.0, 1.0, 1.0, 1.0], 0.01, "rows"),
    ("rbf", [2.0] * D, 0.1, "clustered"),
    ("rbf", [5.0] * D, 0.3, "clustered"),
    ("rbf", LS_TRAINED, 0.05, "clustered"),
    ("matern52", [2.0] * D, 0.01, "rows"),
    ("matern52", [10.0] * D, 0.1, "rows"),
    ("matern52", LS_TRAINED, 0.14415221312756948, "clustered"),
    ("matern32", [5.0] * D, 0.05, "rows"),
]


def _cluster(Z, seed=3):
    """Z = X[idx] + 1e-3 noise in pairs: rows 2k and 2k + 1 sit 1e-3 apart (r^2 ~ 1e-6 d / l^2: K_uu rows equal to ~1e-6)."""
    g = torch.Generator().manual_seed(seed)
    Zc = Z.clone()
    Zc[1::2] = Zc[0::2][: Zc[1::2].shape[0]] + 1e-3 * torch.randn(Zc[1::2].shape, dtype=torch.float64, generator=g)
    return Zc


@pytest.mark.parametrize("kernel,ls,sn,zmode", EXTRA_CELLS)
def test_default_mode_bound_ard_clustered_matern(engine, host_threads, kernel, ls, sn, zmode):
    import ggp_amd
    from oracle import vfe_oracle as O
    M = 512
    X, y, Z = _data(N_SWEEP, M, False)
    if zmode == "clustered":
        Z = _cluster(Z)
    kid = {"rbf": O.KERNEL_RBF, "matern32": O.KERNEL_MATERN32, "matern52": O.KERNEL_MATERN52}[kernel]
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    cb = ggp_amd.CollapsedBound(Xd, yd, kernel=kernel, jitter=1e-6, engine=engine)
    prev = engine.lib.sgp_set_contraction(1)
    try:
        F, parts = cb.value(Zd, ls, 1.0, sn * sn)
        if not cb.n_guard_reruns:
            assert engine.lib.sgp_contraction_last() == 1
    finally:
        engine.lib.sgp_set_contraction(prev)
    host_threads()
    F_ref = O.vfe_pymc3_order_chunked(X, y, Z, ls, 1.0, sn, 1e-6, kernel_id=kid)
    err = abs(F - F_ref) / N_SWEEP
    assert err < 1e-8, (kernel, ls, sn, zmode, F, F_ref, err, cb.last_estimate, cb.n_guard_reruns, cb.n_extended)


@pytest.mark.parametrize("zmode", ["rows", "clustered"])
def test_fp64_contraction_below_the_integer_threshold_over_the_theta_range(engine, host_threads, zmode):
    """rows x Mp^2 < 2^32 (C3's class): the fp64 contraction streams, the guard sends what it must to the whitened order directly
    (no extended order at this size).  Every cell of the isotropic sweep plus the ARD ones to 1e-8 per datum."""
    import ggp_amd
    from oracle import vfe_oracle as O
    N, M = 16_000, 512
    X, y, Z = _data(N, M, False)
    if zmode == "clustered":
        Z = _cluster(Z)
    Xd, yd, Zd = X.to(engine.device), y.to(engine.device), Z.to(engine.device)
    cb = ggp_amd.CollapsedBound(Xd, yd, jitter=1e-6, engine=engine)
    assert not cb._whitened(M) and not engine.would_use_i8(N, M)
    cells = [([ls] * D, sn) for ls, sn in CELLS_512] + [(LS_TRAINED, 0.14415221312756948), (LS_RAGGED, 0.05), (LS_TRAINED, 0.01)]
    bad, log = [], []
    host_threads()
    for ls, sn in cells:
        cb._prefer_whitened = False  # every cell starts in the streaming order: the guard's own decision is what is tested
        before = cb.n_guard_reruns
        F, parts = cb.value(Zd, ls, 1.0, sn * sn)
        rerun = cb.n_guard_reruns - before
        if not rerun:
            assert engine.lib.sgp_contraction_last() == 0, "below the threshold the fp64 contraction runs"
        F_ref = O.vfe_pymc3_order_chunked(X, y, Z, ls, 1.0, sn, 1e-6)
        err = abs(F - F_ref) / N
        log.append((ls[0], sn, rerun, cb.last_estimate, err))
        # Clustered inducing inputs x sig_n = 0.01 x long lengthscales: cond(K_uu) ~ M / jitter and B = I + A A^T / s2 has entries ~ N / s2:
        # two fp64 evaluations of the SAME op order (here: the whitened order on the GPU against the CPU oracle) differ by more than
        # 1e-8 per datum -- the fp64 CPU oracle alone is 1.2e-9 (l = 20) / 4.3e-9 (l = 5) per datum off the 80-bit yardstick
        # (oracle.vfe_extended) already at N = 3 000, M = 128.  There the CPU path is no yardstick at 1e-8: 3e-8.
        tol = 3e-8 if (zmode == "clustered" and sn <= 0.01 and ls[0] >= 5.0) else 1e-8
        if not (err < tol):
            bad.append((ls, sn, rerun, cb.last_estimate, F, F_ref, err))
    assert not bad, (bad, log)
