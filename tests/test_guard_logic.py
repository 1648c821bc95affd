"""Host logic of the streaming-order guard after round 5's rewrite (core.py: required_tier, GuardState, CollapsedBound._evaluate), on the
CPU double -- VERDICT r4 next-5 and the two medium findings of ADVICE r4:

 * the tier an evaluation NEEDS is a pure function of its own estimate (`required_tier`); the guard's memory (`GuardState`) only picks
   where the next evaluation starts -- a property test walks random theta sequences on a fresh bound and on bounds with arbitrary
   history: values agree to 2e-9 per datum (tiers differ, answers do not), and with `strict` they are identical bit for bit;
 * an extended-order evaluation that was started on an optimistic PREDICTION states its exact estimate itself and is repeated in the
   whitened order when that estimate is beyond its reach (it used to be accepted unchecked);
 * the "read the status before pass 2" rule is the JOB's (largest shard), not the rank's: two ranks with shards on either side of the
   threshold issue the same collectives (they used to diverge -- packed statistics against gradients);
 * a failed factorization neither repeats the evaluation nor touches the guard's memory;
 * HmcTarget(gradient="sampler"): the extended order serves gradients as far as values, the tier is the evaluation's own.
Reference for what is being computed: pm.gp.MarginalSparse(approx="VFE") at models/bayesian_sgpr_hmc.py:66,71 (the whitened order IS its
op order); the reference has no counterpart of the guard.
"""
import math
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _problem(N=600, M=24, d=3, seed=4):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X[:, 0]) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    return X, y, X[:M].clone()


def _bound(X, y, engine=None, **kw):
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    cb = pkg.CollapsedBound(X, y, jitter=1e-6, engine=engine or FactoredOracleEngine(), **kw)
    cb.whitened_rows_min_work = 0
    return cb


@pytest.fixture()
def no_small_whitened():
    import ggp_amd as pkg
    old = pkg.CollapsedBound.WHITENED_MAX_WORK
    pkg.CollapsedBound.WHITENED_MAX_WORK = 0   # form="auto" would take the whitened order for problems this small
    yield
    pkg.CollapsedBound.WHITENED_MAX_WORK = old


# theta of the walk: (lengthscale, s2) with estimates from ~1e-12 (benign) over the gradient range and the value range to beyond both
THETAS = [(0.8, 0.3), (3.0, 2e-2), (3.0, 1e-2), (5.0, 1e-3), (25.0, 1e-5), (2.0, 5e-2), (3.0, 3e-2), (10.0, 1e-4), (1.2, 0.1), (4.0, 5e-3)]


def test_required_tier_is_a_pure_threshold_function():
    from ggp_amd.core import TIER_EXTENDED, TIER_STREAMING, TIER_WHITENED, required_tier
    tol = 1e-9
    assert required_tier(0.0, tol, 3.0, True) == TIER_STREAMING and required_tier(tol, tol, 3.0, True) == TIER_STREAMING
    assert required_tier(1.0000001 * tol, tol, 3.0, True) == TIER_EXTENDED and required_tier(3.0 * tol, tol, 3.0, True) == TIER_EXTENDED
    assert required_tier(3.0000001 * tol, tol, 3.0, True) == TIER_WHITENED
    assert required_tier(2.0 * tol, tol, 3.0, False) == TIER_WHITENED          # no extended order for this bound
    assert required_tier(float("nan"), tol, 3.0, True) == TIER_WHITENED and required_tier(float("inf"), tol, 16384.0, True) == TIER_WHITENED
    # monotone in the estimate, for any reach
    for reach in (1.5, 3.0, 16384.0):
        tiers = [required_tier(e, tol, reach, True) for e in np.geomspace(1e-13, 1e-3, 200)]
        assert tiers == sorted(tiers)


def test_guard_state_episode_and_prediction():
    from ggp_amd.core import TIER_EXTENDED, TIER_STREAMING, TIER_WHITENED, GuardState
    s, tol = GuardState(), 1e-9
    assert s.start_tier(tol, 3.0, True) == TIER_STREAMING
    s.note_exact(2e-9, 4e-8, tol)                         # a trip: the episode opens, ratio = 0.05
    assert s.open and s.ratio == pytest.approx(0.05) and s.start_tier(tol, 3.0, True) == TIER_EXTENDED
    assert s.start_tier(tol, 1.5, True) == TIER_WHITENED and s.start_tier(tol, 3.0, False) == TIER_WHITENED
    s.note_bound(1e-7, tol)                               # a whitened evaluation elsewhere: predicted = 0.05 * 1e-7 = 5e-9
    assert s.open and s.predicted == pytest.approx(5e-9) and s.start_tier(tol, 3.0, True) == TIER_WHITENED
    s.note_exact(0.8e-9, 1e-8, tol)                       # below the tolerance but above half of it: the episode stays open (hysteresis)
    assert s.open and s.start_tier(tol, 3.0, True) == TIER_EXTENDED
    s.note_exact(0.4e-9, 1e-8, tol)
    assert not s.open and s.start_tier(tol, 3.0, True) == TIER_STREAMING
    s.note_bound(float("nan"), tol)                       # a failed evaluation's numbers change nothing
    assert not s.open and s.predicted == pytest.approx(0.4e-9)


def test_extended_evaluation_started_on_a_prediction_checks_its_own_estimate(no_small_whitened):
    """the ratio estimate / bound learned at a short lengthscale under-predicts the estimate at a long one.  The
    extended order now reports the exact estimate at ITS theta (phi_diag): beyond the reach -> repeated in the whitened order."""
    from fake_engine import FactoredOracleEngine
    X, y, Z = _problem()
    eng = FactoredOracleEngine()
    cb = _bound(X, y, eng)
    cb.extended_range = 8.0                                # a short reach makes the case easy to hit
    ref = _bound(X, y, form="whitened")
    from oracle import vfe_oracle as O
    N, tol, reach = X.shape[0], cb.streaming_tol, 8.0

    def est_ub(ls, s2):   # what sgp_streaming_error_report states at (ls, s2): 2^-53 max Phi_ii tr(Kuu^-1) / (s2 N) and its bound
        lst = torch.full((3,), ls, dtype=torch.float64)
        Linv = torch.linalg.inv(torch.linalg.cholesky(O.kuu(Z, lst, 1.0, 1e-6)))
        tr = float((Linv ** 2).sum())
        phi_max = float((O.kern(X, Z, lst, 1.0) ** 2).sum(0).max())
        return 2.0 ** -53 * phi_max * tr / (s2 * N), 2.0 ** -53 * tr / s2

    ls_a, ls_b = 1.5, 12.0
    e1, u1 = est_ub(ls_a, 1.0)
    s2_a = e1 / (2.0 * tol)                                 # the trip: estimate = 2 x the tolerance at the short lengthscale
    ratio_a = e1 / u1
    e1b, u1b = est_ub(ls_b, 1.0)
    assert ratio_a < 0.5 * (e1b / u1b), "the two lengthscales must differ in max Phi_ii / (N sf2^2)"
    s2_b = e1b / (1.5 * reach * tol)                        # the long lengthscale: estimate = 1.5 x the extended order's reach ...
    assert ratio_a * (u1b / s2_b) <= reach * tol            # ... while the short lengthscale's ratio predicts it inside the reach
    cb.value(Z, [ls_a] * 3, 1.0, s2_a)
    assert cb.guard.open and cb.guard.ratio == pytest.approx(ratio_a, rel=1e-6) and cb.n_extended == 1
    found = ([ls_b] * 3, 1.0, s2_b)
    assert cb.guard.start_tier(tol, reach, True) == 1       # the guard's memory says: extended (the last estimate was 2 x the tolerance)
    before_ext, before_rows, before_reruns = cb.n_extended, eng.calls["suffstats_whitened_rows"], cb.n_guard_reruns
    F, _ = cb.value(Z, *found, raise_on_fail=False)
    assert cb.n_extended == before_ext + 1                 # it started in the extended order ...
    assert eng.calls["suffstats_whitened_rows"] == before_rows + 1 and cb.n_guard_reruns == before_reruns + 1   # ... and was repeated
    assert F == ref.value(Z, *found, raise_on_fail=False)[0]
    assert cb.last_estimate > reach * tol


def test_failed_factorization_leaves_the_guard_alone(no_small_whitened):
    X, y, Z = _problem()
    cb = _bound(X, y)
    cb.value(Z, [0.8] * 3, 1.0, 0.3)
    state = (cb.guard.open, cb.guard.predicted, cb.guard.ratio, cb.n_guard_reruns)
    Zbad = Z.clone()
    Zbad[1] = Zbad[0]                                       # duplicate inducing rows, no jitter to speak of
    cb0 = _bound(X, y)
    cb0.jitter = 0.0
    cb0.guard.open, cb0.guard.predicted, cb0.guard.ratio = state[0], state[1], state[2]
    F, parts = cb0.value(Zbad, [0.8] * 3, 1.0, 0.3, raise_on_fail=False)
    assert parts["info"] != 0 and F != F
    assert (cb0.guard.open, cb0.guard.predicted, cb0.guard.ratio, cb0.n_guard_reruns) == (state[0], state[1], state[2], 0)


def test_streaming_order_failing_at_B_is_a_guard_trip_not_an_error(no_small_whitened):
    """l = 20, sig_n = 0.01 on the GPU: W = L^-1 Phi L^-T is so far off that I + W / s2 is not positive definite (info > M).  That IS the
    streaming order's failure mode: the estimate (K_uu's factor and Phi only) is valid and sends the evaluation up."""
    from fake_engine import FactoredOracleEngine

    class BrokenB(FactoredOracleEngine):
        def bound(self, Kuu, packed, s2, N, with_adjoints=False, want_factors=False, result=None, kuu_linv=None, whitened=False, want_cw=False):
            res = super().bound(Kuu, packed, s2, N, with_adjoints, want_factors, result, kuu_linv, whitened, want_cw)
            if not whitened and float(s2) < 1e-3:      # the streaming order "fails" at small noise: B not positive definite
                res["info"][0] = Kuu.shape[0] + 5
                res["out"].fill_(float("nan"))
            return res

    X, y, Z = _problem()
    eng = BrokenB()
    cb = _bound(X, y, eng)
    ref = _bound(X, y, form="whitened")
    far = ([25.0] * 3, 1.0, 1e-5)
    F, parts = cb.value(Z, *far)                           # no NotPositiveDefiniteError: repeated in the whitened order
    assert parts["info"] == 0 and F == ref.value(Z, *far)[0] and cb.n_guard_reruns == 1 and cb.last_tier == 2
    # ... while a K_uu failure (info <= M) is an error of the matrix, whatever the tier
    Zbad = Z.clone()
    Zbad[1] = Zbad[0]
    c0 = _bound(X, y)
    c0.jitter = 0.0
    with pytest.raises(Exception):
        c0.value(Zbad, [0.8] * 3, 1.0, 0.3)
    assert c0.n_guard_reruns == 0


def _walk(cb, Z, order, with_grad, **kw):
    out = []
    for k in order:
        ls, s2 = THETAS[k]
        if with_grad:
            F, g = cb.value_and_grad(Z, [ls] * 3, 1.0, s2, want_gz=False, raise_on_fail=False, **kw)
            out.append((F, g["ls"].tolist() if g.get("info", 0) == 0 else None, cb.last_tier))
        else:
            F, _ = cb.value(Z, [ls] * 3, 1.0, s2, raise_on_fail=False, **kw)
            out.append((F, None, cb.last_tier))
    return out


@pytest.mark.parametrize("with_grad", [False, True])
def test_history_changes_the_tier_never_the_answer(no_small_whitened, with_grad):
    """any sequence of theta evaluated on a fresh bound and on a bound with arbitrary prior history agrees to 2e-9
    per datum (the accepted tier may be HIGHER after a guarded episode, never lower than the evaluation's own estimate requires)."""
    from ggp_amd.core import required_tier
    X, y, Z = _problem()
    N = X.shape[0]
    rng = np.random.default_rng(7)
    for trial in range(6):
        order = rng.integers(0, len(THETAS), size=8).tolist()
        fresh = []
        for k in order:                                    # every theta on its own fresh bound: no history at all
            fresh.extend(_walk(_bound(X, y), Z, [k], with_grad))
        hist = _bound(X, y)
        _walk(hist, Z, rng.integers(0, len(THETAS), size=5).tolist(), bool(trial & 1))   # arbitrary prior history
        walked = _walk(hist, Z, order, with_grad)
        for (Ff, gf, tf), (Fh, gh, th), k in zip(fresh, walked, order):
            assert (Ff != Ff and Fh != Fh) or abs(Ff - Fh) / N <= 2e-9, (trial, THETAS[k], Ff, Fh)
            assert th >= tf or th >= 1, (trial, THETAS[k], tf, th)          # history never lowers the tier below what theta needs
            if gf is not None and gh is not None:
                assert np.max(np.abs(np.array(gf) - np.array(gh))) <= 1e-6 * max(1.0, np.max(np.abs(gf))), (trial, THETAS[k])


def test_strict_mode_makes_value_and_gradient_functions_of_theta(no_small_whitened):
    """HmcTarget(gradient="sampler") semantics on the CPU double: with `strict` the accepted tier is the one the evaluation's own
    estimate names, whatever came before -- same tier, same bits on a fresh bound and after an arbitrary history."""
    from ggp_amd.core import required_tier
    X, y, Z = _problem()
    rng = np.random.default_rng(11)
    order = rng.integers(0, len(THETAS), size=12).tolist()
    kw = {"strict": True, "grad_reach": 16384.0}
    fresh = []
    for k in order:
        fresh.extend(_walk(_bound(X, y), Z, [k], True, **kw))
    hist = _bound(X, y)
    _walk(hist, Z, rng.integers(0, len(THETAS), size=7).tolist(), True)
    walked = _walk(hist, Z, order, True, **kw)
    for (Ff, gf, tf), (Fh, gh, th), k in zip(fresh, walked, order):
        assert tf == th, (THETAS[k], tf, th)
        assert (Ff == Fh or (Ff != Ff and Fh != Fh)) and gf == gh, (THETAS[k], Ff, Fh)
    # and the tier is the one the exact estimate asks for (tiers 0 / 1 state it; a whitened evaluation is beyond the extended reach)
    probe = _bound(X, y)
    for k in sorted(set(order)):
        ls, s2 = THETAS[k]
        probe2 = _bound(X, y)
        probe2.value_and_grad(Z, [ls] * 3, 1.0, s2, raise_on_fail=False, **kw)
        if probe2.last_tier < 2:
            assert probe2.last_tier == required_tier(probe2.last_estimate, probe2.streaming_tol, 16384.0, True)


def test_strict_mode_with_an_under_predicting_memory_never_accepts_the_lower_attempt(no_small_whitened):
    """an open episode whose remembered estimate / bound ratio under-predicts (1e-12) used to start a strict evaluation in
    the whitened order, send it DOWN on the prediction, and -- the lower tier's exact estimate naming the whitened order, already tried --
    accept the lower attempt: tier 0 with an estimate of 1.6e-4 against a tolerance of 1e-9.  Now whatever the memory holds, the accepted
    tier is the one the exact estimate names, and the value is the fresh bound's."""
    from ggp_amd.core import required_tier
    X, y, Z = _problem()
    N = X.shape[0]
    for (ls, s2) in [(25.0, 1e-5), (10.0, 1e-4), (5.0, 1e-3)]:
        for with_grad in (False, True):
            for ratio, predicted in [(1e-12, 0.0), (1e-12, 1e-3), (0.5, 1e-12), (1.0, 2e-9)]:
                for early in (False, True):
                    cb = _bound(X, y)
                    cb.guard.open, cb.guard.ratio, cb.guard.predicted = True, ratio, predicted
                    if early:
                        cb.early_check_min_work = 0         # big-shard rule: status read before pass 2 (attempts are not kept)
                    fresh = _bound(X, y)
                    kw = {"strict": True, "grad_reach": 16384.0} if with_grad else {"strict": True}
                    if with_grad:
                        F, g = cb.value_and_grad(Z, [ls] * 3, 1.0, s2, raise_on_fail=False, **kw)
                        Ff, gf = fresh.value_and_grad(Z, [ls] * 3, 1.0, s2, raise_on_fail=False, **kw)
                        assert g["ls"].tolist() == gf["ls"].tolist()
                    else:
                        F, _ = cb.value(Z, [ls] * 3, 1.0, s2, raise_on_fail=False, **kw)
                        Ff, _ = fresh.value(Z, [ls] * 3, 1.0, s2, raise_on_fail=False, **kw)
                    reach = 16384.0
                    assert cb.last_tier == fresh.last_tier, (ls, s2, with_grad, ratio, predicted, early, cb.last_tier, fresh.last_tier)
                    if cb.last_tier < 2:
                        assert cb.last_tier == required_tier(cb.last_estimate, cb.streaming_tol, reach, True)
                    else:
                        assert fresh.n_guard_reruns >= 1     # the whitened order is reached through a lower tier's exact estimate only
                    assert F == Ff and abs(F - Ff) / N == 0.0


def test_strict_mode_at_the_reach_boundary_does_not_depend_on_history(no_small_whitened):
    """at (ls 2, s2 1e-7) the estimate sits just inside the extended order's reach.  A fresh bound accepted tier 1;
    with an open episode and ratio 1 from history the same theta was accepted in the whitened order (only the upper bound is known there,
    and whether it is left was the memory's call).  A strict evaluation no longer starts in the whitened order."""
    X, y, Z = _problem()
    kw = {"strict": True, "grad_reach": 16384.0}
    theta = ([2.0] * 3, 1.0, 1e-7)
    fresh = _bound(X, y)
    Ff, gf = fresh.value_and_grad(Z, *theta, raise_on_fail=False, **kw)
    assert fresh.last_tier == 1 and 0.3 * 16384e-9 < fresh.last_estimate <= 16384e-9, (fresh.last_tier, fresh.last_estimate)
    for ratio, predicted, opened in [(1.0, 1.0, True), (1.0, 1e-3, True), (1e-9, 0.0, True), (1.0, 0.0, False)]:
        cb = _bound(X, y)
        cb.guard.open, cb.guard.ratio, cb.guard.predicted = opened, ratio, predicted
        F, g = cb.value_and_grad(Z, *theta, raise_on_fail=False, **kw)
        assert cb.last_tier == 1 and F == Ff and g["ls"].tolist() == gf["ls"].tolist(), (ratio, predicted, opened, cb.last_tier)
        Fv, _ = cb.value(Z, *theta, raise_on_fail=False, strict=True)
        assert cb.last_tier == 1 and Fv == fresh.value(Z, *theta, raise_on_fail=False, strict=True)[0]
    # a walk that visits the boundary theta in the middle of an arbitrary history: same tier, same bits
    rng = np.random.default_rng(5)
    for trial in range(4):
        hist = _bound(X, y)
        _walk(hist, Z, rng.integers(0, len(THETAS), size=6).tolist(), bool(trial & 1), **(kw if trial & 1 else {"strict": bool(trial & 2)}))
        F, g = hist.value_and_grad(Z, *theta, raise_on_fail=False, **kw)
        assert hist.last_tier == 1 and F == Ff and g["ls"].tolist() == gf["ls"].tolist(), trial


def test_a_device_time_out_switches_to_the_shared_device_mode_and_repeats_once(no_small_whitened):
    """Round 6: SGP_INFO_TIMEOUT from the single-launch Cholesky (two processes on one GPU starving each other's
    spinning workgroups) used to be raised at once.  The first one now sets SGP_OPT_SHARED_DEVICE on the engine's context -- the ticketed
    claim, whose progress does not need the whole launch resident -- and the evaluation is repeated; a second one is raised."""
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    from ggp_amd.core import SgpTimeoutError

    class Starved(FactoredOracleEngine):
        OPTIONS = {"shared_device": 6}

        def __init__(self, heals):
            super().__init__()
            self.opt, self.heals = {"shared_device": 0.0}, heals

        def set_option(self, name, value):
            prev, self.opt[name] = self.opt[name], float(value)
            return prev

        def get_option(self, name):
            return self.opt[name]

        def bound(self, *a, **k):
            res = super().bound(*a, **k)
            if self.opt["shared_device"] == 0.0 or not self.heals:
                res["info"][0] = -7777     # the status word a starved launch leaves
            return res

    X, y, Z = _problem()
    ref = _bound(X, y).value(Z, [0.8] * 3, 1.0, 0.3)[0]
    eng = Starved(heals=True)
    cb = _bound(X, y, eng)
    assert cb.value(Z, [0.8] * 3, 1.0, 0.3)[0] == ref and cb.n_timeout_retries == 1 and eng.get_option("shared_device") == 1.0
    F, g = cb.value_and_grad(Z, [0.8] * 3, 1.0, 0.3)
    assert F == ref and cb.n_timeout_retries == 1          # (no further retries: the mode stays)
    bad = _bound(X, y, Starved(heals=False))
    with pytest.raises(SgpTimeoutError):
        bad.value(Z, [0.8] * 3, 1.0, 0.3)
    assert bad.n_timeout_retries == 1


def test_extended_gradient_is_trusted_by_its_own_trailing_word_correction(no_small_whitened):
    """Round 6: with the double-double Phibar and its trailing word applied in pass 2 (sgp_phibar_dd, sgp_suffstats_bwd_lo)
    a value + gradient evaluation takes the extended order up to `extended_grad_range_lo` x the tolerance -- and is ACCEPTED there only while
    the correction the trailing word made is small against the gradient (`extended_lo_max_correction`); otherwise it is repeated in the
    whitened order, and the next gradient evaluations at such estimates start there."""
    from fake_engine import FactoredOracleEngine
    X, y, Z = _problem()
    theta = ([3.0] * 3, 1.0, 5e-3)        # estimate between 3 x (the old gradient range) and 1000 x the tolerance
    ref = _bound(X, y, form="whitened")
    Fw, gw = ref.value_and_grad(Z, *theta)
    eng = FactoredOracleEngine()
    cb = _bound(X, y, eng)
    F, g = cb.value_and_grad(Z, *theta)
    assert 3.0 * cb.streaming_tol < cb.last_estimate <= cb.extended_grad_range_lo * cb.streaming_tol, cb.last_estimate
    assert cb.last_tier == 1 and eng.calls["phibar_dd"] == 1 and eng.calls["suffstats_bwd_lo"] == 1 and eng.calls["suffstats_bwd_factored"] == 0
    assert eng.calls["f16_handed_over"] == 1   # the fp16 image the assembly left is what the trailing-word product multiplies (checked current there)
    assert cb.last_lo_correction is not None and cb.last_lo_correction <= cb.extended_lo_max_correction and cb.n_lo_rejections == 0
    assert abs(F - Fw) / X.shape[0] < 2e-9 and np.max(np.abs(g["ls"].numpy() - gw["ls"].numpy())) <= 1e-6 * max(1.0, float(gw["ls"].abs().max()))
    # without the trailing word (another kernel, d > 8, no K'_fu kept) the old range holds: the same theta goes to the whitened order
    e2 = FactoredOracleEngine()
    c2 = _bound(X, y, e2)
    c2.extended_lo = False
    c2.value_and_grad(Z, *theta)
    assert c2.last_tier == 2 and e2.calls["suffstats_bwd_factored"] == 1
    # a correction that is NOT small: the evaluation is repeated in the whitened order, the whitened gradient is what comes back ...
    e3 = FactoredOracleEngine()
    e3.lo_delta_scale = 1e12
    c3 = _bound(X, y, e3)
    F3, g3 = c3.value_and_grad(Z, *theta)
    assert c3.last_tier == 2 and c3.n_lo_rejections == 1 and e3.calls["suffstats_bwd_factored"] == 1 and g3["ls"].tolist() == gw["ls"].tolist()
    # ... and the next EIGHT gradient evaluations do not try the extended order again (values still take it: their reach is their own);
    # the ninth does, is rejected again, and the pause doubles
    n_ext = c3.n_extended
    for _ in range(8):
        c3.value_and_grad(Z, *theta)
        assert c3.n_extended == n_ext and c3.last_tier == 2 and c3.n_lo_rejections == 1
    c3.value(Z, *theta)
    assert c3.last_tier == 1
    n_ext = c3.n_extended
    c3.value_and_grad(Z, *theta)
    assert c3.n_extended == n_ext + 1 and c3.last_tier == 2 and c3.n_lo_rejections == 2 and c3._lo_skip == 16
    e3.lo_delta_scale = 1.0                      # the correction is small again: after the pause the extended order is back, the pause forgotten
    c3._lo_skip = 0
    c3.value_and_grad(Z, *theta)
    assert c3.last_tier == 1 and c3._lo_pause == 0
    # a correction reported as NaN (the product's fp16 inputs could not hold this theta: nothing was added) is a rejection, not a pass
    e5 = FactoredOracleEngine()
    e5.lo_delta_scale = float("nan")
    c5 = _bound(X, y, e5)
    F5, g5 = c5.value_and_grad(Z, *theta)
    assert c5.last_tier == 2 and c5.n_lo_rejections == 1 and c5.last_lo_correction == float("inf") and g5["ls"].tolist() == gw["ls"].tolist()
    # the sampler mode takes the extended order's gradient as far as its value holds, whatever the correction says
    e3.lo_delta_scale = 1e12
    F4, g4 = c3.value_and_grad(Z, *theta, grad_reach=16384.0, strict=True)
    assert c3.last_tier == 1 and c3.n_lo_rejections == 2


def test_trailing_word_path_is_taken_for_more_than_eight_dimensions(no_small_whitened):
    """The trailing-word product runs eight dimensions per pass over its accumulators (round 6): d > 8 is no longer a reason to fall back to
    the leading word's 3 x gradient range -- the same rule, the same check, the fp16 image handed over."""
    from fake_engine import FactoredOracleEngine
    X, y, Z = _problem(d=11)
    eng = FactoredOracleEngine()
    cb = _bound(X, y, eng)
    assert cb._bwd_lo_ok(Z.shape[0])
    # a theta whose estimate lies between the old gradient range (3 x) and the new one (1000 x the tolerance)
    for lsv, s2 in ((6.0, 5e-3), (8.0, 5e-3), (10.0, 2e-3), (14.0, 1e-3), (20.0, 1e-3)):
        F, g = cb.value_and_grad(Z, [lsv] * 11, 1.0, s2)
        if 3.0 * cb.streaming_tol < cb.last_estimate <= cb.extended_grad_range_lo * cb.streaming_tol and cb.last_tier == 1:
            break
    else:
        pytest.skip("no theta of the sweep landed between the two ranges")
    assert eng.calls["suffstats_bwd_lo"] >= 1 and eng.calls["f16_handed_over"] >= 1 and cb.n_lo_rejections == 0
    ref = _bound(X, y, form="whitened")
    Fw, gw = ref.value_and_grad(Z, [lsv] * 11, 1.0, s2)
    assert abs(F - Fw) / X.shape[0] < 2e-9 and np.max(np.abs(g["ls"].numpy() - gw["ls"].numpy())) <= 1e-6 * max(1.0, float(gw["ls"].abs().max()))


def test_hmc_target_sampler_mode_uses_the_extended_order_for_gradients(no_small_whitened):
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    from oracle import vfe_oracle as O
    X, y, Z = _problem()
    with pytest.raises(ValueError):
        pkg.HmcTarget(_bound(X, y), Z, gradient="fast")
    # theta with an estimate inside the value reach but beyond the gradient range of the parity mode (1000 x the tolerance since round 6:
    # the double-double Phibar with its trailing word in pass 2)
    th = [math.log(5.0)] * 3 + [0.0, math.log(0.015)]
    ep, es = FactoredOracleEngine(), FactoredOracleEngine()
    parity = pkg.HmcTarget(_bound(X, y, ep), Z)
    sampler = pkg.HmcTarget(_bound(X, y, es), Z, gradient="sampler")
    lp_p, g_p = parity.logp_and_grad(th)
    lp_s, g_s = sampler.logp_and_grad(th)
    est = sampler.bound.last_estimate
    assert 1000.0 * 1e-9 < est <= 16384.0 * 1e-9, est
    assert ep.calls["suffstats_whitened_rows"] == 1 and ep.calls["suffstats_bwd_factored"] == 1    # parity: the whitened order
    assert es.calls["suffstats_whitened_rows"] == 0 and es.calls.get("suffstats_extended", 0) == 1   # sampler: the extended order
    lp_ref, g_ref = O.hmc_logp(torch.tensor(th, dtype=torch.float64), X, y, Z, 1e-6, with_grad=True)
    assert abs(lp_s - float(lp_ref)) / X.shape[0] < 1e-8 and abs(lp_p - float(lp_ref)) / X.shape[0] < 1e-8   # the ENERGY is exact in both
    assert np.max(np.abs(np.array(g_s) - np.asarray(g_ref))) < 1e-5 * max(1.0, float(np.max(np.abs(np.asarray(g_ref)))))


# ---------------------------------------------------------------------------------------------
# two ranks
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _early_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    X, y, Z = _problem(N=700)
    pkg.CollapsedBound.WHITENED_MAX_WORK = 0
    M = Z.shape[0]
    lo, hi = (0, 500) if rank == 0 else (500, 700)         # uneven shards: 500 and 200 rows
    cb = pkg.CollapsedBound(X[lo:hi], y[lo:hi], jitter=1e-6, engine=FactoredOracleEngine())
    cb.whitened_rows_min_work = 0
    cb.early_check_min_work = 300 * M                       # between the two shards: rank 0 alone would read the status first
    out = []
    for ls, s2 in ((0.8, 0.3), (3.0, 2e-2), (25.0, 1e-5), (3.0, 1e-2), (0.8, 0.3)):
        F, g = cb.value_and_grad(Z, [ls] * 3, 1.0, s2, want_gz=False, raise_on_fail=False)
        out.append((F, g["ls"].tolist(), cb.n_guard_reruns, cb.n_direct_whitened, cb.n_extended, cb.n_collectives, cb.last_tier))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_uneven_shards_read_the_status_by_the_same_rule():
    """`early` came from the LOCAL shard; a guard repeat before pass 2 on one rank met pass 2's all-reduce on the
    other.  Both ranks now decide from the job's largest shard: same collectives, same bits, no hang."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_early_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[0][1] == outs[1][1]
    assert outs[0][1][2][2] >= 1                            # the far theta was repeated in a higher tier -- on both ranks alike


def _walk_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    X, y, Z = _problem()
    pkg.CollapsedBound.WHITENED_MAX_WORK = 0
    lo, hi = pkg.shard_rows(X.shape[0], rank, world)
    rng = np.random.default_rng(3)
    order = rng.integers(0, len(THETAS), size=10).tolist()
    cb = pkg.CollapsedBound(X[lo:hi], y[lo:hi], jitter=1e-6, engine=FactoredOracleEngine())
    cb.whitened_rows_min_work = 0
    a = _walk(cb, Z, order, True)
    b = _walk(cb, Z, order, True, strict=True, grad_reach=16384.0)
    q.put((rank, a, b, cb.n_collectives))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_walks_are_bit_identical_across_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_walk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[0][1:] == outs[1][1:]
