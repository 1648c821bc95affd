"""CollapsedBound: the hot path as one object -- F(theta, Z), its gradient and the predictive.

Reference call sites this replaces (paths relative to the reference repo):
  * ``-mll(model(train_x), train_y)`` on ``SparseGPR``            models/sgpr.py:123-125
  * ``pm.gp.MarginalSparse(approx="VFE").marginal_likelihood``   models/bayesian_sgpr_hmc.py:66,71
  * ``likelihood(model(test_x))``                                 models/sgpr.py:150-160

Data layout: X (N_local x d) and y (N_local) are resident on this rank's GPU for the whole run
(row-sharded contiguously across ranks, never moved); Z (M x d) and the hyper-parameters are
replicated.  One evaluation =
    pass 1 on the local rows  ->  ONE all-reduce of the packed [Phi | b | yy | kappa] buffer
    (RCCL over xGMI when world_size > 1)  ->  O(M^3) tail replicated on every rank (deterministic,
    so no broadcast)  [-> pass 2 on the local rows -> one small all-reduce of the gradients].
"""
from __future__ import annotations

import math

import torch

try:  # torch.distributed is plumbing; single-process use never touches it
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None

from ._lib import COMP_LEN, OUT_F, OUT_LOGMARG, OUT_S2BAR, OUT_TRACE


class NotPositiveDefiniteError(RuntimeError):
    """Raised (Adam path) when a Cholesky pivot is non-positive; ``info`` is the LAPACK-style index."""

    def __init__(self, info):
        super().__init__("matrix not positive definite: leading minor of order %d (1..M: Kuu, M+1..2M: B)" % info)
        self.info = info


class SgpTimeoutError(RuntimeError):
    """The single-launch Cholesky gave up waiting for a tile (status word SGP_INFO_TIMEOUT): a device scheduling
    problem, not a property of the matrix.  Always raised -- never turned into logp = -inf, which would bias a chain."""

    def __init__(self):
        super().__init__("device time-out inside the dataflow Cholesky (SGP_INFO_TIMEOUT): outputs are invalid")
        self.info = SGP_INFO_TIMEOUT


SGP_INFO_TIMEOUT = -7777


def few_host_threads(fn):
    """Decorator for the host-driven loops (training, sampling): the host side of a step is a handful of torch operations
    on tensors with d + 2 ... M entries (raw-parameter transforms, a 4 x 4 Cholesky of q(log theta), optimizer updates).
    With torch's default intra-op pool -- one thread per host core, 128-256 on the GPU boxes -- every such operation pays the
    pool's wake-up, and the spinning workers starve the HIP runtime's own threads: a BayesianSVGP minibatch step took 37.6 ms
    with 128 threads and 4.4 ms with one (`tools/bsvgp_rates.py`).  The cap (4) is lifted again when the loop returns."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        n = torch.get_num_threads()
        if n <= 4:
            return fn(*args, **kwargs)
        torch.set_num_threads(4)
        try:
            return fn(*args, **kwargs)
        finally:
            torch.set_num_threads(n)

    return wrapped


# ---------------------------------------------------------------------------------------------
# The streaming-order guard, host side.  Three evaluation orders ("tiers") compute the same bound:
#   0 streaming  Phi = K_uf K_fu streamed, W = L^-1 Phi L^-T in the tail           (the N >> M design; fastest)
#   1 extended   the same from an exact Phi and a double-double sandwich            (values hold 2^14 x further; its explicit-Phibar
#                                                                                    gradients 3 x further)
#   2 whitened   PyMC3's op order, A = L^-1 K_uf first                               (always accurate; two more N M^2 products)
# Every evaluation reports the library's first-order estimate of the streaming order's error at ITS theta (exact in tiers 0 and 1,
# which form Phi; tier 2 can only report the upper bound).  `required_tier` -- a pure function -- says which tier that estimate asks
# for; an evaluation that ran in a lower tier is repeated in the required one.  What is remembered between evaluations (GuardState)
# only decides where the next one STARTS, so that a sampler sitting in a guarded region does not pay a wasted streaming attempt per
# leapfrog; it never decides what is accepted.  Consequence, documented in include/sgp.h / INTEGRATION.md: the bits of F(theta) depend
# on the tier (tiers agree to ~1e-9 per datum), and the tier an evaluation is ACCEPTED in can be higher than required when the
# previous evaluations suggested it (never lower).  All inputs of these decisions are replicated numbers: ranks decide alike.
# ---------------------------------------------------------------------------------------------
TIER_STREAMING, TIER_EXTENDED, TIER_WHITENED = 0, 1, 2


def required_tier(estimate, tol, reach, extended_ok):
    """The cheapest tier that meets `tol` at a theta whose streaming-order error estimate is `estimate`.  `reach`: how many times the
    tolerance the extended order covers for this kind of evaluation (values / gradients); `extended_ok`: that order exists here."""
    if estimate <= tol:
        return TIER_STREAMING
    if extended_ok and estimate <= reach * tol:  # (NaN / inf fail both comparisons: the whitened order)
        return TIER_EXTENDED
    return TIER_WHITENED


class GuardState:
    """All the guard remembers between evaluations."""

    __slots__ = ("open", "predicted", "ratio")

    def __init__(self):
        self.open = False        # an episode is open: evaluations start in the tier the prediction asks for, without a streaming attempt
        self.predicted = 0.0     # the streaming-order estimate expected at the next theta
        self.ratio = 1.0         # estimate / upper bound at the last evaluation that knew both (predicts the estimate from a whitened one)

    def start_tier(self, tol, reach, extended_ok):
        """Where the next evaluation starts.  Inside an episode never in the streaming order (hysteresis: the episode ends once the
        prediction is below half the tolerance), so a theta hovering around the tolerance does not pay a repeat every other time."""
        if not self.open:
            return TIER_STREAMING
        return TIER_EXTENDED if (extended_ok and self.predicted <= reach * tol) else TIER_WHITENED

    def note_exact(self, estimate, bound, tol):
        """After an evaluation that formed Phi (tiers 0, 1): `estimate` is exact at its theta."""
        if math.isfinite(estimate) and math.isfinite(bound) and bound > 0.0:
            self.ratio = min(1.0, max(estimate / bound, 0.0))
        self.predicted = estimate
        self.open = not (estimate < 0.5 * tol) if self.open else not (estimate <= tol)

    def note_bound(self, bound, tol):
        """After a whitened-order evaluation: only the upper bound is known; the last ratio turns it into a prediction."""
        if math.isfinite(bound) and bound > 0.0:
            self.predicted = self.ratio * bound
            self.open = not (self.predicted < 0.5 * tol)


def _world(group):
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(group)


class CollapsedBound:
    """Collapsed (Titsias/VFE) sparse-GP bound over a resident row shard.

    Parameters
    ----------
    X, y : local shard, float64, on the engine's device.
    engine : object with the ``HipEngine`` interface; defaults to ``HipEngine()`` (the only product engine).
    jitter : added to diag(Kuu): 0.0 = GPyTorch parity, 1e-6 = PyMC3 ``stabilize`` parity.
    group : torch.distributed process group (None = default group when initialised).
    form : "streaming" -- Phi = K_uf K_fu streamed over the rows, W = L^-1 Phi L^-T in the tail (the N >> M design);
           "whitened"  -- PyMC3's op order, A = L^-1 K_uf materialised, W = A A^T: B = I + W/s2 is positive definite by
                          construction and F keeps 1e-10 accuracy at cond(Kuu) ~ 1e8, for twice the N M^2 work;
           "auto"      -- whitened up to ``WHITENED_MAX_WORK`` row x inducing pairs per rank (the reference's own
                          workloads: N ~ 250-1300, M ~ 100-480), streaming above.
    """

    WHITENED_MAX_WORK = 1 << 20  # local rows x inducing points up to which form="auto" means "whitened"

    def __init__(self, X, y, kernel="rbf", jitter=0.0, engine=None, group=None, form="auto"):
        if engine is None:
            from .engine import HipEngine
            engine = HipEngine(X.device if X.is_cuda else None)
        self.engine = engine
        if X.dim() == 1:
            X = X[:, None]
        self.X = X.to(dtype=torch.float64, device=engine.device).contiguous()
        self.y = y.to(dtype=torch.float64, device=engine.device).reshape(-1).contiguous()
        if self.X.shape[0] != self.y.shape[0]:
            raise ValueError("X has %d rows, y has %d" % (self.X.shape[0], self.y.shape[0]))
        self.d = int(self.X.shape[1])
        self.kernel = kernel
        self.jitter = float(jitter)
        if form not in ("auto", "streaming", "whitened", "extended"):
            raise ValueError("form must be 'auto', 'streaming', 'extended' or 'whitened'")
        self.form = form
        self.group = group
        self.world = _world(group)
        n_local = int(self.X.shape[0])
        if self.world > 1:
            t = torch.tensor([n_local], dtype=torch.int64, device=engine.device)
            dist.all_reduce(t, group=group)
            self.N = int(t.item())
            # form="auto" must come out the same on every rank (the two orders all-reduce different matrices into the
            # same buffer): decide from the LARGEST shard, not from the local one
            dist.all_reduce(t.fill_(n_local), op=dist.ReduceOp.MAX, group=group)
            self._rows_for_form = int(t.item())
        else:
            self.N = n_local
            self._rows_for_form = n_local
        self.n_evals = 0
        self.n_grads = 0
        self.n_collectives = 0
        # K'_fu of the local shard stays resident between pass 1 and pass 2 of one evaluation when it
        # fits the budget (default 64 GiB of the 288 GB HBM); otherwise the library streams it in
        # 16 GiB super-chunks and pass 2 re-assembles.
        self._kfu = None
        self._kfu_f16 = None
        self.kfu_budget_bytes = 64 << 30
        # local rows x inducing points from which the whitened order runs in the streaming layout (engine.suffstats_whitened_rows)
        self.whitened_rows_min_work = 1 << 26
        self.factored_adjoint = True  # whitened order: pass 2 from L^-T Cw L^-1 applied factor by factor (sgp_suffstats_bwd_factored)
        self.fused = True          # single-launch path for small problems (M <= 128, one rank): sgp_small_eval
        self._small = None         # (pinned host theta, device theta, result buffer) of the single-launch path
        self.overlap_tail = True   # factor Kuu on a second HIP stream while pass 1 runs
        self.overlap_min_work = 1 << 21  # local rows x inducing points below which everything stays on one stream
        self.early_check_min_work = 1 << 28  # ... from which value_and_grad reads the status before enqueueing pass 2
        self._side = None          # (side stream, (z_ready, ready) events, {M: (Kuu, L^-1) buffers}) of the two-stream path
        # Guard of the streaming order (form="auto" only; see the block comment above required_tier): the tolerance is per datum on
        # |dF| / N.  extended_level 2 = 39 digit pairs, what the sandwich amplifies is 2^16 smaller (reach 2^14: a factor 4 kept as margin;
        # measured at C5 over 27 theta: <= 1.6e-10 per datum against the whitened order up to estimates of 3e-5,
        # profiles/r04_extended_order_c5.jsonl); level 1 = 34 pairs, 2^8 (0.9 ms cheaper; reach 2^7 then).
        # GRADIENTS: pass 2 of the extended order takes the explicit Phibar, whose product with K_uf cancels -- against the factored pass 2
        # of the whitened order its lengthscale gradients are off by 7e-7 at an estimate of 4e-9, 3e-5 at 8e-8, 3e-2 at 2e-7 (same file):
        # a value + gradient evaluation that must hold 1e-6 takes the extended order within `extended_grad_range` x the tolerance only.
        # A sampler may ask for more (HmcTarget(gradient="sampler"): the value's reach for the gradient as well, see there).
        self.streaming_tol = 1e-9
        self.extended_level = 2
        self.extended_range = 16384.0
        self.extended_grad_range = 3.0
        self.extended_dd_phibar = True   # the extended order's explicit Phibar formed in double-double (sgp_phibar_dd) ...
        self.extended_lo = True            # ... with its trailing word applied in pass 2 where that exists (RBF, K'_fu kept)
        # ... and its reach with the trailing word applied in pass 2 (sgp_suffstats_bwd_lo).  Calibrated at C5 over 34 theta against the factored
        # pass 2 of the whitened order (profiles/r06_extended_order_gradients_dd_phibar.jsonl, ..._ard_...): with both words every cell
        # whose estimate is <= 1e-6 AND whose correction is <= 1e-4 of the gradient is within 6e-7 (the trained ARD thetas of C5: 3e-9 and
        # 2.6e-8 where the leading word alone is off by 1.3e-6 / 4.5e-5 and the fp64-formed matrix by 4.6e-6 / 4.4e-4); isotropic l = 5 is
        # the counter-example that makes the second condition necessary (estimate 1.8e-7, correction 5.5e-4, 8e-6 left).
        self.extended_grad_range_lo = 1000.0
        self.extended_lo_max_correction = 1e-4
        self._lo_skip, self._lo_pause = 0, 0     # gradient evaluations still to start in the whitened order behind a rejection / the current pause (memory only)
        self.last_lo_correction = None
        self.n_lo_rejections = 0
        self.guard = GuardState()
        self.last_estimate = None   # the estimate (exact, or the bound of a whitened evaluation) of the last guarded evaluation
        self.last_tier = None       # the tier the last evaluation was accepted in
        self.n_guard_reruns = 0     # evaluations repeated in a higher tier
        self.n_direct_whitened = 0  # evaluations that started above the streaming order on the strength of a prediction
        self.n_extended = 0         # evaluations that ran in the extended order
        self.n_timeout_retries = 0  # evaluations repeated behind a device time-out (the engine switched to the shared-device mode)
        self._phi_diag = None

    # ------------------------------------------------------------------ internals
    def _allreduce(self, buf):
        if self.world > 1:
            dist.all_reduce(buf, group=self.group)
            self.n_collectives += 1
        return buf

    def _allreduce_stats(self, stats, M):
        """THE exchange of an evaluation: Phi is symmetric, so only [lower triangle | b | yy | kappa] travels
        (4.2 MB instead of 8.4 MB at M = 1024); unpacking mirrors.  ``n_collectives`` counts what was issued."""
        if self.world > 1:
            e = self.engine
            if hasattr(e, "pack_lower"):
                tri = e.pack_lower(stats, M)
                dist.all_reduce(tri, group=self.group)
                e.unpack_lower(tri, M, stats)
            else:
                dist.all_reduce(stats, group=self.group)
            self.n_collectives += 1
        return stats

    def _fetch(self, res, upto=None):
        """The one host round trip of an evaluation: a single device-to-host copy of the result buffer ([out | status
        word | whatever the caller appended], see ``HipEngine.result_buffer``) -- no cast / concatenate launches."""
        buf = res["buf"] if upto is None else res["buf"][:upto]
        host = buf.detach().to("cpu")
        o, info = self.engine.read_result(host)
        if info < 0:  # not a pivot index: the evaluation itself failed (time-out), with raise_on_fail or without
            raise SgpTimeoutError()
        return o, info, host

    def _prep_Z(self, Z):
        if Z.dim() == 1:
            Z = Z[:, None]
        return Z.detach().to(dtype=torch.float64, device=self.engine.device).contiguous()

    def _kfu_for(self, M):
        e = self.engine
        if not hasattr(e, "kfu_buffer"):
            return None
        n_local = int(self.X.shape[0])
        need = ((max(n_local, 1) + 255) // 256 * 256) * ((M + 127) // 128 * 128)
        if need * 8 > self.kfu_budget_bytes:
            return None
        if self._kfu is None or self._kfu.numel() < need:
            self._kfu = e.kfu_buffer(n_local, M)
        return self._kfu

    def _kfu_f16_for(self, M):
        """The fp16 image of K'_fu beside the fp64 block (the extended order's trailing-word product reads it; written by the assembly
        kernel, which spares sgp_suffstats_bwd_lo a conversion pass over the fp64 block: 2.0 of 4.6 ms at C5)."""
        e = self.engine
        if not hasattr(e, "kfu_f16_buffer"):
            return None
        n_local = int(self.X.shape[0])
        need = ((max(n_local, 1) + 255) // 256 * 256) * ((M + 127) // 128 * 128)
        if need * 10 > self.kfu_budget_bytes:
            return None
        if self._kfu_f16 is None or self._kfu_f16.numel() < need:
            self._kfu_f16 = e.kfu_f16_buffer(n_local, M)
        return self._kfu_f16

    # ------------------------------------------------------------------ single-launch path (small problems)
    def _small_ok(self, M, want_gz=False, sf2=1.0):
        """One cooperative launch instead of ~60: M <= 128, this rank holds all rows, and the caller did not insist on the
        streaming order (the launch evaluates the whitened / PyMC3 order).  Stationary kernels d <= 24; composite kernels
        d <= 8, without dF/dZ (that goes through the materialised path) and with the block's own amplitudes (sf2 = 1)."""
        e = self.engine
        if self.kernel == "composite" and (want_gz or float(sf2) != 1.0):
            return False
        return (self.fused and self.world == 1 and self.form != "streaming" and hasattr(e, "small_eval")
                and e.small_supported(int(self.X.shape[0]), int(M), self.d, self.kernel))

    def _small_eval(self, Z, theta_host, mode, want_grad, want_gz, composite=None):
        """theta_host: the launch's hyper-parameter vector (d + 2 floats; composite kernels: block + [s2], or the log free
        parameters + [log sigma]).  Returns (host out, info, gZ device tensor or None): one launch, one small host-to-device
        copy before it and one device-to-host copy after it; out = [value | gradients (len(theta_host)) | logmarg | trace]."""
        e = self.engine
        nt = len(theta_host)
        if self._small is None or self._small[0].numel() != nt:
            pin = e.device.type == "cuda"
            host = torch.empty(nt, dtype=torch.float64, pin_memory=pin)
            self._small = (host, torch.empty(nt, dtype=torch.float64, device=e.device), e.small_result(nt - 2)[0], host.numpy())
        host, dev_theta, buf, host_np = self._small
        host_np[:] = theta_host  # one vectorised store into the pinned buffer (element-wise tensor stores cost ~1.5 us each)
        dev_theta.copy_(host, non_blocking=True)
        kw = {"composite": composite} if composite is not None else {}
        out, gz, _ = e.small_eval(self.X, self.y, Z, dev_theta, self.jitter, self.kernel, mode=mode, want_grad=want_grad,
                                  want_gz=want_gz, out=buf, **kw)
        h = out.detach().to("cpu")
        info = int(h.numpy()[nt + 3:nt + 4].view("int32")[0])
        if info < 0:
            if hasattr(e, "small_reset"):
                e.small_reset()
            raise SgpTimeoutError()
        return h, info, gz

    def _whitened(self, M):
        if self.form == "auto":
            return hasattr(self.engine, "suffstats_whitened") and self._rows_for_form * int(M) <= self.WHITENED_MAX_WORK
        return self.form == "whitened"

    def _side_state(self, M):
        """(side stream, (z_ready, ready) events, (Kuu, L^-1) buffers for M inducing points) of the two-stream path.  One side stream per
        engine, shared by every bound built on it; events and buffers are this bound's."""
        e = self.engine
        if self._side is None:
            if getattr(e, "_side_stream", None) is None:
                e._side_stream = torch.cuda.Stream(device=e.device, priority=-1)  # ahead of queued pass-1 workgroups
            self._side = (e._side_stream, (torch.cuda.Event(), torch.cuda.Event()), {})
        bufs = self._side[2].get(M)
        if bufs is None:
            self._side[2].clear()  # (one M at a time: a bound whose inducing set changes size does not keep every size)
            bufs = self._side[2][M] = (e.empty(M, M), e.empty(e.lib.sgp_kuu_factor_len(M)))
        return self._side[0], self._side[1], bufs

    def _trace_buf(self):
        if getattr(self, "_trace", None) is None:  # tr(Kuu^-1) + its scratch: one buffer per bound, reused by every evaluation
            e = self.engine
            self._trace = e.empty(e.lib.sgp_kuu_inverse_trace_len() if hasattr(e, "lib") else 2)
        return self._trace

    # -- compatibility of the counters' old names (tests, tools)
    @property
    def _prefer_whitened(self):
        return self.guard.open

    @_prefer_whitened.setter
    def _prefer_whitened(self, v):
        self.guard.open = bool(v)

    def _guard_on(self, M):
        """The guard watches this bound: form="auto" above the size where "auto" means whitened anyway, a tolerance, an engine that reports."""
        return (self.form == "auto" and self.streaming_tol > 0.0 and hasattr(self.engine, "streaming_error_report")
                and not self._whitened(M))

    def _extended_ok(self, M):
        """The extended order exists for this bound (a rank-invariant statement: the shard size is the largest one of the job)."""
        e = self.engine
        return (hasattr(e, "suffstats_extended") and self.kernel != "composite" and self.extended_range > 1.0
                and self._rows_for_form * int(M) >= self.whitened_rows_min_work)

    def _bwd_lo_ok(self, M):
        """Pass 2 of the extended order can apply the trailing word of the double-double Phibar here (RBF, K'_fu kept): a
        rank-invariant statement, like `_extended_ok`."""
        e = self.engine
        return (self.extended_dd_phibar and self.extended_lo and hasattr(e, "suffstats_bwd_lo") and hasattr(e, "phibar_dd")
                and e.bwd_lo_supported(self._rows_for_form, int(M), self.d, self.kernel)
                and ((self._rows_for_form + 255) // 256 * 256) * ((int(M) + 127) // 128 * 128) * 8 <= self.kfu_budget_bytes)

    def _lo_correction_small(self, host, head, nh, lo_slots):
        """The a-posteriori check of the extended order's gradient: the trailing word's correction [d lengthscales | sf2] (tail of the
        buffer) against the gradient it went into, in the parity suite's metric -- lengthscales against their largest component,
        the amplitude against max(1, |g_sf2|)."""
        g = host[head:head + nh + 1].tolist()
        dl = host[host.numel() - lo_slots:].tolist()
        if any(v != v for v in dl):   # NaN: the product's inputs were beyond its fp16 format (an inducing point > 128 lengthscales out) -- no correction was added
            self.last_lo_correction = float("inf")
            return False
        self.last_lo_correction = max(max(abs(v) for v in dl[:nh]) / max(1.0, max(abs(v) for v in g[:nh])), abs(dl[nh]) / max(1.0, abs(g[nh])))
        return self.last_lo_correction <= self.extended_lo_max_correction

    def _fixed_tier(self, M):
        if self.form == "extended":
            return TIER_EXTENDED
        return TIER_WHITENED if self._whitened(M) else TIER_STREAMING

    def _review(self, res, host, info, tier, reach, M, strict):
        """What the evaluation's own numbers say: None = accept, or the tier to repeat it in.  Updates the guard's memory.
        A failed factorization of K_uu (info in 1 .. M), a time-out or a refusal says nothing about the estimate: the state is left alone
        and nothing is repeated.  A failed factorization of B (info > M) below the whitened order is the streaming order's own failure
        mode -- W = L^-1 Phi L^-T so far off that I + W / s2 is not positive definite (l = 20, sig_n = 0.01: profiles/r04_theta_sweep_streaming.jsonl)
        -- and the estimate, which needs K_uu's factor and Phi only, is valid: it is reviewed like any other."""
        if not res.get("reported") or (info != 0 and not (info > M and tier < TIER_WHITENED)):
            return None
        e = self.engine
        est, ub = e.read_estimate(host), e.read_bound(host)
        tol, ext_ok = self.streaming_tol, self._extended_ok(M)
        self.last_estimate = est
        if tier == TIER_WHITENED:
            # (only the upper bound is known here: nothing this order states can send the evaluation DOWN -- a strict evaluation never
            # starts in it, see _evaluate, so whatever is accepted here was sent up by a lower tier's exact estimate)
            self.guard.note_bound(ub, tol)
            return None
        need = required_tier(est, tol, reach, ext_ok)
        self.guard.note_exact(est, ub, tol)
        if need > tier:
            self.guard.open = True
            return need
        if info != 0:  # B failed although the estimate is inside this tier's reach: the next tier up, if there is one
            return tier + 1 if (tier + 1 < TIER_WHITENED and ext_ok) else TIER_WHITENED
        if strict and need < tier:
            return need
        return None

    def _forward(self, Z, ls, sf2, s2, with_adjoints, want_factors=False, extra=0, tier=TIER_STREAMING, report=False, want_lo=True):
        """One attempt in the given tier: pass 1, the exchange, the tail -- everything enqueued, nothing read back.
        report: also the streaming-order estimate / bound into the result buffer (sgp_streaming_error_report)."""
        e = self.engine
        M = int(Z.shape[0])
        result = e.result_buffer(extra)  # (buf, out, info): everything the host reads back, one allocation
        if tier == TIER_EXTENDED:
            # the extended streaming order: the whitened statistics from ONE integer-core contraction (39 digit pairs, double-double fold)
            # and a double-double triple product; pass 2 from the explicit Phibar on the fp64 K'_fu kept here
            self.n_extended += 1
            Kuu = e.kuu(Z, ls, sf2, self.jitter, self.kernel)
            linv, _ = e.kuu_factor(Kuu, info=result[2], trace_out=self._trace_buf() if report else None)
            kfu = self._kfu_for(M) if with_adjoints else None
            diag = None
            if report:
                if self._phi_diag is None or self._phi_diag.numel() < M:
                    self._phi_diag = e.empty(M)
                diag = self._phi_diag
            dd = with_adjoints and self.extended_dd_phibar and hasattr(e, "phibar_dd")
            lo = dd and want_lo and kfu is not None and self._bwd_lo_ok(M)   # (not for the sampler mode: it takes this order's gradient as it is)
            kfu_f16 = self._kfu_f16_for(M) if lo else None
            packed = e.suffstats_extended(self.X, self.y, Z, ls, sf2, linv, self.kernel, kfu=kfu, level=self.extended_level,
                                          **({"phi_diag": diag} if diag is not None else {}), **({"kfu_f16": kfu_f16} if kfu_f16 is not None else {}))
            self._allreduce_stats(packed, M)
            # pass 2 of this order takes the explicit Phibar = L^-T C L^-1 / (2 s2); formed by two fp64 products its own rounding
            # (eps |L^-T| |C| |L^-1| >> eps |Phibar|) is what limits the gradients -- formed in double-double from the whitened core C the
            # bound returns they are 5-15 x closer (tests/studies/explicit_phibar_pass2.py; +1.3 ms at M = 1024)
            res = e.bound(Kuu, packed, s2, self.N, with_adjoints=with_adjoints, want_factors=want_factors, kuu_linv=linv,
                          result=result, whitened=True, **({"want_cw": True} if dd else {}))
            if dd:
                # ... and 35-700 x closer with the trailing word applied as well: an fp16 product K' Phibar_lo beside the fp64 one (_pass2)
                res["Phibar"], res["Phibar_lo"] = e.phibar_dd(res["Cw"], linv, s2, want_lo=lo)
                res["Cw"] = None   # (pass 2 of this order is the explicit one: _pass2 takes the factored route when a core is handed on)
            if report:
                self._allreduce(diag[:M])  # (ranks hold the diagonal of their own shard's Phi)
                e.streaming_error_report(diag, 1, self._trace_buf(), sf2, s2, self.N, M, result)
                res["reported"] = True
            res.update(packed=packed, kfu=kfu, kfu_f16=kfu_f16, t_keep=None, linv=linv)
            return res
        if tier == TIER_WHITENED:
            # PyMC3 op order: chol(Kuu) first, then A = L^-1 K_uf, W = A A^T (one stream; these shards are small)
            Kuu = e.kuu(Z, ls, sf2, self.jitter, self.kernel)
            rep = report and self.kernel != "composite"
            linv, _ = e.kuu_factor(Kuu, info=result[2], trace_out=self._trace_buf() if rep else None)
            # with adjoints: also the whitened core Cw of Phibar, so that pass 2 applies L^-T Cw L^-1 factor by factor
            factored = with_adjoints and self.factored_adjoint and hasattr(e, "suffstats_bwd_factored")
            t_keep = None
            if (hasattr(e, "suffstats_whitened_rows") and self.kernel != "composite"
                    and int(self.X.shape[0]) * M >= self.whitened_rows_min_work):
                # a large shard (the streaming guard's repeats at 10^6 rows): the streaming layout; T = K'_fu L^-T stays for pass 2
                t_keep = self._kfu_for(M) if factored else None
                packed = e.suffstats_whitened_rows(self.X, self.y, Z, ls, sf2, linv, self.kernel, t_out=t_keep)
            else:
                packed = e.suffstats_whitened(self.X, self.y, Z, ls, sf2, linv, self.kernel)
            self._allreduce_stats(packed, M)
            kw = {"want_cw": True} if factored else {}
            res = e.bound(Kuu, packed, s2, self.N, with_adjoints=with_adjoints, want_factors=want_factors, kuu_linv=linv,
                          result=result, whitened=True, **kw)
            if rep:
                e.streaming_error_report(None, 1, self._trace_buf(), sf2, s2, self.N, M, result)
                res["reported"] = True
            res.update(packed=packed, kfu=None, t_keep=t_keep, linv=linv)
            return res
        kfu = self._kfu_for(M) if with_adjoints else None
        gate = None
        guard = report
        trace = self._trace_buf() if guard and hasattr(e, "kuu_factor") else None
        # a second stream only pays once pass 1 is long enough to hide the Kuu chain under it (C3-sized and up; at C1 / C2 sizes
        # the hand-over costs more than the 0.1 ms it could hide)
        overlap = (self.overlap_tail and hasattr(e, "kuu_factor") and e.device.type == "cuda"
                   and int(self.X.shape[0]) * M >= self.overlap_min_work)
        if overlap:
            # chol(Kuu), its inverse, the conditioning gate and tr(Kuu^-1) depend on (Z, theta) only: engine.kuu + engine.kuu_factor, SIX
            # launches since round 5 (the factorization's launch forms L^-1 itself), on a side stream beside pass 1.  The calling thread
            # enqueues them first (~25 us; through round 4 the chain was ~50 launches, replayed from a hipGraph by a helper thread whose
            # wake-up alone cost 70 us: profiles/r05_v2_c3_timeline.txt), into buffers the bound keeps (no allocation, no stream
            # bookkeeping per evaluation: the previous evaluation's readers are ahead of `z_ready` in the main stream's order).
            main = torch.cuda.current_stream(e.device)
            side, (z_ready, ready), (Kuu, linv) = self._side_state(M)
            z_ready.record(main)  # Z is materialised on the main stream
            side.wait_event(z_ready)
            # (the side stream's accesses to the result buffer and to Kuubar end before events the main stream waits for ahead of its
            # own last use: no record_stream -- the allocator's per-block event bookkeeping costs the host ~25 us per evaluation)
            e.kuu(Z, ls, sf2, self.jitter, self.kernel, out=Kuu, stream=side)
            e.kuu_factor(Kuu, info=result[2], trace_out=trace, out=linv, stream=side)  # the evaluation's status word starts as the Kuu status
            ready.record(side)
            # Big shards contract on the integer matrix cores, beside which nothing co-schedules: the contraction is gated on the
            # chain's end, so the chain runs beside kernel assembly (only where assembly outlasts the chain -- 2.0 ms vs 0.6 at
            # N = 1M, M = 1024, but 0.3 vs 0.6 at an eighth of it, where gating costs 0.1 ms: rows >= 300 M) ... and only where the
            # integer cores will actually contract (the library's own rule for this engine's context)
            if (hasattr(e, "would_use_i8") and self.kernel != "composite" and int(self.X.shape[0]) >= 300 * M
                    and e.would_use_i8(int(self.X.shape[0]), M)):
                gate = ready
        try:
            packed = e.suffstats(self.X, self.y, Z, ls, sf2, self.kernel, kfu=kfu, **({"gate": gate} if gate is not None else {}))
            self._allreduce_stats(packed, M)
        finally:
            # the side stream writes `result`, the trace buffer and (Kuu, L^-1) with no record_stream: the main stream must be behind
            # `ready` before any of them can go back to its allocator pool -- also when pass 1 or the exchange raises
            if overlap:
                main.wait_event(ready)
        if overlap:
            res = e.bound(Kuu, packed, s2, self.N, with_adjoints=with_adjoints, want_factors=want_factors, kuu_linv=linv,
                          result=result)
        elif guard and hasattr(e, "kuu_factor"):
            Kuu = e.kuu(Z, ls, sf2, self.jitter, self.kernel)
            linv, _ = e.kuu_factor(Kuu, info=result[2], trace_out=trace)
            res = e.bound(Kuu, packed, s2, self.N, with_adjoints=with_adjoints, want_factors=want_factors, kuu_linv=linv,
                          result=result)
        else:
            Kuu = e.kuu(Z, ls, sf2, self.jitter, self.kernel)
            res = e.bound(Kuu, packed, s2, self.N, with_adjoints=with_adjoints, want_factors=want_factors, result=result)
        if guard and trace is not None:
            e.streaming_error_report(packed, M + 1, trace, sf2, s2, self.N, M, result)
            res["reported"] = True
        res["packed"] = packed
        res["kfu"] = kfu
        return res

    def _pass2(self, res, Z, ls, sf2, s2, want_gz, g):
        """Pass 2 on the local rows + the all-reduce of the gradients + the K_uu path, into the packed gradient slice `g`."""
        e = self.engine
        # kappabar = dF/dkappa = -1 / (2 s2) needs nothing from the device (same value the tail writes to out)
        if res.get("Cw") is not None:
            # whitened order: Phibar's cond(K_uu)-sized entries would cancel in Phibar K_uf -- its factors are applied instead
            e.suffstats_bwd_factored(self.X, self.y, Z, ls, sf2, res["linv"], res["Cw"], s2, res["bbar"], -1.0 / (2.0 * float(s2)),
                                     self.kernel, want_gz=want_gz, out=g,
                                     **({"t_in": res["t_keep"]} if res.get("t_keep") is not None else {}))
        else:
            e.suffstats_bwd(self.X, self.y, Z, ls, sf2, res["Phibar"], res["bbar"], -1.0 / (2.0 * float(s2)),
                            self.kernel, want_gz=want_gz, out=g, kfu=res["kfu"])
            if res.get("Phibar_lo") is not None:   # the extended order: the trailing word of its double-double Phibar (lengthscales, amplitude)
                n_lo = int(res.get("lo_slots", 0))   # (the correction itself goes to the tail of `g`: all-reduced with it, read back with it)
                e.suffstats_bwd_lo(self.X, self.y, Z, ls, sf2, res["Phibar_lo"], res["kfu"], g, self.kernel,
                                   **({"delta": g[g.numel() - n_lo:]} if n_lo else {}),
                                   **({"kfu_f16": res["kfu_f16"]} if res.get("kfu_f16") is not None else {}))
        if int(res.get("lo_slots", 0)) and res.get("Phibar_lo") is None:
            g[g.numel() - int(res["lo_slots"]):].zero_()   # (the correction's slots travel with g: nothing undefined into the all-reduce)
        self._allreduce(g)
        # (measured and dropped, round 5: this launch pair on the side stream beside pass 2 -- it needs Kuubar only -- ends the evaluation no
        # earlier: pass 2 fills the chip and runs the 28 us longer that the side stream takes from it, profiles/r05_kuu_bwd_side_stream_c3_timeline.txt)
        e.kuu_bwd(Z, ls, sf2, res["Kuubar"], g, self.kernel, want_gz=want_gz)

    def _evaluate(self, *args, **kw):
        """`_evaluate_once`, and ONE more attempt behind a device time-out: the single-launch Cholesky deals its work items statically
        and needs all its workgroups resident (the fast way while this process has the GPU to itself); a second process on the same
        device -- joblib workers as in the reference's experiments/regression.py:219-231, ranks sharing a GPU -- can starve such a launch
        into SGP_INFO_TIMEOUT.  The first time-out switches this engine's context to SGP_OPT_SHARED_DEVICE (items claimed by ticket, by
        workgroups that are running: include/sgp.h) for good and the evaluation is repeated; a second time-out is raised."""
        try:
            return self._evaluate_once(*args, **kw)
        except SgpTimeoutError:
            e = self.engine
            if "shared_device" not in getattr(e, "OPTIONS", {}) or e.get_option("shared_device") != 0:
                raise
            e.set_option("shared_device", 1)
            self.n_timeout_retries += 1
            return self._evaluate_once(*args, **kw)

    def _evaluate_once(self, Z, ls, sf2, s2, with_grad=False, want_gz=False, want_factors=False, grad_reach=None, strict=False):
        """ONE evaluation through the guard -- the driver value / value_and_grad / factors share.  Runs attempts (every repeat goes to a
        strictly higher tier; strict: once, first, to the LOWER tier the evaluation's own estimate names) until `_review` accepts.
        Returns (res, out (host), info, host buffer, head) of the accepted attempt; with_grad: the packed gradient sits behind `head`."""
        e = self.engine
        M, d = int(Z.shape[0]), int(Z.shape[1])
        guard = self._guard_on(M)
        reach = self.extended_range
        if with_grad:   # (how far the extended order's gradient holds 1e-6: further with both words of its double-double Phibar)
            reach = float(grad_reach) if grad_reach is not None else (self.extended_grad_range_lo if self._bwd_lo_ok(M) else self.extended_grad_range)
        ext_ok = self._extended_ok(M)
        if guard:
            tier = self.guard.start_tier(self.streaming_tol, reach, ext_ok)
            if strict and tier == TIER_WHITENED:
                # strict: the ACCEPTED tier is the one the evaluation's own exact estimate names.  The whitened order states only an upper
                # bound -- whether it would be left for a lower tier was a matter of the guard's memory (the same theta accepted
                # in tier 1 on a fresh bound and in tier 2 after some history).  So a strict evaluation starts at most in the highest tier
                # that states its exact estimate, and reaches the whitened order only when that estimate sends it there.
                tier = TIER_EXTENDED if ext_ok else TIER_STREAMING
            if tier != TIER_STREAMING:
                self.n_direct_whitened += 1
        else:
            tier = self._fixed_tier(M)
        nh = (e.hyper_len(self.kernel, d) if hasattr(e, "hyper_len") else d) if with_grad else 0
        # (behind the packed gradient: d + 1 slots for the trailing word's correction, which decides whether the explicit pass 2 is trusted)
        lo_slots = (nh + 1) if (with_grad and guard and ext_ok and self._bwd_lo_ok(M)) else 0
        extra = nh + 1 + (M * d if want_gz else 0) + lo_slots if with_grad else 0
        if guard and with_grad and tier == TIER_EXTENDED and grad_reach is None and self._lo_skip > 0:
            # the trailing word's correction was too large a moment ago: the next few gradient evaluations go straight to the whitened order (no
            # wasted attempt per leapfrog); the pause doubles with every rejection in a row (8, 16, 32) and ends with the first acceptance
            self._lo_skip -= 1
            tier = TIER_WHITENED
        # Small shards: pass 2 is enqueued straight behind the tail and ONE copy at the very end brings back F, the status and the
        # gradient -- a failed factorization or a guard repeat then costs a wasted pass 2, which is cheaper than idling the GPU for
        # a host round trip on every leapfrog.  Big shards check first.  The rule is the job's (largest shard), not this rank's:
        # ranks must issue the same collectives in the same order.
        early = with_grad and self._rows_for_form * M >= self.early_check_min_work
        # Repeats: UP whenever the attempt's own estimate asks for a higher tier; DOWN (strict only) at most once, and never again after
        # it -- so the walk ends after at most four attempts (1 -> 0 -> 1 -> 2) and what is accepted is never an attempt whose own estimate
        # asked for more (`nxt in tried` used to fall through to the LOWER attempt).  Attempts that are complete (everything
        # already on the host) are kept, so coming back to a tier costs nothing; an incomplete one (big shards: pass 2 not yet run, and
        # its K'_fu buffer since overwritten) is run again.
        done = {}
        may_go_down = strict
        while True:
            res = self._forward(Z, ls, sf2, s2, with_adjoints=with_grad, want_factors=want_factors, extra=extra, tier=tier, report=guard,
                                want_lo=grad_reach is None)
            head = res["buf"].numel() - extra  # [out | status word | estimate | bound | pad], then the packed gradient (16-byte aligned)
            g = res["buf"][head:] if with_grad else None
            res["lo_slots"] = lo_slots
            grad_upto = None if lo_slots else head + nh + 1   # (with the correction's slots: the whole buffer -- they sit at its end)
            if with_grad and not early:
                self._pass2(res, Z, ls, sf2, s2, want_gz, g)
            o, info, host = self._fetch(res, upto=grad_upto if (with_grad and not early) else head)
            nxt = self._review(res, host, info, tier, reach, M, may_go_down) if guard else None
            if nxt is not None:
                if not (with_grad and early) and not want_factors:
                    done[tier] = (res, o, info, host, head)
                if nxt < tier:
                    may_go_down = False
                self.n_guard_reruns += 1
                tier = nxt
                if tier in done:
                    res, o, info, host, head = done[tier]
                    self.last_tier = tier
                    return res, o, info, host, head
                continue
            if with_grad and early and info == 0:
                self._pass2(res, Z, ls, sf2, s2, want_gz, g)
                _, _, host = self._fetch(res, upto=grad_upto)
            if (with_grad and info == 0 and tier == TIER_EXTENDED and res.get("Phibar_lo") is not None and lo_slots and grad_reach is None
                    and not self._lo_correction_small(host, head, nh, lo_slots)):
                # The explicit pass 2 of this order is trusted only while its trailing-word correction is small against the gradient (what is
                # left behind the correction is a few per cent of it: profiles/r06_extended_order_gradients_*): repeated in the whitened order
                self.n_guard_reruns += 1
                self.n_lo_rejections += 1
                self._lo_pause = min(32, 2 * self._lo_pause) if self._lo_pause else 8
                self._lo_skip = self._lo_pause
                tier = TIER_WHITENED
                continue
            if with_grad and tier == TIER_EXTENDED and res.get("Phibar_lo") is not None and lo_slots and grad_reach is None:
                self._lo_pause = 0   # accepted: the next rejection starts with the short pause again
            self.last_tier = tier
            return res, o, info, host, head

    # ------------------------------------------------------------------ public
    def value(self, Z, ls, sf2, s2, raise_on_fail=True, strict=False):
        """F (not divided by N).  Returns (F, parts) with parts = dict(logmarg, trace_term, info)."""
        Z = self._prep_Z(Z)
        if self._small_ok(Z.shape[0], sf2=sf2):
            th, comp = self._natural_theta(ls, sf2, s2)
            nh = len(th) - 1
            h, info, _ = self._small_eval(Z, th, 0, False, False, comp)
            self.n_evals += 1
            if info != 0:
                if raise_on_fail:
                    raise NotPositiveDefiniteError(info)
                return float("nan"), {"info": info}
            hl = h.tolist()  # (plain floats once: indexing the tensor costs ~1.5 us per element)
            return hl[0], {"logmarg": hl[nh + 2], "trace_term": hl[nh + 3], "info": 0}
        res, o, info, host, _ = self._evaluate(Z, ls, sf2, s2, strict=strict)
        self.n_evals += 1
        if info != 0:
            if raise_on_fail:
                raise NotPositiveDefiniteError(info)
            return float("nan"), {"info": info}
        return float(o[OUT_F]), {"logmarg": float(o[OUT_LOGMARG]), "trace_term": float(o[OUT_TRACE]), "info": 0}

    def _natural_theta(self, ls, sf2, s2):
        """(theta of the single launch in natural parameters, composite description or None)."""
        vals = [float(v) for v in (ls.tolist() if hasattr(ls, "tolist") else ls)]
        if self.kernel == "composite":
            if len(vals) != COMP_LEN:
                raise ValueError("composite kernels take the %d-entry parameter block as `ls`" % COMP_LEN)
            return vals + [float(s2)], {"structure": vals}
        if len(vals) == 1 and self.d > 1:
            vals = vals * self.d
        if len(vals) != self.d:
            raise ValueError("lengthscale has %d entries, expected %d" % (len(vals), self.d))
        return vals + [float(sf2), float(s2)], None

    def value_and_grad(self, Z, ls, sf2, s2, want_gz=False, raise_on_fail=True, grad_reach=None, strict=False):
        """F and dF/d{lengthscale_j, sf2, s2[, Z]} (natural parameters, not their raw transforms).

        Returns (F, grads) with grads = dict(ls=tensor[d] (cpu), sf2=float, s2=float, Z=device tensor or None).
        grad_reach / strict: see HmcTarget(gradient="sampler") -- how far (x the tolerance) the extended order's explicit-Phibar gradient
        is accepted (default: `extended_grad_range`, what keeps 1e-6), and whether the accepted tier must be the one the evaluation's own
        estimate names (so that value and gradient are functions of theta alone).
        """
        e = self.engine
        Z = self._prep_Z(Z)
        M, d = Z.shape
        if self._small_ok(M, want_gz, sf2):
            th, comp = self._natural_theta(ls, sf2, s2)
            h, info, gz = self._small_eval(Z, th, 0, True, want_gz, comp)
            self.n_evals += 1
            self.n_grads += 1
            if info != 0:
                if raise_on_fail:
                    raise NotPositiveDefiniteError(info)
                return float("nan"), {"info": info}
            nh = len(th) - 1  # kernel hyper-parameter entries: d lengthscales + sf2, or the composite block
            hl = h.tolist()
            if comp is not None:
                return hl[0], {"ls": h[1:1 + nh].clone(), "sf2": 0.0, "s2": hl[1 + nh], "Z": None, "info": 0,
                               "logmarg": hl[nh + 2], "trace_term": hl[nh + 3]}
            return hl[0], {"ls": h[1:1 + d].clone(), "sf2": hl[1 + d], "s2": hl[2 + d], "Z": gz, "info": 0,
                           "logmarg": hl[d + 3], "trace_term": hl[d + 4]}
        nh = e.hyper_len(self.kernel, d) if hasattr(e, "hyper_len") else d  # composite kernels: the parameter block
        res, o, info, host, head = self._evaluate(Z, ls, sf2, s2, with_grad=True, want_gz=want_gz, grad_reach=grad_reach, strict=strict)
        self.n_evals += 1
        self.n_grads += 1
        if info != 0:
            if raise_on_fail:
                raise NotPositiveDefiniteError(info)
            return float("nan"), {"info": info}
        gh = host[head:]
        g = res["buf"][head:]
        grads = {"ls": gh[:nh].clone(), "sf2": float(gh[nh]), "s2": float(o[OUT_S2BAR]),
                 "Z": g[nh + 1:nh + 1 + M * d].reshape(M, d) if want_gz else None, "info": 0,
                 "logmarg": float(o[OUT_LOGMARG]), "trace_term": float(o[OUT_TRACE])}
        return float(o[OUT_F]), grads

    def factors(self, Z, ls, sf2, s2):
        """Device tensor [Linv | G | q] for ``predict`` (computed from the current statistics)."""
        Z = self._prep_Z(Z)
        res, _, info, _, _ = self._evaluate(Z, ls, sf2, s2, want_factors=True)
        if info != 0:
            raise NotPositiveDefiniteError(info)
        return res["factors"]

    def predict(self, Xs, Z, ls, sf2, s2, pred_noise=True, full_cov=False, factors=None):
        """Posterior predictive mean / variance (/ covariance) at Xs -- models/sgpr.py:150-160, 256-286."""
        Z = self._prep_Z(Z)
        if Xs.dim() == 1:
            Xs = Xs[:, None]
        Xs = Xs.detach().to(dtype=torch.float64, device=self.engine.device).contiguous()
        if factors is None:
            factors = self.factors(Z, ls, sf2, s2)
        return self.engine.predict(Xs, Z, ls, sf2, s2, factors, self.kernel, pred_noise=pred_noise, full_cov=full_cov)


def device_run_fits(n_rows: int, n_draws_total: int, max_treedepth: int) -> bool:
    """sgp_small_nuts refuses (SGP_ERR_DIM) a run whose worst case -- (draws + 2) x 2^depth evaluations x workgroups of the launch
    (3 + one per 64-row slab, at most 211) -- could pass 2^31, the range of its cumulative sync counters."""
    grid = 3 + min((int(n_rows) + 63) // 64, 208)
    return (float(n_draws_total) + 2.0) * float(1 << int(max_treedepth)) * grid <= 2.0e9


def shard_rows(N: int, rank: int, world: int):
    """Contiguous row block [lo, hi) of rank ``rank`` (X[g N/G : (g+1) N/G])."""
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


# ---------------------------------------------------------------------------------------------
# HMC target: VFE logp + priors + log-Jacobians  (reference models/bayesian_sgpr_hmc.py:60-71)
# ---------------------------------------------------------------------------------------------
class HmcTarget:
    """logp(theta_unc) and its gradient, theta_unc = [log ls_1..d, log sig_f, log sig_n].

    ls ~ Gamma(alpha=2, beta=1), sig_f ~ HalfCauchy(1), sig_n ~ HalfCauchy(1), all log-transformed
    as PyMC3 does for positive variables; covariance sig_f**2 * ExpQuad(ls), noise sig_n, Kuu jitter
    1e-6 (``stabilize``).  A failed Cholesky gives logp = -inf (PyMC3: ``on_error='nan'``), which the
    sampler treats as a divergence, never an exception.
    """

    def __init__(self, bound: CollapsedBound, Z, gradient="parity"):
        """gradient: "parity" (default) -- every gradient holds 1e-6 against the CPU path (north_star): where the streaming order's error
        estimate is beyond 3 x its tolerance a leapfrog runs in the whitened order (74 instead of 49 ms at C5).
        "sampler" -- opt-in for NUTS in such a region: the extended order serves value AND gradient as far as its VALUE holds (2^14 x the
        tolerance; 55 ms per leapfrog at C5).  The energy still meets 1e-8 per datum; the force is the extended order's explicit-Phibar
        gradient, off by up to ~1e-4 relative at the far end of that range (profiles/r04_extended_order_c5.jsonl) -- but a deterministic
        function of theta: the tier of every evaluation is the one its OWN error estimate names (`strict`), never the guard's memory of
        earlier evaluations.  Leapfrog with a deterministic approximate force is still volume preserving and reversible, and the
        accept step uses the accurate energy, so the chain still targets the exact posterior (tests/test_posterior_pin.py holds the
        mode to the same 4 MCSE pin as the default); only the acceptance rate pays for the force error."""
        if gradient not in ("parity", "sampler"):
            raise ValueError("gradient must be 'parity' or 'sampler'")
        self.bound = bound
        self.Z = bound._prep_Z(Z)
        self.d = bound.d
        self.ndim = self.d + 2
        self.gradient = gradient

    def start(self):
        """PyMC3's test point in the unconstrained space: Gamma(2,1) -> mean 2, HalfCauchy(1) -> 1."""
        return [math.log(2.0)] * self.d + [0.0, 0.0]

    def device_sampler_ok(self, n_draws_total=None, max_treedepth=10):
        """True when ``hmc.sample_nuts_device`` can run this target: the bound takes the single-launch path -- and, when the run
        length is given, its worst case (every tree at the depth limit) stays inside the persistent kernel's cumulative int
        counters (``device_run_fits``); the host-driven sampler over the same single launch takes the longer runs."""
        ok = hasattr(self.bound.engine, "small_nuts") and self.bound._small_ok(self.Z.shape[0])
        return ok and (n_draws_total is None or device_run_fits(int(self.bound.X.shape[0]), n_draws_total, max_treedepth))

    @staticmethod
    def _prior(ls, sf, sn):
        lp = sum(math.log(v) - v for v in ls)
        g_ls = [1.0 / v - 1.0 for v in ls]
        c = math.log(2.0) - math.log(math.pi)
        lp += (c - math.log1p(sf * sf)) + (c - math.log1p(sn * sn))
        return lp, g_ls, -2.0 * sf / (1.0 + sf * sf), -2.0 * sn / (1.0 + sn * sn)

    @staticmethod
    def _in_range(theta):
        # exp() of the log-transformed variables must stay representable; beyond it the density is treated as
        # zero (PyMC3: non-finite logp -> divergence), never an exception
        return all(math.isfinite(float(v)) and abs(float(v)) < 300.0 for v in theta)

    def constrain(self, theta):
        th = [float(v) for v in theta]
        return {"ls": [math.exp(v) for v in th[: self.d]], "sig_f": math.exp(th[self.d]), "sig_n": math.exp(th[self.d + 1])}

    def logp(self, theta):
        if not self._in_range(theta):
            return -math.inf
        if self.bound._small_ok(self.Z.shape[0]):
            return self.logp_and_grad(theta)[0]
        p = self.constrain(theta)
        F, parts = self.bound.value(self.Z, p["ls"], p["sig_f"] ** 2, p["sig_n"] ** 2, raise_on_fail=False,
                                    **({"strict": True} if self.gradient == "sampler" else {}))
        if parts.get("info", 0) != 0 or not math.isfinite(F):
            return -math.inf
        lp, _, _, _ = self._prior(p["ls"], p["sig_f"], p["sig_n"])
        return F + lp + sum(float(v) for v in theta)

    def logp_and_grad(self, theta):
        """Returns (logp, grad list[d+2]).  One call = one HMC leapfrog's worth of device work."""
        theta = theta.tolist() if hasattr(theta, "tolist") else [float(v) for v in theta]  # plain floats once (the sampler hands an ndarray)
        if not self._in_range(theta):
            return -math.inf, [0.0] * self.ndim
        b = self.bound
        if b._small_ok(self.Z.shape[0]):
            # ONE launch: transforms, priors and Jacobians are applied on the device (mode SGP_SMALL_HMC)
            h, info, _ = b._small_eval(self.Z, [float(v) for v in theta], 1, True, False)
            b.n_evals += 1
            b.n_grads += 1
            hl = h.tolist()
            lp = hl[0]
            if info != 0 or not math.isfinite(lp):
                return -math.inf, [0.0] * self.ndim
            return lp, hl[1:1 + self.ndim]
        p = self.constrain(theta)
        ls, sf, sn = p["ls"], p["sig_f"], p["sig_n"]
        kw = {"grad_reach": self.bound.extended_range, "strict": True} if self.gradient == "sampler" else {}
        F, g = self.bound.value_and_grad(self.Z, ls, sf * sf, sn * sn, want_gz=False, raise_on_fail=False, **kw)
        if g.get("info", 0) != 0 or not math.isfinite(F):
            return -math.inf, [0.0] * self.ndim
        lp, pg_ls, pg_sf, pg_sn = self._prior(ls, sf, sn)
        gl = g["ls"].tolist()  # (one conversion: indexing a tensor element by element costs ~1.5 us each -- 27 us per leapfrog at d = 18)
        grad = []
        for j in range(self.d):  # d/d log ls = ls * d/d ls ; + Jacobian term 1
            grad.append(ls[j] * (gl[j] + pg_ls[j]) + 1.0)
        grad.append(sf * (2.0 * sf * g["sf2"] + pg_sf) + 1.0)
        grad.append(sn * (2.0 * sn * g["s2"] + pg_sn) + 1.0)
        return F + lp + sum(float(v) for v in theta), grad
