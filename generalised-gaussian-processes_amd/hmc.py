"""Host-side NUTS driver over the device log-density -- the caller of one hot-path evaluation per leapfrog.

Stands in for what the reference gets from ``pm.NUTS()`` / ``pm.sample(n, tune=tune, chains=1,
return_inferencedata=False)`` (reference models/bayesian_sgpr_hmc.py:73-78): multinomial NUTS with a
generalised U-turn criterion, dual-averaging step-size adaptation (target_accept 0.8) and windowed
diagonal mass-matrix adaptation started from ``jitter+adapt_diag``.  PyMC3 itself is not vendored in
the reference and not installed here; this is a restatement of the published algorithm (Hoffman &
Gelman 2014; Betancourt 2017) with PyMC3's documented defaults, not a transcription, and chains are
comparable statistically, not draw-for-draw (SURVEY.md section 7 "Hard parts").

Only the tree logic runs on the host (theta has d+2 entries); every leapfrog calls
``target.logp_and_grad(theta)`` = pass 1 + tail + pass 2 on the GPU(s).
"""
from __future__ import annotations

import math
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


# ---------------------------------------------------------------------------------------------
# random numbers shared with the device sampler
# ---------------------------------------------------------------------------------------------
class SplitMix:
    """splitmix64 -> uniforms / Box-Muller normals: the generator of the device-resident sampler (csrc/sgp_nuts.hpp),
    restated so that ``NUTS(..., rng=SplitMix(seed))`` draws the same stream -- the CPU tests compare the two samplers
    draw for draw.  Implements the three ``numpy.random.Generator`` methods the sampler uses."""

    MASK = (1 << 64) - 1

    def __init__(self, seed: int):
        self.s = int(seed) & self.MASK
        self.spare = None

    def _u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.MASK
        return z ^ (z >> 31)

    def random(self):
        return float(self._u64() >> 11) * (1.0 / 9007199254740992.0)

    def _normal(self):
        if self.spare is not None:
            v, self.spare = self.spare, None
            return v
        u1, u2 = 1.0 - self.random(), self.random()
        rad, ang = math.sqrt(-2.0 * math.log(u1)), 6.283185307179586 * u2
        self.spare = rad * math.sin(ang)
        return rad * math.cos(ang)

    def standard_normal(self, n):
        return np.array([self._normal() for _ in range(n)], dtype=np.float64)

    def uniform(self, lo, hi, n):
        return np.array([lo + (hi - lo) * self.random() for _ in range(n)], dtype=np.float64)


def _logaddexp(a, b):
    """log(exp(a) + exp(b)), the formula of csrc/sgp_nuts.hpp (so both samplers round alike)."""
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    return max(a, b) + math.log1p(math.exp(-abs(a - b)))


# ---------------------------------------------------------------------------------------------
# adaptation
# ---------------------------------------------------------------------------------------------
class DualAveraging:
    """Nesterov dual averaging of log(step size) towards a target mean acceptance (Hoffman & Gelman 2014, Alg. 5)."""

    def __init__(self, initial_step, target=0.8, gamma=0.05, t0=10.0, kappa=0.75):
        self.mu = math.log(10.0 * initial_step)
        self.target, self.gamma, self.t0, self.kappa = target, gamma, t0, kappa
        self.log_step = math.log(initial_step)
        self.log_bar = math.log(initial_step)  # tune = 0 samples with the initial step (PyMC3 behaviour), not exp(0)
        self.hbar = 0.0
        self.count = 1

    def update(self, accept_stat):
        w = 1.0 / (self.count + self.t0)
        self.hbar = (1.0 - w) * self.hbar + w * (self.target - accept_stat)
        self.log_step = self.mu - self.hbar * math.sqrt(self.count) / self.gamma
        mk = self.count ** (-self.kappa)  # count = 1: the average starts from the first adapted step
        self.log_bar = mk * self.log_step + (1.0 - mk) * self.log_bar
        self.count += 1

    def current(self, tuning):
        return math.exp(self.log_step if tuning else self.log_bar)


class _WeightedVariance:
    """Running mean / variance that starts from a prior guess carrying ``weight`` pseudo-observations -- PyMC3's
    ``_WeightedVariance`` behind ``init='jitter+adapt_diag'`` [UPSTREAM pymc3/step_methods/hmc/quadpotential.py]:
    the foreground estimator starts at mean = start point, variance = 1, weight = 10, so the first few (possibly
    identical) draws cannot collapse the metric; the background estimator starts empty."""

    def __init__(self, n, mean=None, var=None, weight=0.0):
        self.n = float(weight)
        self.mean = np.zeros(n) if mean is None else np.array(mean, dtype=np.float64)
        self.m2 = np.zeros(n) if var is None else np.array(var, dtype=np.float64) * float(weight)

    def add(self, x):
        self.n += 1.0
        d = x - self.mean
        self.mean = self.mean + d / self.n
        self.m2 = self.m2 + d * (x - self.mean)

    def var(self):
        return self.m2 / self.n if self.n > 0 else None


class DiagMassAdapter:
    """Windowed diagonal mass matrix (inverse metric = posterior variance estimate), foreground / background estimators
    switched every ``window`` tuning draws (PyMC3 ``QuadPotentialDiagAdapt``, adaptation_window = 101)."""

    def __init__(self, n, window=101, growth=1.0, initial_mean=None, initial_weight=10.0):
        self.n = n
        self.var = np.ones(n)
        self.window = window
        self.growth = growth
        self.fg = _WeightedVariance(n, np.zeros(n) if initial_mean is None else initial_mean, np.ones(n), initial_weight)
        self.bg = _WeightedVariance(n)
        self.count = 0

    def update(self, sample, tuning):
        if not tuning:
            return
        self.fg.add(sample)
        self.bg.add(sample)
        v = self.fg.var()
        if v is not None and np.all(np.isfinite(v)) and np.all(v > 0.0):
            self.var = v
        if self.count > 0 and self.count % self.window == 0:
            self.fg = self.bg
            self.bg = _WeightedVariance(self.n)
            self.window = int(self.window * self.growth)
        self.count += 1


# ---------------------------------------------------------------------------------------------
# NUTS
# ---------------------------------------------------------------------------------------------
class _State:
    __slots__ = ("q", "p", "v", "grad", "logp", "energy")

    def __init__(self, q, p, v, grad, logp, energy):
        self.q, self.p, self.v, self.grad, self.logp, self.energy = q, p, v, grad, logp, energy


class _Tree:
    __slots__ = ("left", "right", "p_sum", "proposal", "log_size", "accept_sum", "n", "diverging", "turning")


class NUTS:
    def __init__(self, logp_and_grad: Callable[[Sequence[float]], tuple], ndim: int, step_scale=0.25, target_accept=0.8,
                 max_treedepth=10, Emax=1000.0, seed: Optional[int] = None, rng=None):
        self.f = logp_and_grad
        self.ndim = ndim
        self.rng = rng if rng is not None else np.random.default_rng(seed)
        self.step0 = step_scale / ndim ** 0.25
        self.da = DualAveraging(self.step0, target=target_accept)
        self.mass = DiagMassAdapter(ndim)
        self.max_treedepth = max_treedepth
        self.Emax = Emax
        self.n_leapfrog = 0

    # -- Hamiltonian pieces
    def _eval(self, q):
        lp, g = self.f(q)
        self.n_leapfrog += 1
        return float(lp), np.asarray(g, dtype=np.float64)

    @staticmethod
    def _dot(a, b):
        s = 0.0
        for x, y in zip(a.tolist(), b.tolist()):  # left-to-right like csrc/sgp_nuts.hpp (BLAS may reorder)
            s += x * y
        return s

    def _state(self, q, p, lp, g):
        v = self.mass.var * p
        energy = -lp + 0.5 * self._dot(p, v) if math.isfinite(lp) else math.inf
        return _State(q, p, v, g, lp, energy)

    def _leapfrog(self, s: _State, eps):
        p_half = s.p + 0.5 * eps * s.grad
        q_new = s.q + eps * (self.mass.var * p_half)
        lp, g = self._eval(q_new)
        if not math.isfinite(lp) or not np.all(np.isfinite(g)):
            return self._state(q_new, p_half, -math.inf, np.zeros_like(g))
        p_new = p_half + 0.5 * eps * g
        return self._state(q_new, p_new, lp, g)

    # -- recursive doubling
    def _leaf(self, s_new: _State, e0):
        t = _Tree()
        de = s_new.energy - e0
        t.left = t.right = t.proposal = s_new
        t.p_sum = s_new.p.copy()
        if not math.isfinite(de):
            de = math.inf
        t.diverging = de > self.Emax
        t.log_size = -de if math.isfinite(de) else -math.inf
        t.accept_sum = min(1.0, math.exp(-de)) if de > -700 and math.isfinite(de) else (1.0 if de <= -700 else 0.0)
        t.n = 1
        t.turning = False
        return t

    @staticmethod
    def _uturn(p_sum, left: _State, right: _State):
        return NUTS._dot(p_sum, left.v) <= 0.0 or NUTS._dot(p_sum, right.v) <= 0.0

    def _build(self, edge: _State, direction, depth, eps, e0):
        if depth == 0:
            return self._leaf(self._leapfrog(edge, direction * eps), e0)
        a = self._build(edge, direction, depth - 1, eps, e0)
        if a.diverging or a.turning:
            return a
        a_edge = a.right if direction > 0 else a.left
        b = self._build(a_edge, direction, depth - 1, eps, e0)
        t = _Tree()
        t.left, t.right = (a.left, b.right) if direction > 0 else (b.left, a.right)
        t.p_sum = a.p_sum + b.p_sum
        t.log_size = _logaddexp(a.log_size, b.log_size)
        t.accept_sum = a.accept_sum + b.accept_sum
        t.n = a.n + b.n
        t.diverging = b.diverging
        # multinomial sampling inside the subtree
        if not (b.diverging or b.turning) and math.log(self.rng.random() + 1e-300) < b.log_size - t.log_size:
            t.proposal = b.proposal
        else:
            t.proposal = a.proposal
        t.turning = b.turning
        if not (t.diverging or t.turning):
            first, second = (a, b) if direction > 0 else (b, a)
            t.turning = (self._uturn(t.p_sum, t.left, t.right)
                         or self._uturn(first.p_sum + second.left.p, t.left, second.left)
                         or self._uturn(first.right.p + second.p_sum, first.right, t.right))
        return t

    def draw(self, q, lp, g, tuning):
        """One NUTS transition from (q, logp, grad).  Returns (q', logp', grad', stats)."""
        eps = self.da.current(tuning)
        p0 = self.rng.standard_normal(self.ndim) / np.sqrt(self.mass.var)
        s0 = self._state(q, p0, lp, g)
        e0 = s0.energy
        left = right = s0
        proposal = s0
        p_sum = p0.copy()
        log_size = 0.0
        accept_sum, n_states = 0.0, 0
        depth = 0
        diverging = False
        while depth < self.max_treedepth:
            direction = 1 if self.rng.random() < 0.5 else -1
            sub = self._build(right if direction > 0 else left, direction, depth, eps, e0)
            accept_sum += sub.accept_sum
            n_states += sub.n
            if sub.diverging:
                diverging = True
                break
            if sub.turning:
                break
            # biased progressive sampling at the top level
            if math.log(self.rng.random() + 1e-300) < sub.log_size - log_size:
                proposal = sub.proposal
            log_size = _logaddexp(log_size, sub.log_size)
            first_p, first_right = p_sum, right
            if direction > 0:
                right = sub.right
            else:
                left = sub.left
            p_sum = p_sum + sub.p_sum
            depth += 1
            if self._uturn(p_sum, left, right):
                break
        accept = accept_sum / max(1, n_states)
        stats = {"step_size": eps, "tree_size": n_states, "depth": depth, "mean_tree_accept": accept,
                 "diverging": diverging, "energy": proposal.energy, "tune": tuning}
        if tuning:
            self.da.update(accept)
            self.mass.update(proposal.q, tuning)
        return proposal.q, proposal.logp, proposal.grad, stats


class Trace:
    """What the reference reads from a PyMC3 MultiTrace (models/bayesian_sgpr_hmc.py:123-157,206-216;
    experiments/demo_1d_regression.py:199-206): ``len``, ``trace[i]`` -> dict, ``trace['ls']`` -> column,
    slices, ``get_sampler_stats(name)``."""

    def __init__(self, samples: List[Dict[str, np.ndarray]], stats: Dict[str, np.ndarray], varnames=("ls", "sig_f", "sig_n")):
        self._samples = samples
        self._stats = stats
        self.varnames = list(varnames)

    def __len__(self):
        return len(self._samples)

    def __getitem__(self, idx):
        if isinstance(idx, str):
            return np.stack([np.asarray(s[idx]) for s in self._samples]) if self._samples else np.zeros((0,))
        if isinstance(idx, slice):
            return Trace(self._samples[idx], {k: v[idx] for k, v in self._stats.items()}, self.varnames)
        return self._samples[idx]

    def __iter__(self):
        return iter(self._samples)

    def get_values(self, name):
        return self[name]

    def get_sampler_stats(self, name):
        return self._stats[name]

    @property
    def stat_names(self):
        return set(self._stats)


def shared_seed(seed: Optional[int], group=None) -> Optional[int]:
    """One seed for every rank of a ``torch.distributed`` job (the tree logic runs redundantly
    on every rank "from a shared seed").  Each leapfrog issues collectives, so ranks whose random streams differ build
    different trees, issue different numbers of all-reduces and hang.  Rank 0's seed (fresh entropy when ``seed`` is
    None) is broadcast; without an initialised process group ``seed`` is returned unchanged."""
    try:
        import torch
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return seed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return seed
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0] >> 1)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([int(seed)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.item())


def sample_nuts_device(target, n_samples: int, tune: int, seed: Optional[int] = None, start: Optional[Sequence[float]] = None,
                       step_scale=0.25, target_accept=0.8, max_treedepth=10) -> Trace:
    """``pm.sample(n_samples, tune=tune, chains=1)`` entirely on the GPU: one persistent launch (``sgp_small_nuts``) runs the
    sampler and every leapfrog's evaluation; theta, the momentum and the sampler state never visit the host (SURVEY section 8
    f-1).  ``target`` is an ``HmcTarget`` or ``CompositeHmcTarget`` whose bound takes the single-launch path (M <= 128, one rank).  Same algorithm and
    random stream as ``NUTS(..., rng=SplitMix(seed))``; the trace has the surface the reference reads
    (``trace['ls']``, ``trace[i]``, ``get_sampler_stats('step_size' | 'perf_counter_diff')``)."""
    b = target.bound
    if not target.device_sampler_ok():
        raise ValueError("the device sampler needs the single-launch path (M <= 128, one rank; stationary kernels d <= 24, "
                         "composite kernels d <= 8)")
    from .core import device_run_fits
    if not device_run_fits(int(b.X.shape[0]), n_samples + tune, max_treedepth):
        raise ValueError("a run of %d draws at tree depth %d could overflow the persistent kernel's counters: use sample_nuts "
                         "(host-driven, same single-launch evaluations) or a smaller max_treedepth" % (n_samples + tune, max_treedepth))
    nd = target.ndim
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0] >> 1)
    rng = SplitMix(seed)
    if start is None:  # PyMC3's jitter around the test point; redrawn while the density there is zero
        q = np.asarray(target.start(), dtype=np.float64) + rng.uniform(-1.0, 1.0, nd)
        tries = 0
        while not math.isfinite(target.logp(q)) and tries < 20:
            q = np.asarray(target.start(), dtype=np.float64) + rng.uniform(-1.0, 1.0, nd)
            tries += 1
    else:
        q = np.asarray(start, dtype=np.float64).copy()
    if not math.isfinite(target.logp(q)):
        raise RuntimeError("could not find a starting point with finite log-density" if start is None
                           else "the log-density is not finite at the supplied start")
    t0 = time.perf_counter()
    extra = target.device_sampler_args() if hasattr(target, "device_sampler_args") else {}  # composite kernels: the structure
    r = b.engine.small_nuts(b.X, b.y, target.Z, q, tune, n_samples, rng.s, jitter=b.jitter, kernel=b.kernel,
                            max_treedepth=max_treedepth, step_scale=step_scale, target_accept=target_accept, **extra)
    wall = time.perf_counter() - t0
    if r["info"] < 0:
        from .core import SgpTimeoutError
        raise SgpTimeoutError()
    if r["draws"] != tune + n_samples:
        raise RuntimeError("the device sampler stopped after %d of %d draws" % (r["draws"], tune + n_samples))
    b.n_evals += r["evaluations"]
    b.n_grads += r["evaluations"]
    th = r["samples"].numpy()
    samples = []
    for row in th:
        c = target.constrain(row)
        samples.append({"ls": np.asarray(c["ls"], dtype=np.float64), "sig_f": float(c["sig_f"]), "sig_n": float(c["sig_n"]),
                        "theta_unc": row.copy()})
    st = r["stats"].numpy()
    stats = {"step_size": st[:, 0], "tree_size": st[:, 1], "depth": st[:, 2], "mean_tree_accept": st[:, 3],
             "diverging": st[:, 4] != 0.0, "energy": st[:, 5], "logp": st[:, 6], "perf_counter_diff": r["seconds"].numpy()}
    tr = Trace(samples, stats)
    tr.n_leapfrog = r["evaluations"]
    tr.wall_clock_secs = wall
    tr.device_resident = True
    return tr


def _few_threads(fn):
    from .core import few_host_threads  # (core imports nothing from this module)
    return few_host_threads(fn)


@_few_threads
def sample_nuts(target, n_samples: int, tune: int, seed: Optional[int] = None, start: Optional[Sequence[float]] = None,
                step_scale=0.25, target_accept=0.8, max_treedepth=10, progress: Optional[Callable[[int, dict], None]] = None,
                group=None) -> Trace:
    """``pm.sample(n_samples, tune=tune, chains=1)`` for an ``HmcTarget``-like object
    (``ndim``, ``logp_and_grad``, ``constrain``).  Returns the post-tuning draws as a ``Trace``.
    Under ``torch.distributed`` every rank calls this; the seed is made common first (``shared_seed``)."""
    nd = target.ndim
    seed = shared_seed(seed, group)
    nuts = NUTS(target.logp_and_grad, nd, step_scale=step_scale, target_accept=target_accept, max_treedepth=max_treedepth, seed=seed)

    def test_point():
        # PyMC3 test point (Gamma(2,1) -> mean 2 ; HalfCauchy(1) -> 1) in log space, plus U(-1,1) jitter
        base = np.asarray(target.start(), dtype=np.float64) if hasattr(target, "start") else np.array([math.log(2.0)] * (nd - 2) + [0.0, 0.0])
        return base + nuts.rng.uniform(-1.0, 1.0, nd)

    q = test_point() if start is None else np.asarray(start, dtype=np.float64).copy()
    lp, g = nuts._eval(q)
    tries = 0
    while start is None and not math.isfinite(lp) and tries < 20:  # unlucky jitter: redraw (a user-supplied start is never replaced)
        q = test_point()
        lp, g = nuts._eval(q)
        tries += 1
    if not math.isfinite(lp):
        raise RuntimeError("could not find a starting point with finite log-density" if start is None
                           else "the log-density is not finite at the supplied start")
    nuts.mass = DiagMassAdapter(nd, initial_mean=q)  # jitter+adapt_diag: mean = start, variance 1, weight 10
    samples, stat_rows = [], []
    for it in range(tune + n_samples):
        tuning = it < tune
        t0 = time.perf_counter()
        q, lp, g, st = nuts.draw(q, lp, g, tuning)
        st["perf_counter_diff"] = time.perf_counter() - t0
        st["logp"] = lp
        if progress is not None:
            progress(it, st)
        if not tuning:
            c = target.constrain(q)
            samples.append({"ls": np.asarray(c["ls"], dtype=np.float64), "sig_f": float(c["sig_f"]), "sig_n": float(c["sig_n"]),
                            "theta_unc": q.copy()})
            stat_rows.append(st)
    keys = ("step_size", "tree_size", "depth", "mean_tree_accept", "diverging", "energy", "perf_counter_diff", "logp")
    stats = {k: np.array([r[k] for r in stat_rows]) for k in keys}
    tr = Trace(samples, stats)
    tr.n_leapfrog = nuts.n_leapfrog
    return tr
