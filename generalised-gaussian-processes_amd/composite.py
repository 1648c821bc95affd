"""Sum-of-products covariance functions on the HIP core.

The reference's CO2 workload builds its covariance twice -- GPyTorch kernels for the optimisation stage
(experiments/co2_bayesian_sgpr_hmc.py:74-83) and PyMC3 ``pm.gp.cov`` objects for the NUTS stage (:107-149):

    n_per**2 * Periodic(1, period=1, ls=l_psmooth) * ExpQuad(1, l_pdecay)  +  n_med**2 * RatQuad(1, l_med, alpha)
      +  n_trend**2 * ExpQuad(1, l_trend)  +  n_noise**2 * Matern32(1, l_noise)

``CompositeKernel`` describes such a kernel as data (terms of amplitude * factors) and packs it into the parameter
block ``include/sgp.h`` documents (SGP_KERNEL_COMPOSITE); ``CollapsedBound(kernel="composite")`` evaluates the same
collapsed bound, its gradient with respect to every entry of the block, Z and the noise, and the predictive.
``CompositeHmcTarget`` is the NUTS target of the reference's PyMC3 model: Normal priors on the log-parameters,
HalfNormal(1) on the noise standard deviation.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import COMP_LEN
from .core import CollapsedBound

EXPQUAD, MATERN32, MATERN52, RATQUAD, PERIODIC = 0, 1, 2, 3, 4
_FACTOR_IDS = {"expquad": EXPQUAD, "rbf": EXPQUAD, "matern32": MATERN32, "matern52": MATERN52, "ratquad": RATQUAD,
               "rq": RATQUAD, "periodic": PERIODIC}
MAX_TERMS, MAX_FACTORS = 4, 2


class Factor:
    """One isotropic factor: kind in {expquad, matern32, matern52, ratquad, periodic}; ``aux`` is RatQuad's alpha or
    Periodic's period.  ``fixed_aux`` keeps it out of the sampled / optimised parameters (the reference pins
    period_length = 1, co2_bayesian_sgpr_hmc.py:79-80)."""

    def __init__(self, kind: str, ls: float, aux: float = 0.0, fixed_aux: bool = False):
        self.kind = _FACTOR_IDS[kind.lower()]
        self.ls, self.aux, self.fixed_aux = float(ls), float(aux), bool(fixed_aux)
        if self.kind in (RATQUAD, PERIODIC) and not self.aux > 0.0:
            raise ValueError("ratquad needs alpha > 0, periodic needs period > 0")


class CompositeKernel:
    """terms: [(amplitude_sd, [Factor, ...]), ...]; k = sum_t amplitude_sd_t**2 * prod_f factor_tf."""

    def __init__(self, terms: Sequence[Tuple[float, Sequence[Factor]]]):
        if not 1 <= len(terms) <= MAX_TERMS:
            raise ValueError("1..%d terms" % MAX_TERMS)
        self.terms = [(float(a), list(f)) for a, f in terms]
        for _, f in self.terms:
            if not 1 <= len(f) <= MAX_FACTORS:
                raise ValueError("1..%d factors per term" % MAX_FACTORS)

    # ------------------------------------------------------------------ parameter block <-> named parameters
    def block(self) -> List[float]:
        b = [0.0] * COMP_LEN
        b[0] = float(len(self.terms))
        for t, (amp, facs) in enumerate(self.terms):
            base = 1 + 8 * t
            b[base], b[base + 1] = amp * amp, float(len(facs))
            for f, fac in enumerate(facs):
                fb = base + 2 + 3 * f
                b[fb], b[fb + 1], b[fb + 2] = float(fac.kind), fac.ls, fac.aux
        return b

    def free_parameters(self) -> List[Tuple[str, int, str]]:
        """(name, block slot, role) of every positive parameter, in a fixed order: role 'amp' (the block stores
        amp**2), 'ls' or 'aux'."""
        out = []
        for t, (_, facs) in enumerate(self.terms):
            base = 1 + 8 * t
            out.append(("amp_%d" % t, base, "amp"))
            for f, fac in enumerate(facs):
                fb = base + 2 + 3 * f
                out.append(("ls_%d_%d" % (t, f), fb + 1, "ls"))
                if fac.kind in (RATQUAD, PERIODIC) and not fac.fixed_aux:
                    out.append(("aux_%d_%d" % (t, f), fb + 2, "aux"))
        return out

    def values(self) -> List[float]:
        """Current values of ``free_parameters()`` (amplitudes as standard deviations)."""
        b = self.block()
        return [math.sqrt(b[s]) if role == "amp" else b[s] for _, s, role in self.free_parameters()]

    def with_values(self, vals: Sequence[float]) -> "CompositeKernel":
        it = iter(float(v) for v in vals)
        terms = []
        for amp, facs in self.terms:
            a = next(it)
            nf = []
            for fac in facs:
                ls = next(it)
                aux = fac.aux
                if fac.kind in (RATQUAD, PERIODIC) and not fac.fixed_aux:
                    aux = next(it)
                nf.append(_clone_factor(fac, ls, aux))
            terms.append((a, nf))
        return CompositeKernel(terms)


def _clone_factor(fac: Factor, ls: float, aux: float) -> Factor:
    f = Factor.__new__(Factor)
    f.kind, f.ls, f.aux, f.fixed_aux = fac.kind, float(ls), float(aux), fac.fixed_aux
    return f


def co2_kernel(n_per=1.0, l_psmooth=1.0, l_pdecay=1.0, n_med=1.0, l_med=1.0, alpha=1.0, n_trend=1.0, l_trend=1.0,
               n_noise=1.0, l_noise=1.0, period=1.0) -> CompositeKernel:
    """The reference's CO2 covariance, PyMC3 side (experiments/co2_bayesian_sgpr_hmc.py:107-149); period fixed."""
    return CompositeKernel([
        (n_per, [Factor("periodic", l_psmooth, period, fixed_aux=True), Factor("expquad", l_pdecay)]),
        (n_med, [Factor("ratquad", l_med, alpha)]),
        (n_trend, [Factor("expquad", l_trend)]),
        (n_noise, [Factor("matern32", l_noise)]),
    ])


# the reference's priors on the log-parameters of the CO2 model (co2_bayesian_sgpr_hmc.py:107-141): Normal(0, sd)
CO2_LOG_PRIOR_SD = {"amp_0": 3.0, "ls_0_0": 1.0, "ls_0_1": 0.1, "amp_1": 3.0, "ls_1_0": 3.0, "aux_1_0": 0.1,
                    "amp_2": 3.0, "ls_2_0": 1.0, "amp_3": 3.0, "ls_3_0": 1.0}


class CompositeHmcTarget:
    """logp(theta) and gradient for theta = [log of every free kernel parameter ..., log sigma].

    Kernel parameters: log p ~ Normal(0, sd_p) (the reference declares ``log_x = pm.Normal`` and uses exp(log_x), so the
    sampled variable is the log itself and there is no Jacobian term); sigma ~ HalfNormal(1), log-transformed as PyMC3
    does for positive variables (Jacobian + log sigma).  A failed Cholesky gives -inf, never an exception.
    """

    def __init__(self, bound: CollapsedBound, Z, kernel: CompositeKernel, log_prior_sd: Optional[dict] = None):
        if bound.kernel != "composite":
            raise ValueError("CompositeHmcTarget needs CollapsedBound(kernel='composite')")
        self.bound, self.kernel = bound, kernel
        self.Z = bound._prep_Z(Z)
        self.params = kernel.free_parameters()
        sd = log_prior_sd or {}
        self.sd = [float(sd.get(name, 3.0)) for name, _, _ in self.params]
        self.ndim = len(self.params) + 1

    def constrain(self, theta):
        """Trace row: 'ls' holds every free kernel parameter (``kernel.free_parameters()`` order, amplitudes as
        standard deviations), 'sig_n' the noise sd; 'sig_f' is kept at 1 so ``Trace`` keeps the reference's columns."""
        th = [float(v) for v in theta]
        vals = [math.exp(v) for v in th[:-1]]
        return {"ls": vals, "sig_f": 1.0, "sig_n": math.exp(th[-1]), "kernel": self.kernel.with_values(vals)}

    def start(self):
        """PyMC3's test point: the prior mean of every Normal log-parameter (0) and sigma = 1."""
        return [0.0] * self.ndim

    _ROLE_ID = {"amp": 0, "ls": 1, "aux": 2}

    def device_description(self):
        """The ``composite=`` argument of ``engine.small_eval`` / ``small_nuts``: the structure (term / factor types, fixed
        periods) and the table (block slot, role, prior sd) of the sampled parameters."""
        return {"structure": self.kernel.block(),
                "free": [(slot, self._ROLE_ID[role], sd) for (_, slot, role), sd in zip(self.params, self.sd)]}

    def device_sampler_ok(self, n_draws_total=None, max_treedepth=10):
        """True when ``hmc.sample_nuts_device`` can run this target (single-launch path, at most 17 sampled parameters, and -- when
        the run length is given -- a worst case inside the persistent kernel's counters, ``core.device_run_fits``)."""
        from .core import device_run_fits
        b = self.bound
        ok = hasattr(b, "_small_ok") and hasattr(b.engine, "small_nuts") and b._small_ok(self.Z.shape[0]) and self.ndim <= 18
        return ok and (n_draws_total is None or device_run_fits(int(b.X.shape[0]), n_draws_total, max_treedepth))

    def device_sampler_args(self):
        return {"composite": self.device_description()}

    def logp_and_grad(self, theta):
        th = [float(v) for v in theta]
        if not all(math.isfinite(v) and abs(v) < 300.0 for v in th):
            return -math.inf, [0.0] * self.ndim
        b = self.bound
        if hasattr(b, "_small_ok") and b._small_ok(self.Z.shape[0]):
            # ONE launch: exp transforms, priors and the chain rule run on the device (sgp_small_eval_composite, SGP_SMALL_HMC)
            if not all(abs(v) < 150.0 for v in th):
                return -math.inf, [0.0] * self.ndim
            h, info, _ = b._small_eval(self.Z, th, 1, True, False, self.device_description())
            b.n_evals += 1
            b.n_grads += 1
            lp = float(h[0])
            if info != 0 or not math.isfinite(lp):
                return -math.inf, [0.0] * self.ndim
            return lp, h[1:1 + self.ndim].tolist()
        vals = [math.exp(v) for v in th[:-1]]
        sigma = math.exp(th[-1])
        kern = self.kernel.with_values(vals)
        F, g = self.bound.value_and_grad(self.Z, kern.block(), 1.0, sigma * sigma, raise_on_fail=False)
        if not math.isfinite(F):
            return -math.inf, [0.0] * self.ndim
        gb = g["ls"]
        lp, grad = F, []
        for (name, slot, role), v, t, sd in zip(self.params, vals, th[:-1], self.sd):
            dF = float(gb[slot])
            dF_dlog = 2.0 * v * v * dF if role == "amp" else v * dF  # block stores amp**2
            lp += -0.5 * (t / sd) ** 2 - math.log(sd) - 0.5 * math.log(2.0 * math.pi)
            grad.append(dF_dlog - t / (sd * sd))
        # sigma ~ HalfNormal(1): log density 0.5 log(2/pi) - sigma^2/2, plus the log-transform Jacobian log sigma
        lp += 0.5 * math.log(2.0 / math.pi) - 0.5 * sigma * sigma + th[-1]
        grad.append(2.0 * sigma * sigma * g["s2"] - sigma * sigma + 1.0)
        return lp, grad

    def logp(self, theta):
        return self.logp_and_grad(theta)[0]


# ---------------------------------------------------------------------------------------------------------------------
# The reference's CO2 model class (experiments/co2_bayesian_sgpr_hmc.py:58-300): its own copy of BayesianSparseGPR_HMC with
# the composite covariance -- warm start with Adam on every raw parameter, then Adam on Z alone against the bound averaged
# over the current NUTS trace, with NUTS phases at the scheduled iterations.
# ---------------------------------------------------------------------------------------------------------------------
def _inv_softplus(v: float) -> float:
    return v + math.log(-math.expm1(-v)) if v < 30.0 else v


class _CompositeBoundFn(torch.autograd.Function):
    """F(values, s2, Z) / N for a composite kernel; ``values`` = the free parameters in ``CompositeKernel.free_parameters()``
    order (amplitudes as standard deviations).  Gradients from the HIP library (dF/d block -> chain rule to the values)."""

    @staticmethod
    def forward(ctx, values, s2, Z, model):
        cb = model.bound
        kern = model.kernel.with_values([float(v) for v in values.detach().tolist()])
        need = any(ctx.needs_input_grad[:3])
        if need:
            Fv, g = cb.value_and_grad(Z.detach(), kern.block(), 1.0, float(s2), want_gz=bool(ctx.needs_input_grad[2]))
            gv = []
            for (_, slot, role), v in zip(model.params, values.detach().tolist()):
                dF = float(g["ls"][slot])
                gv.append(2.0 * v * dF if role == "amp" else dF)  # the block stores amp**2
            ctx.g = (torch.tensor(gv, dtype=torch.float64), float(g["s2"]), g["Z"])
        else:
            Fv, _ = cb.value(Z.detach(), kern.block(), 1.0, float(s2))
            ctx.g = None
        ctx.N = cb.N
        ctx.meta = (values.shape, values.device, Z.shape)
        return torch.tensor(Fv / cb.N, dtype=torch.float64, device=values.device)

    @staticmethod
    def backward(ctx, gout):
        gv, gs2, gz = ctx.g
        vshape, vdev, zshape = ctx.meta
        s = gout / ctx.N
        out_v = (gv.to(vdev).reshape(vshape) * s) if ctx.needs_input_grad[0] else None
        out_s = (torch.as_tensor(gs2, dtype=torch.float64, device=vdev) * s).reshape(()) if ctx.needs_input_grad[1] else None
        out_z = (gz.reshape(zshape) * s.to(gz.device)) if ctx.needs_input_grad[2] else None
        return out_v, out_s, out_z, None


class CompositeBayesianSparseGPR_HMC(torch.nn.Module):  # noqa: N801  (after the reference's class name)
    """Collapsed sparse GP regression with a sum-of-products covariance, hyper-parameters sampled by NUTS at scheduled
    iterations -- the class of experiments/co2_bayesian_sgpr_hmc.py:58-300 on the HIP core.

    Parameters (all learnable in the warm start): ``raw_values`` (softplus -> the kernel's free parameters, amplitudes as
    standard deviations), ``raw_noise`` (softplus -> noise variance, + 1e-4 as GPyTorch's GaussianLikelihood) and
    ``inducing_points``.  ``train_model`` returns (losses, trace_hyper, trace_step_size, trace_perf_time) like the reference
    (:186-253); ``train_fixed_model`` is its HMC-only run (:257-277, 500 tune / 100 draws)."""

    def __init__(self, train_x, train_y, kernel: CompositeKernel, Z_init, log_prior_sd: Optional[dict] = None, engine=None,
                 jitter: float = 1e-6, noise: float = 0.1, seed: Optional[int] = None):
        super().__init__()
        if train_x.dim() == 1:
            train_x = train_x[:, None]
        self.bound = CollapsedBound(train_x, train_y, kernel="composite", jitter=jitter, engine=engine)
        self.kernel = kernel
        self.params = kernel.free_parameters()
        self.log_prior_sd = dict(log_prior_sd or {})
        dev = self.bound.engine.device
        self.raw_values = torch.nn.Parameter(torch.tensor([_inv_softplus(v) for v in kernel.values()], dtype=torch.float64))
        self.raw_noise = torch.nn.Parameter(torch.tensor(_inv_softplus(max(noise - 1e-4, 1e-6)), dtype=torch.float64))
        Z = torch.as_tensor(Z_init, dtype=torch.float64)
        self.inducing_points = torch.nn.Parameter((Z[:, None] if Z.dim() == 1 else Z).clone().to(dev))
        self._seed, self._n_hmc_calls = seed, 0
        self.device_sampler = True

    # ------------------------------------------------------------------ parameters
    def values(self) -> torch.Tensor:
        return torch.nn.functional.softplus(self.raw_values)

    def noise(self) -> torch.Tensor:
        return torch.nn.functional.softplus(self.raw_noise) + 1e-4

    def current_kernel(self) -> CompositeKernel:
        return self.kernel.with_values([float(v) for v in self.values().detach().tolist()])

    def freeze_kernel_hyperparameters(self):
        self.raw_values.requires_grad = False
        self.raw_noise.requires_grad = False

    def update_model_to_hyper(self, hyper_sample):
        """Set kernel parameters and noise to one draw of the trace (reference :162-184); period lengths stay fixed."""
        with torch.no_grad():
            vals = np.asarray(hyper_sample["ls"], dtype=np.float64)
            self.raw_values.copy_(torch.tensor([_inv_softplus(float(v)) for v in vals], dtype=torch.float64))
            self.raw_noise.copy_(torch.tensor(_inv_softplus(max(float(hyper_sample["sig_n"]) ** 2 - 1e-4, 1e-12)), dtype=torch.float64))

    # ------------------------------------------------------------------ bound
    def neg_bound_per_datum(self):
        return -_CompositeBoundFn.apply(self.values(), self.noise(), self.inducing_points, self)

    def sample_optimal_variational_hyper_dist(self, n_samples, Z_opt, tune, sampler_params=None):
        """NUTS over the log-parameters with Z fixed (reference :99-160).  Starts at the current parameter values."""
        from .hmc import sample_nuts, sample_nuts_device
        Z = torch.as_tensor(np.asarray(Z_opt), dtype=torch.float64)
        target = CompositeHmcTarget(self.bound, Z, self.current_kernel(), self.log_prior_sd)
        start = [math.log(float(v)) for v in self.values().detach().tolist()] + [0.5 * math.log(float(self.noise().detach()))]
        scale = 0.25 if not sampler_params else sampler_params.get("step_scale", 0.25)
        seed = None if self._seed is None else self._seed + self._n_hmc_calls
        self._n_hmc_calls += 1
        fn = sample_nuts_device if (self.device_sampler and target.device_sampler_ok(n_samples + tune)) else sample_nuts
        return fn(target, n_samples, tune, seed=seed, start=start, step_scale=scale)

    def train_model(self, optimizer, max_steps=10000, hmc_scheduler=(200, 500, 1000, 1500), verbose=False,
                    num_tune_long=200, num_samples_long=50, num_tune_short=25, num_samples_short=10):
        from .core import few_host_threads
        return few_host_threads(self._train_model)(optimizer, max_steps, list(hmc_scheduler), verbose, num_tune_long,
                                                    num_samples_long, num_tune_short, num_samples_short)

    def _train_model(self, optimizer, max_steps, hmc_scheduler, verbose, num_tune_long, num_samples_long, num_tune_short,
                     num_samples_short):
        self.train()
        losses, trace_hyper, trace_step_size, trace_perf_time = [], None, [], []
        for n_iter in range(max_steps):
            optimizer.zero_grad()
            if n_iter < hmc_scheduler[0]:  # warm start: every raw parameter
                loss = self.neg_bound_per_datum()
                losses.append(loss.item())
                loss.backward()
                optimizer.step()
                continue
            self.freeze_kernel_hyperparameters()
            if trace_hyper is not None:  # the bound averaged over the current trace, differentiable in Z only
                loss = 0.0
                for i in range(len(trace_hyper)):
                    self.update_model_to_hyper(trace_hyper[i])
                    loss = loss + self.neg_bound_per_datum() / len(trace_hyper)
                if verbose:
                    print('Iter %d/%d - Loss: %.3f ' % (n_iter, max_steps, loss.item()))
                losses.append(loss.item())
                loss.backward()
                optimizer.step()
            if n_iter in hmc_scheduler:
                Z_opt = self.inducing_points.detach().cpu().numpy()
                long_phase = n_iter in (hmc_scheduler[0], hmc_scheduler[-1])
                trace_hyper = self.sample_optimal_variational_hyper_dist(num_samples_long if long_phase else num_samples_short, Z_opt,
                                                                         num_tune_long if long_phase else num_tune_short)
                trace_step_size.append(trace_hyper.get_sampler_stats('step_size')[0])
                trace_perf_time.append(trace_hyper.get_sampler_stats('perf_counter_diff').sum())
        return losses, trace_hyper, trace_step_size, trace_perf_time

    def train_fixed_model(self, num_tune=500, num_samples=100):
        """NUTS over the hyper-parameters with Z fixed at its current value (reference :257-277)."""
        trace = self.sample_optimal_variational_hyper_dist(num_samples, self.inducing_points.detach().cpu().numpy(), num_tune)
        return trace, [trace.get_sampler_stats('step_size')[0]], [trace.get_sampler_stats('perf_counter_diff').sum()]

    # ------------------------------------------------------------------ predictive
    def posterior_predictive(self, test_x, full_cov=False):
        """(mean, variance[, covariance]) of y* at test_x, observation noise included (reference :283-300)."""
        with torch.no_grad():
            mean, var, cov = self.bound.predict(test_x, self.inducing_points.detach(), self.current_kernel().block(), 1.0,
                                                float(self.noise()), pred_noise=True, full_cov=full_cov)
        return (mean, var, cov) if full_cov else (mean, var)

    def mixture_posterior_predictive(self, test_x, trace_hyper):
        """One (mean, variance) per draw of the trace (reference :302-340); the model is left at the last draw."""
        out = []
        for i in range(len(trace_hyper)):
            self.update_model_to_hyper(trace_hyper[i])
            out.append(self.posterior_predictive(test_x))
        return out
