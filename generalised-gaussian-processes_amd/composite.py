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

    def device_sampler_ok(self):
        """True when ``hmc.sample_nuts_device`` can run this target (single-launch path, at most 17 sampled parameters)."""
        b = self.bound
        return (hasattr(b, "_small_ok") and hasattr(b.engine, "small_nuts") and b._small_ok(self.Z.shape[0]) and self.ndim <= 18)

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
            return lp, [float(v) for v in h[1:1 + self.ndim]]
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
