"""The handful of GPyTorch names the reference's model classes touch, re-created so those classes keep
their shape.  GPyTorch is not installed on the target and is not forked here: these
are minimal parameter containers; every heavy method routes to ``CollapsedBound`` (the HIP library).

Imported by the reference as (models/sgpr.py:9-14, models/bayesian_sgpr_hmc.py:8-15):
    gpytorch.models.ExactGP, means.ZeroMean, kernels.{ScaleKernel, RBFKernel, InducingPointKernel},
    distributions.MultivariateNormal, likelihoods.GaussianLikelihood, mlls.ExactMarginalLogLikelihood

Parametrisation follows GPyTorch's defaults: lengthscale = softplus(raw), outputscale = softplus(raw),
noise = softplus(raw) + 1e-4, every raw parameter initialised at 0 (so lengthscale = outputscale = log 2).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .core import SgpTimeoutError

NOISE_FLOOR = 1e-4


def _inv_softplus(v: torch.Tensor) -> torch.Tensor:
    return v + torch.log(-torch.expm1(-v))


class ZeroMean(nn.Module):
    def forward(self, x):
        return torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)


class _PositiveParam(nn.Module):
    """raw parameter + softplus (+ floor); assignment of the constrained value writes the raw one."""

    def __init__(self, shape, floor=0.0):
        super().__init__()
        self.raw = nn.Parameter(torch.zeros(shape, dtype=torch.float64))
        self.floor = floor

    @property
    def value(self):
        return F.softplus(self.raw) + self.floor

    def set(self, v):
        v = torch.as_tensor(v, dtype=torch.float64, device=self.raw.device)
        v = v.expand(self.raw.shape) if v.numel() == 1 else v.reshape(self.raw.shape)  # scalars broadcast (ARD)
        with torch.no_grad():
            self.raw.copy_(_inv_softplus(torch.clamp(v - self.floor, min=1e-300)))


class RBFKernel(nn.Module):
    kernel_name = "rbf"

    def __init__(self, ard_num_dims: Optional[int] = None):
        super().__init__()
        self.ard_num_dims = ard_num_dims or 1
        self._ls = _PositiveParam((1, self.ard_num_dims))

    @property
    def raw_lengthscale(self):
        return self._ls.raw

    @property
    def lengthscale(self):
        return self._ls.value

    @lengthscale.setter
    def lengthscale(self, v):
        self._ls.set(v)


class MaternKernel(RBFKernel):
    def __init__(self, nu=2.5, ard_num_dims: Optional[int] = None):
        super().__init__(ard_num_dims)
        if nu not in (1.5, 2.5):
            raise ValueError("only nu = 1.5 and 2.5 have HIP kernels")
        self.kernel_name = "matern32" if nu == 1.5 else "matern52"


class ScaleKernel(nn.Module):
    def __init__(self, base_kernel):
        super().__init__()
        self.base_kernel = base_kernel
        self._os = _PositiveParam(())

    @property
    def raw_outputscale(self):
        return self._os.raw

    @property
    def outputscale(self):
        return self._os.value

    @outputscale.setter
    def outputscale(self, v):
        self._os.set(v)


class _NoiseCovar(nn.Module):
    def __init__(self):
        super().__init__()
        self._n = _PositiveParam((1,), floor=NOISE_FLOOR)

    @property
    def noise(self):
        return self._n.value

    @noise.setter
    def noise(self, v):
        self._n.set(v)


class GaussianLikelihood(nn.Module):
    name = "gaussian"

    def __init__(self):
        super().__init__()
        self.noise_covar = _NoiseCovar()

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, v):
        self.noise_covar.noise = v

    def forward(self, dist):
        return self(dist)

    def __call__(self, dist):
        """likelihood(f) adds the observation noise; a lazy predictive resolves itself with pred_noise=True."""
        if isinstance(dist, LazyPredictive):
            return dist.resolve(pred_noise=True)
        if isinstance(dist, TrainPrior):
            return dist
        n = dist.loc.shape[0]
        return MultivariateNormal(dist.loc, dist.covariance_matrix + self.noise.to(dist.loc.device) * torch.eye(n, dtype=dist.loc.dtype, device=dist.loc.device))


class BernoulliLikelihood(nn.Module):
    """Probit link, labels in {-1, +1} inside the bound ({0, 1} inputs are mapped); no noise parameter
    (the reference's classification scratch uses gpytorch.likelihoods.BernoulliLikelihood, scratch_pymc3.py:78-88)."""
    name = "bernoulli"

    def forward(self, dist):
        return self(dist)

    def __call__(self, dist):
        # predictive class-1 probability  Phi(mu / sqrt(1 + v))
        z = dist.loc / torch.sqrt(1.0 + dist.variance)
        return 0.5 * torch.erfc(-z * 0.7071067811865476)


class InducingPointKernel(nn.Module):
    def __init__(self, base_kernel, inducing_points, likelihood):
        super().__init__()
        self.base_kernel = base_kernel
        self.likelihood = likelihood
        Z = torch.as_tensor(inducing_points).detach().clone().to(torch.float64)
        if Z.dim() == 1:
            Z = Z[:, None]
        self.inducing_points = nn.Parameter(Z)


class MultivariateNormal:
    """loc / covariance container with the members the reference's metrics and plots use
    (utils/metrics.py:44,53-54; utils/visualisation.py:22,41)."""

    def __init__(self, mean, covariance_matrix, variance=None, engine=None):
        self.loc = mean
        self._cov = covariance_matrix
        self._var = variance
        self._engine = engine  # HipEngine when the covariance lives on the GPU: factorizations stay there

    @property
    def mean(self):
        return self.loc

    @property
    def covariance_matrix(self):
        return self._cov

    @property
    def variance(self):
        return self._var if self._var is not None else torch.diagonal(self._cov)

    @property
    def stddev(self):
        return torch.sqrt(self.variance)

    def confidence_region(self):
        s = 2.0 * self.stddev
        return self.loc - s, self.loc + s

    def _on_device(self):
        return self._engine is not None and self._cov is not None and self._cov.is_cuda and hasattr(self._engine, "chol_lower")

    def is_psd(self, jitter=1e-4):
        """The reference's gate on a predictive covariance, ``torch.linalg.cholesky(cov + 1e-4 I)`` succeeding
        (models/bayesian_sgpr_hmc.py:225-229).  On the GPU the T x T matrix is factored where it is (sgp_chol_lower) and
        only the 4-byte status word comes back."""
        if self._cov is None:
            return True
        if self._on_device():
            A = self._cov.detach().clone()
            A.diagonal().add_(jitter)
            _, info = self._engine.chol_lower(A)
            info = int(info.to("cpu")[0])
            if info < 0:  # SGP_INFO_TIMEOUT from the dataflow factorization: not a statement about the matrix
                raise SgpTimeoutError()
            return info == 0
        cov = self._cov.detach().to("cpu", torch.float64)
        try:
            torch.linalg.cholesky(cov + torch.eye(cov.shape[0], dtype=cov.dtype) * jitter)
            return True
        except RuntimeError:
            return False

    def log_prob(self, y):
        """Joint log-density (needs the T x T covariance).  On the GPU: sgp_chol_lower, sgp_trsm_lower, sgp_logdiag_sum on
        the device-resident covariance (one scalar comes back); otherwise host LAPACK."""
        T = self.loc.shape[0]
        if self._on_device():
            e = self._engine
            L, info = e.chol_lower(self._cov.detach())
            r = (y.detach().to(device=self._cov.device, dtype=torch.float64).reshape(-1) - self.loc.detach()).reshape(T, 1).contiguous()
            a = e.trsm_lower(L, r)
            ld = e.logdiag_sum(L)
            if int(info.to("cpu")[0]) < 0:
                raise SgpTimeoutError()
            if int(info.to("cpu")[0]) != 0:
                raise RuntimeError("predictive covariance is not positive definite (leading minor %d)" % int(info.to("cpu")[0]))
            return (-0.5 * (a * a).sum() - ld[0] - 0.5 * T * math.log(2.0 * math.pi)).to("cpu")
        cov = self._cov.detach().to("cpu", torch.float64)
        r = (y.detach().to("cpu", torch.float64).reshape(-1) - self.loc.detach().to("cpu", torch.float64))
        L = torch.linalg.cholesky(cov)
        a = torch.linalg.solve_triangular(L, r[:, None], upper=False)[:, 0]
        return -0.5 * (a @ a) - torch.log(torch.diagonal(L)).sum() - 0.5 * T * math.log(2.0 * math.pi)


class TrainPrior:
    """What ``model(train_x)`` returns in training mode: a handle the marginal log-likelihood evaluates."""

    def __init__(self, model):
        self.model = model


class LazyPredictive:
    """What ``model(test_x)`` returns in eval mode; ``likelihood(...)`` resolves it on the device."""

    def __init__(self, model, test_x):
        self.model = model
        self.test_x = test_x

    def resolve(self, pred_noise=True):
        return self.model._predict(self.test_x, pred_noise=pred_noise)


class _VFEBoundFn(torch.autograd.Function):
    """F(ls, sf2, s2, Z) / N with the gradient computed by the HIP library (pass 2 + adjoint tail)."""

    @staticmethod
    def forward(ctx, ls, sf2, s2, Z, model):
        cb = model._bound()
        need_grad = any(ctx.needs_input_grad[:4])
        lsv = ls.detach().reshape(-1).tolist()
        if need_grad:
            Fv, g = cb.value_and_grad(Z.detach(), lsv, float(sf2), float(s2), want_gz=ctx.needs_input_grad[3])
            ctx.g = g
        else:
            Fv, _ = cb.value(Z.detach(), lsv, float(sf2), float(s2))
            ctx.g = None
        ctx.N = cb.N
        ctx.shapes = (ls.shape, Z.shape)
        return torch.tensor(Fv / cb.N, dtype=torch.float64, device=ls.device)

    @staticmethod
    def backward(ctx, gout):
        g, N = ctx.g, ctx.N
        ls_shape, Z_shape = ctx.shapes
        s = gout / N
        g_ls = (g["ls"].to(gout.device).reshape(ls_shape) * s) if ctx.needs_input_grad[0] else None
        g_sf2 = (torch.as_tensor(g["sf2"], dtype=torch.float64, device=gout.device) * s) if ctx.needs_input_grad[1] else None
        g_s2 = (torch.as_tensor(g["s2"], dtype=torch.float64, device=gout.device) * s).reshape(1) if ctx.needs_input_grad[2] else None
        g_Z = (g["Z"].reshape(Z_shape) * s) if ctx.needs_input_grad[3] else None
        return g_ls, g_sf2, g_s2, g_Z, None


class ExactMarginalLogLikelihood(nn.Module):
    """``mll(model(train_x), train_y)`` -> collapsed bound / N (GPyTorch divides by the number of data)."""

    def __init__(self, likelihood, model):
        super().__init__()
        self.likelihood = likelihood
        self.model = model

    def forward(self, output, target=None):
        m = self.model
        ls = m.base_covar_module.base_kernel.lengthscale
        sf2 = m.base_covar_module.outputscale
        s2 = m.likelihood.noise
        Z = m.covar_module.inducing_points
        return _VFEBoundFn.apply(ls, sf2, s2, Z, m)


class ExactGP(nn.Module):
    def __init__(self, train_x, train_y, likelihood):
        super().__init__()
        self.likelihood = likelihood

    def named_hyperparameters(self):
        return self.named_parameters()


class settings:  # noqa: N801  (mirrors gpytorch.settings)
    class cholesky_jitter:  # noqa: N801
        """The reference writes ``gpytorch.settings.cholesky_jitter(float=1e-5)`` as a bare statement
        (experiments/regression.py:34) which has no effect; kept as an inert context manager."""

        def __init__(self, float=None, double=None, half=None):
            self.value = double if double is not None else float

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False
