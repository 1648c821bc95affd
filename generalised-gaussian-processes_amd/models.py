"""Model classes with the reference's signatures, running on the HIP core.

Mirrors (same names, argument meaning, return shapes and error behaviour):
  * ``SparseGPR``                      reference models/sgpr.py:22-160
  * ``BayesianSparseGPR_HMC``          reference models/bayesian_sgpr_hmc.py:26-196
  * ``mixture_posterior_predictive``   reference models/bayesian_sgpr_hmc.py:198-231

Differences that are deliberate and documented: the device is taken from the inputs
at construction (R16); ``model.inducing_points`` always tracks the optimised Z (R8); ``train_model``
accepts ``num_steps`` as an alias of ``max_steps`` (R5); the bound subtracts the trace term (R1).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from .core import CollapsedBound, HmcTarget, NotPositiveDefiniteError, SgpTimeoutError, few_host_threads
from .gp_shim import (ExactGP, ExactMarginalLogLikelihood, GaussianLikelihood, InducingPointKernel, LazyPredictive,
                      MultivariateNormal, RBFKernel, ScaleKernel, TrainPrior, ZeroMean)
from .hmc import Trace, sample_nuts, sample_nuts_device

FULL_COV_MAX_T = 4096  # predictive covariance is T x T; beyond this only mean / variance are formed


class SparseGPR(ExactGP):
    """The sparse GP class for regression with the collapsed bound; q*(u) is implicit (reference models/sgpr.py:22)."""

    def __init__(self, train_x, train_y, likelihood, Z_init, engine=None, jitter: float = 0.0, kernel=None):
        super().__init__(train_x, train_y, likelihood)
        if train_x.dim() == 1:
            train_x = train_x[:, None]
        self.train_x = train_x
        self.train_y = train_y
        self.num_inducing = len(Z_init)
        self.likelihood = likelihood
        self.mean_module = ZeroMean()
        base = kernel if kernel is not None else RBFKernel(ard_num_dims=self.train_x.shape[-1])
        self.base_covar_module = ScaleKernel(base)
        self.covar_module = InducingPointKernel(self.base_covar_module, inducing_points=Z_init, likelihood=self.likelihood)
        self.jitter = float(jitter)
        self._engine = engine
        self._cb: Optional[CollapsedBound] = None
        dev = engine.device if engine is not None else train_x.device
        self.to(dev)

    # reference attribute: aliases the optimised inducing inputs (not the initial tensor)
    @property
    def inducing_points(self):
        return self.covar_module.inducing_points.data

    def _bound(self) -> CollapsedBound:
        if self._cb is None:
            self._cb = CollapsedBound(self.train_x, self.train_y, kernel=self.base_covar_module.base_kernel.kernel_name,
                                      jitter=self.jitter, engine=self._engine)
        return self._cb

    def _hypers(self):
        ls = self.base_covar_module.base_kernel.lengthscale.detach().reshape(-1).tolist()
        return ls, float(self.base_covar_module.outputscale.detach()), float(self.likelihood.noise.detach())

    def forward(self, x):
        if self.training:
            return TrainPrior(self)
        return LazyPredictive(self, x)

    def _predict(self, test_x, pred_noise=True):
        if test_x.dim() == 1:
            test_x = test_x[:, None]
        ls, sf2, s2 = self._hypers()
        full = test_x.shape[0] <= FULL_COV_MAX_T
        mean, var, cov = self._bound().predict(test_x, self.covar_module.inducing_points, ls, sf2, s2,
                                               pred_noise=pred_noise, full_cov=full)
        eng = self._bound().engine
        return MultivariateNormal(mean, cov, variance=var, engine=eng if getattr(eng, "device", None) is not None and eng.device.type == "cuda" else None)

    @few_host_threads
    def train_model(self, optimizer, combine_terms=True, n_restarts=10, max_steps=10000, num_steps=None, verbose=True):
        """Full-batch optimisation of -ELBO/N (reference models/sgpr.py:110-144); one list entry per step."""
        if num_steps is not None:
            max_steps = num_steps
        self.train()
        self.likelihood.train()
        mll = ExactMarginalLogLikelihood(self.likelihood, self)
        losses = []
        for j in range(max_steps):
            optimizer.zero_grad()
            output = self.forward(self.train_x)
            loss = -mll(output, self.train_y).sum()
            losses.append(loss.item())
            loss.backward()
            if verbose and j % 1000 == 0:
                print('Iter %d/%d - Loss: %.3f   outputscale: %.3f  lengthscale: %s   noise: %.3f ' % (
                    j + 1, max_steps, loss.item(), self.base_covar_module.outputscale.item(),
                    self.base_covar_module.base_kernel.lengthscale, self.likelihood.noise.item()))
            optimizer.step()
        return losses

    def optimal_q_u(self):
        was = self.training
        self.eval()
        out = self._predict(self.covar_module.inducing_points.detach(), pred_noise=False)
        self.train(was)
        return out

    def posterior_predictive(self, test_x):
        """Returns the posterior predictive multivariate normal (observation noise included)."""
        self.eval()
        self.likelihood.eval()
        with torch.no_grad():
            y_star = self.likelihood(self(test_x))
        return y_star


class _ThetaAveragedLossFn(torch.autograd.Function):
    """mean_s( -F(theta_s, Z) / N ) over the samples of a trace, differentiable in Z only -- the loss of the reference's
    alternating schedule once the kernel hyper-parameters are frozen (models/bayesian_sgpr_hmc.py:117-134).  All S bounds
    and their dF/dZ come from ONE launch (sgp_small_eval_batch)."""

    @staticmethod
    def forward(ctx, Z, model, thetas):
        cb = model._bound()
        e = cb.engine
        Zd = cb._prep_Z(Z)
        outs, gz, infos = e.small_eval_batch(cb.X, cb.y, Zd, thetas, cb.jitter, cb.kernel, mode=0, want_grad=True, want_gz=True)
        info = infos.to("cpu")
        if int(info.min()) < 0:
            e.small_reset()
            raise SgpTimeoutError()
        if int(info.max()) != 0:
            raise NotPositiveDefiniteError(int(info[info != 0][0]))
        n = float(cb.N)
        ctx.gz = -(gz.mean(0)) / n
        ctx.zshape, ctx.zdev = Z.shape, Z.device
        cb.n_evals += thetas.shape[0]
        cb.n_grads += thetas.shape[0]
        return (-(outs[:, 0].mean()) / n).to(Z.device)

    @staticmethod
    def backward(ctx, gout):
        return (ctx.gz.to(ctx.zdev) * gout).reshape(ctx.zshape), None, None


class BayesianSparseGPR_HMC(SparseGPR):  # noqa: N801  (reference class name)
    """Doubly collapsed SGPR: q(u) implicit, theta sampled with NUTS at scheduled iterations
    (reference models/bayesian_sgpr_hmc.py:26)."""

    def __init__(self, train_x, train_y, likelihood, Z_init, engine=None, jitter: float = 0.0, seed: Optional[int] = None):
        super().__init__(train_x, train_y, likelihood, Z_init, engine=engine, jitter=jitter)
        self.data_dim = self.train_x.shape[1]
        self._hmc_cb: Optional[CollapsedBound] = None
        self._seed = seed
        self._n_hmc_calls = 0
        self.device_sampler = True  # NUTS on the device when the problem takes the single-launch path (M <= 128)
        self.hmc_gradient = "parity"  # or "sampler": see core.HmcTarget (large shards in the streaming-order guard's regime)
        self.batched_theta_loss = True  # ... and the theta-averaged loss of the alternating schedule in one launch

    def freeze_kernel_hyperparameters(self):
        for name, parameter in self.named_hyperparameters():
            if name != 'covar_module.inducing_points':
                parameter.requires_grad = False

    def _hmc_bound(self):
        # PyMC3's MarginalSparse always stabilises Kuu with 1e-6 I (reference models/bayesian_sgpr_hmc.py:66)
        if self._hmc_cb is None:
            self._hmc_cb = CollapsedBound(self.train_x, self.train_y, kernel="rbf", jitter=1e-6, engine=self._bound().engine)
        return self._hmc_cb

    def sample_optimal_variational_hyper_dist(self, n_samples, input_dim, Z_opt, tune, sampler_params=None) -> Trace:
        """NUTS over (ls, sig_f, sig_n) with Z fixed at Z_opt (reference models/bayesian_sgpr_hmc.py:58-80)."""
        Z = torch.as_tensor(np.asarray(Z_opt), dtype=torch.float64)
        target = HmcTarget(self._hmc_bound(), Z, gradient=self.hmc_gradient)
        scale = 0.25 if not sampler_params else sampler_params.get('step_scale', 0.25)
        seed = None if self._seed is None else self._seed + self._n_hmc_calls
        self._n_hmc_calls += 1
        if self.device_sampler and target.device_sampler_ok(n_samples + tune):
            # the whole of pm.sample() in one persistent launch: theta, momentum and the tree never leave the GPU
            return sample_nuts_device(target, n_samples, tune, seed=seed, step_scale=scale)
        return sample_nuts(target, n_samples, tune, seed=seed, step_scale=scale)

    def update_model_to_hyper(self, elbo, hyper_sample):
        elbo.likelihood.noise_covar.noise = hyper_sample['sig_n'] ** 2
        elbo.model.base_covar_module.outputscale = hyper_sample['sig_f'] ** 2
        elbo.model.base_covar_module.base_kernel.lengthscale = hyper_sample['ls']

    @few_host_threads
    def train_model(self, optimizer, max_steps=10000, hmc_scheduler=(200, 500, 1000, 1500), verbose=True,
                    num_tune_long=100, num_samples_long=20, num_tune_short=25, num_samples_short=10):
        """Alternates Adam on Z with NUTS on theta (reference models/bayesian_sgpr_hmc.py:88-158).
        Returns (losses, trace_hyper, trace_step_size, trace_perf_time)."""
        hmc_scheduler = list(hmc_scheduler)
        self.train()
        self.likelihood.train()
        elbo = ExactMarginalLogLikelihood(self.likelihood, self)
        losses, trace_hyper, trace_step_size, trace_perf_time = [], None, [], []
        thetas_dev = None  # the current trace's hyper-parameters on the device (rebuilt after every NUTS phase)
        for n_iter in range(max_steps):
            optimizer.zero_grad()
            if n_iter < hmc_scheduler[0]:  # warm start: plain SGPR optimisation
                output = self(self.train_x)
                loss = -elbo(output, self.train_y)
                losses.append(loss.item())
                loss.backward()
                optimizer.step()
            else:
                self.freeze_kernel_hyperparameters()
                if trace_hyper is not None:  # stochastic ELBO: average over the current theta samples
                    Zp = self.covar_module.inducing_points
                    if self.batched_theta_loss and self._bound()._small_ok(Zp.shape[0]):
                        # all samples in ONE launch; the model is left at the last sample's hypers like the loop below
                        cb = self._bound()
                        if thetas_dev is None:
                            rows = [list(np.asarray(h['ls'], dtype=np.float64).reshape(-1)) + [float(h['sig_f']) ** 2, float(h['sig_n']) ** 2]
                                    for h in trace_hyper]
                            thetas_dev = torch.tensor(rows, dtype=torch.float64).to(cb.engine.device)
                        loss = _ThetaAveragedLossFn.apply(Zp, self, thetas_dev)
                        self.update_model_to_hyper(elbo, trace_hyper[len(trace_hyper) - 1])
                    else:
                        loss = 0.0
                        for i in range(len(trace_hyper)):
                            self.update_model_to_hyper(elbo, trace_hyper[i])
                            output = self(self.train_x)
                            loss = loss + (-elbo(output, self.train_y).sum() / len(trace_hyper))
                    if verbose:
                        print('Iter %d/%d - Loss: %.3f ' % (n_iter, max_steps, loss.item()))
                    losses.append(loss.item())
                    loss.backward()
                    optimizer.step()
                if n_iter in hmc_scheduler:
                    Z_opt = self.inducing_points.detach().cpu().numpy()
                    if n_iter in (hmc_scheduler[0], hmc_scheduler[-1]):
                        num_tune, num_samples = num_tune_long, num_samples_long
                    else:
                        num_tune, num_samples = num_tune_short, num_samples_short
                    trace_hyper = self.sample_optimal_variational_hyper_dist(num_samples, self.data_dim, Z_opt, num_tune,
                                                                             sampler_params=None)
                    thetas_dev = None
                    trace_step_size.append(trace_hyper.get_sampler_stats('step_size')[0])
                    trace_perf_time.append(trace_hyper.get_sampler_stats('perf_counter_diff').sum())
        return losses, trace_hyper, trace_step_size, trace_perf_time

    def train_fixed_model(self, num_tune=500, num_samples=500):
        """HMC over theta with Z fixed at its initial value (reference models/bayesian_sgpr_hmc.py:160-180)."""
        self.train()
        self.likelihood.train()
        Z_opt = self.inducing_points.detach().cpu().numpy()
        trace_hyper = self.sample_optimal_variational_hyper_dist(num_samples, self.data_dim, Z_opt, num_tune, None)
        return (trace_hyper, [trace_hyper.get_sampler_stats('step_size')[0]],
                [trace_hyper.get_sampler_stats('perf_counter_diff').sum()])


def _mixture_batched(model, test_x, trace_hyper):
    """The same list of predictives from sgp_mixture_predict: eight theta samples per chain of launches, the PSD gate
    cholesky(cov + 1e-4 I) of all of them in one dataflow launch, ONE status copy for the whole trace.
    Returns None when the batched call does not apply (test double, several ranks, composite kernels, T beyond the full-
    covariance limit, ``model.batched_mixture = False``): the caller then runs the reference's per-sample loop."""
    b = model._bound()
    e = b.engine
    if (not getattr(model, "batched_mixture", True) or not hasattr(e, "mixture_predict") or b.world != 1 or b.kernel == "composite"
            or len(trace_hyper) == 0):
        return None
    tx = test_x[:, None] if test_x.dim() == 1 else test_x
    if tx.shape[0] > FULL_COV_MAX_T or tx.shape[0] > 8192:
        return None
    S = len(trace_hyper)
    ls = [np.asarray(trace_hyper[i]['ls'], dtype=np.float64).reshape(-1) for i in range(S)]
    d = b.d
    ls = [np.repeat(v, d) if v.size == 1 and d > 1 else v for v in ls]
    sf2 = [float(trace_hyper[i]['sig_f']) ** 2 for i in range(S)]
    s2 = [float(trace_hyper[i]['sig_n']) ** 2 for i in range(S)]
    Xs = tx.detach().to(dtype=torch.float64, device=e.device).contiguous()
    with torch.no_grad():
        r = e.mixture_predict(b.X, b.y, Xs, b._prep_Z(model.covar_module.inducing_points), ls, sf2, s2, jitter=b.jitter, kernel=b.kernel,
                              pred_noise=True, full_cov=True, gate_jitter=1e-4)
        status = torch.stack([r["info"], r["gate"]]).to("cpu")  # the one host round trip of the whole mixture
    # the loop leaves the model at the last sample's hyper-parameters (models/bayesian_sgpr_hmc.py:206-208): so does this
    last = trace_hyper[S - 1]
    model.likelihood.noise_covar.noise = last['sig_n'] ** 2
    model.base_covar_module.outputscale = last['sig_f'] ** 2
    model.base_covar_module.base_kernel.lengthscale = last['ls']
    model.eval()
    model.likelihood.eval()
    preds = []
    for i in range(S):
        info, gate = int(status[0, i]), int(status[1, i])
        if info < 0 or gate < 0:
            raise SgpTimeoutError()
        if info != 0 or gate != 0:
            print('Not psd for sample ' + str(i))
            continue
        preds.append(MultivariateNormal(r["mean"][i], r["cov"][i], variance=r["var"][i], engine=e))
    return preds


def mixture_posterior_predictive(model, test_x, trace_hyper):
    """One predictive per theta sample; samples whose predictive covariance fails the reference's PSD gate
    (cholesky(cov + 1e-4 I)) are skipped, never raised (reference models/bayesian_sgpr_hmc.py:198-231)."""
    batched = _mixture_batched(model, test_x, trace_hyper)
    if batched is not None:
        return batched
    preds = []
    for i in range(len(trace_hyper)):
        hyper_sample = trace_hyper[i]
        model.train()
        model.likelihood.train()
        model.likelihood.noise_covar.noise = hyper_sample['sig_n'] ** 2
        model.base_covar_module.outputscale = hyper_sample['sig_f'] ** 2
        model.base_covar_module.base_kernel.lengthscale = hyper_sample['ls']
        with torch.no_grad():
            model.eval()
            model.likelihood.eval()
            try:
                pred = model.likelihood(model(test_x))
                # the reference's PSD gate, cholesky(cov + 1e-4 I): on the device the T x T covariance is factored where
                # it is and only the status word crosses PCIe (no per-sample T x T copy)
                if not pred.is_psd(1e-4):
                    raise RuntimeError("predictive covariance not positive definite")
                preds.append(pred)
            except SgpTimeoutError:
                raise  # a device scheduling fault is never a numerical outcome: it must not thin the mixture silently
            except (RuntimeError, NotPositiveDefiniteError):
                print('Not psd for sample ' + str(i))
    return preds


# ---------------------------------------------------------------------------------------------
# SVGP
# ---------------------------------------------------------------------------------------------
def _raise_on_info(info: int):
    if info < 0:
        raise SgpTimeoutError()
    if info != 0:
        raise NotPositiveDefiniteError(info)


class _SVGPBoundFn(torch.autograd.Function):
    """ELBO per datum of one minibatch; forward and the whole reverse pass run in sgp_svgp_elbo."""

    @staticmethod
    def forward(ctx, ls, sf2, s2, Z, m, LS, model, xb, yb):
        eng = model._engine_obj()
        need = any(ctx.needs_input_grad[:6])
        res = eng.svgp_elbo(xb, yb, Z.detach().contiguous(), ls.detach().reshape(-1).tolist(), float(sf2), float(s2),
                            m.detach().contiguous(), LS.detach().contiguous(), model.num_data, jitter=model.jitter,
                            kernel=model.covar_module.base_kernel.kernel_name, likelihood=model.likelihood.name, with_grads=need)
        pending = getattr(model, "_pending_infos", None)
        if pending is not None:  # the caller checks the status words of several bounds with ONE host round trip
            pending.append(res["info"])
        else:
            _raise_on_info(int(res["info"].to("cpu").item()))
        ctx.res = res if need else None
        ctx.meta = [(t.shape, t.device) for t in (ls, sf2, s2, Z, m, LS)]
        return res["out"][0].clone()

    @staticmethod
    def backward(ctx, gout):
        r = ctx.res
        n = ctx.needs_input_grad
        keys = ("g_ls", "g_sf2", "g_s2", "g_Z", "g_m", "g_LS")
        outs = []
        for i, k in enumerate(keys):  # hyper-parameters may live on the host (Bayesian variant), Z / m / LS on the device
            if not n[i]:
                outs.append(None)
                continue
            shape, dev = ctx.meta[i]
            outs.append((r[k] * gout).to(dev).reshape(shape))
        return (*outs, None, None, None)


class _SVGPBatchBoundFn(torch.autograd.Function):
    """ELBO per datum of one minibatch at S hyper-parameter samples: ONE sgp_svgp_elbo_batch call evaluates all S bounds and
    their whole reverse pass (~36 launches whatever S is).  theta: S x (d + 2) host tensor [sf2 | ls_1..d | s2] (s2 column
    ignored by the Bernoulli likelihood).  Returns the S bounds as a HOST tensor -- the one device-to-host copy of the forward
    pass brings them together with the S status words (``model._last_infos``) -- so the incoming gradient is a host tensor too,
    and backward is one launch (sgp_svgp_batch_combine) plus one copy of the S x (d + 2) hyper-parameter gradients."""

    @staticmethod
    def forward(ctx, theta, Z, m, LS, model, xb, yb):
        eng = model._engine_obj()
        need = any(ctx.needs_input_grad[:4])
        th = theta.detach().to("cpu", torch.float64)
        d = Z.shape[1]
        asyn = bool(getattr(model, "_async_bounds", False)) and hasattr(eng, "lib")
        kw = {"defer_reverse": True} if (asyn and need) else {}
        res = eng.svgp_elbo_batch(xb, yb, Z.detach().contiguous(), th[:, 1:1 + d].tolist(), th[:, 0].tolist(), th[:, 1 + d].tolist(),
                                  m.detach().contiguous(), LS.detach().contiguous(), model.num_data, jitter=model.jitter,
                                  kernel=model.covar_module.base_kernel.kernel_name, likelihood=model.likelihood.name, with_grads=need, **kw)
        ctx.res = res if need else None
        ctx.eng = eng
        ctx.theta_meta = (theta.shape, theta.device, theta.dtype)
        if asyn and res["out"].is_cuda:
            # the training loop's variant: the forward half of the chain, then the copy of the bounds into pinned memory and an
            # event, then the reverse half.  The caller waits for the EVENT (model._wait_bounds()) after it has formed the KL term,
            # and runs its autograd bookkeeping while the device is still busy with the gradients.
            pin = getattr(model, "_pinned_out", None)
            if pin is None or pin.shape != res["out"].shape:
                pin = model._pinned_out = torch.empty(res["out"].shape, dtype=torch.float64, pin_memory=True)
                model._bounds_event = torch.cuda.Event()
            pin.copy_(res["out"], non_blocking=True)
            model._bounds_event.record(torch.cuda.current_stream(res["out"].device))
            if "reverse" in res:
                res.pop("reverse")()
            model._pending_out = pin
            return pin[:, 0]  # valid once the event has completed; NOT read before (see train_model)
        host = res["out"].to("cpu")
        model._last_infos = host[:, 3].to(torch.int32)
        for v in model._last_infos.tolist():  # synchronous use: a failed factorization raises here, as sgp_svgp_elbo's wrapper does
            _raise_on_info(int(v))
        return host[:, 0].clone()

    @staticmethod
    def backward(ctx, gout):
        r = ctx.res
        n = ctx.needs_input_grad
        if hasattr(ctx.eng, "svgp_batch_combine"):
            gm, gLS, gZ, gth = ctx.eng.svgp_batch_combine(r, gout.detach().reshape(-1).tolist())
        else:  # test double: the same sums in torch
            g = gout.to(r["g_Z"].device, torch.float64)
            gZ, gm, gLS = torch.einsum("s,smd->md", g, r["g_Z"]), torch.einsum("s,sm->m", g, r["g_m"]), torch.einsum("s,smk->mk", g, r["g_LS"])
            gth = torch.cat([r["g_sf2"][:, None], r["g_ls"], r["g_s2"][:, None]], 1) * g[:, None]
        g_theta = None
        if n[0]:
            shape, dev, dt = ctx.theta_meta
            g_theta = gth.to(dev, dt).reshape(shape)
        return g_theta, gZ if n[1] else None, gm if n[2] else None, gLS if n[3] else None, None, None, None


class StochasticVariationalGP(torch.nn.Module):
    """The sparse GP class with the uncollapsed stochastic bound; q(u) = N(m, S) is learnt numerically
    (reference models/svgp.py:24-141): whitened variational strategy, Cholesky variational distribution
    (m = 0, L_S = I at start), learnable inducing locations, ScaleKernel(RBFKernel(ard)).  Minibatches run
    through ``sgp_svgp_elbo`` on the device."""

    def __init__(self, train_x, train_y, likelihood, Z_init, num_tasks=None, engine=None, jitter: float = 1e-6):
        super().__init__()
        if train_x.dim() == 1:
            train_x = train_x[:, None]
        self.train_x, self.train_y = train_x, train_y
        self.likelihood = likelihood
        self.num_inducing = len(Z_init)
        self.num_data = int(train_y.shape[0])
        self.jitter = float(jitter)
        self.mean_module = ZeroMean()
        self.covar_module = ScaleKernel(RBFKernel(ard_num_dims=train_x.shape[-1]))
        Z = torch.as_tensor(Z_init).detach().clone().to(torch.float64)
        self.inducing_inputs = torch.nn.Parameter(Z[:, None] if Z.dim() == 1 else Z)
        self.variational_mean = torch.nn.Parameter(torch.zeros(self.num_inducing, dtype=torch.float64))
        self.chol_variational_covar = torch.nn.Parameter(torch.eye(self.num_inducing, dtype=torch.float64))
        self._engine = engine
        self.to(engine.device if engine is not None else train_x.device)
        # The d + 2 kernel / likelihood hyper-parameters stay on the HOST (the inducing inputs and q(u) live on the device): the
        # library takes them as kernel arguments, and reading d + 2 scalars from device tensors cost three blocking copies per
        # minibatch step (the step was host-bound: 0.95 ms around 0.40 ms of device work)
        self.covar_module.to("cpu")
        self.likelihood.to("cpu")
        self.batched = True  # bounds through sgp_svgp_elbo_batch (S = 1 here): asynchronous forward, one-launch reverse combine

    def _engine_obj(self):
        if self._engine is None:
            from .engine import HipEngine
            self._engine = HipEngine(self.train_x.device if self.train_x.is_cuda else None)
        return self._engine

    def _theta_row(self):
        """[outputscale | lengthscales | noise variance] as a 1 x (d + 2) host tensor (differentiable wrt the raw parameters)."""
        sf2 = self.covar_module.outputscale.reshape(1)
        ls = self.covar_module.base_kernel.lengthscale.reshape(-1)
        if getattr(self.likelihood, "name", "gaussian") == "bernoulli":
            s2 = torch.ones(1, dtype=torch.float64, device=sf2.device)
        else:
            s2 = self.likelihood.noise.reshape(1).to(sf2.device)
        return torch.cat([sf2, ls, s2])[None, :]

    def _wait_bounds(self):
        """Completes an asynchronous bound evaluation (training loops only): waits for the event recorded behind the forward
        half of the chain, returns the status words."""
        pin = getattr(self, "_pending_out", None)
        self._pending_out = None
        if pin is None:
            return self._last_infos
        self._bounds_event.synchronize()
        return pin[:, 3].to(torch.int32)

    def _dev(self, t):
        return t.detach().to(dtype=torch.float64, device=self._engine_obj().device).contiguous()

    def elbo_minibatch(self, x_batch, y_batch):
        """mean_b E_q log p(y_b | f_b) - KL / N  (GPyTorch ``VariationalELBO`` convention), differentiable."""
        if x_batch.dim() == 1:
            x_batch = x_batch[:, None]
        yb = self._dev(y_batch).reshape(-1)
        if getattr(self.likelihood, "name", "gaussian") == "bernoulli":
            yb = torch.where(yb > 0, torch.ones_like(yb), -torch.ones_like(yb))
        if getattr(self, "batched", True) and hasattr(self._engine_obj(), "svgp_elbo_batch"):
            # the batched chain with one sample: a host tensor comes back (the bound and its status word in one copy; inside a
            # training loop asynchronously, see train_model)
            return _SVGPBatchBoundFn.apply(self._theta_row(), self.inducing_inputs, self.variational_mean, self.chol_variational_covar,
                                           self, self._dev(x_batch), yb)[0]
        if getattr(self.likelihood, "name", "gaussian") == "bernoulli":
            s2 = torch.ones(1, dtype=torch.float64, device=yb.device)
        else:
            s2 = self.likelihood.noise
        return _SVGPBoundFn.apply(self.covar_module.base_kernel.lengthscale, self.covar_module.outputscale, s2,
                                  self.inducing_inputs, self.variational_mean, self.chol_variational_covar, self,
                                  self._dev(x_batch), yb)

    @few_host_threads
    def train_model(self, optimizer, train_loader, minibatch_size=100, num_epochs=25, combine_terms=True, verbose=False):
        """Minibatch Adam on -ELBO (reference models/svgp.py:88-127); one loss entry per minibatch."""
        losses = []
        for i in range(num_epochs):
            for x_batch, y_batch in train_loader:
                self.train()
                self.likelihood.train()
                optimizer.zero_grad()
                asyn = getattr(self, "batched", True) and hasattr(self._engine_obj(), "svgp_elbo_batch")
                self._async_bounds = asyn  # forward half, copy of the bound + event, reverse half: all enqueued, nothing waited for
                try:
                    elbo = self.elbo_minibatch(x_batch, y_batch)  # asynchronous: a view of pinned memory, NOT read before the wait
                finally:
                    self._async_bounds = False
                if asyn:
                    for v in self._wait_bounds().tolist():
                        _raise_on_info(int(v))
                loss = -elbo.sum()
                losses.append(loss.item())
                loss.backward()
                optimizer.step()
            if verbose:
                print("Epoch %d  loss %.4f" % (i, losses[-1]))
        return losses

    def latent_predictive(self, test_x):
        if test_x.dim() == 1:
            test_x = test_x[:, None]
        ls = self.covar_module.base_kernel.lengthscale.detach().reshape(-1).tolist()
        mean, var, info = self._engine_obj().svgp_predict(self._dev(test_x), self._dev(self.inducing_inputs), ls,
                                                          float(self.covar_module.outputscale.detach()),
                                                          self._dev(self.variational_mean), self._dev(self.chol_variational_covar),
                                                          jitter=self.jitter, kernel=self.covar_module.base_kernel.kernel_name)
        if int(info.to("cpu").item()) != 0:
            raise NotPositiveDefiniteError(int(info.to("cpu").item()))
        return mean, var

    def posterior_predictive(self, test_x):
        """Predictive through the likelihood (reference models/svgp.py:132-141): a diagonal Gaussian with the
        noise added, or class-1 probabilities for the Bernoulli likelihood."""
        self.eval()
        self.likelihood.eval()
        with torch.no_grad():
            mean, var = self.latent_predictive(test_x)
            if getattr(self.likelihood, "name", "gaussian") == "bernoulli":
                return self.likelihood(MultivariateNormal(mean, None, variance=var))
            return MultivariateNormal(mean, None, variance=var + self.likelihood.noise.detach().to(var.device))


class VariationalHyperDist(torch.nn.Module):
    """q(log theta) = N(q_mu, L L^T + 1e-5 I) with the Cholesky factor stored as a vector
    (reference models/bayesian_svgp.py:30-71).  Host-side: hyper_dim = d + 2."""

    def __init__(self, hyper_dim, prior_var=0.01, n=1, seed=None):
        super().__init__()
        self.hyper_dim, self.n, self.prior_var = hyper_dim, n, prior_var
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        self.q_mu = torch.nn.Parameter(torch.randn(hyper_dim, dtype=torch.float64, generator=g) * 1e-3)
        self.q_sigma_vec = torch.nn.Parameter(torch.randn(hyper_dim * (hyper_dim + 1) // 2, dtype=torch.float64, generator=g) * 1e-3)
        self._gen = g

    def construct_sigma(self):
        rows, cols = torch.tril_indices(self.hyper_dim, self.hyper_dim)
        lower = torch.zeros(self.hyper_dim, self.hyper_dim, dtype=torch.float64)
        lower = lower.index_put((rows, cols), self.q_sigma_vec)   # the vector also fills the diagonal, as upstream
        return lower @ lower.T + 1e-5 * torch.eye(self.hyper_dim, dtype=torch.float64)

    def kl_per_point(self):
        """KL(q(log theta) || N(0, prior_var I)) / n  -- the reference's added loss term (:73-84)."""
        S = self.construct_sigma()
        k = self.hyper_dim
        kl = 0.5 * (torch.trace(S) / self.prior_var + (self.q_mu @ self.q_mu) / self.prior_var - k
                    + k * math.log(self.prior_var) - torch.logdet(S))
        return kl / self.n

    def forward(self, num_samples, chol=None):
        L = torch.linalg.cholesky(self.construct_sigma()) if chol is None else chol
        eps = torch.randn(num_samples, self.hyper_dim, dtype=torch.float64, generator=self._gen)
        return self.q_mu[None, :] + eps @ L.T

    def draws(self, count):
        """``count`` reparametrised samples drawn ONE AT A TIME from the random stream, as the reference's training loop does
        (models/bayesian_svgp.py:159-160), with Sigma and its Cholesky factor formed once instead of once per draw (the
        parameters do not change inside a minibatch step; ~40 host operations fewer).  Returns count x hyper_dim."""
        L = torch.linalg.cholesky(self.construct_sigma())
        eps = torch.cat([torch.randn(1, self.hyper_dim, dtype=torch.float64, generator=self._gen) for _ in range(count)])
        return self.q_mu[None, :] + eps @ L.T


class BayesianStochasticVariationalGP(StochasticVariationalGP):
    """SVGP with a variational distribution over log-hyperparameters, 5 reparametrised theta samples per
    minibatch (reference models/bayesian_svgp.py:87-181): theta = exp(log theta), outputscale = theta_0,
    lengthscale = theta_1..d, noise = theta_{d+1}^2.  The five bounds of a minibatch are ONE sgp_svgp_elbo_batch call.
    Bernoulli-probit likelihood (BASELINE config C4; the reference's classification use is scratch_pymc3.py:78-88): labels
    are mapped to {-1, +1} and there is no noise hyper-parameter, so q(log theta) has d + 1 dimensions.
    Deliberate deviations: the reparametrisation gradient reaches q(log theta) (upstream
    assigns theta through GPyTorch setters, which detaches it), and prediction uses exp() like training."""

    def __init__(self, train_x, train_y, likelihood, Z_init, engine=None, jitter: float = 1e-6, seed=None, num_hyper_samples=5):
        super().__init__(train_x, train_y, likelihood, Z_init, engine=engine, jitter=jitter)
        self.n = self.num_data
        self.input_dim = self.train_x.shape[1]
        self.num_hyper_samples = num_hyper_samples
        self.bernoulli = getattr(self.likelihood, "name", "gaussian") == "bernoulli"
        self.hyper_dim = self.input_dim + (1 if self.bernoulli else 2)
        self.log_theta = VariationalHyperDist(self.hyper_dim, prior_var=0.01, n=self.n, seed=seed)
        self.log_theta.to("cpu")
        self.batched = True  # one launch chain for all hyper-samples; False = one sgp_svgp_elbo chain per sample (A/B, tests)

    def sample_variational_log_hyper(self, num_samples):
        return self.log_theta(num_samples)

    def _labels(self, y_batch):
        yb = self._dev(y_batch).reshape(-1)
        if self.bernoulli:  # {0, 1} or {-1, +1} labels -> {-1, +1}, as elbo_minibatch does
            yb = torch.where(yb > 0, torch.ones_like(yb), -torch.ones_like(yb))
        return yb

    def _theta_of(self, log_theta):
        """[outputscale | lengthscales | noise variance] of one or several log-theta samples (rows); the Bernoulli likelihood
        has no noise: its column is the constant 1 the kernel ignores."""
        theta = torch.exp(log_theta)
        if self.bernoulli:
            return torch.cat([theta, torch.ones_like(theta[..., :1])], -1)
        return torch.cat([theta[..., :-1], theta[..., -1:] ** 2], -1)

    def _elbo_at(self, x_batch, y_batch, log_theta):
        th = self._theta_of(log_theta)
        if x_batch.dim() == 1:
            x_batch = x_batch[:, None]
        s2 = torch.ones(1, dtype=torch.float64, device=self._engine_obj().device) if self.bernoulli else th[-1]
        return _SVGPBoundFn.apply(th[1:-1], th[0], s2, self.inducing_inputs, self.variational_mean,
                                  self.chol_variational_covar, self, self._dev(x_batch), self._labels(y_batch))

    def elbo_hyper_samples(self, x_batch, y_batch, log_thetas):
        """The S bounds (host tensor, differentiable wrt q(u), Z and log_thetas) of one minibatch at the rows of log_thetas."""
        if x_batch.dim() == 1:
            x_batch = x_batch[:, None]
        return _SVGPBatchBoundFn.apply(self._theta_of(log_thetas), self.inducing_inputs, self.variational_mean,
                                       self.chol_variational_covar, self, self._dev(x_batch), self._labels(y_batch))

    @few_host_threads
    def train_model(self, optimizer, train_loader, minibatch_size=100, num_epochs=25, combine_terms=True):
        """Returns (epoch_losses, batch_losses) like the reference (:144-181)."""
        self.train()
        self.likelihood.train()
        epoch_losses, batch_losses = [], []
        S = self.num_hyper_samples
        for i in range(num_epochs):
            batch_losses = []
            for x_batch, y_batch in train_loader:
                optimizer.zero_grad()
                xb = self._dev(x_batch[:, None] if x_batch.dim() == 1 else x_batch)  # one host-to-device copy per minibatch
                # one reparametrised draw at a time, as the reference consumes its random stream (:159-160)
                lts = self.log_theta.draws(S)
                if self.batched and hasattr(self._engine_obj(), "svgp_elbo_batch"):
                    # the chain is enqueued first; the KL term of q(log theta) (a dozen host operations) is formed while the
                    # device works; then ONE wait brings the S bounds and their status words
                    self._async_bounds = True
                    try:
                        host = self.elbo_hyper_samples(xb, y_batch, lts)
                    finally:
                        self._async_bounds = False
                    kl = self.log_theta.kl_per_point()
                    infos = self._wait_bounds()
                else:
                    kl = self.log_theta.kl_per_point()
                    # one launch chain per sample, enqueued back to back; status words read together
                    self._pending_infos = []
                    try:
                        es = [self._elbo_at(xb, y_batch, lts[k]) for k in range(S)]
                        infos = torch.cat(self._pending_infos).to("cpu") if self._pending_infos else torch.zeros(1, dtype=torch.int32)
                    finally:
                        self._pending_infos = None
                    host = torch.stack(es).to("cpu")
                for v in infos.tolist():
                    _raise_on_info(int(v))
                loss = kl - host.sum() / S  # = sum_k (-elbo_k + kl) / S, the reference's accumulation (:161-166)
                batch_losses.append(loss.item())
                loss.backward()
                optimizer.step()
            epoch_losses.append(float(np.sum(batch_losses)))
        return epoch_losses, batch_losses

    def mixture_posterior_predictive(self, test_x, num_samples=100):
        """One predictive per hyper-sample from q(log theta) (reference :183-207)."""
        self.eval()
        self.likelihood.eval()
        out = []
        eng = self._engine_obj()
        if getattr(self, "batched", True) and hasattr(eng, "svgp_predict_batch"):
            # every hyper-sample's q(f*) from ONE chain of launches per eight samples, one status copy for all of them
            with torch.no_grad():
                th = self._theta_of(self.sample_variational_log_hyper(num_samples))
                tx = self._dev(test_x[:, None] if test_x.dim() == 1 else test_x)
                mean, var, info = eng.svgp_predict_batch(tx, self._dev(self.inducing_inputs), th[:, 1:-1].tolist(), th[:, 0].tolist(),
                                                         self._dev(self.variational_mean), self._dev(self.chol_variational_covar),
                                                         jitter=self.jitter, kernel=self.covar_module.base_kernel.kernel_name)
                ok = info.to("cpu").tolist()
                for k in range(num_samples):
                    if ok[k] < 0:
                        raise SgpTimeoutError()
                    if ok[k] != 0:
                        continue
                    if self.bernoulli:
                        out.append(self.likelihood(MultivariateNormal(mean[k], None, variance=var[k])))
                    else:
                        out.append(MultivariateNormal(mean[k], None, variance=var[k] + float(th[k, -1])))
            return out
        with torch.no_grad():
            for lt in self.sample_variational_log_hyper(num_samples):
                th = self._theta_of(lt)
                if test_x.dim() == 1:
                    test_x = test_x[:, None]
                mean, var, info = self._engine_obj().svgp_predict(self._dev(test_x), self._dev(self.inducing_inputs), th[1:-1].tolist(),
                                                                  float(th[0]), self._dev(self.variational_mean),
                                                                  self._dev(self.chol_variational_covar), jitter=self.jitter)
                if int(info.to("cpu").item()) != 0:
                    continue
                if self.bernoulli:  # class-1 probabilities through the probit link
                    out.append(self.likelihood(MultivariateNormal(mean, None, variance=var)))
                else:
                    out.append(MultivariateNormal(mean, None, variance=var + float(th[-1])))
        return out
