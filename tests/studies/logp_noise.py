#!/usr/bin/env python3
"""A/B of the rounding noise on logp and its gradient ("What's weak" 2): for ill-conditioned K_uu, how far
is each evaluation of the SAME scalar from an extended-precision yardstick (oracle/vfe_extended.py, 64-bit mantissa)?

    (i)   the HIP path            CollapsedBound.value / value_and_grad through the C ABI
    (ii)  oracle, PyMC3 op order  torch fp64, LAPACK solve_triangular on the materialised A = L^-1 K_uf
    (iii) oracle, streaming form  torch fp64, Phi formed first, then L^-1 Phi L^-T (the algebra of the HIP tail, LAPACK solves)
    (iv)  dense scipy definition  multivariate_normal.logpdf on Q_ff + s2 I  (N <= 2000 only)

Cases: the duplicate-inducing-row fixture; a C2-shaped RBF problem whose inducing inputs are closer than the
lengthscale; the CO2 composite model at its MAP point (jitter 1e-4, the demo default, and 1e-6, PyMC3's stabilize()).
For every case: |F - F_ext| at the base point and at 12 points theta + t v (t = 1e-7 .. 1e-3, random direction v in
log-parameter space), and the gradient along v against extended-precision central differences.  Also, for the CO2
case, the status words of 64 evaluations scattered 1e-3 around the mode (is the sampler seeing failed factorizations?).

Test infrastructure (imports oracle/): run on the GPU box,  python3 tests/studies/logp_noise.py > profiles/r02_logp_noise.json
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ggp_amd  # noqa: E402
from oracle import composite_oracle as CO  # noqa: E402
from oracle import vfe_extended as E  # noqa: E402
from oracle import vfe_oracle as O  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "experiments"))


def rbf_case(name, X, y, Z, ls, sf2, s2, jitter):
    """theta = [log ls_1..d, log sf2, log s2]"""
    d = X.shape[1]
    th0 = np.concatenate([np.log(ls), [math.log(sf2), math.log(s2)]])

    def unpack(th):
        return np.exp(th[:d]), float(np.exp(th[d])), float(np.exp(th[d + 1]))

    def ext(th):
        l, a, n = unpack(np.asarray(th, dtype=np.float64))
        return E.vfe(X, y, Z, l, a, n, jitter, 0)

    def ext_exact(th):  # theta itself in extended precision (for the central differences)
        th = E._ld(th)
        return E.vfe(X, y, Z, np.exp(th[:d]), np.exp(th[d]), np.exp(th[d + 1]), jitter, 0)

    def pymc3(th):
        l, a, n = unpack(th)
        return float(O.vfe_pymc3_order(X, y, Z, l, math.sqrt(a), math.sqrt(n), jitter, 0))

    def stream(th):
        l, a, n = unpack(th)
        return O.bound_from_stats(O.kuu(Z, l, a, jitter, 0), O.suffstats(X, y, Z, l, a, 0), n)["F"]

    def dense(th):
        l, a, n = unpack(th)
        return O.vfe_dense(X, y, Z, l, a, n, jitter, 0)[0]

    def hip_factory(eng):
        cb = ggp_amd.CollapsedBound(torch.as_tensor(X).to(eng.device), torch.as_tensor(y).to(eng.device), jitter=jitter, engine=eng)
        Zd = torch.as_tensor(Z).to(eng.device)

        def val(th):
            l, a, n = unpack(th)
            F, parts = cb.value(Zd, l.tolist(), a, n, raise_on_fail=False)
            return F, parts.get("info", 0)

        def grad(th):  # d/d theta (log-parameters)
            l, a, n = unpack(th)
            F, g = cb.value_and_grad(Zd, l.tolist(), a, n, raise_on_fail=False)
            if g.get("info", 0) != 0:
                return float("nan"), np.full(d + 2, np.nan)
            return F, np.concatenate([np.asarray(g["ls"]) * l, [g["sf2"] * a, g["s2"] * n]])
        return val, grad

    def oracle_grad(th):
        l, a, n = unpack(th)
        g = O.grads_autograd(X, y, Z, torch.as_tensor(l), a, n, jitter)
        return np.concatenate([g["g_ls"].numpy() * l, [g["g_sf2"] * a, g["g_s2"] * n]])

    def oracle_grad_closed(th, whitened=True):
        l, a, n = unpack(th)
        g = O.grads_analytic(X, y, Z, torch.as_tensor(l), a, n, jitter, 0, whitened=whitened)
        return np.concatenate([g["g_ls"].numpy() * l, [g["g_sf2"] * a, g["g_s2"] * n]])

    return dict(name=name, th0=th0, ext=ext, ext_exact=ext_exact, methods={"oracle_pymc3_order": pymc3, "oracle_streaming": stream, "dense_scipy": dense},
                hip_factory=hip_factory, grads={"oracle_autograd_pymc3_order": oracle_grad, "oracle_closed_form_whitened": oracle_grad_closed,
                       "oracle_closed_form_textbook_A5": lambda th: oracle_grad_closed(th, whitened=False)})


def composite_case(name, X, y, Z, kernel, theta, jitter, target_factory):
    nk = len(theta) - 1

    def block_of(th):
        return kernel.with_values([math.exp(v) for v in th[:nk]]).block()

    def ext(th):
        th = np.asarray(th, dtype=np.float64)
        return E.vfe_composite(X, y, Z, block_of(th), math.exp(2 * th[nk]), jitter)

    def ext_exact(th):
        th = E._ld(th)
        blk = E._ld(block_of([float(v) for v in th]))
        vals = np.exp(th[:nk])
        for (_, slot, role), v in zip(kernel.free_parameters(), vals):
            blk[slot] = v * v if role == "amp" else v
        return E.vfe_composite(X, y, Z, blk, np.exp(2 * th[nk]), jitter)

    def pymc3(th):
        return float(CO.vfe_composite(X, y, Z, np.asarray(block_of(th)), math.exp(2 * th[nk]), jitter))

    def hip_factory(eng):
        cb, tgt = target_factory(eng, jitter)
        Zd = tgt.Z

        def val(th):
            F, parts = cb.value(Zd, block_of(th), 1.0, math.exp(2 * th[nk]), raise_on_fail=False)
            return F, parts.get("info", 0)

        def grad(th):
            F, g = cb.value_and_grad(Zd, block_of(th), 1.0, math.exp(2 * th[nk]), raise_on_fail=False)
            if g.get("info", 0) != 0:  # failed factorization
                return float("nan"), np.full(nk + 1, np.nan)
            gb = g["ls"]
            out = []
            for (_, slot, role), t in zip(kernel.free_parameters(), th[:nk]):
                v = math.exp(t)
                out.append((2.0 * v * v if role == "amp" else v) * float(gb[slot]))
            out.append(2.0 * math.exp(2 * th[nk]) * g["s2"])
            return F, np.asarray(out)
        return val, grad

    def oracle_grad(th):
        blk = np.asarray(block_of(th))
        _, g = CO.vfe_composite_and_grads(X, y, Z, blk, math.exp(2 * th[nk]), jitter)
        out = []
        for (_, slot, role), t in zip(kernel.free_parameters(), th[:nk]):
            v = math.exp(t)
            out.append((2.0 * v * v if role == "amp" else v) * float(g["block"][slot]))
        out.append(2.0 * math.exp(2 * th[nk]) * g["s2"])
        return np.asarray(out)

    return dict(name=name, th0=np.asarray(theta, dtype=np.float64), ext=ext, ext_exact=ext_exact, methods={"oracle_pymc3_order": pymc3},
                hip_factory=hip_factory, grads={"oracle_autograd_pymc3_order": oracle_grad})


def run_case(case, eng, rng):
    th0 = case["th0"]
    v = rng.standard_normal(th0.size)
    v /= np.linalg.norm(v)
    hip_val, hip_grad = case["hip_factory"](eng)
    methods = dict(case["methods"])
    ts = [0.0] + [s * 10.0 ** k for k in range(-7, -2) for s in (1.0, 3.0)] + [1e-3 * 3, -1e-4]
    errs = {k: [] for k in list(methods) + ["hip_value", "hip_value_and_grad"]}
    F0 = None
    for t in ts:
        th = th0 + t * v
        Fx = case["ext"](th)
        if F0 is None:
            F0 = float(Fx)
        for k, f in methods.items():
            try:
                errs[k].append(abs(float(E.LD(f(th)) - Fx)))
            except Exception:  # a failed fp64 factorization
                errs[k].append(float("nan"))
        Fh, info = hip_val(th)
        errs["hip_value"].append(abs(float(E.LD(Fh) - Fx)) if info == 0 else float("nan"))
        Fg, _ = hip_grad(th)
        errs["hip_value_and_grad"].append(abs(float(E.LD(Fg) - Fx)) if math.isfinite(Fg) else float("nan"))
    # directional derivative along v: extended-precision central difference of the extended-precision bound
    h = E.LD(1e-6)
    dd = float((case["ext_exact"](E._ld(th0) + h * E._ld(v)) - case["ext_exact"](E._ld(th0) - h * E._ld(v))) / (2 * h))
    gerr = {}
    _, gh = hip_grad(th0)
    gerr["hip"] = abs(float(gh @ v) - dd)
    for k, f in case["grads"].items():
        try:
            gerr[k] = abs(float(f(th0) @ v) - dd)
        except Exception:
            gerr[k] = float("nan")
    return {"case": case["name"], "F_ext": F0, "dirderiv_ext": dd,
            "abs_err_logp_max": {k: float(np.nanmax(e)) for k, e in errs.items()},
            "abs_err_logp_median": {k: float(np.nanmedian(e)) for k, e in errs.items()},
            "failed_evaluations": {k: int(np.isnan(e).sum()) for k, e in errs.items()},
            "abs_err_dirderiv": gerr, "rel_err_dirderiv": {k: e / max(1e-300, abs(dd)) for k, e in gerr.items()}}


def main():
    if os.environ.get("SGP_STUDY_DRY_RUN"):  # CPU dry run of the script itself through the test double
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from fake_engine import OracleEngine
        eng = OracleEngine()
    else:
        eng = ggp_amd.HipEngine()
    rng = np.random.default_rng(0)
    out = {"yardstick": "numpy.longdouble (x87 80-bit) restatement, oracle/vfe_extended.py", "cases": []}

    G = np.load(os.path.join(ROOT, "tests", "golden", "rbf_d2_dupZ.npz"))
    c = rbf_case("dupZ fixture (N 300, M 24, d 2, jitter 1e-6, duplicate inducing rows)", G["X"], G["y"], G["Z"], G["ls"],
                 float(G["sf2"]), float(G["s2"]), float(G["jitter"]))
    out["cases"].append(run_case(c, eng, rng))

    g = torch.Generator().manual_seed(3)
    X = torch.linspace(0, 52.8, 634, dtype=torch.float64)[:, None]
    y = torch.sin(X[:, 0] * 2 * math.pi) * 0.3 + 0.04 * X[:, 0] + 0.05 * torch.randn(634, dtype=torch.float64, generator=g)
    y = (y - y.mean()) / y.std()
    Z = X[torch.linspace(0, 633, 128).round().long()].clone()
    c = rbf_case("C2-shaped RBF (N 634, M 128, d 1, ls 3.0 over inducing spacing 0.42, jitter 1e-6)", X.numpy(), y.numpy(), Z.numpy(),
                 np.array([3.0]), 1.0, 0.01, 1e-6)
    out["cases"].append(run_case(c, eng, rng))

    if os.environ.get("SGP_STUDY_DRY_RUN"):  # the test double has no composite kernels
        print(json.dumps(out, indent=1))
        return
    # CO2 composite at the MAP point of the demo
    import co2_composite_hmc as demo
    y_tr, t_tr, _, _, _ = demo.synthetic_keeling(seed=47)
    Xc = torch.as_tensor(t_tr, dtype=torch.float64)
    yc = torch.as_tensor(y_tr, dtype=torch.float64)
    M = 64
    Zc = Xc[torch.linspace(0, Xc.shape[0] - 1, M).round().long()].clone()
    kernel = ggp_amd.co2_kernel()

    def target_factory(eng_, jitter):
        cb = ggp_amd.CollapsedBound(Xc.to(eng_.device), yc.to(eng_.device), kernel="composite", jitter=jitter, engine=eng_)
        return cb, ggp_amd.CompositeHmcTarget(cb, Zc.to(eng_.device), kernel, ggp_amd.CO2_LOG_PRIOR_SD)

    for jitter in (1e-4, 1e-6):
        cb, tgt = target_factory(eng, jitter)
        theta = list(tgt.start())
        m1 = [0.0] * len(theta)
        m2 = [0.0] * len(theta)
        for it in range(1, 401):
            lp, gth = tgt.logp_and_grad(theta)
            if not math.isfinite(lp):
                break
            for k in range(len(theta)):
                m1[k] = 0.9 * m1[k] + 0.1 * gth[k]
                m2[k] = 0.999 * m2[k] + 0.001 * gth[k] * gth[k]
                theta[k] += 0.05 * (m1[k] / (1 - 0.9 ** it)) / (math.sqrt(m2[k] / (1 - 0.999 ** it)) + 1e-8)
        c = composite_case("CO2 composite at the MAP point (N 634, M 64, jitter %g)" % jitter, Xc.numpy(), yc.numpy(), Zc.numpy(), kernel,
                           theta, jitter, target_factory)
        r = run_case(c, eng, rng)
        r["map_theta"] = [float(v) for v in theta]
        r["map_steps_done"] = it
        # what the sampler sees around the mode: status words, logp spread
        infos, lps = [], []
        for _ in range(64):
            th = np.asarray(theta) + 1e-3 * rng.standard_normal(len(theta))
            blk = kernel.with_values([math.exp(v) for v in th[:-1]]).block()
            F, parts = cb.value(tgt.Z, blk, 1.0, math.exp(2 * th[-1]), raise_on_fail=False)
            infos.append(int(parts.get("info", 0)))
            lps.append(F)
        r["status_words_around_mode"] = {str(k): infos.count(k) for k in sorted(set(infos))}
        r["logp_range_around_mode"] = [float(np.nanmin(lps)), float(np.nanmax(lps))]
        try:
            blk0 = np.asarray(kernel.with_values([math.exp(v) for v in theta[:-1]]).block())
            Kuu = CO.composite_k(torch.as_tensor(Zc), torch.as_tensor(Zc), torch.as_tensor(blk0)).numpy() + jitter * np.eye(M)
            r["cond_Kuu"] = float(np.linalg.cond(Kuu))
        except Exception as ex:  # pragma: no cover
            r["cond_Kuu"] = str(ex)
        out["cases"].append(r)

    # Above the single launch's M <= 128 (the reference runs this model with M = 480 random training times): the multi-launch
    # whitened order at the last MAP point (jitter 1e-6), with pass 2 from the factored adjoint (default) and from an explicit Phibar
    for M2 in (200, 480):
        Z2 = Xc[torch.linspace(0, Xc.shape[0] - 1, M2).round().long()].clone()
        for factored in (True, False):
            def target_factory2(eng_, jitter, _Z=Z2, _f=factored):
                cb = ggp_amd.CollapsedBound(Xc.to(eng_.device), yc.to(eng_.device), kernel="composite", jitter=jitter, engine=eng_)
                cb.factored_adjoint = _f
                return cb, ggp_amd.CompositeHmcTarget(cb, _Z.to(eng_.device), kernel, ggp_amd.CO2_LOG_PRIOR_SD)
            c = composite_case("CO2 composite, theta of the M = 64 MAP point (N 634, M %d, jitter 1e-06), multi-launch whitened order, %s"
                               % (M2, "factored pass 2" if factored else "explicit Phibar"), Xc.numpy(), yc.numpy(), Z2.numpy(), kernel, theta,
                               1e-6, target_factory2)
            r = run_case(c, eng, np.random.default_rng(1))
            blk0 = np.asarray(kernel.with_values([math.exp(v) for v in theta[:-1]]).block())
            Kuu = CO.composite_k(torch.as_tensor(Z2), torch.as_tensor(Z2), torch.as_tensor(blk0)).numpy() + 1e-6 * np.eye(M2)
            r["cond_Kuu"] = float(np.linalg.cond(Kuu))
            out["cases"].append(r)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
