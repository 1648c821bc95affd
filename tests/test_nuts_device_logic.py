"""The device-resident sampler's state machine (csrc/sgp_nuts.hpp), compiled for the host, against the Python sampler
(hmc.NUTS) fed the same random stream: draws, step sizes, tree sizes and leapfrog counts must coincide -- on an
analytic Gaussian and on the reference's NUTS target evaluated by the oracle-backed test double."""
import ctypes as C
import math
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from fake_engine import OracleEngine

import ggp_amd
from ggp_amd.hmc import NUTS, DiagMassAdapter, SplitMix

CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def nuts_lib(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    so = str(tmp_path_factory.mktemp("nuts") / "libnuts_host.so")
    inc = os.path.join(ROOT, "generalised-gaussian-processes_amd", "csrc")
    # -ffp-contract=off: no fused multiply-adds the Python arithmetic does not have
    subprocess.run([gxx, "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I", inc, "-o", so,
                    os.path.join(ROOT, "tests", "native", "nuts_host.cpp")], check=True, timeout=300)
    lib = C.CDLL(so)
    lib.nuts_host_run.restype = C.c_long
    lib.nuts_host_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_ulonglong,
                                  C.POINTER(C.c_double), CB, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return lib


def run_host(lib, f, ndim, q0, tune, draws, seed, depth=10):
    samples = np.zeros((draws, ndim))
    stats = np.zeros((draws, 8))

    def cb(qp, lpp, gp):
        q = np.array([qp[i] for i in range(ndim)])
        lp, g = f(q)
        lpp[0] = lp
        for i in range(ndim):
            gp[i] = g[i]

    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    nl = lib.nuts_host_run(ndim, tune, draws, depth, 0.25, 0.8, seed, q0.ctypes.data_as(C.POINTER(C.c_double)), CB(cb),
                           samples.ctypes.data_as(C.POINTER(C.c_double)), stats.ctypes.data_as(C.POINTER(C.c_double)), None)
    return samples, stats, nl


def run_python(f, ndim, q0, tune, draws, seed, depth=10):
    nuts = NUTS(f, ndim, max_treedepth=depth, rng=SplitMix(seed))
    q = np.array(q0, dtype=np.float64)
    lp, g = nuts._eval(q)
    nuts.mass = DiagMassAdapter(ndim, initial_mean=q)
    samples, stats = [], []
    for it in range(tune + draws):
        q, lp, g, st = nuts.draw(q, lp, g, it < tune)
        if it >= tune:
            samples.append(q.copy())
            stats.append([st["step_size"], st["tree_size"], st["depth"], st["mean_tree_accept"], float(st["diverging"]), st["energy"], lp,
                          nuts.n_leapfrog])
    return np.array(samples), np.array(stats), nuts.n_leapfrog


@pytest.mark.parametrize("ndim,seed", [(1, 3), (3, 11), (10, 5)])
def test_state_machine_equals_recursive_sampler_on_a_gaussian(nuts_lib, ndim, seed):
    mu = np.linspace(-1.0, 2.0, ndim)
    sd = np.logspace(-1.0, 0.7, ndim)

    def f(q):
        z = (np.asarray(q) - mu) / sd
        return float(-0.5 * np.sum(z * z)), (-z / sd).tolist()

    q0 = mu + 0.3
    a = run_host(nuts_lib, f, ndim, q0, 60, 40, seed)
    b = run_python(f, ndim, q0, 60, 40, seed)
    assert a[2] == b[2], "leapfrog counts differ: the trees differ"
    assert np.array_equal(a[1][:, 1:3], b[1][:, 1:3])                     # tree sizes, depths
    assert np.allclose(a[1][:, 0], b[1][:, 0], rtol=1e-12, atol=0)        # step sizes
    assert np.allclose(a[0], b[0], rtol=1e-10, atol=1e-12)
    # and it samples the right thing
    c = run_host(nuts_lib, f, ndim, q0, 300, 1500, seed + 1)
    assert np.all(np.abs(c[0].mean(0) - mu) < 0.25 * sd) and np.all(np.abs(c[0].std(0) / sd - 1.0) < 0.2)


def test_divergences_and_zero_density_regions_are_handled_alike(nuts_lib):
    """A target with a hard wall (logp = -inf beyond it) and a funnel-like scale: exercises the diverging / non-finite
    leaf and the folding of a bad sub-tree into its pending siblings."""
    def f(q):
        x, y = float(q[0]), float(q[1])
        if x > 1.5:
            return -math.inf, [0.0, 0.0]
        s = math.exp(-x)
        return -0.5 * x * x - 0.5 * y * y * s * s * 50.0 - 2.0 * x * 0.0, [-x + y * y * s * s * 50.0, -y * s * s * 50.0]

    a = run_host(nuts_lib, f, 2, [0.2, 0.1], 40, 60, 7, depth=8)
    b = run_python(f, 2, [0.2, 0.1], 40, 60, 7, depth=8)
    assert a[2] == b[2] and np.array_equal(a[1][:, 1:3], b[1][:, 1:3]) and np.array_equal(a[1][:, 4], b[1][:, 4])
    assert np.allclose(a[0], b[0], rtol=1e-9, atol=1e-12)


def test_state_machine_on_the_reference_target(nuts_lib):
    """The NUTS target of models/bayesian_sgpr_hmc.py:60-71 (oracle-backed engine): same chain from both samplers."""
    G = load_golden("rbf_d3_small")
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
    cb = ggp_amd.CollapsedBound(T(G["X"]), T(G["y"]), jitter=1e-6, engine=OracleEngine())
    tgt = ggp_amd.HmcTarget(cb, T(G["Z"]))
    q0 = np.array(tgt.start()) + 0.1
    a = run_host(nuts_lib, tgt.logp_and_grad, tgt.ndim, q0, 15, 10, 21, depth=6)
    b = run_python(tgt.logp_and_grad, tgt.ndim, q0, 15, 10, 21, depth=6)
    assert a[2] == b[2] and np.array_equal(a[1][:, 1:3], b[1][:, 1:3])
    assert np.allclose(a[0], b[0], rtol=1e-8, atol=1e-10)

