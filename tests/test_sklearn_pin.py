"""Third-party pin (scikit-learn, the one GP library the reference uses that is installed here:
experiments/lml_surface.py, hyperparameter_identification.py): kernel values and the Z = X limit of the collapsed bound
against fixtures produced by ``tests/golden/make_golden_sklearn.py``.  CPU: the oracles; ``-m gpu``: the HIP path."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, dev
from oracle import composite_oracle as CO
from oracle import vfe_extended as E
from oracle import vfe_oracle as O

SK = os.path.join(GOLDEN_DIR, "sklearn")


def _load(name):
    z = np.load(os.path.join(SK, name + ".npz"))
    return {k: z[k] for k in z.files}


def _composite_cases(K):
    return [("ratquad", [(1.0, [(CO.RATQUAD, float(K["ratquad_ls"]), float(K["ratquad_alpha"]))])]),
            ("periodic", [(1.0, [(CO.PERIODIC, float(K["periodic_ls_pymc3"]), float(K["periodic_period"]))])]),
            ("rbf_iso", [(1.0, [(CO.EXPQUAD, 0.9)])]), ("matern32_iso", [(1.0, [(CO.MATERN32, 0.9)])]),
            ("matern52_iso", [(1.0, [(CO.MATERN52, 0.9)])])]


def test_oracle_kernels_match_sklearn():
    K = _load("kernel_values")
    for key, kid in (("rbf_ard", 0), ("matern32_ard", 1), ("matern52_ard", 2)):
        got = O.kern(K["A"], K["B"], K["ls"], 1.0, kid).numpy()
        assert np.max(np.abs(got - K[key])) < 1e-14, key
        assert np.max(np.abs(np.asarray(E.stationary_k(K["A"], K["B"], K["ls"], 1.0, kid), dtype=np.float64) - K[key])) < 1e-14
    for key, terms in _composite_cases(K):
        blk = CO.make_block(terms)
        got = CO.composite_k(torch.as_tensor(K["a1"]), torch.as_tensor(K["b1"]), torch.as_tensor(blk)).numpy()
        assert np.max(np.abs(got - K[key])) < 1e-14, key
        assert np.max(np.abs(np.asarray(E.composite_k(K["a1"], K["b1"], blk), dtype=np.float64) - K[key])) < 1e-14, key


@pytest.mark.parametrize("name", ["lml_d1", "lml_d3"])
def test_oracle_bound_with_Z_equal_X_is_sklearns_log_marginal_likelihood(name):
    G = _load(name)
    X, y, ls, sf2, s2 = G["X"], G["y"], G["ls"], float(G["sf2"]), float(G["s2"])
    lml = float(G["lml"])
    assert abs(O.vfe_dense(X, y, X, ls, sf2, s2, 0.0)[0] - lml) < 1e-9 * abs(lml)
    assert abs(float(O.vfe_pymc3_order(X, y, X, ls, sf2 ** 0.5, s2 ** 0.5, 0.0)) - lml) < 1e-9 * abs(lml)
    assert abs(float(E.vfe(X, y, X, ls, sf2, s2, 0.0)) - lml) < 1e-9 * abs(lml)
    g = O.grads_autograd(X, y, X, torch.as_tensor(ls), sf2, s2, 0.0)
    # Z = X moves with X in sklearn's derivative: dF/d ls through both K_uf and K_uu is what autograd returns for fixed Z = X
    assert np.max(np.abs(g["g_ls"].numpy() * ls - G["dlml_dlog_ls"])) < 1e-6 * np.max(np.abs(G["dlml_dlog_ls"]))
    assert abs(g["g_sf2"] * sf2 - float(G["dlml_dlog_sf2"])) < 1e-6 * abs(float(G["dlml_dlog_sf2"]))
    assert abs(g["g_s2"] * s2 - float(G["dlml_dlog_s2"])) < 1e-6 * abs(float(G["dlml_dlog_s2"]))


@pytest.mark.gpu
def test_hip_kernels_match_sklearn(engine):
    K = _load("kernel_values")
    A, B = dev(K["A"], engine), dev(K["B"], engine)
    Zall = torch.cat([A, B])  # Kuu of the stacked inputs holds k(A, B) as an off-diagonal block
    for key, kern in (("rbf_ard", "rbf"), ("matern32_ard", "matern32"), ("matern52_ard", "matern52")):
        Kuu = engine.kuu(Zall, K["ls"], 1.0, 0.0, kern).cpu().numpy()
        assert np.max(np.abs(Kuu[:9, 9:] - K[key])) < 1e-14, key
    z1 = torch.cat([dev(K["a1"], engine), dev(K["b1"], engine)])
    for key, terms in _composite_cases(K):
        Kuu = engine.kuu(z1, list(CO.make_block(terms)), 1.0, 0.0, "composite").cpu().numpy()
        assert np.max(np.abs(Kuu[:9, 9:] - K[key])) < 1e-14, key


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["streaming", "whitened"])
@pytest.mark.parametrize("name", ["lml_d1", "lml_d3"])
def test_hip_bound_with_Z_equal_X_is_sklearns_log_marginal_likelihood(engine, name, form):
    import ggp_amd
    G = _load(name)
    X, y, ls, sf2, s2 = dev(G["X"], engine), dev(G["y"], engine), G["ls"], float(G["sf2"]), float(G["s2"])
    lml = float(G["lml"])
    cb = ggp_amd.CollapsedBound(X, y, jitter=0.0, engine=engine, form=form)
    F, parts = cb.value(X, ls.tolist(), sf2, s2)
    assert abs(F - lml) < 1e-8 * abs(lml) and abs(parts["trace_term"]) < 1e-8
    F2, g = cb.value_and_grad(X, ls.tolist(), sf2, s2)
    assert abs(F2 - lml) < 1e-8 * abs(lml)
    assert np.max(np.abs(g["ls"].numpy() * ls - G["dlml_dlog_ls"])) < 1e-5 * np.max(np.abs(G["dlml_dlog_ls"]))
    assert abs(g["sf2"] * sf2 - float(G["dlml_dlog_sf2"])) < 1e-5 * abs(float(G["dlml_dlog_sf2"]))
    assert abs(g["s2"] * s2 - float(G["dlml_dlog_s2"])) < 1e-5 * abs(float(G["dlml_dlog_s2"]))
