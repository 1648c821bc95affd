"""CPU: libsgp_hip.so builds, loads and exports every function include/sgp.h declares; host-side argument
checks and workspace queries behave (no kernel is launched here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "sgp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sgp_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    import ggp_amd
    return ggp_amd.load_library()


def test_every_declared_symbol_is_exported_and_bound(lib):
    import ggp_amd._lib as L
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libsgp_hip.so does not export %s" % n
        assert n in L.PROTOTYPES, "python binding lacks a prototype for %s" % n
    assert sorted(L.PROTOTYPES) == names, "binding declares functions the header does not"


def test_abi_version_and_status_strings(lib):
    assert lib.sgp_abi_version() == 1
    assert lib.sgp_status_string(0) == b"ok"
    assert b"workspace" in lib.sgp_status_string(-3)
    assert b"positive definite" in lib.sgp_status_string(7)


def test_workspace_queries(lib):
    a = lib.sgp_suffstats_workspace_bytes(1000, 100, 3)
    b = lib.sgp_suffstats_workspace_bytes(1_000_000, 1024, 8)
    assert 0 < a < b
    assert lib.sgp_suffstats_workspace_bytes(10, 100, 33) == 0      # d > SGP_MAX_DIM
    assert lib.sgp_suffstats_workspace_bytes(10, 5000, 3) == 0      # M > SGP_MAX_INDUCING
    assert lib.sgp_suffstats_workspace_bytes(0, 10, 2) > 0          # empty shard is legal
    assert lib.sgp_bound_workspace_bytes(1024, 1) >= 9 * 1024 * 1024 * 8
    assert lib.sgp_bound_factors_len(7) == 2 * 49 + 7
    assert lib.sgp_predict_workspace_bytes(100, 50, 2, 1) > lib.sgp_predict_workspace_bytes(100, 50, 2, 0)
    assert lib.sgp_suffstats_bwd_workspace_bytes(1000, 100, 3) > 0
    assert lib.sgp_kuu_bwd_workspace_bytes(100, 3) > 0
    assert lib.sgp_chol_workspace_bytes(100) > 0 and lib.sgp_trsm_workspace_bytes(100, 4) > 0


def test_bad_arguments_are_rejected_before_any_launch(lib):
    null = C.c_void_p(0)
    inv = (C.c_double * 2)(1.0, 1.0)
    one = C.c_void_p(8)  # non-null dummy, never dereferenced: every call below fails validation first
    assert lib.sgp_kuu(null, 2, inv, 1.0, 0.0, 4, 2, 0, one, null) == -1
    assert lib.sgp_kuu(one, 2, inv, 1.0, 0.0, 4, 2, 9, one, null) == -1            # kernel id
    assert lib.sgp_kuu(one, 40, inv, 1.0, 0.0, 4, 40, 0, one, null) == -2          # d too large
    assert lib.sgp_suffstats_fwd(one, 2, one, one, 2, inv, 1.0, 10, 4, 2, 0, one, one, one, one, null, null, 0, null) == -3
    assert lib.sgp_suffstats_fwd(one, 1, one, one, 2, inv, 1.0, 10, 4, 2, 0, one, one, one, one, null, one, 1 << 30, null) == -1  # ldx < d
    assert lib.sgp_kfu_len(1000, 100) == 1024 * 128 and lib.sgp_kfu_len(0, 5) == 256 * 128
    assert lib.sgp_bound_from_stats(one, one, one, one, one, -1.0, 10, 4, 0, one, null, null, null, null, null, one, one, 1 << 30, null) == -1
    assert lib.sgp_bound_from_stats(one, one, one, one, one, 0.1, 10, 4, 1, one, null, null, null, null, null, one, one, 1 << 30, null) == -1
    assert lib.sgp_bound_from_stats(null, one, one, one, one, 0.1, 10, 4, 0, one, null, null, null, null, null, one, one, 1 << 30, null) == -1
    assert lib.sgp_kuu_factor(one, 4, one, one, null, 0, null) == -3 and lib.sgp_kuu_factor_len(100) == 128 * 128
    assert lib.sgp_chol_lower(one, 4, 4, one, null, 0, null) == -3


def test_product_has_no_cpu_fallback():
    import torch
    import ggp_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ggp_amd.SgpLibraryError):
        ggp_amd.HipEngine()
    with pytest.raises(ggp_amd.SgpLibraryError):
        ggp_amd.CollapsedBound(torch.zeros(4, 2, dtype=torch.float64), torch.zeros(4, dtype=torch.float64))


def test_gauss_hermite_rule_matches_numpy(lib):
    """Host utility behind the Bernoulli-probit expectation: nodes / weights for the standard normal."""
    import numpy as np
    for n in (3, 10, 20):
        x = (C.c_double * n)()
        w = (C.c_double * n)()
        assert lib.sgp_gauss_hermite(n, x, w) == 0
        xr, wr = np.polynomial.hermite.hermgauss(n)
        order = np.argsort(np.array(x[:]))
        assert np.allclose(np.array(x[:])[order], np.sort(xr * np.sqrt(2.0)), atol=1e-13)
        assert np.allclose(np.array(w[:])[order], (wr / np.sqrt(np.pi))[np.argsort(xr)], rtol=1e-11, atol=1e-300)
        assert abs(sum(w[:]) - 1.0) < 1e-13


def test_only_the_c_abi_is_exported(lib):
    """-fvisibility=hidden + csrc/libsgp.map: the dynamic symbol table holds the header's functions and nothing else
    (no sgp:: internals, no kernel handles) -- VERDICT r2 weak-13."""
    import shutil
    import subprocess
    import ggp_amd._lib as L
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", L.lib_path() if hasattr(L, "lib_path") else
                          os.path.join(ROOT, "generalised-gaussian-processes_amd", "csrc", "libsgp_hip.so")],
                         capture_output=True, text=True, check=True).stdout
    syms = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert syms == header_functions(), sorted(set(syms) ^ set(header_functions()))


def test_digest_covers_every_header_a_source_includes(tmp_path, monkeypatch):
    """build() reuses the .so while the source digest is unchanged: editing ANY file a translation unit includes (the sampler
    header sgp_nuts.hpp was once left out) must change it."""
    import importlib.util
    import shutil
    pkg = os.path.join(ROOT, "generalised-gaussian-processes_amd")
    spec = importlib.util.spec_from_file_location("_sgp_build_t", os.path.join(pkg, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    covered = set(b.SOURCES) | {os.path.basename(h) for h in b._headers()}
    for src in b.SOURCES + [h for h in b._headers() if h.endswith(".hpp")]:
        for inc in re.findall(r'#include\s+"([^"]+)"', open(os.path.join(b.CSRC, src)).read()):
            assert os.path.basename(inc) in covered, "%s includes %s, which the digest ignores" % (src, inc)
    # and for real: a comment added to the sampler header changes the digest
    work = tmp_path / "pkg" / "csrc"
    work.mkdir(parents=True)
    for f in os.listdir(b.CSRC):
        if f.endswith((".hip", ".hpp", ".map")):
            shutil.copy(os.path.join(b.CSRC, f), work / f)
    (tmp_path / "include").mkdir()
    shutil.copy(os.path.join(ROOT, "include", "sgp.h"), tmp_path / "include" / "sgp.h")
    monkeypatch.setattr(b, "CSRC", str(work))
    before = b._digest()
    assert before == open(b.LIB_PATH + ".sha256").read().strip()  # same bytes, same digest as the shipped stamp
    with open(work / "sgp_nuts.hpp", "a") as fh:
        fh.write("// a comment\n")
    assert b._digest() != before
