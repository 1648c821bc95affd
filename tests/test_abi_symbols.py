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
    assert lib.sgp_abi_version() == 3
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


def test_contexts_carry_their_own_options_and_the_setters_are_shims_over_the_default_one(lib):
    """ABI version 2 (include/sgp.h: sgp_ctx_*): no kernel is launched -- options, the contraction rule, the workspace query."""
    import ggp_amd._lib as L
    null = C.c_void_p(0)
    a, b = C.c_void_p(lib.sgp_ctx_create(0)), C.c_void_p(lib.sgp_ctx_create(0))
    assert a.value and b.value and a.value != b.value and not lib.sgp_ctx_create(-1)
    try:
        assert lib.sgp_ctx_get_option(a, L.OPT_CONTRACTION) == 1.0 and lib.sgp_ctx_get_option(a, L.OPT_COND_LIMIT) == 1e13
        assert lib.sgp_ctx_set_option(a, L.OPT_CONTRACTION, 0.0) == 0
        assert lib.sgp_ctx_set_option(a, L.OPT_CONTRACTION, 3.0) == -1 and lib.sgp_ctx_set_option(a, 99, 1.0) == -1
        assert lib.sgp_ctx_get_option(a, L.OPT_CONTRACTION) == 0.0 and lib.sgp_ctx_get_option(b, L.OPT_CONTRACTION) == 1.0
        assert lib.sgp_ctx_get_option(null, L.OPT_CONTRACTION) == 1.0                       # the default context is untouched
        # the rule follows the context: rows x Mp^2 >= 2^32 in mode 1, never in mode 0
        assert lib.sgp_ctx_contraction_would_use_i8(b, 1_000_000, 1024) == 1 and lib.sgp_ctx_contraction_would_use_i8(a, 1_000_000, 1024) == 0
        assert lib.sgp_ctx_contraction_would_use_i8(b, 30_000, 100) == 0
        # ... and so does the workspace of a value + gradient call (digit planes beside a caller-owned K'_fu only when they are needed)
        wa = lib.sgp_ctx_suffstats_workspace_bytes(a, 1_000_000, 1024, 8, 1)
        wb = lib.sgp_ctx_suffstats_workspace_bytes(b, 1_000_000, 1024, 8, 1)
        assert 0 < wa < wb and wb - wa >= 7 * 1024 * 1_000_000
        assert lib.sgp_ctx_suffstats_workspace_bytes(null, 1_000_000, 1024, 8, 1) == lib.sgp_suffstats_workspace_bytes_ex(1_000_000, 1024, 8, 1) == wb
        # the deprecated setters write the DEFAULT context, and only it
        prev = lib.sgp_set_contraction(2)
        assert prev == 1 and lib.sgp_ctx_get_option(null, L.OPT_CONTRACTION) == 2.0 and lib.sgp_ctx_get_option(b, L.OPT_CONTRACTION) == 1.0
        lib.sgp_set_contraction(prev)
        lib.sgp_set_cond_limit(5.0)
        assert lib.sgp_ctx_get_option(null, L.OPT_COND_LIMIT) == 5.0 and lib.sgp_ctx_get_option(a, L.OPT_COND_LIMIT) == 1e13
        lib.sgp_set_cond_limit(-1.0)
        assert lib.sgp_ctx_get_option(null, L.OPT_COND_LIMIT) == 1e13
        lib.sgp_set_kfu_budget_bytes(1 << 20)
        assert lib.sgp_ctx_get_option(null, L.OPT_KFU_BUDGET_BYTES) == float(1 << 20) and lib.sgp_ctx_get_option(a, L.OPT_KFU_BUDGET_BYTES) == float(16 << 30)
        lib.sgp_set_kfu_budget_bytes(0)
        # argument checks run in a context too
        one = C.c_void_p(8)
        inv = (C.c_double * 2)(1.0, 1.0)
        assert lib.sgp_ctx_suffstats_fwd(a, one, 2, one, one, 2, inv, 1.0, 10, 4, 2, 0, one, one, one, one, null, null, 0, null) == -3
        assert lib.sgp_ctx_kuu_factor(a, one, 4, one, one, null, 0, null) == -3
        assert lib.sgp_ctx_contraction_last(a) == 0 and lib.sgp_ctx_timing_last_rows(a, 1) == 0 and lib.sgp_ctx_timing_last_rows(a, 0) == -1
    finally:
        lib.sgp_ctx_destroy(a)
        lib.sgp_ctx_destroy(b)
        lib.sgp_ctx_destroy(null)   # no-op


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
