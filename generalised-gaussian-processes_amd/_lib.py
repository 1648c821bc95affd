"""ctypes binding of libsgp_hip.so -- one Python prototype per entry point of include/sgp.h.

There is deliberately no fallback: if the HIP library is missing or does not export a symbol the
import fails loudly (``SgpLibraryError``).  The product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libsgp_hip.so")

SGP_ABI_VERSION = 3
SGP_MAX_DIM = 32
SGP_MAX_INDUCING = 4096
KERNEL_IDS = {"rbf": 0, "matern32": 1, "matern52": 2, "composite": 3}
COMP_LEN = 33  # SGP_COMP_LEN: doubles in a composite-kernel parameter block (include/sgp.h)
OPT_CONTRACTION, OPT_ASM_OVERLAP, OPT_KFU_BUDGET_BYTES, OPT_COND_LIMIT, OPT_CU_BUDGET, OPT_TIMING, OPT_SHARED_DEVICE = range(7)
OUT_F, OUT_LOGMARG, OUT_TRACE, OUT_LOGDETB, OUT_QUAD, OUT_TRW, OUT_S2BAR, OUT_KAPPABAR, OUT_LEN = range(9)


class SgpLibraryError(RuntimeError):
    pass


class SgpStatusError(RuntimeError):
    def __init__(self, fn, status, text):
        super().__init__("%s returned %d: %s" % (fn, status, text))
        self.status = status


_vp, _i64, _i32, _dbl, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_size_t
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes) ; mirrors include/sgp.h line by line
PROTOTYPES = {
    "sgp_abi_version": (_i32, []),
    "sgp_status_string": (C.c_char_p, [_i32]),
    # contexts (ABI version 2): first argument = sgp_ctx* (NULL = the default context)
    "sgp_ctx_create": (_vp, [_i32]),
    "sgp_ctx_destroy": (None, [_vp]),
    "sgp_ctx_device": (_i32, [_vp]),
    "sgp_ctx_set_option": (_i32, [_vp, _i32, _dbl]),
    "sgp_ctx_get_option": (_dbl, [_vp, _i32]),
    "sgp_ctx_bind_thread": (None, [_vp]),
    "sgp_ctx_set_pass1_gate": (None, [_vp, _vp]),
    "sgp_ctx_contraction_last": (_i32, [_vp]),
    "sgp_ctx_contraction_would_use_i8": (_i32, [_vp, _i64, _i32]),
    "sgp_ctx_timing_last_ms": (_i32, [_vp, _i32, C.POINTER(C.c_float)]),
    "sgp_ctx_timing_last_rows": (_i64, [_vp, _i32]),
    "sgp_ctx_suffstats_workspace_bytes": (_sz, [_vp, _i64, _i32, _i32, _i32]),
    "sgp_ctx_suffstats_fwd": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_suffstats_bwd": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _dp, _dbl, _vp, _vp, _dbl, _vp, _i64, _i32, _i32, _i32,
                                     _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_kuu_factor": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_kuu_factor_ex": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_bound_from_stats": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _i64, _i32, _i32, _vp,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    # ABI version 3: the guard's fallback orders, the factored pass 2 and the whitened bound with a context
    "sgp_ctx_suffstats_whitened_workspace_bytes": (_sz, [_vp, _i64, _i32, _i32]),
    "sgp_ctx_suffstats_fwd_whitened": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp,
                                              _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_suffstats_whitened_rows_workspace_bytes": (_sz, [_vp, _i64, _i32, _i32, _i32]),
    "sgp_ctx_suffstats_fwd_whitened_rows": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp,
                                                   _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_suffstats_extended_workspace_bytes": (_sz, [_vp, _i64, _i32, _i32]),
    "sgp_ctx_suffstats_fwd_extended": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp, _i32,
                                              _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_suffstats_fwd_extended_f16": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp, _i32,
                                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_suffstats_bwd_factored_workspace_bytes": (_sz, [_vp, _i64, _i32, _i32, _i32]),
    "sgp_ctx_suffstats_bwd_factored": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _dp, _dbl, _vp, _vp, _dbl, _vp, _dbl, _i64, _i32, _i32, _i32,
                                              _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_bound_from_whitened_stats": (_i32, [_vp, _vp, _vp, _vp, _vp, _dbl, _i64, _i32, _i32, _vp,
                                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_ctx_mixture_predict": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i32, _dp, _dp, _dp, _dbl, _i32, _i32, _i32, _i32,
                                       _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_timing_enable": (None, [_i32]),
    "sgp_timing_last_ms": (_i32, [_i32, C.POINTER(C.c_float)]),
    "sgp_timing_last_rows": (_i64, [_i32]),
    "sgp_set_asm_overlap": (None, [_i32]),
    "sgp_set_cond_limit": (None, [_dbl]),
    "sgp_suffstats_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sgp_suffstats_workspace_bytes_ex": (_sz, [_i64, _i32, _i32, _i32]),
    "sgp_suffstats_bwd_workspace_bytes_ex": (_sz, [_i64, _i32, _i32, _i32]),
    "sgp_kfu_len": (_sz, [_i64, _i32]),
    "sgp_set_kfu_budget_bytes": (None, [_sz]),
    "sgp_set_contraction": (_i32, [_i32]),
    "sgp_contraction_last": (_i32, []),
    "sgp_set_pass1_gate": (None, [_vp]),
    "sgp_suffstats_fwd": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_stats_packed_len": (_sz, [_i32]),
    "sgp_stats_pack_lower": (_i32, [_vp, _i32, _vp, _vp]),
    "sgp_stats_unpack_lower": (_i32, [_vp, _i32, _vp, _vp]),
    "sgp_kuu": (_i32, [_vp, _i64, _dp, _dbl, _dbl, _i32, _i32, _i32, _vp, _vp]),
    "sgp_chol_workspace_bytes": (_sz, [_i32]),
    "sgp_chol_lower": (_i32, [_vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "sgp_trsm_workspace_bytes": (_sz, [_i32, _i32]),
    "sgp_trsm_lower": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "sgp_logdiag_sum": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "sgp_bound_workspace_bytes": (_sz, [_i32, _i32]),
    "sgp_bound_factors_len": (_sz, [_i32]),
    "sgp_kuu_inverse_trace_len": (_sz, []),
    "sgp_kuu_inverse_trace": (_i32, [_vp, _i32, _vp, _vp]),
    "sgp_streaming_error_estimate": (_i32, [_vp, _vp, _dbl, _i64, _i32, _vp, _vp]),
    "sgp_streaming_error_bound": (_i32, [_vp, _dbl, _dbl, _vp, _vp]),
    "sgp_streaming_error_report": (_i32, [_vp, _i64, _vp, _dbl, _dbl, _i64, _i32, _vp, _vp]),
    "sgp_kuu_factor_len": (_sz, [_i32]),
    "sgp_kuu_factor_workspace_bytes": (_sz, [_i32]),
    "sgp_kuu_factor": (_i32, [_vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "sgp_kuu_factor_ex": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_bound_from_stats": (_i32, [_vp, _vp, _vp, _vp, _vp, _dbl, _i64, _i32, _i32, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_whitened_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sgp_suffstats_fwd_whitened": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp,
                                          _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_whitened_rows_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "sgp_suffstats_fwd_whitened_rows": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp,
                                               _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_extended_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sgp_suffstats_fwd_extended": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp, _i32,
                                          _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_fwd_extended_ex": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp, _i32,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_phibar_dd_workspace_bytes": (_sz, [_i32]),
    "sgp_phibar_dd": (_i32, [_vp, _vp, _i32, _dbl, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_bwd_lo_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sgp_suffstats_bwd_lo": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_bwd_lo_workspace_bytes_ex": (_sz, [_i64, _i32, _i32, _i32]),
    "sgp_suffstats_bwd_lo_f16": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_fwd_extended_f16": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _i64, _i32, _i32, _i32, _vp, _i32,
                                              _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_bound_from_whitened_stats": (_i32, [_vp, _vp, _vp, _vp, _dbl, _i64, _i32, _i32, _vp,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_bound_from_whitened_stats_ex": (_i32, [_vp, _vp, _vp, _vp, _dbl, _i64, _i32, _i32, _vp,
                                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_bwd_factored_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sgp_suffstats_bwd_factored": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _vp, _vp, _dbl, _vp, _dbl, _i64, _i32, _i32, _i32,
                                          _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_bwd_factored_workspace_bytes_ex": (_sz, [_i64, _i32, _i32, _i32]),
    "sgp_suffstats_bwd_factored_ex": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _vp, _vp, _dbl, _vp, _dbl, _i64, _i32, _i32, _i32,
                                             _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_set_cu_budget": (None, [_i32]),
    "sgp_small_supported": (_i32, [_i64, _i32, _i32, _i32]),
    "sgp_small_debug_stamps": (None, [_vp]),
    "sgp_small_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sgp_small_sync_bytes": (_sz, []),
    "sgp_small_eval": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _dbl, _i32, _i32, _vp, _vp, _vp,
                              _vp, _sz, _vp]),
    "sgp_small_eval_composite": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _dp, _i32, _ip, _ip, _dp, _i64, _i32, _i32, _dbl, _i32, _i32,
                                        _vp, _vp, _vp, _sz, _vp]),
    "sgp_small_nuts_composite": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _dp, _i32, _ip, _ip, _dp, _i64, _i32, _i32, _dbl, _i32, _i32,
                                        _i32, _dbl, _dbl, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_small_eval_batch": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i32, _i64, _i32, _i32, _i32, _dbl, _i32, _i32, _vp, _vp, _vp, _vp,
                                    _vp, _sz, _vp]),
    "sgp_small_nuts_stat_cols": (_sz, []),
    "sgp_small_nuts": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _dbl, _i32, _i32, _i32, _dbl, _dbl,
                              C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_suffstats_bwd_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sgp_suffstats_bwd": (_i32, [_vp, _i64, _vp, _vp, _i64, _dp, _dbl, _vp, _vp, _dbl, _vp, _i64, _i32, _i32, _i32,
                                 _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_kuu_bwd_workspace_bytes": (_sz, [_i32, _i32]),
    "sgp_kuu_bwd": (_i32, [_vp, _i64, _dp, _dbl, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_svgp_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sgp_svgp_elbo": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _dp, _dbl, _dbl, _dbl, _vp, _vp, _i64, _i32, _i32, _i32, _i32,
                             _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_svgp_batch_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "sgp_svgp_elbo_batch": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _dp, _dp, _dp, _dbl, _vp, _vp, _i64, _i32, _i32, _i32, _i32,
                                   _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_svgp_elbo_batch_forward": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _dp, _dp, _dp, _dbl, _vp, _vp, _i64, _i32, _i32, _i32, _i32,
                                           _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_svgp_elbo_batch_reverse": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _dp, _dp, _dp, _dbl, _vp, _vp, _i64, _i32, _i32, _i32, _i32,
                                           _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_svgp_predict_batch": (_i32, [_vp, _i64, _i64, _vp, _i64, _i32, _dp, _dp, _dbl, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_svgp_batch_combine": (_i32, [_i32, _dp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgp_mixture_predict_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32, _i32, _i32]),
    "sgp_mixture_predict": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i32, _dp, _dp, _dp, _dbl, _i32, _i32, _i32, _i32, _dbl,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_svgp_predict": (_i32, [_vp, _i64, _i64, _vp, _i64, _dp, _dbl, _dbl, _vp, _vp, _i32, _i32, _i32,
                                _vp, _vp, _vp, _vp, _sz, _vp]),
    "sgp_gauss_hermite": (_i32, [_i32, _dp, _dp]),
    "sgp_predict_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "sgp_predict": (_i32, [_vp, _i64, _i64, _vp, _i64, _dp, _dbl, _dbl, _vp, _i32, _i32, _i32, _i32,
                           _vp, _vp, _vp, _vp, _sz, _vp]),
}

_LIB = None


def load_library(path: str | None = None):
    """dlopen the HIP library and bind every prototype; raises SgpLibraryError when anything is missing."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise SgpLibraryError(
            "%s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc); this package has no CPU fallback." % p)
    try:
        lib = C.CDLL(p)
    except OSError as e:  # pragma: no cover - depends on the ROCm runtime being present
        raise SgpLibraryError("cannot load %s: %s" % (p, e)) from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SgpLibraryError("%s does not export %s" % (p, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.sgp_abi_version() != SGP_ABI_VERSION:
        raise SgpLibraryError("ABI version mismatch: library %d, binding %d" % (lib.sgp_abi_version(), SGP_ABI_VERSION))
    if path is None:
        _LIB = lib
    return lib


def check(fn_name: str, status: int):
    if status != 0:
        lib = load_library()
        raise SgpStatusError(fn_name, status, lib.sgp_status_string(status).decode())
