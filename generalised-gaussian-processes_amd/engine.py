"""HipEngine: torch-tensor front end of the C ABI (include/sgp.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and (in ``core``) the
RCCL process group.  All arithmetic happens inside libsgp_hip.so.  Every method takes / returns
fp64 CUDA(=HIP) tensors; hyper-parameters are host floats (they travel as kernel arguments).

The engine interface (``suffstats`` / ``bound`` / ``suffstats_bwd`` / ``kuu_bwd`` / ``predict``) is
the seam the CPU-only tests use to exercise the host logic above it with a test double; the product
constructs ``HipEngine`` and nothing else.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import COMP_LEN, KERNEL_IDS, OUT_LEN

RESULT_HEAD = OUT_LEN + 4  # doubles ahead of the caller's extras in a result buffer: out, status word, estimate, bound, pad


def _kernel_id(kernel) -> int:
    if isinstance(kernel, int):
        if kernel not in KERNEL_IDS.values():
            raise ValueError("unknown kernel id %r" % (kernel,))
        return kernel
    try:
        return KERNEL_IDS[str(kernel).lower()]
    except KeyError:
        raise ValueError("unknown kernel %r (expected one of %s)" % (kernel, sorted(KERNEL_IDS))) from None


class HipEngine:
    """Calls the HIP library on ``device`` (default: current CUDA device).  No CPU fallback.

    ``own_context=True`` gives the engine a library context of its own (include/sgp.h: sgp_ctx_create): its contraction mode,
    conditioning limit, budgets and timing events are then independent of every other engine in the process (``set_option`` /
    ``get_option`` / ``contraction_last``).  The default engine runs in the library's DEFAULT context -- the one the deprecated
    process-wide setters (``lib.sgp_set_contraction`` ...) act on."""

    OPTIONS = {"contraction": _lib.OPT_CONTRACTION, "asm_overlap": _lib.OPT_ASM_OVERLAP, "kfu_budget_bytes": _lib.OPT_KFU_BUDGET_BYTES,
               "cond_limit": _lib.OPT_COND_LIMIT, "cu_budget": _lib.OPT_CU_BUDGET, "timing": _lib.OPT_TIMING,
               "shared_device": _lib.OPT_SHARED_DEVICE}

    def __init__(self, device: Optional[torch.device] = None, own_context: bool = False):
        self.lib = _lib.load_library()
        if not torch.cuda.is_available():
            raise _lib.SgpLibraryError("HipEngine needs a HIP device (torch.cuda.is_available() is False); "
                                       "there is no CPU implementation of the sparse-GP core")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._ws = {}
        self._ctx = None  # NULL = the default context
        if own_context:
            self._ctx = self.lib.sgp_ctx_create(self.device.index if self.device.index is not None else torch.cuda.current_device())
            if not self._ctx:
                raise _lib.SgpLibraryError("sgp_ctx_create failed")

    def __del__(self):
        ctx, self._ctx = getattr(self, "_ctx", None), None
        if ctx:
            try:
                self.lib.sgp_ctx_destroy(C.c_void_p(ctx))
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass

    # ------------------------------------------------------------------ context
    class _ContextFree:
        """The entry points that take no context argument (sgp_chol_lower, sgp_predict, the sgp_svgp_* and sgp_small_* families), called
        with this engine's context bound to the calling thread for the duration of the call (include/sgp.h: sgp_ctx_bind_thread), so
        that they read ITS options -- CU budget, conditioning limit, shared-device mode.  An engine on the default context calls straight
        through."""

        def __init__(self, eng):
            self._eng = eng

        def __getattr__(self, name):
            eng = self._eng
            fn = getattr(eng.lib, name)
            if not eng._ctx:
                return fn
            bind = eng.lib.sgp_ctx_bind_thread

            def call(*a):
                bind(C.c_void_p(eng._ctx))
                try:
                    return fn(*a)
                finally:
                    bind(C.c_void_p(0))
            return call

    @property
    def _cf(self):
        cf = self.__dict__.get("_cf_obj")
        if cf is None:
            cf = self.__dict__["_cf_obj"] = HipEngine._ContextFree(self)
        return cf

    def _c(self):
        return C.c_void_p(self._ctx) if self._ctx else C.c_void_p(0)

    def set_option(self, name: str, value) -> float:
        """Sets an option of this engine's context (``OPTIONS``); returns the previous value."""
        prev = self.lib.sgp_ctx_get_option(self._c(), self.OPTIONS[name])
        _lib.check("sgp_ctx_set_option", self.lib.sgp_ctx_set_option(self._c(), self.OPTIONS[name], float(value)))
        return prev

    def get_option(self, name: str) -> float:
        return self.lib.sgp_ctx_get_option(self._c(), self.OPTIONS[name])

    def contraction_last(self) -> int:
        """What this engine's last pass 1 ran: 0 fp64 matrix cores, 1 integer matrix cores."""
        return int(self.lib.sgp_ctx_contraction_last(self._c()))

    def would_use_i8(self, N: int, M: int) -> bool:
        """The contraction rule of this engine's context for an N-row shard with M inducing inputs."""
        return bool(self.lib.sgp_ctx_contraction_would_use_i8(self._c(), int(N), int(M)))

    # ------------------------------------------------------------------ helpers
    def _workspace(self, name: str, nbytes: int) -> torch.Tensor:
        buf = self._ws.get(name)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._ws[name] = buf
        return buf

    def _stream(self, stream=None):
        """The HIP stream handle of `stream` (a torch.cuda.Stream), default torch's current one on this device."""
        return C.c_void_p((stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream)

    def _chk(self, t: torch.Tensor, name: str) -> torch.Tensor:
        if t.dtype != torch.float64 or t.device != self.device or not t.is_contiguous():
            raise ValueError("%s must be a contiguous float64 tensor on %s (got %s, %s)" % (name, self.device, t.dtype, t.device))
        return t

    @staticmethod
    def hyper_len(kernel, d: int) -> int:
        """Entries of the hyper-parameter gradient: d lengthscales, or the composite kernel's parameter block."""
        return COMP_LEN if _kernel_id(kernel) == KERNEL_IDS["composite"] else d

    def _inv_ls(self, ls: Sequence[float], d: int, kernel="rbf"):
        """1 / lengthscale as the C array the entry points take.  An evaluation passes the same list to four or five calls: the last
        array is kept (the library copies it during the call)."""
        c = self.__dict__.get("_ils")
        if c is not None and type(ls) is list and c[1] == d and c[2] == kernel and c[0] == ls:
            return c[3]
        vals = [float(v) for v in (ls.tolist() if hasattr(ls, "tolist") else ls)]
        if _kernel_id(kernel) == KERNEL_IDS["composite"]:  # the parameter block travels in place of 1 / lengthscale
            if len(vals) != COMP_LEN:
                raise ValueError("a composite kernel takes a %d-entry parameter block (got %d)" % (COMP_LEN, len(vals)))
            arr = (C.c_double * COMP_LEN)(*vals)
        else:
            if len(vals) == 1 and d > 1:
                vals = vals * d
            if len(vals) != d:
                raise ValueError("lengthscale has %d entries, expected %d" % (len(vals), d))
            arr = (C.c_double * d)(*[1.0 / v for v in vals])
        if type(ls) is list:
            self._ils = (list(ls), d, kernel, arr)
        return arr

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]):
        return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    # ------------------------------------------------------------------ pass 1
    def kfu_buffer(self, N: int, M: int) -> torch.Tensor:
        """Caller-owned K'_fu block (see include/sgp.h: sgp_kfu_len) for ``suffstats(..., kfu=)``."""
        return self.empty(self.lib.sgp_kfu_len(N, M))

    def kfu_f16_buffer(self, N: int, M: int) -> torch.Tensor:
        """Caller-owned fp16 image of K'_fu (sgp_kfu_len 16-bit words) for ``suffstats_extended(..., kfu_f16=)`` / ``suffstats_bwd_lo(..., kfu_f16=)``."""
        return torch.empty(self.lib.sgp_kfu_len(N, M), dtype=torch.float16, device=self.device)

    def suffstats(self, X, y, Z, ls, sf2, kernel="rbf", out: Optional[torch.Tensor] = None,
                  kfu: Optional[torch.Tensor] = None, gate: Optional[torch.cuda.Event] = None) -> torch.Tensor:
        """Packed local statistics [Phi (M*M) | b (M) | yy | kappa] -- the buffer the all-reduce sums.

        ``kfu`` (optional, from ``kfu_buffer``) keeps the assembled kernel block for ``suffstats_bwd``.  ``gate``: an event
        already recorded on another stream that the integer-core contraction waits for (include/sgp.h: sgp_set_pass1_gate)."""
        N, d = X.shape
        M = Z.shape[0]
        self._chk(Z, "Z")
        if N > 0:
            self._chk(X, "X"), self._chk(y, "y")
        if out is None:
            out = self.empty(M * M + M + 2)
        # with a caller-owned K'_fu the library's own super-chunk (up to 16 GiB) is not part of the workspace
        nbytes = self.lib.sgp_ctx_suffstats_workspace_bytes(self._c(), N, M, d, 1 if kfu is not None else 0)
        if nbytes == 0:
            raise ValueError("unsupported shape N=%d M=%d d=%d (d <= %d, M <= %d)" % (N, M, d, _lib.SGP_MAX_DIM, _lib.SGP_MAX_INDUCING))
        ws = self._workspace("fwd_kfu" if kfu is not None else "fwd", nbytes)
        base = out.data_ptr()
        if gate is not None:
            self.lib.sgp_ctx_set_pass1_gate(self._c(), C.c_void_p(gate.cuda_event))
        st = self.lib.sgp_ctx_suffstats_fwd(
            self._c(), self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), N, M, d, _kernel_id(kernel),
            C.c_void_p(base), C.c_void_p(base + 8 * M * M), C.c_void_p(base + 8 * (M * M + M)),
            C.c_void_p(base + 8 * (M * M + M + 1)), self._ptr(kfu), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_suffstats_fwd", st)
        return out

    def suffstats_whitened(self, X, y, Z, ls, sf2, kuu_linv, kernel="rbf", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Packed local statistics in the whitened basis, [W = A A^T (M*M) | u = A y (M) | yy | kappa] with A = L^-1 K_uf
        (PyMC3's op order; ``kuu_linv`` from ``kuu_factor``).  Same layout and all-reduce as ``suffstats``."""
        N, d = X.shape
        M = Z.shape[0]
        self._chk(Z, "Z"), self._chk(kuu_linv, "kuu_linv")
        if N > 0:
            self._chk(X, "X"), self._chk(y, "y")
        if out is None:
            out = self.empty(M * M + M + 2)
        nbytes = self.lib.sgp_ctx_suffstats_whitened_workspace_bytes(self._c(), N, M, d)
        if nbytes == 0:
            raise ValueError("unsupported shape N=%d M=%d d=%d" % (N, M, d))
        ws = self._workspace("fwd_whitened", nbytes)
        base = out.data_ptr()
        st = self.lib.sgp_ctx_suffstats_fwd_whitened(
            self._c(), self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), N, M, d, _kernel_id(kernel),
            self._ptr(kuu_linv), C.c_void_p(base), C.c_void_p(base + 8 * M * M), C.c_void_p(base + 8 * (M * M + M)),
            C.c_void_p(base + 8 * (M * M + M + 1)), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_suffstats_fwd_whitened", st)
        return out

    def suffstats_whitened_rows(self, X, y, Z, ls, sf2, kuu_linv, kernel="rbf", out: Optional[torch.Tensor] = None,
                                t_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``suffstats_whitened`` for a large shard, in the streaming layout (include/sgp.h: sgp_suffstats_fwd_whitened_rows; stationary
        kernels).  ``t_out`` (from ``kfu_buffer``) keeps T = K'_fu L^-T for ``suffstats_bwd_factored(..., t_in=)``."""
        N, d = X.shape
        M = Z.shape[0]
        self._chk(Z, "Z"), self._chk(kuu_linv, "kuu_linv")
        if N > 0:
            self._chk(X, "X"), self._chk(y, "y")
        if out is None:
            out = self.empty(M * M + M + 2)
        if t_out is not None:
            self._chk(t_out, "t_out")
            if t_out.numel() < self.lib.sgp_kfu_len(N, M):
                raise ValueError("t_out holds %d doubles, sgp_kfu_len(N, M) = %d" % (t_out.numel(), self.lib.sgp_kfu_len(N, M)))
        nbytes = self.lib.sgp_ctx_suffstats_whitened_rows_workspace_bytes(self._c(), N, M, d, 1 if t_out is not None else 0)
        if nbytes == 0:
            raise ValueError("unsupported shape N=%d M=%d d=%d" % (N, M, d))
        ws = self._workspace("fwd_whitened_rows_t" if t_out is not None else "fwd_whitened_rows", nbytes)
        base = out.data_ptr()
        st = self.lib.sgp_ctx_suffstats_fwd_whitened_rows(
            self._c(), self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), N, M, d, _kernel_id(kernel),
            self._ptr(kuu_linv), C.c_void_p(base), C.c_void_p(base + 8 * M * M), C.c_void_p(base + 8 * (M * M + M)),
            C.c_void_p(base + 8 * (M * M + M + 1)), self._ptr(t_out) if t_out is not None else C.c_void_p(0), self._ptr(ws), ws.numel(),
            self._stream())
        _lib.check("sgp_suffstats_fwd_whitened_rows", st)
        return out

    def suffstats_extended(self, X, y, Z, ls, sf2, kuu_linv, kernel="rbf", out: Optional[torch.Tensor] = None,
                           kfu: Optional[torch.Tensor] = None, level: int = 1, phi_diag: Optional[torch.Tensor] = None,
                           kfu_f16: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The whitened statistics [W | u | yy | kappa] from the EXTENDED streaming order (include/sgp.h: sgp_suffstats_fwd_extended):
        Phi on the integer matrix cores to 2^-61, the triple product in double-double.  ``kfu`` (from ``kfu_buffer``) keeps the fp64
        K'_fu for ``suffstats_bwd``.  ``level`` 1: 34 digit pairs (Phi to 2^-61), 2: 39 pairs (2^-69).  Stationary kernels.
        ``phi_diag`` (M doubles): receives diag(K_uf K_fu) of this shard for ``streaming_error_report`` (ranks add theirs up).
        ``kfu_f16`` (from ``kfu_f16_buffer``, with ``kfu``): receives the fp16 image of K'_fu for ``suffstats_bwd_lo``."""
        N, d = X.shape
        M = Z.shape[0]
        self._chk(Z, "Z"), self._chk(kuu_linv, "kuu_linv")
        if N > 0:
            self._chk(X, "X"), self._chk(y, "y")
        if out is None:
            out = self.empty(M * M + M + 2)
        if kfu is not None:
            self._chk(kfu, "kfu")
            if kfu.numel() < self.lib.sgp_kfu_len(N, M):
                raise ValueError("kfu holds %d doubles, sgp_kfu_len(N, M) = %d" % (kfu.numel(), self.lib.sgp_kfu_len(N, M)))
        nbytes = self.lib.sgp_ctx_suffstats_extended_workspace_bytes(self._c(), N, M, d)
        if nbytes == 0:
            raise ValueError("unsupported shape N=%d M=%d d=%d" % (N, M, d))
        ws = self._workspace("fwd_extended", nbytes)
        base = out.data_ptr()
        if phi_diag is not None:
            self._chk(phi_diag, "phi_diag")
            if phi_diag.numel() < M:
                raise ValueError("phi_diag holds %d doubles, M = %d" % (phi_diag.numel(), M))
        if kfu_f16 is not None:
            if kfu is None or kfu_f16.dtype != torch.float16 or not kfu_f16.is_contiguous() or kfu_f16.device != self.device \
                    or kfu_f16.numel() < self.lib.sgp_kfu_len(N, M):
                raise ValueError("kfu_f16: a contiguous float16 tensor of sgp_kfu_len(N, M) elements on this device, together with kfu")
        st = self.lib.sgp_ctx_suffstats_fwd_extended_f16(
            self._c(), self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), N, M, d, _kernel_id(kernel),
            self._ptr(kuu_linv), int(level), C.c_void_p(base), C.c_void_p(base + 8 * M * M), C.c_void_p(base + 8 * (M * M + M)),
            C.c_void_p(base + 8 * (M * M + M + 1)), self._ptr(kfu) if kfu is not None else C.c_void_p(0),
            C.c_void_p(kfu_f16.data_ptr()) if kfu_f16 is not None else C.c_void_p(0),
            self._ptr(phi_diag) if phi_diag is not None else C.c_void_p(0), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_suffstats_fwd_extended", st)
        return out

    def phibar_dd(self, Cw: torch.Tensor, kuu_linv: torch.Tensor, s2: float, want_lo: bool = False):
        """Phibar = L^-T (Cw / 2 s2) L^-1 formed in double-double (include/sgp.h: sgp_phibar_dd): (leading word, trailing word or None),
        M x M each.  The leading word replaces the fp64-formed Phibar of ``bound(..., whitened=True)`` in the extended order's pass 2."""
        M = Cw.shape[0]
        self._chk(Cw, "Cw"), self._chk(kuu_linv, "kuu_linv")
        hi = self.empty(M, M)
        lo = self.empty(M, M) if want_lo else None
        ws = self._workspace("phibar_dd", self.lib.sgp_phibar_dd_workspace_bytes(M))
        st = self.lib.sgp_phibar_dd(self._ptr(Cw), self._ptr(kuu_linv), M, float(s2), self._ptr(hi), self._ptr(lo) if want_lo else C.c_void_p(0),
                                    self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_phibar_dd", st)
        return hi, lo

    def bwd_lo_supported(self, N: int, M: int, d: int, kernel="rbf") -> bool:
        """``suffstats_bwd_lo`` exists for this shape: RBF (any d the streaming kernels take)."""
        return kernel == "rbf" and self.lib.sgp_suffstats_bwd_lo_workspace_bytes(int(N), int(M), int(d)) > 0

    def suffstats_bwd_lo(self, X, y, Z, ls, sf2, Phibar_lo, kfu, grads: torch.Tensor, kernel="rbf", delta: Optional[torch.Tensor] = None,
                         kfu_f16: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Adds the trailing word's share of pass 2 into the packed gradients ``suffstats_bwd`` wrote (include/sgp.h: sgp_suffstats_bwd_lo);
        ``delta`` (d + 1 doubles) receives the correction itself.  ``kfu_f16``: the fp16 image ``suffstats_extended(..., kfu_f16=)`` left
        (the conversion pass is skipped; ``kfu`` may then be None)."""
        N, d = X.shape
        M = Z.shape[0]
        self._chk(Phibar_lo, "Phibar_lo")
        if kfu is not None:
            self._chk(kfu, "kfu")
        elif kfu_f16 is None:
            raise ValueError("suffstats_bwd_lo needs kfu or kfu_f16")
        if kfu_f16 is not None and (kfu_f16.dtype != torch.float16 or kfu_f16.device != self.device or kfu_f16.numel() < self.lib.sgp_kfu_len(N, M)):
            raise ValueError("kfu_f16: a float16 tensor of sgp_kfu_len(N, M) elements on this device")
        nh = self.hyper_len(kernel, d)
        ws = self._workspace("bwd_lo", self.lib.sgp_suffstats_bwd_lo_workspace_bytes_ex(N, M, d, 1 if kfu_f16 is not None else 0))
        base = grads.data_ptr()
        st = self.lib.sgp_suffstats_bwd_lo_f16(self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2),
                                               self._ptr(Phibar_lo), self._ptr(kfu) if kfu is not None else C.c_void_p(0),
                                               C.c_void_p(kfu_f16.data_ptr()) if kfu_f16 is not None else C.c_void_p(0), N, M, d,
                                               _kernel_id(kernel), C.c_void_p(base), C.c_void_p(base + 8 * nh),
                                               self._ptr(delta) if delta is not None else C.c_void_p(0), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_suffstats_bwd_lo", st)
        return grads

    def pack_lower(self, stats: torch.Tensor, M: int) -> torch.Tensor:
        """[lower triangle of Phi | b | yy | kappa]: what crosses xGMI (half the bytes of ``stats``)."""
        tri = self.empty(self.lib.sgp_stats_packed_len(M))
        _lib.check("sgp_stats_pack_lower", self.lib.sgp_stats_pack_lower(self._ptr(stats), M, self._ptr(tri), self._stream()))
        return tri

    def unpack_lower(self, tri: torch.Tensor, M: int, stats: torch.Tensor) -> torch.Tensor:
        _lib.check("sgp_stats_unpack_lower", self.lib.sgp_stats_unpack_lower(self._ptr(tri), M, self._ptr(stats), self._stream()))
        return stats

    def kuu(self, Z, ls, sf2, jitter, kernel="rbf", out: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
        """K_uu + jitter I.  ``stream``: enqueue there instead of on torch's current stream (the caller orders it against the producers /
        consumers of Z and ``out`` itself)."""
        M, d = Z.shape
        self._chk(Z, "Z")
        K = out if out is not None else self.empty(M, M)
        st = self.lib.sgp_kuu(self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), float(jitter), M, d, _kernel_id(kernel),
                              self._ptr(K), self._stream(stream))
        _lib.check("sgp_kuu", st)
        return K

    # ------------------------------------------------------------------ tail
    def result_buffer(self, extra: int = 0):
        """One allocation for everything the host reads back after an evaluation: (buf, out, info) with
        buf = [out (OUT_LEN doubles) | status word (int32, in the low half of one double) | streaming-order estimate | its upper bound |
               pad | extra doubles]
        (RESULT_HEAD doubles ahead of the extras: they start 16-byte aligned, collectives run on that slice), so a single
        device-to-host copy of ``buf`` -- and no cast / concatenate launches -- ends the evaluation."""
        buf = self.empty(RESULT_HEAD + extra)
        return buf, buf[:OUT_LEN], buf[OUT_LEN:OUT_LEN + 1].view(torch.int32)[:1]

    @staticmethod
    def read_result(host_buf):
        """(out, info) from a host copy of a ``result_buffer``; out as plain floats (indexing the tensor costs ~1.5 us per element)."""
        return host_buf[:OUT_LEN].tolist(), int(host_buf.numpy()[OUT_LEN:OUT_LEN + 1].view("int32")[0])

    def kuu_factor(self, Kuu, info: Optional[torch.Tensor] = None, trace_out: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, stream=None):
        """Padded L^-1 of chol(Kuu) and its info flag; independent of the streamed statistics (side-stream work).
        ``trace_out`` (``sgp_kuu_inverse_trace_len()`` doubles): also tr(K_uu^-1) in trace_out[0], from the call's own last launch
        (include/sgp.h: sgp_kuu_factor_ex; the same bits as ``kuu_inverse_trace``).  ``out`` (``sgp_kuu_factor_len(M)`` doubles) /
        ``stream``: a buffer and a stream of the caller's, as for ``kuu``."""
        M = Kuu.shape[0]
        Linv = out if out is not None else self.empty(self.lib.sgp_kuu_factor_len(M))
        if info is None:
            info = torch.empty(1, dtype=torch.int32, device=self.device)  # cleared by the call itself
        ws = self._workspace("kuu_factor", self.lib.sgp_kuu_factor_workspace_bytes(M))
        st = self.lib.sgp_ctx_kuu_factor_ex(self._c(), self._ptr(Kuu), M, self._ptr(Linv), self._ptr(info), self._ptr(trace_out), self._ptr(ws),
                                            ws.numel(), self._stream(stream))
        _lib.check("sgp_kuu_factor_ex", st)
        return Linv, info

    def kuu_inverse_trace(self, Linv, M: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tr(K_uu^-1) = ||L^-1||_F^2 in out[0] (the other entries are scratch) from ``kuu_factor``'s output: the ingredient of the
        streaming-order guard (include/sgp.h: sgp_streaming_error_estimate) that does not depend on the streamed statistics."""
        if out is None:
            out = self.empty(self.lib.sgp_kuu_inverse_trace_len())
        _lib.check("sgp_kuu_inverse_trace", self.lib.sgp_kuu_inverse_trace(self._ptr(Linv), int(M), self._ptr(out), self._stream()))
        return out

    def streaming_error_estimate(self, packed, trace, s2, N, M, result) -> None:
        """Writes the first-order estimate of |dF| / N of the streaming order into the pad word of ``result`` (a
        ``result_buffer``): it comes back with the evaluation's one host copy (``read_estimate``)."""
        buf = result[0]
        est = C.c_void_p(buf.data_ptr() + 8 * (OUT_LEN + 1))
        _lib.check("sgp_streaming_error_estimate",
                   self.lib.sgp_streaming_error_estimate(self._ptr(packed), self._ptr(trace), float(s2), int(N), int(M), est, self._stream()))

    def streaming_error_bound(self, trace, sf2, s2, result) -> None:
        """The estimate's upper bound (max Phi_ii <= N sf2^2) into the pad word of ``result``: what a whitened-order evaluation can
        still report (include/sgp.h: sgp_streaming_error_bound)."""
        est = C.c_void_p(result[0].data_ptr() + 8 * (OUT_LEN + 1))
        _lib.check("sgp_streaming_error_bound", self.lib.sgp_streaming_error_bound(self._ptr(trace), float(sf2), float(s2), est, self._stream()))

    def streaming_error_report(self, diag, stride, trace, sf2, s2, N, M, result) -> None:
        """Estimate AND upper bound into the two words behind the status word of ``result`` (include/sgp.h:
        sgp_streaming_error_report).  ``diag``: the (all-reduced) packed statistics of the streaming order with ``stride`` M + 1, the
        (all-reduced) ``phi_diag`` of the extended order with ``stride`` 1, or None (whitened order: the estimate IS the bound)."""
        est = C.c_void_p(result[0].data_ptr() + 8 * (OUT_LEN + 1))
        _lib.check("sgp_streaming_error_report",
                   self.lib.sgp_streaming_error_report(self._ptr(diag) if diag is not None else C.c_void_p(0), int(stride), self._ptr(trace),
                                                       float(sf2), float(s2), int(N), int(M), est, self._stream()))

    @staticmethod
    def read_estimate(host_buf) -> float:
        return float(host_buf[OUT_LEN + 1])

    @staticmethod
    def read_bound(host_buf) -> float:
        return float(host_buf[OUT_LEN + 2])

    def bound(self, Kuu, packed, s2, N, with_adjoints=False, want_factors=False, kuu_linv=None, kuu_info=None, result=None,
              whitened=False, want_cw=False):
        """Runs the O(M^3) tail on (already all-reduced) packed statistics.

        Returns dict(out=[8] device tensor, info=int32 device tensor, buf=the ``result_buffer`` both live in, and
        when asked Phibar, bbar, Kuubar, factors; ``want_cw`` (whitened order, with adjoints) adds Cw, the whitened core
        of the adjoint that ``suffstats_bwd_factored`` takes).  Nothing is synchronised.  With ``kuu_linv`` (from ``kuu_factor``)
        the status word must already hold that call's status: pass ``result`` whose info word ``kuu_factor`` wrote,
        or ``kuu_info`` (copied in with one tiny launch).
        """
        M = Kuu.shape[0]
        self._chk(Kuu, "Kuu"), self._chk(packed, "packed")
        if whitened and kuu_linv is None:
            raise ValueError("bound(whitened=True) needs kuu_linv (the factor the statistics were whitened with)")
        buf, out, info = result if result is not None else self.result_buffer()
        if kuu_linv is not None:
            if kuu_info is not None:
                info.copy_(kuu_info)
            elif result is None:
                raise ValueError("bound(kuu_linv=...) needs the status of kuu_factor: kuu_info= or result=")
        res = {"out": out, "info": info, "buf": buf}
        Phibar = bbar = Kuubar = factors = None
        if with_adjoints:
            Phibar, bbar, Kuubar = self.empty(M, M), self.empty(M), self.empty(M, M)
            res.update(Phibar=Phibar, bbar=bbar, Kuubar=Kuubar)
        if want_factors:
            factors = self.empty(self.lib.sgp_bound_factors_len(M))
            res["factors"] = factors
        nbytes = self.lib.sgp_bound_workspace_bytes(M, 1 if with_adjoints else 0)
        ws = self._workspace("bound", nbytes)
        base = packed.data_ptr()
        stats = (C.c_void_p(base), C.c_void_p(base + 8 * M * M), C.c_void_p(base + 8 * (M * M + M)),
                 C.c_void_p(base + 8 * (M * M + M + 1)))
        tail = (float(s2), int(N), M, 1 if with_adjoints else 0, self._ptr(out), self._ptr(Phibar), self._ptr(bbar),
                self._ptr(Kuubar), self._ptr(factors), self._ptr(kuu_linv), self._ptr(info), self._ptr(ws), ws.numel(), self._stream())
        if whitened:  # (Cw = NULL: the plain whitened bound)
            if want_cw and with_adjoints:
                res["Cw"] = self.empty(M, M)
            _lib.check("sgp_bound_from_whitened_stats",
                       self.lib.sgp_ctx_bound_from_whitened_stats(self._c(), *stats, *tail[:-3], self._ptr(res.get("Cw")), *tail[-3:]))
        else:
            _lib.check("sgp_bound_from_stats", self.lib.sgp_ctx_bound_from_stats(self._c(), self._ptr(Kuu), *stats, *tail))
        return res

    # ------------------------------------------------------------------ single-launch path for small problems
    def small_supported(self, N: int, M: int, d: int, kernel="rbf") -> bool:
        return bool(self._cf.sgp_small_supported(int(N), int(M), int(d), _kernel_id(kernel)))

    def _small_ws(self, N, M, d):
        nbytes = self.lib.sgp_small_workspace_bytes(N, M, d)
        if nbytes == 0:
            raise ValueError("shape N=%d M=%d d=%d is outside the single-launch path" % (N, M, d))
        # one workspace per HIP stream: the launch's sync words live in it, so two evaluations in flight on different streams
        # of one engine must not share them
        key = ("small", int(torch.cuda.current_stream(self.device).cuda_stream) if self.device.type == "cuda" else 0)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.device)  # zeroed once: the sync words
            self._ws[key] = ws
        return ws

    @staticmethod
    def _composite_args(composite, hmc: bool):
        """(structure, n_free, slots, roles, sds) ctypes arguments of the *_composite entry points (host arrays).
        ``composite`` = {"structure": the SGP_COMP_LEN parameter block, "free": [(slot, role 0 amp / 1 ls / 2 aux, prior sd)]}."""
        blk = [float(v) for v in composite["structure"]]
        if len(blk) != _lib.COMP_LEN:
            raise ValueError("composite structure must have %d entries" % _lib.COMP_LEN)
        free = list(composite.get("free") or []) if hmc else []
        n = len(free)
        return ((C.c_double * len(blk))(*blk), n, (C.c_int * max(n, 1))(*[int(f[0]) for f in free]),
                (C.c_int * max(n, 1))(*[int(f[1]) for f in free]), (C.c_double * max(n, 1))(*[float(f[2]) for f in free]))

    @staticmethod
    def small_out_len(d: int, mode: int = 0, composite=None) -> int:
        """Entries of the result of one evaluation: [value | kernel hyper-parameter gradients | noise | logmarg | trace]."""
        if composite is None:
            return d + 5
        return (len(composite["free"]) if mode else _lib.COMP_LEN) + 4

    def small_eval(self, X, y, Z, theta: torch.Tensor, jitter, kernel="rbf", mode=0, want_grad=True, want_gz=False,
                   out: Optional[torch.Tensor] = None, composite=None):
        """ONE kernel launch: the bound (mode 0) or the NUTS target (mode 1) and its gradient, hyper-parameters read from
        the device tensor ``theta``.  Returns (out device tensor, gZ or None, info int32 device tensor); nothing is
        synchronised.  ``out`` may be a caller-owned buffer of ``small_out_len`` (+1 for the status word, see
        ``small_result``) entries.  ``kernel="composite"``: ``composite`` carries the structure (see ``_composite_args``);
        theta = [parameter block | s2] (mode 0) or [log free parameters | log sigma] (mode 1); no dF/dZ."""
        N, d = X.shape
        M = Z.shape[0]
        for t, n in ((X, "X"), (y, "y"), (Z, "Z"), (theta, "theta")):
            self._chk(t, n)
        is_comp = _kernel_id(kernel) == _kernel_id("composite")
        if is_comp != (composite is not None):
            raise ValueError("composite= goes with kernel='composite'")
        nout = self.small_out_len(d, mode, composite)
        if theta.numel() != nout - 3:
            raise ValueError("theta has %d entries, expected %d" % (theta.numel(), nout - 3))
        ws = self._small_ws(N, M, d)
        if out is None:
            out, info = self.small_result(nout - 5)
        else:
            info = out[nout:nout + 1].view(torch.int32)[:1]
        if is_comp:
            if want_gz:
                raise ValueError("the single-launch path has no dF/dZ for composite kernels")
            cargs = self._composite_args(composite, bool(mode))
            st = self._cf.sgp_small_eval_composite(self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._ptr(theta), *cargs, N, M, d,
                                                   float(jitter), int(mode), 1 if want_grad else 0, self._ptr(out),
                                                   C.c_void_p(info.data_ptr()), self._ptr(ws), ws.numel(), self._stream())
            _lib.check("sgp_small_eval_composite", st)
            return out, None, info
        gz = self.empty(M, d) if (want_grad and want_gz) else None
        st = self._cf.sgp_small_eval(self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._ptr(theta), N, M, d, _kernel_id(kernel),
                                     float(jitter), int(mode), 1 if want_grad else 0, self._ptr(out), self._ptr(gz),
                                     C.c_void_p(info.data_ptr()), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_small_eval", st)
        return out, gz, info

    def small_eval_batch(self, X, y, Z, thetas: torch.Tensor, jitter, kernel="rbf", mode=0, want_grad=True, want_gz=False):
        """S evaluations (``thetas``: S x (d + 2) on the device) in one launch.  Returns (outs [S, d + 5], gZ [S, M, d] or None,
        infos [S] int32), device tensors; nothing is synchronised."""
        N, d = X.shape
        M = Z.shape[0]
        for t, n in ((X, "X"), (y, "y"), (Z, "Z"), (thetas, "thetas")):
            self._chk(t, n)
        S = thetas.shape[0]
        if thetas.shape[1] != d + 2:
            raise ValueError("thetas must be S x (d + 2)")
        ws = self._small_ws(N, M, d)
        outs = self.empty(S, d + 5)
        gz = self.empty(S, M, d) if (want_grad and want_gz) else None
        infos = torch.zeros(S, dtype=torch.int32, device=self.device)
        scratch = self.empty(d + 2)
        st = self._cf.sgp_small_eval_batch(self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._ptr(thetas), S, N, M, d,
                                           _kernel_id(kernel), float(jitter), int(mode), 1 if want_grad else 0, self._ptr(scratch),
                                           self._ptr(outs), self._ptr(gz), C.c_void_p(infos.data_ptr()), self._ptr(ws), ws.numel(),
                                           self._stream())
        _lib.check("sgp_small_eval_batch", st)
        return outs, gz, infos

    def small_nuts(self, X, y, Z, q0, n_tune, n_draws, seed, jitter=1e-6, kernel="rbf", max_treedepth=10, step_scale=0.25,
                   target_accept=0.8, composite=None):
        """The whole NUTS run in one persistent launch (sgp_small_nuts / sgp_small_nuts_composite).  Returns dict(samples
        [n_draws, ndim] unconstrained, stats [n_draws, 8], seconds [n_draws], evaluations, draws, info) as host tensors /
        ints (synchronises).  ndim = d + 2, or the composite kernel's free parameters + 1."""
        N, d = X.shape
        M = Z.shape[0]
        for t, n in ((X, "X"), (y, "y"), (Z, "Z")):
            self._chk(t, n)
        is_comp = _kernel_id(kernel) == _kernel_id("composite")
        if is_comp != (composite is not None):
            raise ValueError("composite= goes with kernel='composite'")
        nout = self.small_out_len(d, 1, composite)
        nd = nout - 3
        q0d = torch.as_tensor([float(v) for v in q0], dtype=torch.float64).to(self.device)
        if q0d.numel() != nd:
            raise ValueError("q0 has %d entries, expected %d" % (q0d.numel(), nd))
        ws = self._small_ws(N, M, d)
        cols = int(self.lib.sgp_small_nuts_stat_cols())
        samples = self.empty(n_draws, nd)
        stats = torch.zeros(n_draws * cols, dtype=torch.float64, device=self.device)
        counters = torch.zeros(2, dtype=torch.int64, device=self.device)
        theta, out = self.empty(nd), self.empty(nout)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        head = (self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._ptr(q0d))
        tail = (float(jitter), int(n_tune), int(n_draws), int(max_treedepth), float(step_scale), float(target_accept),
                int(seed) & ((1 << 64) - 1), self._ptr(theta), self._ptr(samples), self._ptr(stats),
                C.c_void_p(counters.data_ptr()), self._ptr(out), C.c_void_p(info.data_ptr()), self._ptr(ws), ws.numel(),
                self._stream())
        if is_comp:
            cargs = self._composite_args(composite, True)
            _lib.check("sgp_small_nuts_composite", self._cf.sgp_small_nuts_composite(*head, *cargs, N, M, d, *tail))
        else:
            _lib.check("sgp_small_nuts", self._cf.sgp_small_nuts(*head, N, M, d, _kernel_id(kernel), *tail))
        h = stats.to("cpu")
        c = counters.to("cpu")
        res = {"samples": samples.to("cpu"), "stats": h[: n_draws * (cols - 1)].reshape(n_draws, cols - 1),
               "seconds": h[n_draws * (cols - 1):], "evaluations": int(c[0]), "draws": int(c[1]), "info": int(info.to("cpu")[0])}
        if res["info"] < 0:
            self.small_reset()
        return res

    def small_result(self, d: int):
        """[out (d + 5) | status word] in one buffer, so one device-to-host copy ends an evaluation."""
        buf = self.empty(d + 6)
        return buf, buf[d + 5:d + 6].view(torch.int32)[:1]

    def small_reset(self):
        """Zero the sync words again (only needed after a SGP_INFO_TIMEOUT)."""
        for key, ws in self._ws.items():
            if isinstance(key, tuple) and key[0] == "small":
                ws[: self.lib.sgp_small_sync_bytes()].zero_()

    # ------------------------------------------------------------------ pass 2
    def suffstats_bwd(self, X, y, Z, ls, sf2, Phibar, bbar, kappabar, kernel="rbf", want_gz=False,
                      out: Optional[torch.Tensor] = None, kfu: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Packed local gradients [g_hyper (d lengthscales, or the composite block) | g_sf2 | g_Z (M*d, only when want_gz)]."""
        N, d = X.shape
        M = Z.shape[0]
        nh = self.hyper_len(kernel, d)
        if out is None:
            out = self.empty(nh + 1 + (M * d if want_gz else 0))
        nbytes = self.lib.sgp_suffstats_bwd_workspace_bytes_ex(N, M, d, 1 if kfu is not None else 0)
        ws = self._workspace("bwd_kfu" if kfu is not None else "bwd", nbytes)
        base = out.data_ptr()
        st = self.lib.sgp_ctx_suffstats_bwd(
            self._c(), self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), self._ptr(Phibar),
            self._ptr(bbar), float(kappabar), self._ptr(kfu), N, M, d, _kernel_id(kernel), C.c_void_p(base),
            C.c_void_p(base + 8 * nh),
            C.c_void_p(base + 8 * (nh + 1)) if want_gz else C.c_void_p(0), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_suffstats_bwd", st)
        return out

    def suffstats_bwd_factored(self, X, y, Z, ls, sf2, kuu_linv, Cw, s2, bbar, kappabar, kernel="rbf", want_gz=False,
                               out: Optional[torch.Tensor] = None, t_in: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``suffstats_bwd`` from the factored adjoint 2 s2 Phibar = L^-T Cw L^-1 (``kuu_linv`` from ``kuu_factor``, ``Cw``
        from ``bound(..., whitened=True, want_cw=True)``): same packed gradients, without the cancellation of an explicit
        Phibar on ill-conditioned K_uu.  ``t_in``: T = K'_fu L^-T kept by ``suffstats_whitened_rows(..., t_out=)``."""
        N, d = X.shape
        M = Z.shape[0]
        nh = self.hyper_len(kernel, d)
        if out is None:
            out = self.empty(nh + 1 + (M * d if want_gz else 0))
        if t_in is not None:
            self._chk(t_in, "t_in")
        nbytes = self.lib.sgp_ctx_suffstats_bwd_factored_workspace_bytes(self._c(), N, M, d, 1 if t_in is not None else 0)
        ws = self._workspace("bwd_factored_t" if t_in is not None else "bwd_factored", nbytes)
        base = out.data_ptr()
        st = self.lib.sgp_ctx_suffstats_bwd_factored(
            self._c(), self._ptr(X), d, self._ptr(y), self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), self._ptr(kuu_linv),
            self._ptr(Cw), float(s2), self._ptr(bbar), float(kappabar), N, M, d, _kernel_id(kernel),
            self._ptr(t_in) if t_in is not None else C.c_void_p(0), C.c_void_p(base),
            C.c_void_p(base + 8 * nh), C.c_void_p(base + 8 * (nh + 1)) if want_gz else C.c_void_p(0), self._ptr(ws), ws.numel(),
            self._stream())
        _lib.check("sgp_suffstats_bwd_factored_ex", st)
        return out

    def kuu_bwd(self, Z, ls, sf2, Kuubar, grads: torch.Tensor, kernel="rbf", want_gz=False) -> torch.Tensor:
        """Adds the Kuu path into the packed gradient buffer produced by ``suffstats_bwd``."""
        M, d = Z.shape
        nbytes = self.lib.sgp_kuu_bwd_workspace_bytes(M, d)
        ws = self._workspace("kuu_bwd", nbytes)
        base = grads.data_ptr()
        nh = self.hyper_len(kernel, d)
        st = self.lib.sgp_kuu_bwd(
            self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), self._ptr(Kuubar), M, d, _kernel_id(kernel),
            C.c_void_p(base), C.c_void_p(base + 8 * nh), C.c_void_p(base + 8 * (nh + 1)) if want_gz else C.c_void_p(0),
            self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_kuu_bwd", st)
        return grads

    # ------------------------------------------------------------------ predictive
    def predict(self, Xs, Z, ls, sf2, s2, factors, kernel="rbf", pred_noise=True, full_cov=False):
        T, d = Xs.shape
        M = Z.shape[0]
        self._chk(Xs, "Xs"), self._chk(factors, "factors")
        mean, var = self.empty(T), self.empty(T)
        cov = self.empty(T, T) if full_cov else None
        nbytes = self.lib.sgp_predict_workspace_bytes(T, M, d, 1 if full_cov else 0)
        ws = self._workspace("predict", nbytes)
        st = self._cf.sgp_predict(
            self._ptr(Xs), d, T, self._ptr(Z), d, self._inv_ls(ls, d, kernel), float(sf2), float(s2), self._ptr(factors), M, d,
            _kernel_id(kernel), 1 if pred_noise else 0, self._ptr(mean), self._ptr(var), self._ptr(cov),
            self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_predict", st)
        return mean, var, cov

    # ------------------------------------------------------------------ SVGP minibatch bound
    def svgp_elbo(self, Xb, yb, Z, ls, sf2, s2, m, LS, N_total, jitter=1e-6, kernel="rbf", likelihood="gaussian",
                  with_grads=False):
        """One minibatch of the whitened SVGP bound.  Returns dict(out=[elbo/datum, sum ELL, KL], info, and with
        ``with_grads`` g_m, g_LS (lower), g_Z, g_ls, g_sf2, g_s2 as device tensors).  Nothing is synchronised."""
        B, d = Xb.shape
        M = Z.shape[0]
        for t, n in ((Xb, "Xb"), (yb, "yb"), (Z, "Z"), (m, "m"), (LS, "LS")):
            self._chk(t, n)
        lik = {"gaussian": 0, "bernoulli": 1, "bernoulli_probit": 1}[likelihood]
        out = self.empty(3)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        res = {"out": out, "info": info}
        g = {}
        if with_grads:
            g = {"g_m": self.empty(M), "g_LS": self.empty(M, M), "g_Z": self.empty(M, d), "g_ls": self.empty(d),
                 "g_sf2": self.empty(1), "g_s2": self.empty(1)}
            res.update(g)
        nbytes = self.lib.sgp_svgp_workspace_bytes(B, M, d)
        if nbytes == 0:
            raise ValueError("unsupported SVGP shape B=%d M=%d d=%d" % (B, M, d))
        ws = self._workspace("svgp", nbytes)
        st = self._cf.sgp_svgp_elbo(
            self._ptr(Xb), d, self._ptr(yb), B, self._ptr(Z), d, self._inv_ls(ls, d), float(sf2), float(s2), float(jitter),
            self._ptr(m), self._ptr(LS), int(N_total), M, d, _kernel_id(kernel), lik, 1 if with_grads else 0, self._ptr(out),
            self._ptr(g.get("g_m")), self._ptr(g.get("g_LS")), self._ptr(g.get("g_Z")), self._ptr(g.get("g_ls")),
            self._ptr(g.get("g_sf2")), self._ptr(g.get("g_s2")), self._ptr(info), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_svgp_elbo", st)
        return res

    def svgp_elbo_batch(self, Xb, yb, Z, ls, sf2, s2, m, LS, N_total, jitter=1e-6, kernel="rbf", likelihood="gaussian",
                        with_grads=False, defer_reverse=False):
        """The bound of one minibatch at S hyper-parameter samples in ONE chain of launches (sgp_svgp_elbo_batch).
        ls: S x d, sf2, s2: S (host sequences / arrays).  Returns dict(out [S, 4] = [bound per datum | sum E log p | KL | status], info [S], and with ``with_grads``
        g_m [S, M], g_LS [S, M, M], g_Z [S, M, d], g_ls [S, d], g_sf2 [S], g_s2 [S]); device tensors, nothing synchronised."""
        import ctypes
        B, d = Xb.shape
        M = Z.shape[0]
        for t, n in ((Xb, "Xb"), (yb, "yb"), (Z, "Z"), (m, "m"), (LS, "LS")):
            self._chk(t, n)
        lsv = [[float(v) for v in row] for row in ls]
        S = len(lsv)
        if any(len(row) != d for row in lsv) or len(sf2) != S or len(s2) != S:
            raise ValueError("ls must be S x d, sf2 and s2 of length S")
        lik = {"gaussian": 0, "bernoulli": 1, "bernoulli_probit": 1}[likelihood]
        inv = (ctypes.c_double * (S * d))(*[1.0 / v for row in lsv for v in row])
        sf2c = (ctypes.c_double * S)(*[float(v) for v in sf2])
        s2c = (ctypes.c_double * S)(*[float(v) for v in s2])
        out = self.empty(S, 4)
        info = torch.empty(S, dtype=torch.int32, device=self.device)
        res = {"out": out, "info": info}
        g = {}
        if with_grads:
            g = {"g_m": self.empty(S, M), "g_LS": self.empty(S, M, M), "g_Z": self.empty(S, M, d), "g_ls": self.empty(S, d),
                 "g_sf2": self.empty(S), "g_s2": self.empty(S)}
            res.update(g)
        nbytes = self.lib.sgp_svgp_batch_workspace_bytes(B, M, d, S)
        if nbytes == 0:
            raise ValueError("unsupported SVGP batch shape B=%d M=%d d=%d S=%d (S <= 8)" % (B, M, d, S))
        ws = self._workspace("svgp_batch", nbytes)
        if with_grads and defer_reverse:
            # two halves: the forward now; res["reverse"]() enqueues the reverse chain (the caller copies the bounds out in
            # between, so that they reach the host while the device is still busy with the gradients)
            head = (self._ptr(Xb), d, self._ptr(yb), B, self._ptr(Z), d, S, inv, sf2c, s2c, float(jitter), self._ptr(m), self._ptr(LS),
                    int(N_total), M, d, _kernel_id(kernel), lik)
            st = self._cf.sgp_svgp_elbo_batch_forward(*head, self._ptr(out), self._ptr(g["g_s2"]), self._ptr(info), self._ptr(ws),
                                                      ws.numel(), self._stream())
            _lib.check("sgp_svgp_elbo_batch_forward", st)
            keep = (Xb, yb, Z, m, LS, ws)  # the buffers the deferred call reads

            def reverse():
                st2 = self._cf.sgp_svgp_elbo_batch_reverse(*head, self._ptr(g["g_m"]), self._ptr(g["g_LS"]), self._ptr(g["g_Z"]),
                                                           self._ptr(g["g_ls"]), self._ptr(g["g_sf2"]), self._ptr(keep[5]), keep[5].numel(),
                                                           self._stream())
                _lib.check("sgp_svgp_elbo_batch_reverse", st2)

            res["reverse"] = reverse
            return res
        st = self._cf.sgp_svgp_elbo_batch(
            self._ptr(Xb), d, self._ptr(yb), B, self._ptr(Z), d, S, inv, sf2c, s2c, float(jitter), self._ptr(m), self._ptr(LS),
            int(N_total), M, d, _kernel_id(kernel), lik, 1 if with_grads else 0, self._ptr(out),
            self._ptr(g.get("g_m")), self._ptr(g.get("g_LS")), self._ptr(g.get("g_Z")), self._ptr(g.get("g_ls")),
            self._ptr(g.get("g_sf2")), self._ptr(g.get("g_s2")), self._ptr(info), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_svgp_elbo_batch", st)
        return res

    def svgp_predict_batch(self, Xs, Z, ls, sf2, m, LS, jitter=1e-6, kernel="rbf"):
        """q(f*) at the rows of Xs for S hyper-parameter samples (ls: S x d, sf2: S), eight per chain of launches.
        Returns (mean [S, T], var [S, T], info [S] int32) on the device."""
        import ctypes
        T, d = Xs.shape
        M = Z.shape[0]
        lsv = [[float(v) for v in row] for row in ls]
        S = len(lsv)
        mean, var = self.empty(S, T), self.empty(S, T)
        info = torch.zeros(S, dtype=torch.int32, device=self.device)
        # test rows in chunks: the library carves S x 6 x Mp x Tp doubles for its "minibatch" -- 10 GB at S = 8,
        # M = 512, T = 50k in one piece, and T > 2^20 is refused outright; 8192 rows at a time as sgp_mixture_predict does
        TC = self.SVGP_PREDICT_CHUNK
        for s0 in range(0, S, 8):
            n = min(8, S - s0)
            inv = (ctypes.c_double * (n * d))(*[1.0 / v for row in lsv[s0:s0 + n] for v in row])
            sf2c = (ctypes.c_double * n)(*[float(v) for v in sf2[s0:s0 + n]])
            for t0 in range(0, T, TC):
                tn = min(TC, T - t0)
                nbytes = self.lib.sgp_svgp_batch_workspace_bytes(tn, M, d, n)
                if nbytes == 0:
                    raise ValueError("unsupported SVGP predictive shape T=%d M=%d d=%d" % (T, M, d))
                ws = self._workspace("svgp_batch", nbytes)
                whole = tn == T
                mc, vc = (mean[s0:s0 + n], var[s0:s0 + n]) if whole else (self.empty(n, tn), self.empty(n, tn))
                ic = info[s0:s0 + n] if whole else torch.empty(n, dtype=torch.int32, device=self.device)
                st = self._cf.sgp_svgp_predict_batch(self._ptr(Xs[t0:t0 + tn]), d, tn, self._ptr(Z), d, n, inv, sf2c, float(jitter),
                                                     self._ptr(m), self._ptr(LS), M, d, _kernel_id(kernel), self._ptr(mc), self._ptr(vc),
                                                     C.c_void_p(ic.data_ptr()), self._ptr(ws), ws.numel(), self._stream())
                _lib.check("sgp_svgp_predict_batch", st)
                if not whole:
                    mean[s0:s0 + n, t0:t0 + tn] = mc
                    var[s0:s0 + n, t0:t0 + tn] = vc
                    info[s0:s0 + n] = torch.where(info[s0:s0 + n] != 0, info[s0:s0 + n], ic)  # the first failure stays
        return mean, var, info

    SVGP_PREDICT_CHUNK = 8192  # test rows per sgp_svgp_predict_batch call

    def svgp_batch_combine(self, res, weights):
        """Reverse pass of sum_s weights[s] * bound_s from a ``svgp_elbo_batch(..., with_grads=True)`` result, one launch.
        Returns (g_m [M], g_LS [M, M], g_Z [M, d], g_theta [S, d + 2] = w_s [d/dsf2 | d/dls | d/ds2]) on the device."""
        import ctypes
        S, M, d = res["g_Z"].shape
        w = (ctypes.c_double * S)(*[float(v) for v in weights])
        gm, gLS, gZ, gth = self.empty(M), self.empty(M, M), self.empty(M, d), self.empty(S, d + 2)
        st = self.lib.sgp_svgp_batch_combine(S, w, M, d, self._ptr(res["g_m"]), self._ptr(res["g_LS"]), self._ptr(res["g_Z"]),
                                             self._ptr(res["g_ls"]), self._ptr(res["g_sf2"]), self._ptr(res["g_s2"]), self._ptr(gm),
                                             self._ptr(gLS), self._ptr(gZ), self._ptr(gth), self._stream())
        _lib.check("sgp_svgp_batch_combine", st)
        return gm, gLS, gZ, gth

    MIXTURE_BATCH = 8  # hyper-parameter samples per sgp_mixture_predict call

    def mixture_predict(self, X, y, Xs, Z, ls, sf2, s2, jitter=1e-6, kernel="rbf", pred_noise=True, full_cov=False, gate_jitter=None):
        """Predictive of the collapsed bound at S hyper-parameter samples (ls: S x d, sf2, s2: S host sequences), eight samples
        per chain of launches.  Returns dict(mean [S, T], var [S, T], cov [S, T, T] or None, info [S] int32, gate [S] int32 or
        None: status of cholesky(cov + gate_jitter I), the reference's PSD gate) -- device tensors, nothing synchronised."""
        import ctypes
        N, d = X.shape
        T = Xs.shape[0]
        M = Z.shape[0]
        for t, n in ((X, "X"), (y, "y"), (Xs, "Xs"), (Z, "Z")):
            self._chk(t, n)
        lsv = [[float(v) for v in row] for row in ls]
        S = len(lsv)
        if any(len(row) != d for row in lsv) or len(sf2) != S or len(s2) != S:
            raise ValueError("ls must be S x d, sf2 and s2 of length S")
        want_gate = full_cov and gate_jitter is not None
        mean, var = self.empty(S, T), self.empty(S, T)
        cov = self.empty(S, T, T) if full_cov else None
        info = torch.empty(S, dtype=torch.int32, device=self.device)
        gate = torch.empty(S, dtype=torch.int32, device=self.device) if want_gate else None
        for s0 in range(0, S, self.MIXTURE_BATCH):
            n = min(self.MIXTURE_BATCH, S - s0)
            nbytes = self.lib.sgp_mixture_predict_workspace_bytes(N, T, M, d, n, 1 if full_cov else 0, 1 if want_gate else 0)
            if nbytes == 0:
                raise ValueError("unsupported mixture-predictive shape N=%d T=%d M=%d d=%d (full covariance needs T <= 8192)" % (N, T, M, d))
            ws = self._workspace("mixture", nbytes)
            inv = (ctypes.c_double * (n * d))(*[1.0 / v for row in lsv[s0:s0 + n] for v in row])
            sf2c = (ctypes.c_double * n)(*[float(v) for v in sf2[s0:s0 + n]])
            s2c = (ctypes.c_double * n)(*[float(v) for v in s2[s0:s0 + n]])
            st = self.lib.sgp_ctx_mixture_predict(
                self._c(), self._ptr(X), d, self._ptr(y), N, self._ptr(Xs), d, T, self._ptr(Z), d, n, inv, sf2c, s2c, float(jitter), M, d,
                _kernel_id(kernel), 1 if pred_noise else 0, float(gate_jitter) if want_gate else 0.0, self._ptr(mean[s0:]), self._ptr(var[s0:]),
                self._ptr(cov[s0:]) if full_cov else None, C.c_void_p(info[s0:].data_ptr()),
                C.c_void_p(gate[s0:].data_ptr()) if want_gate else None, self._ptr(ws), ws.numel(), self._stream())
            _lib.check("sgp_mixture_predict", st)
        return {"mean": mean, "var": var, "cov": cov, "info": info, "gate": gate}

    def svgp_predict(self, Xs, Z, ls, sf2, m, LS, jitter=1e-6, kernel="rbf"):
        T, d = Xs.shape
        M = Z.shape[0]
        mean, var = self.empty(T), self.empty(T)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        ws = self._workspace("svgp", self.lib.sgp_svgp_workspace_bytes(T, M, d))
        st = self._cf.sgp_svgp_predict(self._ptr(Xs), d, T, self._ptr(Z), d, self._inv_ls(ls, d), float(sf2), float(jitter),
                                       self._ptr(m), self._ptr(LS), M, d, _kernel_id(kernel), self._ptr(mean), self._ptr(var),
                                       self._ptr(info), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_svgp_predict", st)
        return mean, var, info

    # ------------------------------------------------------------------ stand-alone M x M entry points
    def chol_lower(self, A: torch.Tensor):
        M = A.shape[0]
        A = A.clone().contiguous()
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        ws = self._workspace("chol", self.lib.sgp_chol_workspace_bytes(M))
        st = self._cf.sgp_chol_lower(self._ptr(A), M, M, self._ptr(info), self._ptr(ws), ws.numel(), self._stream())
        _lib.check("sgp_chol_lower", st)
        return A, info

    def trsm_lower(self, L: torch.Tensor, B: torch.Tensor, trans=False):
        M, k = B.shape
        B = B.clone().contiguous()
        ws = self._workspace("trsm", self.lib.sgp_trsm_workspace_bytes(M, k))
        st = self._cf.sgp_trsm_lower(self._ptr(L), M, self._ptr(B), k, 1 if trans else 0, M, k, self._ptr(ws), ws.numel(),
                                     self._stream())
        _lib.check("sgp_trsm_lower", st)
        return B

    def logdiag_sum(self, L: torch.Tensor):
        out = self.empty(1)
        st = self.lib.sgp_logdiag_sum(self._ptr(L), L.shape[1], L.shape[0], self._ptr(out), self._stream())
        _lib.check("sgp_logdiag_sum", st)
        return out
