"""Test double for HipEngine: same interface, arithmetic delegated to the CPU oracle.

TEST INFRASTRUCTURE ONLY -- lets the CPU-only suite exercise the host logic above the C ABI
(CollapsedBound, HmcTarget, the model classes, the NUTS driver, the gloo world_size-2 sharding)
in a container without a GPU.  The product never constructs it.
"""
import math

import torch

from oracle import vfe_oracle as O

KID = {"rbf": 0, "matern32": 1, "matern52": 2, 0: 0, 1: 1, 2: 2}
OUT_LEN = 8


class OracleEngine:
    def __init__(self):
        self.device = torch.device("cpu")
        self.calls = {"suffstats": 0, "suffstats_whitened": 0, "bound": 0, "suffstats_bwd": 0, "kuu_bwd": 0, "predict": 0}

    @staticmethod
    def _ls(ls, d):
        t = torch.as_tensor([float(v) for v in (ls.tolist() if hasattr(ls, "tolist") else ls)], dtype=torch.float64)
        return t.expand(d).clone() if t.numel() == 1 else t

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64)

    def result_buffer(self, extra=0):
        buf = torch.zeros(OUT_LEN + 4 + extra, dtype=torch.float64)  # [out | status word | estimate | bound | pad | extras]
        return buf, buf[:OUT_LEN], buf[OUT_LEN:OUT_LEN + 1].view(torch.int32)[:1]

    @staticmethod
    def read_result(host_buf):
        return host_buf[:OUT_LEN], int(host_buf[OUT_LEN:OUT_LEN + 1].view(torch.int32)[0])

    def suffstats(self, X, y, Z, ls, sf2, kernel="rbf", out=None, kfu=None):
        self.calls["suffstats"] += 1
        M, d = Z.shape
        if X.shape[0] == 0:
            packed = torch.zeros(M * M + M + 2, dtype=torch.float64)
        else:
            st = O.suffstats(X, y, Z, self._ls(ls, d), float(sf2), KID[kernel])
            packed = torch.cat([st.Phi.reshape(-1), st.b, torch.tensor([st.yy, st.kappa], dtype=torch.float64)])
        if out is not None:
            out.copy_(packed)
            return out
        return packed

    def pack_lower(self, stats, M):
        r, c = torch.tril_indices(M, M)
        return torch.cat([stats[: M * M].reshape(M, M)[r, c], stats[M * M:]])

    def unpack_lower(self, tri, M, stats):
        r, c = torch.tril_indices(M, M)
        nt = M * (M + 1) // 2
        P = torch.zeros(M, M, dtype=torch.float64)
        P[r, c] = tri[:nt]
        P = P + P.T - torch.diag(torch.diagonal(P))
        stats[: M * M] = P.reshape(-1)
        stats[M * M:] = tri[nt:]
        return stats

    def kuu(self, Z, ls, sf2, jitter, kernel="rbf"):
        return O.kuu(Z, self._ls(ls, Z.shape[1]), float(sf2), float(jitter), KID[kernel])

    def kuu_factor(self, Kuu, info=None, trace_out=None):
        """(L^-1, info); like the HIP engine the status word is written, never raised.  trace_out[0] <- tr(K_uu^-1) (sgp_kuu_factor_ex)."""
        M = Kuu.shape[0]
        if info is None:
            info = torch.zeros(1, dtype=torch.int32)
        info.zero_()
        try:
            L = torch.linalg.cholesky(Kuu)
        except Exception:
            info[0] = 1
            if trace_out is not None:
                trace_out[0] = float(M)
            return torch.eye(M, dtype=torch.float64), info
        Li = torch.linalg.solve_triangular(L, torch.eye(M, dtype=torch.float64), upper=False)
        if trace_out is not None:
            trace_out[0] = float((Li ** 2).sum())
        return Li, info

    def suffstats_whitened(self, X, y, Z, ls, sf2, kuu_linv, kernel="rbf", out=None):
        self.calls["suffstats_whitened"] += 1
        M, d = Z.shape
        if X.shape[0] == 0:
            packed = torch.zeros(M * M + M + 2, dtype=torch.float64)
        else:
            L = torch.linalg.solve_triangular(kuu_linv, torch.eye(M, dtype=torch.float64), upper=False)
            st = O.suffstats_whitened(X, y, Z, self._ls(ls, d), float(sf2), L, KID[kernel])
            packed = torch.cat([st.Phi.reshape(-1), st.b, torch.tensor([st.yy, st.kappa], dtype=torch.float64)])
        if out is not None:
            out.copy_(packed)
            return out
        return packed

    def bound(self, Kuu, packed, s2, N, with_adjoints=False, want_factors=False, result=None, kuu_linv=None, whitened=False):
        self.calls["bound"] += 1
        M = Kuu.shape[0]
        st = O.SuffStats(packed[: M * M].reshape(M, M), packed[M * M: M * M + M], float(packed[M * M + M]),
                         float(packed[M * M + M + 1]), int(N))
        buf, out, info = result if result is not None else self.result_buffer()
        out.zero_()
        res = {"out": out, "info": info, "buf": buf}
        if kuu_linv is not None and int(info[0]) != 0:  # kuu_factor already failed: the status word stays
            if with_adjoints:
                res.update(Phibar=torch.zeros(M, M, dtype=torch.float64), bbar=torch.zeros(M, dtype=torch.float64),
                           Kuubar=torch.zeros(M, M, dtype=torch.float64))
            return res
        info.zero_()
        try:
            r = O.bound_from_stats(Kuu, st, float(s2), with_adjoints=with_adjoints, stats_whitened=whitened)
        except Exception:  # torch.linalg.cholesky failure -> LAPACK-style info like the HIP path: 1 .. M = K_uu, M + 1 .. 2 M = B
            try:
                torch.linalg.cholesky(Kuu)
                info[0] = M + 1
            except Exception:
                info[0] = 1
            if with_adjoints:
                res.update(Phibar=torch.zeros(M, M, dtype=torch.float64), bbar=torch.zeros(M, dtype=torch.float64),
                           Kuubar=torch.zeros(M, M, dtype=torch.float64))
            return res
        out[0], out[1], out[2] = r["F"], r["logmarg"], r["trace_term"]
        if with_adjoints:
            out[6], out[7] = r["s2bar"], r["kappabar"]
            res.update(Phibar=r["Phibar"], bbar=r["bbar"], Kuubar=r["Kuubar"])
        if want_factors:
            I = torch.eye(M, dtype=torch.float64)
            Linv = torch.linalg.solve_triangular(r["L"], I, upper=False)
            LBinv = torch.linalg.solve_triangular(r["LB"], I, upper=False)
            res["factors"] = torch.cat([Linv.reshape(-1), LBinv.reshape(-1), r["q"]])
        return res

    def suffstats_bwd(self, X, y, Z, ls, sf2, Phibar, bbar, kappabar, kernel="rbf", want_gz=False, out=None, kfu=None):
        self.calls["suffstats_bwd"] += 1
        M, d = Z.shape
        lst = self._ls(ls, d)
        kid = KID[kernel]
        g_ls = torch.zeros(d, dtype=torch.float64)
        g_Z = torch.zeros(M, d, dtype=torch.float64)
        g_sf2 = torch.zeros((), dtype=torch.float64)
        Zs = Z / lst
        for s in range(0, X.shape[0], 4096):
            Xc = X[s:s + 4096]
            r2 = O.sqdist(Z, Xc, lst)
            K = O.kernel_from_r2(r2, float(sf2), kid)
            Kbar = 2.0 * Phibar @ K + torch.outer(bbar, y[s:s + 4096])
            g_sf2 = g_sf2 + (Kbar * K).sum() / sf2
            E = Kbar * O._dk_factors(r2, K, float(sf2), kid)
            diff = Zs[:, None, :] - (Xc / lst)[None, :, :]
            g_ls = g_ls + (-2.0 * (E[:, :, None] * diff * diff).sum((0, 1)) / lst)
            g_Z = g_Z + 2.0 * (E[:, :, None] * diff).sum(1) / lst
        g_sf2 = g_sf2 + float(kappabar) * X.shape[0]
        parts = [g_ls, g_sf2.reshape(1)] + ([g_Z.reshape(-1)] if want_gz else [])
        packed = torch.cat(parts)
        if out is not None:
            out[: packed.numel()].copy_(packed)   # (the caller's buffer may carry more behind the gradient: the correction's slots)
            return out
        return packed

    def kuu_bwd(self, Z, ls, sf2, Kuubar, grads, kernel="rbf", want_gz=False):
        self.calls["kuu_bwd"] += 1
        M, d = Z.shape
        lst = self._ls(ls, d)
        kid = KID[kernel]
        Zs = Z / lst
        r2u = O.sqdist(Z, Z, lst)
        Ku = O.kernel_from_r2(r2u, float(sf2), kid)
        Eu = Kuubar * O._dk_factors(r2u, Ku, float(sf2), kid)
        diffu = Zs[:, None, :] - Zs[None, :, :]
        grads[:d] += -2.0 * (Eu[:, :, None] * diffu * diffu).sum((0, 1)) / lst
        grads[d] += (Kuubar * Ku).sum() / sf2
        if want_gz:
            grads[d + 1:d + 1 + M * d] += (2.0 * ((Eu + Eu.T)[:, :, None] * diffu).sum(1) / lst).reshape(-1)   # (more may follow: the correction's slots)
        return grads

    def predict(self, Xs, Z, ls, sf2, s2, factors, kernel="rbf", pred_noise=True, full_cov=False):
        self.calls["predict"] += 1
        M, d = Z.shape
        lst = self._ls(ls, d)
        Linv = factors[: M * M].reshape(M, M)
        LBinv = factors[M * M: 2 * M * M].reshape(M, M)
        q = factors[2 * M * M:]
        Kus = O.kernel_from_r2(O.sqdist(Z, Xs, lst), float(sf2), KID[kernel])
        As = Linv @ Kus
        C = LBinv @ As
        mean = C.T @ q / s2
        var = sf2 - (As * As).sum(0) + (C * C).sum(0) + (s2 if pred_noise else 0.0)
        cov = None
        if full_cov:
            cov = O.kernel_from_r2(O.sqdist(Xs, Xs, lst), float(sf2), KID[kernel]) - As.T @ As + C.T @ C
            if pred_noise:
                cov = cov + s2 * torch.eye(Xs.shape[0], dtype=torch.float64)
        return mean, var, cov

    # ---- SVGP (oracle.svgp_oracle) ----
    def svgp_elbo(self, Xb, yb, Z, ls, sf2, s2, m, LS, N_total, jitter=1e-6, kernel="rbf", likelihood="gaussian",
                  with_grads=False):
        from oracle import svgp_oracle as S
        lik = {"gaussian": 0, "bernoulli": 1, "bernoulli_probit": 1}[likelihood]
        d = Z.shape[1]
        info = torch.zeros(1, dtype=torch.int32)
        if with_grads:
            r = S.svgp_elbo_and_grads(Xb, yb, Z, self._ls(ls, d), float(sf2), float(s2), m, LS, N_total, jitter, KID[kernel], lik)
            ell, kl, _, _ = S.svgp_terms(Xb, yb, Z, self._ls(ls, d), float(sf2), float(s2), m, LS, jitter, KID[kernel], lik)
            out = torch.tensor([r["elbo"], float(ell.sum()), float(kl)], dtype=torch.float64)
            return {"out": out, "info": info, "g_m": r["g_m"], "g_LS": r["g_LS"], "g_Z": r["g_Z"], "g_ls": r["g_ls"],
                    "g_sf2": torch.tensor([r["g_sf2"]], dtype=torch.float64), "g_s2": torch.tensor([r["g_s2"]], dtype=torch.float64)}
        ell, kl, _, _ = S.svgp_terms(Xb, yb, Z, self._ls(ls, d), float(sf2), float(s2), m, LS, jitter, KID[kernel], lik)
        out = torch.tensor([float(ell.mean() - kl / N_total), float(ell.sum()), float(kl)], dtype=torch.float64)
        return {"out": out, "info": info}

    def svgp_elbo_batch(self, Xb, yb, Z, ls, sf2, s2, m, LS, N_total, jitter=1e-6, kernel="rbf", likelihood="gaussian",
                        with_grads=False):
        """S hyper-parameter samples: the oracle, sample by sample (layout of HipEngine.svgp_elbo_batch)."""
        rs = [self.svgp_elbo(Xb, yb, Z, list(ls[k]), float(sf2[k]), float(s2[k]) if likelihood == "gaussian" else 1.0, m, LS, N_total,
                             jitter, kernel, likelihood, with_grads) for k in range(len(ls))]
        info = torch.cat([r["info"] for r in rs])
        res = {"out": torch.cat([torch.stack([r["out"] for r in rs]), info.to(torch.float64)[:, None]], 1), "info": info}
        if with_grads:
            for key in ("g_m", "g_LS", "g_Z", "g_ls"):
                res[key] = torch.stack([r[key] for r in rs])
            for key in ("g_sf2", "g_s2"):
                res[key] = torch.cat([r[key] for r in rs])
        return res

    def svgp_predict(self, Xs, Z, ls, sf2, m, LS, jitter=1e-6, kernel="rbf"):
        from oracle import svgp_oracle as S
        mu, v = S.svgp_predict(Xs, Z, self._ls(ls, Z.shape[1]), float(sf2), m, LS, jitter, KID[kernel])
        return mu, v, torch.zeros(1, dtype=torch.int32)


class GuardedOracleEngine(OracleEngine):
    """OracleEngine + the three entry points of the streaming-order guard (include/sgp.h: sgp_kuu_inverse_trace,
    sgp_streaming_error_estimate), so that CollapsedBound's estimate / repeat logic runs on the CPU -- also over gloo ranks."""

    def kuu_inverse_trace(self, Linv, M, out=None):
        t = torch.zeros(2, dtype=torch.float64) if out is None else out
        t[0] = float((Linv[:M, :M] ** 2).sum()) if Linv.dim() == 2 else float((Linv.reshape(-1) ** 2).sum())
        return t

    def streaming_error_report(self, diag, stride, trace, sf2, s2, N, M, result):
        """sgp_streaming_error_report: estimate from the diagonal an evaluation order holds of Phi (None: the bound), and the bound."""
        ub = 2.0 ** -53 * float(sf2) ** 2 * float(trace[0]) / float(s2)
        result[0][OUT_LEN + 2] = ub
        if diag is None:
            result[0][OUT_LEN + 1] = ub
        else:
            phi_max = float(diag.reshape(-1)[: (M - 1) * int(stride) + 1: int(stride)].max())
            result[0][OUT_LEN + 1] = 2.0 ** -53 * phi_max * float(trace[0]) / (float(s2) * max(1, int(N)))

    @staticmethod
    def read_estimate(host_buf):
        return float(host_buf[OUT_LEN + 1])

    @staticmethod
    def read_bound(host_buf):
        return float(host_buf[OUT_LEN + 2])


class FactoredOracleEngine(GuardedOracleEngine):
    """GuardedOracleEngine + the whitened order in the streaming layout and the factored pass 2 (include/sgp.h:
    sgp_suffstats_fwd_whitened_rows, sgp_suffstats_bwd_factored_ex), so that CollapsedBound's hand-over of T = K'_fu L^-T from pass 1 to
    pass 2 runs on the CPU.  T is kept in the caller's K'_fu block exactly as the HIP engine keeps it; pass 2 CHECKS what it is handed."""

    def __init__(self):
        super().__init__()
        self.calls.update({"suffstats_whitened_rows": 0, "suffstats_bwd_factored": 0, "t_handed_over": 0})

    def kfu_buffer(self, N, M):
        return torch.full((max(1, int(N)) * int(M),), float("nan"), dtype=torch.float64)

    def kfu_f16_buffer(self, N, M):
        return torch.full((max(1, int(N)) * int(M),), float("nan"), dtype=torch.float16)

    def suffstats_whitened_rows(self, X, y, Z, ls, sf2, kuu_linv, kernel="rbf", out=None, t_out=None):
        self.calls["suffstats_whitened_rows"] += 1
        n = self.calls["suffstats_whitened"]
        packed = self.suffstats_whitened(X, y, Z, ls, sf2, kuu_linv, kernel, out)
        self.calls["suffstats_whitened"] = n                       # (delegation, not a call of the chunked routine)
        if t_out is not None and X.shape[0] > 0:
            M, d = Z.shape
            Kp = O.kern(X, Z, self._ls(ls, d), 1.0, KID[kernel])  # unit amplitude
            t_out[: X.shape[0] * M] = (Kp @ kuu_linv.T).reshape(-1)
        return packed

    def bound(self, Kuu, packed, s2, N, with_adjoints=False, want_factors=False, result=None, kuu_linv=None, whitened=False,
              want_cw=False):
        res = super().bound(Kuu, packed, s2, N, with_adjoints, want_factors, result, kuu_linv, whitened)
        if want_cw and with_adjoints and kuu_linv is not None:
            M = Kuu.shape[0]
            L = torch.linalg.solve_triangular(kuu_linv, torch.eye(M, dtype=torch.float64), upper=False)
            res["Cw"] = 2.0 * float(s2) * (L.T @ res["Phibar"] @ L)   # 2 s2 Phibar = L^-T Cw L^-1
        return res

    def phibar_dd(self, Cw, kuu_linv, s2, want_lo=False):
        """sgp_phibar_dd: L^-T (Cw / 2 s2) L^-1 formed with extra precision (x87 long double standing in for double-double), split in two words."""
        import numpy as np
        self.calls["phibar_dd"] = self.calls.get("phibar_dd", 0) + 1
        M = Cw.shape[0]
        Li = kuu_linv.view(-1)[: M * M].view(M, M).numpy().astype(np.longdouble) if kuu_linv.dim() == 1 else kuu_linv[:M, :M].numpy().astype(np.longdouble)
        P = Li.T @ (Cw.numpy().astype(np.longdouble) / (2 * np.longdouble(float(s2)))) @ Li
        P = np.tril(P) + np.tril(P, -1).T
        hi = np.asarray(P, dtype=np.float64)
        lo = np.asarray(P - hi.astype(np.longdouble), dtype=np.float64)
        return torch.from_numpy(hi), (torch.from_numpy(lo) if want_lo else None)

    def bwd_lo_supported(self, N, M, d, kernel="rbf"):
        return kernel == "rbf" and d <= 32

    def suffstats_bwd_lo(self, X, y, Z, ls, sf2, Phibar_lo, kfu, grads, kernel="rbf", delta=None, kfu_f16=None):
        """sgp_suffstats_bwd_lo: what pass 2 would have added had its Phibar carried the trailing word (lengthscales and amplitude only)."""
        self.calls["suffstats_bwd_lo"] = self.calls.get("suffstats_bwd_lo", 0) + 1
        d = Z.shape[1]
        if kfu_f16 is not None:   # the image the forward pass of THIS theta left
            self.calls["f16_handed_over"] = self.calls.get("f16_handed_over", 0) + 1
            if X.shape[0] > 0:
                Kp = O.kern(X, Z, self._ls(ls, d), 1.0, KID[kernel])
                assert torch.equal(kfu_f16[: X.shape[0] * Z.shape[0]], Kp.reshape(-1).to(torch.float16)), "the trailing-word product was handed a stale fp16 image"
        n = self.calls["suffstats_bwd"]
        corr = self.suffstats_bwd(X, y, Z, ls, sf2, Phibar_lo, torch.zeros(Z.shape[0], dtype=torch.float64), 0.0, kernel, False, None)
        self.calls["suffstats_bwd"] = n
        grads[: d + 1] += corr[: d + 1]
        if delta is not None:
            delta[: d + 1] = corr[: d + 1] * getattr(self, "lo_delta_scale", 1.0)   # (tests inflate it to exercise the a-posteriori check)
        return grads

    def suffstats_bwd_factored(self, X, y, Z, ls, sf2, kuu_linv, Cw, s2, bbar, kappabar, kernel="rbf", want_gz=False, out=None,
                               t_in=None):
        self.calls["suffstats_bwd_factored"] += 1
        M, d = Z.shape
        if t_in is not None:
            self.calls["t_handed_over"] += 1
            if X.shape[0] > 0:
                Kp = O.kern(X, Z, self._ls(ls, d), 1.0, KID[kernel])
                assert torch.equal(t_in[: X.shape[0] * M], (Kp @ kuu_linv.T).reshape(-1)), "pass 2 was handed a stale T"
        Phibar = kuu_linv.T @ Cw @ kuu_linv / (2.0 * float(s2))
        n = self.calls["suffstats_bwd"]
        g = self.suffstats_bwd(X, y, Z, ls, sf2, Phibar, bbar, kappabar, kernel, want_gz, out)
        self.calls["suffstats_bwd"] = n
        return g

    def suffstats_extended(self, X, y, Z, ls, sf2, kuu_linv, kernel="rbf", out=None, kfu=None, level=1, phi_diag=None, kfu_f16=None):
        """sgp_suffstats_fwd_extended: the same whitened statistics (the oracle has one way to compute them), K'_fu kept in `kfu`."""
        self.calls["suffstats_extended"] = self.calls.get("suffstats_extended", 0) + 1
        self.calls["extended_level_%d" % level] = self.calls.get("extended_level_%d" % level, 0) + 1
        n = self.calls["suffstats_whitened"]
        packed = self.suffstats_whitened(X, y, Z, ls, sf2, kuu_linv, kernel, out)
        self.calls["suffstats_whitened"] = n
        M, d = Z.shape
        if kfu is not None and X.shape[0] > 0:
            kfu[: X.shape[0] * M] = O.kern(X, Z, self._ls(ls, d), 1.0, KID[kernel]).reshape(-1)
        if kfu_f16 is not None:
            assert kfu is not None
            if X.shape[0] > 0:
                kfu_f16[: X.shape[0] * M] = O.kern(X, Z, self._ls(ls, d), 1.0, KID[kernel]).reshape(-1).to(torch.float16)
        if phi_diag is not None:  # diag(K_uf K_fu) of this shard, with its amplitude
            phi_diag[:M] = (O.kern(X, Z, self._ls(ls, d), float(sf2), KID[kernel]) ** 2).sum(0) if X.shape[0] > 0 else 0.0
        return packed
