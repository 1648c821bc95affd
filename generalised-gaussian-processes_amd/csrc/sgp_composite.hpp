// Sum-of-products covariance functions (SGP_KERNEL_COMPOSITE): the reference's CO2 covariance
//   n_per^2 Periodic * ExpQuad + n_med^2 RatQuad + n_trend^2 ExpQuad + n_noise^2 Matern32
// (experiments/co2_bayesian_sgpr_hmc.py:107-149, PyMC3 pm.gp.cov classes; GPyTorch twin :74-83) and anything of the
// same shape: up to 4 terms, each amp2 * (1 or 2 isotropic factors).  The parameter block layout is include/sgp.h's
// SGP_COMP_*.  These workloads are small (C2: N = 634, M = 128), so the composite path is a plain materialised
// one -- K_fu by an elementwise kernel, Phi / Kbar through the MFMA GEMM of sgp_dense.hip, gradients by an
// elementwise contraction with fixed-order reductions -- and shares the whole O(M^3) tail with the fast path.
#pragma once
#include "sgp_common.hpp"

namespace sgp {

constexpr int COMP_MAX_DIM = 8;  // input dimension the composite path accepts

struct CompSpec {
  int nterms;
  int nfac[SGP_COMP_MAX_TERMS];
  int type[SGP_COMP_MAX_TERMS][SGP_COMP_MAX_FACTORS];
  double amp2[SGP_COMP_MAX_TERMS];
  double ls[SGP_COMP_MAX_TERMS][SGP_COMP_MAX_FACTORS];
  double aux[SGP_COMP_MAX_TERMS][SGP_COMP_MAX_FACTORS];
  double kdiag;  // k(x, x) = sum of the amplitudes
};

// host: validate and unpack a parameter block; SGP_OK or SGP_ERR_ARG
inline int comp_parse(const double* blk, int d, CompSpec* out) {
  if (!blk || d <= 0 || d > COMP_MAX_DIM) return SGP_ERR_ARG;
  CompSpec s{};
  const int nt = (int)blk[0];
  if ((double)nt != blk[0] || nt < 1 || nt > SGP_COMP_MAX_TERMS) return SGP_ERR_ARG;
  s.nterms = nt;
  s.kdiag = 0.0;
  for (int t = 0; t < nt; ++t) {
    const double* tb = blk + 1 + 8 * t;
    const int nf = (int)tb[1];
    if (!(tb[0] > 0.0) || (double)nf != tb[1] || nf < 1 || nf > SGP_COMP_MAX_FACTORS) return SGP_ERR_ARG;
    s.amp2[t] = tb[0];
    s.nfac[t] = nf;
    s.kdiag += tb[0];
    for (int f = 0; f < nf; ++f) {
      const double* fb = tb + 2 + 3 * f;
      const int ty = (int)fb[0];
      if ((double)ty != fb[0] || ty < SGP_FAC_EXPQUAD || ty > SGP_FAC_PERIODIC || !(fb[1] > 0.0)) return SGP_ERR_ARG;
      if ((ty == SGP_FAC_RATQUAD || ty == SGP_FAC_PERIODIC) && !(fb[2] > 0.0)) return SGP_ERR_ARG;
      s.type[t][f] = ty;
      s.ls[t][f] = fb[1];
      s.aux[t][f] = fb[2];
    }
  }
  *out = s;
  return SGP_OK;
}

// One factor: value F, dF/d ls, dF/d aux, and the two handles of its input derivative:
//   dF/d b_j = hr * (-2 delta_j) + hs * sin(2 pi delta_j / T) * pi / T        (delta = a - b)
struct FacOut {
  double F, dls, daux, hr, hs;
};
__device__ __forceinline__ FacOut comp_factor(int ty, double ls, double aux, double r2, const double* delta, int d) {
  FacOut o{0.0, 0.0, 0.0, 0.0, 0.0};
  const double il2 = 1.0 / (ls * ls);
  if (ty == SGP_FAC_EXPQUAD) {
    o.F = exp(-0.5 * r2 * il2);
    o.dls = o.F * r2 * il2 / ls;
    o.hr = -0.5 * o.F * il2;
  } else if (ty == SGP_FAC_MATERN32) {
    const double a = 1.7320508075688772 * sqrt(r2) / ls, e = exp(-a);
    o.F = (1.0 + a) * e;
    o.dls = a * a * e / ls;
    o.hr = -1.5 * il2 * e;
  } else if (ty == SGP_FAC_MATERN52) {
    const double a = 2.23606797749979 * sqrt(r2) / ls, e = exp(-a);
    o.F = (1.0 + a + a * a * (1.0 / 3.0)) * e;
    o.dls = a * a * (1.0 + a) * e / (3.0 * ls);
    o.hr = -(5.0 / 6.0) * il2 * (1.0 + a) * e;
  } else if (ty == SGP_FAC_RATQUAD) {
    const double w = 1.0 + 0.5 * r2 * il2 / aux, lw = log(w);
    o.F = exp(-aux * lw);
    o.dls = o.F / w * r2 * il2 / ls;
    o.daux = o.F * ((w - 1.0) / w - lw);
    o.hr = -0.5 * o.F * il2 / w;
  } else {  // SGP_FAC_PERIODIC: exp(-sum_j sin^2(pi delta_j / T) / (2 ls^2))
    const double w = 3.141592653589793 / aux;
    double S = 0.0, dST = 0.0;
    for (int j = 0; j < d; ++j) {
      const double s = sin(w * delta[j]);
      S = fma(s, s, S);
      dST = fma(sin(2.0 * w * delta[j]), delta[j], dST);
    }
    o.F = exp(-0.5 * S * il2);
    o.dls = o.F * S * il2 / ls;
    o.daux = 0.5 * o.F * il2 * dST * w / aux;
    o.hs = 0.5 * o.F * il2;
  }
  return o;
}

// k(a, b)
__device__ __forceinline__ double comp_value(const CompSpec& cs, const double* a, const double* b, int d) {
  double delta[COMP_MAX_DIM];
  double r2 = 0.0;
  for (int j = 0; j < d; ++j) {
    delta[j] = a[j] - b[j];
    r2 = fma(delta[j], delta[j], r2);
  }
  double k = 0.0;
  for (int t = 0; t < cs.nterms; ++t) {
    double term = cs.amp2[t];
    for (int f = 0; f < cs.nfac[t]; ++f) term *= comp_factor(cs.type[t][f], cs.ls[t][f], cs.aux[t][f], r2, delta, d).F;
    k += term;
  }
  return k;
}

// k(a, b), its derivative with respect to every slot of the parameter block (gpar[SGP_COMP_LEN], only the slots that
// carry a parameter are written; the caller zero-initialises) and with respect to b (dkdb[d]).
__device__ __forceinline__ double comp_grad(const CompSpec& cs, const double* a, const double* b, int d, double* gpar,
                                            double* dkdb) {
  double delta[COMP_MAX_DIM];
  double r2 = 0.0;
  for (int j = 0; j < d; ++j) {
    delta[j] = a[j] - b[j];
    r2 = fma(delta[j], delta[j], r2);
    dkdb[j] = 0.0;
  }
  double k = 0.0;
  for (int t = 0; t < cs.nterms; ++t) {
    const int base = 1 + 8 * t;
    FacOut fo[SGP_COMP_MAX_FACTORS];
    double prod = 1.0;
    for (int f = 0; f < cs.nfac[t]; ++f) {
      fo[f] = comp_factor(cs.type[t][f], cs.ls[t][f], cs.aux[t][f], r2, delta, d);
      prod *= fo[f].F;
    }
    k = fma(cs.amp2[t], prod, k);
    gpar[base] = prod;
    for (int f = 0; f < cs.nfac[t]; ++f) {
      const double other = cs.amp2[t] * (cs.nfac[t] == 2 ? fo[1 - f].F : 1.0);
      const int fb = base + 2 + 3 * f;
      gpar[fb + 1] = other * fo[f].dls;
      gpar[fb + 2] = other * fo[f].daux;
      const double hr = other * fo[f].hr, hs = other * fo[f].hs;
      if (cs.type[t][f] == SGP_FAC_PERIODIC) {
        const double w = 3.141592653589793 / cs.aux[t][f];
        for (int j = 0; j < d; ++j) dkdb[j] = fma(hs * w, sin(2.0 * w * delta[j]), dkdb[j]);
      } else {
        for (int j = 0; j < d; ++j) dkdb[j] = fma(-2.0 * hr, delta[j], dkdb[j]);
      }
    }
  }
  return k;
}

// ---- host entry points of the materialised composite path (sgp_composite.hip) --------------------------------------
// out[i][j] = k(a_i, b_j) (+ jitter on i == j) for i < na, j < nb, zero in the padding; rows_p x cols_p, ld cols_p
void comp_kmatrix(const double* A, int64_t lda, int64_t na, const double* B, int64_t ldb, int nb, const CompSpec& cs, int d,
                  int64_t rows_p, int cols_p, double jitter, double* out, hipStream_t st);
size_t comp_fwd_workspace_bytes(int64_t N, int M);
int comp_suffstats_fwd(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const CompSpec& cs,
                       int64_t N, int M, int d, double* Phi, double* b, double* yy, double* kappa, void* ws, size_t ws_bytes,
                       hipStream_t st);
size_t comp_bwd_workspace_bytes(int64_t N, int M, int d);
size_t comp_bwd_factored_workspace_bytes(int64_t N, int M, int d);
// g_blk[SGP_COMP_LEN] and g_Z (may be null) are OVERWRITTEN; includes the kappa term kappabar * N on every amplitude
int comp_suffstats_bwd(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const CompSpec& cs,
                       const double* Phibar, const double* bbar, double kappabar, int64_t N, int M, int d, double* g_blk,
                       double* g_Z, void* ws, size_t ws_bytes, hipStream_t st);
// The same gradients from the FACTORED adjoint 2 Phibar = L^-T (Cw / s2) L^-1 (Linv: Mp x Mp from sgp_kuu_factor, Cw: M x M):
// the three factors are applied to every K_fu chunk one after the other.  Phibar itself has entries of size cond(K_uu)
// that cancel in Phibar K_uf; formed explicitly it left 1e-2 .. 1e-1 relative error on the gradients of the CO2 model at
// cond 3e9 .. 5e10 (M = 64 .. 480), applied as factors 1e-5 (tests/studies/logp_noise.py).
int comp_suffstats_bwd_factored(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const CompSpec& cs,
                                const double* Linv, const double* Cw, double s2, const double* bbar, double kappabar, int64_t N,
                                int M, int d, double* g_blk, double* g_Z, void* ws, size_t ws_bytes, hipStream_t st);
size_t comp_kuu_bwd_workspace_bytes(int M, int d);
// ADDS the Kuu contribution (Kuubar used as a symmetric matrix)
int comp_kuu_bwd(const double* Z, int64_t ldz, const CompSpec& cs, const double* Kuubar, int M, int d, double* g_blk,
                 double* g_Z, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace sgp
