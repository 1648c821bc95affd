// Uncollapsed (SVGP) minibatch bound and its gradients -- SURVEY.md section 8 (f-3).
//
// Replaces, for one minibatch, what the reference obtains from GPyTorch's
//   VariationalELBO(likelihood, model, num_data=N)(model(x_batch), y_batch)  +  loss.backward()
// over VariationalStrategy(CholeskyVariationalDistribution(M), learn_inducing_locations=True)
// (reference models/svgp.py:37,46,88-127; models/bayesian_svgp.py:144-181):
//
//   L = chol(Kuu + J I) ; A = L^-1 K_ub ; T = L_S^T A ; mu = A^T m ; v = k_bb - colsum(A o A) + colsum(T o T)
//   ELBO / datum = mean_b E_{N(mu_b, v_b)} log p(y_b | f) - KL(N(m, L_S L_S^T) || N(0, I)) / N
//
// Likelihoods: Gaussian (closed form) and Bernoulli-probit (20-point Gauss-Hermite, y in {-1, +1}).
// The shapes are small (M <= a few hundred, B = minibatch), so this path is launch-latency bound; every
// O(M^2 B) / O(M^3) product runs on the fp64-MFMA GEMM of sgp_dense.hip and the reverse pass is the
// closed-form adjoint (Cholesky adjoint  Kbar = sym(L^-T Phi(L^T Lbar) L^-1), Murray 2016) -- no autograd.
#include "sgp_dense.hpp"

namespace sgp {

constexpr int GH_N = 20;
struct GHTable {
  double x[GH_N];  // nodes of  int f(x) N(x; 0, 1) dx
  double w[GH_N];  // weights (sum to 1)
};
// Gauss-Hermite nodes / weights by Newton iteration on the orthonormal Hermite recurrence (host, once):
// physicists' rule (weight exp(-x^2)) rescaled to the standard normal:  x * sqrt(2),  w / sqrt(pi).
static void gauss_hermite_host(int n, double* xs, double* ws) {
  const double PIM4 = 0.7511255444649425;  // pi^(-1/4)
  double z = 0.0, pp = 1.0;
  const int half = (n + 1) / 2;
  for (int i = 0; i < half; ++i) {
    if (i == 0) z = sqrt(2.0 * n + 1.0) - 1.85575 * pow(2.0 * n + 1.0, -0.16667);
    else if (i == 1) z -= 1.14 * pow((double)n, 0.426) / z;
    else if (i == 2) z = 1.86 * z - 0.86 * xs[0];
    else if (i == 3) z = 1.91 * z - 0.91 * xs[1];
    else z = 2.0 * z - xs[i - 2];
    for (int its = 0; its < 200; ++its) {
      double p1 = PIM4, p2 = 0.0;
      for (int j = 1; j <= n; ++j) {
        const double p3 = p2;
        p2 = p1;
        p1 = z * sqrt(2.0 / j) * p2 - sqrt((double)(j - 1) / j) * p3;
      }
      pp = sqrt(2.0 * n) * p2;
      const double z1 = z;
      z = z1 - p1 / pp;
      if (fabs(z - z1) <= 1e-15 * (1.0 + fabs(z))) break;
    }
    xs[i] = z;
    xs[n - 1 - i] = -z;
    ws[i] = ws[n - 1 - i] = 2.0 / (pp * pp);
  }
  for (int i = 0; i < n; ++i) {
    xs[i] *= 1.4142135623730951;
    ws[i] *= 0.5641895835477563;
  }
}
static GHTable make_gh() {
  GHTable t;
  gauss_hermite_host(GH_N, t.x, t.w);
  return t;
}

// Kub[m][b] = sf2 k'(z_m, x_b) (zero in the padding), Mp x Bp
template <int KID>
__global__ __launch_bounds__(256) void svgp_kub_kernel(const double* __restrict__ Z, int64_t ldz, const double* __restrict__ Xb,
                                                       int64_t ldx, KernArgs ka, int M, int Mp, int B, int Bp,
                                                       double* __restrict__ Kub) {
  const int64_t total = (int64_t)Mp * Bp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / Bp), b = (int)(e - (int64_t)m * Bp);
    double v = 0.0;
    if (m < M && b < B) {
      double r2 = 0.0;
      for (int q = 0; q < ka.d; ++q) {
        const double df = (Z[m * ldz + q] - Xb[b * ldx + q]) * ka.inv_ls[q];
        r2 = fma(df, df, r2);
      }
      v = ka.sf2 * kprofile<KID>(r2);
    }
    Kub[e] = v;
  }
}

// mu[b] = sum_m A[m][b] mp[m] ; v[b] = kbb - sum A^2 + sum T^2   (block = 64 columns x 4 row groups)
__global__ __launch_bounds__(256) void svgp_cols_kernel(const double* __restrict__ A, const double* __restrict__ T,
                                                        const double* __restrict__ mp, int Mp, int Bp, int B, double kbb,
                                                        double* __restrict__ mu, double* __restrict__ v) {
  __shared__ double pm[4][64], pa[4][64], pt[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6, l = threadIdx.x & 63;
  double sm = 0.0, sa = 0.0, st = 0.0;
  for (int m = w; m < Mp; m += 4) {
    const double a = A[(int64_t)m * Bp + col], t = T[(int64_t)m * Bp + col];
    sm = fma(a, mp[m], sm);
    sa = fma(a, a, sa);
    st = fma(t, t, st);
  }
  pm[w][l] = sm; pa[w][l] = sa; pt[w][l] = st;
  __syncthreads();
  if (w == 0) {
    mu[col] = (pm[0][l] + pm[1][l]) + (pm[2][l] + pm[3][l]);
    v[col] = col < B ? kbb - ((pa[0][l] + pa[1][l]) + (pa[2][l] + pa[3][l])) + ((pt[0][l] + pt[1][l]) + (pt[2][l] + pt[3][l])) : 1.0;
  }
}

__device__ __forceinline__ double log_ndtr_dev(double z) { return log(0.5 * erfc(-z * 0.7071067811865476)); }
// phi(z) / Phi(z), stable for the range Gauss-Hermite nodes reach
__device__ __forceinline__ double mills_dev(double z) {
  const double phi = 0.3989422804014327 * exp(-0.5 * z * z);
  const double Phi = 0.5 * erfc(-z * 0.7071067811865476);
  return Phi > 0.0 ? phi / Phi : -z;  // asymptote phi/Phi -> -z as z -> -inf
}

// per-point expected log-likelihood and its derivatives; fixed grid of 64 blocks, partial sums per block
__global__ __launch_bounds__(256) void svgp_ell_kernel(const double* __restrict__ y, const double* __restrict__ mu,
                                                       const double* __restrict__ v, int B, double s2, int lik, GHTable gh,
                                                       double* __restrict__ dmu, double* __restrict__ dv,
                                                       double* __restrict__ part /* [64][2]: ell, ds2 */) {
  __shared__ double red[4];
  double se = 0.0, ss = 0.0;
  for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
    const double yb = y[b], m = mu[b], vv = v[b];
    double ell, gm, gv, gs = 0.0;
    if (lik == 0) {
      const double r = yb - m, q = r * r + vv;
      ell = -0.9189385332046727 - 0.5 * log(s2) - q / (2.0 * s2);
      gm = r / s2;
      gv = -0.5 / s2;
      gs = -0.5 / s2 + q / (2.0 * s2 * s2);
    } else {
      const double sd = sqrt(vv);
      ell = 0.0; gm = 0.0; gv = 0.0;
      for (int i = 0; i < GH_N; ++i) {
        const double z = yb * (m + sd * gh.x[i]);
        ell = fma(gh.w[i], log_ndtr_dev(z), ell);
        const double r = gh.w[i] * yb * mills_dev(z);
        gm += r;
        gv = fma(r, gh.x[i], gv);
      }
      gv = gv / (2.0 * sd);
    }
    dmu[b] = gm;
    dv[b] = gv;
    se += ell;
    ss += gs;
  }
  se = block_sum256(se, red);
  ss = block_sum256(ss, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = se;
    part[2 * blockIdx.x + 1] = ss;
  }
}

// KL(N(m, LS LS^T) || N(0, I)) and, on request, its gradients into g_m / g_LS (scaled by kl_scale = -1/N)
// klpart[block] = partial of  sum m^2 - 2 sum log diag(LS) + ||tril(LS)||_F^2  (fixed grid of 64 blocks; one block
// walking the M^2 entries took 90 us at M = 256)
__global__ __launch_bounds__(256) void svgp_kl_kernel(const double* __restrict__ m, const double* __restrict__ LS, int M,
                                                      double* __restrict__ klpart) {
  __shared__ double red[4];
  double s = 0.0;
  const int t = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
  for (int i = t; i < M; i += nt) s += m[i] * m[i] - 2.0 * log(LS[(int64_t)i * M + i]);
  for (int64_t e = t; e < (int64_t)M * M; e += nt) {
    const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
    if (c <= r) s = fma(LS[e], LS[e], s);
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) klpart[blockIdx.x] = s;
}

__global__ void svgp_finalize_kernel(const double* __restrict__ part, int nparts, const double* __restrict__ kl, int M, int B,
                                     double N_total, const double* __restrict__ dvbuf, double* __restrict__ out,
                                     double* __restrict__ g_s2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double se = 0.0, ss = 0.0;
  for (int i = 0; i < nparts; ++i) {
    se += part[2 * i];
    ss += part[2 * i + 1];
  }
  double klsum = 0.0;
  for (int i = 0; i < 64; ++i) klsum += kl[i];
  const double klv = 0.5 * (klsum - (double)M);
  out[0] = se / (double)B - klv / N_total;
  out[1] = se;
  out[2] = klv;
  if (g_s2) *g_s2 = ss / (double)B;
}

// Abar = mp mubar^T + 2 (U - A) diag(vbar), with mubar = dmu / B, vbar = dv / B ; Av = A diag(vbar)
__global__ __launch_bounds__(256) void svgp_abar_kernel(const double* __restrict__ A, const double* __restrict__ U,
                                                        const double* __restrict__ mp, const double* __restrict__ dmu,
                                                        const double* __restrict__ dv, int Mp, int Bp, int B, double invB,
                                                        double* __restrict__ Abar, double* __restrict__ Av) {
  const int64_t total = (int64_t)Mp * Bp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / Bp), b = (int)(e - (int64_t)m * Bp);
    double ab = 0.0, av = 0.0;
    if (b < B) {
      const double mb = dmu[b] * invB, vb = dv[b] * invB;
      ab = mp[m] * mb + 2.0 * vb * (U[e] - A[e]);
      av = A[e] * vb;
    }
    Abar[e] = ab;
    Av[e] = av;
  }
}

// g_m[m] = sum_b A[m][b] dmu[b] / B - m[m] / N      (one wave per row)
__global__ __launch_bounds__(256) void svgp_gm_kernel(const double* __restrict__ A, const double* __restrict__ dmu,
                                                      const double* __restrict__ m, int M, int Bp, int B, double invB,
                                                      double invN, double* __restrict__ g_m) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  double s = 0.0;
  for (int b = lane; b < B; b += 64) s = fma(A[(int64_t)row * Bp + b], dmu[b], s);
  s = wave_sum(s);
  if (lane == 0) g_m[row] = s * invB - m[row] * invN;
}

// g_LS = tril(2 G) - (LS - diag(1 / diag LS)) / N   (M x M, ld M; strictly-upper part zero)
__global__ __launch_bounds__(256) void svgp_gls_kernel(const double* __restrict__ G, int Mp, const double* __restrict__ LS,
                                                       int M, double invN, double* __restrict__ g_LS) {
  const int64_t total = (int64_t)M * M;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
    double v = 0.0;
    if (c <= r) {
      v = 2.0 * G[(int64_t)r * Mp + c] - LS[e] * invN;
      if (c == r) v += invN / LS[e];
    }
    g_LS[e] = v;
  }
}

// in place: X <- -tril(X)   /  X <- tril(X) with halved diagonal
__global__ __launch_bounds__(256) void svgp_tril_kernel(double* __restrict__ X, int Mp, double scale, int halve_diag) {
  const int64_t total = (int64_t)Mp * Mp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    double v = X[e] * scale;
    if (c > r) v = 0.0;
    else if (c == r && halve_diag) v *= 0.5;
    X[e] = v;
  }
}
// Kuubar (M x M, ld M) = (P + P^T) / 2 cropped
__global__ __launch_bounds__(256) void svgp_symcrop_kernel(const double* __restrict__ P, int Mp, int M, double* __restrict__ out) {
  const int64_t total = (int64_t)M * M;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
    out[e] = 0.5 * (P[(int64_t)r * Mp + c] + P[(int64_t)c * Mp + r]);
  }
}

// per-inducing-row partials of sum_{m,b} Kubbar[m][b] dKub[m][b]/d(.) :
//   part[m][q] = sum_b E df_q^2 (q < d), part[m][d] = sum_b Kubbar k', gz[m][q] = sum_b E df_q, E = Kubbar sf2 dk'/dr2
template <int KID>
__global__ __launch_bounds__(256) void svgp_kub_bwd_kernel(const double* __restrict__ Z, int64_t ldz, const double* __restrict__ Xb,
                                                           int64_t ldx, KernArgs ka, const double* __restrict__ Kbb, int Bp,
                                                           int B, double* __restrict__ part, double* __restrict__ gzraw) {
  __shared__ double red[4];
  const int m = blockIdx.x, d = ka.d;
  for (int q = 0; q <= d; ++q) {
    double s2 = 0.0, s1 = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
      double r2 = 0.0, dfq = 0.0;
      for (int j = 0; j < d; ++j) {
        const double df = (Z[m * ldz + j] - Xb[b * ldx + j]) * ka.inv_ls[j];
        r2 = fma(df, df, r2);
        if (j == q) dfq = df;
      }
      double kp, hp;
      kprofile_grad<KID>(r2, kp, hp);
      const double kb = Kbb[(int64_t)m * Bp + b];
      if (q == d) {
        s2 = fma(kb, kp, s2);
      } else {
        const double E = kb * ka.sf2 * hp;
        s1 = fma(E, dfq, s1);
        s2 = fma(E * dfq, dfq, s2);
      }
    }
    s2 = block_sum256(s2, red);
    s1 = block_sum256(s1, red);
    if (threadIdx.x == 0) {
      part[(size_t)m * (d + 1) + q] = s2;
      if (q < d) gzraw[(size_t)m * d + q] = s1;
    }
  }
}
// g_ls[q] = -2 inv_ls_q sum_m part[m][q] ; g_sf2 = sum_m part[m][d] + sum_b dv[b] / B ; g_Z[m][q] = 2 inv_ls_q gz[m][q]
__global__ __launch_bounds__(256) void svgp_kub_bwd_reduce_kernel(const double* __restrict__ part, const double* __restrict__ gzraw,
                                                                  int M, KernArgs ka, const double* __restrict__ dv, int B,
                                                                  double invB, double* g_ls, double* g_sf2, double* g_Z) {
  __shared__ double red[4];
  const int d = ka.d;
  if (blockIdx.x == 0) {
    for (int q = 0; q <= d; ++q) {
      double s = 0.0;
      for (int m = threadIdx.x; m < M; m += 256) s += part[(size_t)m * (d + 1) + q];
      s = block_sum256(s, red);
      if (q == d) {
        double t = 0.0;
        for (int b = threadIdx.x; b < B; b += 256) t += dv[b];
        t = block_sum256(t, red);
        if (threadIdx.x == 0) *g_sf2 = s + t * invB;
      } else if (threadIdx.x == 0) {
        g_ls[q] = -2.0 * ka.inv_ls[q] * s;
      }
    }
  }
  const int64_t total = (int64_t)M * d;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    g_Z[e] = 2.0 * ka.inv_ls[(int)(e % d)] * gzraw[e];
}

static int grid_for_s(int64_t total, int cap = 2048) {
  int64_t g = (total + 255) / 256;
  if (g < 1) g = 1;
  return (int)(g < cap ? g : cap);
}

constexpr int SVGP_SPLITK = 16;  // k-slices of the two M x M x B products of the reverse pass
struct SvgpWs {
  double *Kuu, *Kp, *Linv, *tmp, *LSp, *Kub, *A, *T, *U, *Abar, *Av, *S1, *Q, *P, *Kuubar;
  double *mp, *mu, *v, *dmu, *dv, *part, *kl, *kpart, *gzraw, *splitk;
  void* kuu_ws;
  int* flags;
  size_t kuu_ws_bytes, bytes;
};
static SvgpWs carve_svgp(void* ws, int Mp, int Bp, int M, int d) {
  Carver c(ws);
  SvgpWs w;
  const size_t mm = (size_t)Mp * Mp, mb = (size_t)Mp * Bp;
  w.Kuu = c.take<double>((size_t)M * M);
  w.Kp = c.take<double>(mm);
  w.Linv = c.take<double>(mm);
  w.tmp = c.take<double>(mm);
  w.LSp = c.take<double>(mm);
  w.S1 = c.take<double>(mm);
  w.Q = c.take<double>(mm);
  w.P = c.take<double>(mm);
  w.Kuubar = c.take<double>((size_t)M * M);
  w.Kub = c.take<double>(mb);
  w.A = c.take<double>(mb);
  w.T = c.take<double>(mb);
  w.U = c.take<double>(mb);
  w.Abar = c.take<double>(mb);
  w.Av = c.take<double>(mb);
  w.mp = c.take<double>(Mp);
  w.mu = c.take<double>(Bp);
  w.v = c.take<double>(Bp);
  w.dmu = c.take<double>(Bp);
  w.dv = c.take<double>(Bp);
  w.part = c.take<double>(128);
  w.kl = c.take<double>(64);
  w.splitk = c.take<double>((size_t)SVGP_SPLITK * mm);
  w.kpart = c.take<double>((size_t)M * (d + 1));
  w.gzraw = c.take<double>((size_t)M * d);
  w.kuu_ws_bytes = sgp_kuu_bwd_workspace_bytes(M, d);
  w.kuu_ws = c.take<char>(w.kuu_ws_bytes);
  w.flags = c.take<int>(potrf_scratch_ints(Mp));
  w.bytes = c.used();
  return w;
}

static KernArgs make_ka_s(const double* inv_ls, double sf2, int d) {
  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;
  return ka;
}

// shared forward: fills w.A, w.T, w.mu, w.v (and L in Kp, L^-1 in Linv, padded m, padded tril(LS))
static void svgp_forward(const SvgpWs& w, const double* Xb, int64_t ldx, int64_t B, const double* Z, int64_t ldz,
                         const double* inv_ls, double sf2, double jitter, const double* m, const double* LS, int M, int d,
                         int kernel_id, int Mp, int Bp, int* info, hipStream_t st) {
  const KernArgs ka = make_ka_s(inv_ls, sf2, d);
  zero_ints(info, 1, st);
  sgp_kuu(Z, ldz, inv_ls, sf2, jitter, M, d, kernel_id, w.Kuu, st);
  pad_copy(w.Kuu, M, M, M, w.Kp, Mp, Mp, Mp, 1.0, st);
  if (!potrf_lower(w.Kp, w.Linv, Mp, Mp, info, 0, w.flags, st)) tri_inverse(w.Kp, w.Linv, w.tmp, Mp, Mp, st);
  pad_copy(LS, M, M, M, w.LSp, Mp, Mp, Mp, 1.0, st);
  svgp_tril_kernel<<<grid_for_s((int64_t)Mp * Mp), 256, 0, st>>>(w.LSp, Mp, 1.0, 0);
  pad_copy(m, 1, M, 1, w.mp, 1, Mp, 1, 0.0, st);
  const int g = grid_for_s((int64_t)Mp * Bp);
  switch (kernel_id) {
    case SGP_KERNEL_RBF: svgp_kub_kernel<SGP_KERNEL_RBF><<<g, 256, 0, st>>>(Z, ldz, Xb, ldx, ka, M, Mp, (int)B, Bp, w.Kub); break;
    case SGP_KERNEL_MATERN32: svgp_kub_kernel<SGP_KERNEL_MATERN32><<<g, 256, 0, st>>>(Z, ldz, Xb, ldx, ka, M, Mp, (int)B, Bp, w.Kub); break;
    default: svgp_kub_kernel<SGP_KERNEL_MATERN52><<<g, 256, 0, st>>>(Z, ldz, Xb, ldx, ka, M, Mp, (int)B, Bp, w.Kub); break;
  }
  GemmDesc a;  // A = L^-1 Kub
  a.A = w.Linv; a.lda = Mp; a.B = w.Kub; a.ldb = Bp; a.C = w.A; a.ldc = Bp;
  a.m = Mp; a.n = Bp; a.k = Mp; a.khi_mask = 1;
  gemm(a, st);
  GemmDesc t;  // T = LS^T A
  t.A = w.LSp; t.lda = Mp; t.ta = true; t.B = w.A; t.ldb = Bp; t.C = w.T; t.ldc = Bp;
  t.m = Mp; t.n = Bp; t.k = Mp; t.klo_mask = 1;
  gemm(t, st);
  svgp_cols_kernel<<<Bp / 64, 256, 0, st>>>(w.A, w.T, w.mp, Mp, Bp, (int)B, sf2, w.mu, w.v);
}


// =====================================================================================================================
// The bound at S hyper-parameter samples in ONE chain of launches (sgp_svgp_elbo_batch) -- the five theta samples per
// minibatch of the reference's BayesianStochasticVariationalGP (models/bayesian_svgp.py:156-167).  Every launch below
// carries the sample index in blockIdx.y (GEMMs: GemmDesc::batch2), every per-sample buffer is an array of S equal
// slices, so the chain is ~36 launches whatever S is (5 x 41 before: the step was bound by the host enqueueing them).
// What does not depend on theta is done once: padded tril(L_S), padded m, the KL term.
// =====================================================================================================================
constexpr int SVGP_MAX_S = 8;
struct SvgpThetaS {  // a kernel argument (2.2 KB): theta never makes a host-to-device copy of its own
  KernArgs ka[SVGP_MAX_S];
  double s2[SVGP_MAX_S];
};

// LSp = tril(LS) padded to Mp x Mp with a unit diagonal ; mp = m zero padded ; status words <- 0
__global__ __launch_bounds__(256) void svgp_prep_batch_kernel(const double* __restrict__ LS, const double* __restrict__ m, int M, int Mp,
                                                              double* __restrict__ LSp, double* __restrict__ mp, int* __restrict__ info, int S) {
  const int64_t total = (int64_t)Mp * Mp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    double v = 0.0;
    if (r < M && c <= r) v = LS[(int64_t)r * M + c];
    else if (r >= M && r == c) v = 1.0;
    LSp[e] = v;
    if (e < Mp) mp[e] = e < M ? m[e] : 0.0;
    if (e < S) info[e] = 0;
  }
}

// Kp[s] = Kuu(theta_s) + jitter I, padded with the identity (the factorization's input), Mp x Mp
template <int KID>
__global__ __launch_bounds__(256) void svgp_kuu_batch_kernel(const double* __restrict__ Z, int64_t ldz, SvgpThetaS th, double jitter, int M,
                                                             int Mp, double* __restrict__ Kp) {
  const KernArgs& ka = th.ka[blockIdx.y];
  double* K = Kp + (int64_t)blockIdx.y * Mp * Mp;
  const int64_t total = (int64_t)Mp * Mp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / Mp), j = (int)(e - (int64_t)i * Mp);
    double v = (i == j) ? 1.0 : 0.0;
    if (i < M && j < M) {
      double r2 = 0.0;
      for (int q = 0; q < ka.d; ++q) {
        const double df = (Z[i * ldz + q] - Z[j * ldz + q]) * ka.inv_ls[q];
        r2 = fma(df, df, r2);
      }
      v = ka.sf2 * kprofile<KID>(r2);
      if (i == j) v += jitter;
    }
    K[e] = v;
  }
}

template <int KID>
__global__ __launch_bounds__(256) void svgp_kub_batch_kernel(const double* __restrict__ Z, int64_t ldz, const double* __restrict__ Xb,
                                                             int64_t ldx, SvgpThetaS th, int M, int Mp, int B, int Bp,
                                                             double* __restrict__ Kub) {
  const KernArgs& ka = th.ka[blockIdx.y];
  const int64_t total = (int64_t)Mp * Bp;
  double* K = Kub + (int64_t)blockIdx.y * total;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / Bp), b = (int)(e - (int64_t)m * Bp);
    double v = 0.0;
    if (m < M && b < B) {
      double r2 = 0.0;
      for (int q = 0; q < ka.d; ++q) {
        const double df = (Z[m * ldz + q] - Xb[b * ldx + q]) * ka.inv_ls[q];
        r2 = fma(df, df, r2);
      }
      v = ka.sf2 * kprofile<KID>(r2);
    }
    K[e] = v;
  }
}

__global__ __launch_bounds__(256) void svgp_cols_batch_kernel(const double* __restrict__ A, const double* __restrict__ T,
                                                              const double* __restrict__ mp, int Mp, int Bp, int B, SvgpThetaS th,
                                                              double* __restrict__ mu, double* __restrict__ v) {
  __shared__ double pm[4][64], pa[4][64], pt[4][64];
  const int64_t off = (int64_t)blockIdx.y * Mp * Bp;
  A += off; T += off; mu += (int64_t)blockIdx.y * Bp; v += (int64_t)blockIdx.y * Bp;
  const double kbb = th.ka[blockIdx.y].sf2;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6, l = threadIdx.x & 63;
  double sm = 0.0, sa = 0.0, st = 0.0;
  for (int m = w; m < Mp; m += 4) {
    const double a = A[(int64_t)m * Bp + col], t = T[(int64_t)m * Bp + col];
    sm = fma(a, mp[m], sm);
    sa = fma(a, a, sa);
    st = fma(t, t, st);
  }
  pm[w][l] = sm; pa[w][l] = sa; pt[w][l] = st;
  __syncthreads();
  if (w == 0) {
    mu[col] = (pm[0][l] + pm[1][l]) + (pm[2][l] + pm[3][l]);
    v[col] = col < B ? kbb - ((pa[0][l] + pa[1][l]) + (pa[2][l] + pa[3][l])) + ((pt[0][l] + pt[1][l]) + (pt[2][l] + pt[3][l])) : 1.0;
  }
}

// expected log-likelihood of sample blockIdx.y: same arithmetic as svgp_ell_kernel, 64 blocks per sample
__global__ __launch_bounds__(256) void svgp_ell_batch_kernel(const double* __restrict__ y, const double* __restrict__ mu,
                                                             const double* __restrict__ v, int B, int Bp, SvgpThetaS th, int lik, GHTable gh,
                                                             double* __restrict__ dmu, double* __restrict__ dv, double* __restrict__ part) {
  __shared__ double red[4];
  const int64_t ob = (int64_t)blockIdx.y * Bp;
  mu += ob; v += ob; dmu += ob; dv += ob; part += (int64_t)blockIdx.y * 128;
  const double s2 = th.s2[blockIdx.y];
  double se = 0.0, ss = 0.0;
  for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
    const double yb = y[b], m = mu[b], vv = v[b];
    double ell, gm, gv, gs = 0.0;
    if (lik == 0) {
      const double r = yb - m, q = r * r + vv;
      ell = -0.9189385332046727 - 0.5 * log(s2) - q / (2.0 * s2);
      gm = r / s2;
      gv = -0.5 / s2;
      gs = -0.5 / s2 + q / (2.0 * s2 * s2);
    } else {
      const double sd = sqrt(vv);
      ell = 0.0; gm = 0.0; gv = 0.0;
      for (int i = 0; i < GH_N; ++i) {
        const double z = yb * (m + sd * gh.x[i]);
        ell = fma(gh.w[i], log_ndtr_dev(z), ell);
        const double r = gh.w[i] * yb * mills_dev(z);
        gm += r;
        gv = fma(r, gh.x[i], gv);
      }
      gv = gv / (2.0 * sd);
    }
    dmu[b] = gm;
    dv[b] = gv;
    se += ell;
    ss += gs;
  }
  se = block_sum256(se, red);
  ss = block_sum256(ss, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = se;
    part[2 * blockIdx.x + 1] = ss;
  }
}

// out[s] = [ELBO per datum | sum_b E log p | KL | status word as a double], g_s2[s]; one thread per sample, fixed summation order
__global__ void svgp_finalize_batch_kernel(const double* __restrict__ part, const double* __restrict__ kl, int M, int B, double N_total,
                                           int S, const int* __restrict__ info, double* __restrict__ out, double* __restrict__ g_s2) {
  const int s = threadIdx.x;
  if (blockIdx.x != 0 || s >= S) return;
  double se = 0.0, ss = 0.0;
  for (int i = 0; i < 64; ++i) {
    se += part[(int64_t)s * 128 + 2 * i];
    ss += part[(int64_t)s * 128 + 2 * i + 1];
  }
  double klsum = 0.0;
  for (int i = 0; i < 64; ++i) klsum += kl[i];
  const double klv = 0.5 * (klsum - (double)M);
  out[4 * s] = se / (double)B - klv / N_total;
  out[4 * s + 1] = se;
  out[4 * s + 2] = klv;
  out[4 * s + 3] = (double)info[s];  // the factorization's status rides along: ONE device-to-host copy brings bounds and statuses
  if (g_s2) g_s2[s] = ss / (double)B;
}

__global__ __launch_bounds__(256) void svgp_abar_batch_kernel(const double* __restrict__ A, const double* __restrict__ U,
                                                              const double* __restrict__ mp, const double* __restrict__ dmu,
                                                              const double* __restrict__ dv, int Mp, int Bp, int B, double invB,
                                                              double* __restrict__ Abar, double* __restrict__ Av) {
  const int64_t total = (int64_t)Mp * Bp, off = (int64_t)blockIdx.y * total, ob = (int64_t)blockIdx.y * Bp;
  A += off; U += off; Abar += off; Av += off; dmu += ob; dv += ob;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / Bp), b = (int)(e - (int64_t)m * Bp);
    double ab = 0.0, av = 0.0;
    if (b < B) {
      const double mb = dmu[b] * invB, vb = dv[b] * invB;
      ab = mp[m] * mb + 2.0 * vb * (U[e] - A[e]);
      av = A[e] * vb;
    }
    Abar[e] = ab;
    Av[e] = av;
  }
}

__global__ __launch_bounds__(256) void svgp_gm_batch_kernel(const double* __restrict__ A, const double* __restrict__ dmu,
                                                            const double* __restrict__ m, int M, int Mp, int Bp, int B, double invB,
                                                            double invN, double* __restrict__ g_m) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  A += (int64_t)blockIdx.y * Mp * Bp; dmu += (int64_t)blockIdx.y * Bp; g_m += (int64_t)blockIdx.y * M;
  double s = 0.0;
  for (int b = lane; b < B; b += 64) s = fma(A[(int64_t)row * Bp + b], dmu[b], s);
  s = wave_sum(s);
  if (lane == 0) g_m[row] = s * invB - m[row] * invN;
}

__global__ __launch_bounds__(256) void svgp_gls_batch_kernel(const double* __restrict__ G, int Mp, const double* __restrict__ LS,
                                                             int M, double invN, double* __restrict__ g_LS) {
  const int64_t total = (int64_t)M * M;
  G += (int64_t)blockIdx.y * Mp * Mp; g_LS += (int64_t)blockIdx.y * total;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
    double v = 0.0;
    if (c <= r) {
      v = 2.0 * G[(int64_t)r * Mp + c] - LS[e] * invN;
      if (c == r) v += invN / LS[e];
    }
    g_LS[e] = v;
  }
}

__global__ __launch_bounds__(256) void svgp_tril_batch_kernel(double* __restrict__ X, int Mp, double scale, int halve_diag) {
  const int64_t total = (int64_t)Mp * Mp;
  X += (int64_t)blockIdx.y * total;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    double v = X[e] * scale;
    if (c > r) v = 0.0;
    else if (c == r && halve_diag) v *= 0.5;
    X[e] = v;
  }
}

// Kernel-derivative contractions of sample blockIdx.y, one workgroup per inducing row and source:
//   blockIdx.x <  M : row m of Kubbar (Mp x Bp)   against d Kub[m][.]  (data = the minibatch rows)
//   blockIdx.x >= M : row m of sym(P) (Mp x Mp)   against d Kuu[m][.]  (data = the inducing inputs; Kuubar = (P + P^T) / 2)
// part[src][m][q] = sum E df_q^2 (q < d), part[src][m][d] = sum Kbar k', gz[src][m][q] = sum E df_q, E = Kbar sf2 dk'/dr2
template <int KID>
__global__ __launch_bounds__(256) void svgp_kbwd_batch_kernel(const double* __restrict__ Z, int64_t ldz, const double* __restrict__ Xb,
                                                              int64_t ldx, SvgpThetaS th, const double* __restrict__ Kbb,
                                                              const double* __restrict__ P, int M, int Mp, int Bp, int B,
                                                              double* __restrict__ part, double* __restrict__ gzraw) {
  __shared__ double red[4];
  const KernArgs& ka = th.ka[blockIdx.y];
  const int d = ka.d;
  const bool uu = (int)blockIdx.x >= M;
  const int m = uu ? (int)blockIdx.x - M : (int)blockIdx.x;
  const double* data = uu ? Z : Xb;
  const int64_t ldd = uu ? ldz : ldx;
  const int n = uu ? M : B;
  Kbb += (int64_t)blockIdx.y * Mp * Bp;
  P += (int64_t)blockIdx.y * Mp * Mp;
  part += ((int64_t)blockIdx.y * 2 + (uu ? 1 : 0)) * M * (d + 1);
  gzraw += ((int64_t)blockIdx.y * 2 + (uu ? 1 : 0)) * M * d;
  for (int q = 0; q <= d; ++q) {
    double s2 = 0.0, s1 = 0.0;
    for (int b = threadIdx.x; b < n; b += 256) {
      double r2 = 0.0, dfq = 0.0;
      for (int j = 0; j < d; ++j) {
        const double df = (Z[m * ldz + j] - data[b * ldd + j]) * ka.inv_ls[j];
        r2 = fma(df, df, r2);
        if (j == q) dfq = df;
      }
      double kp, hp;
      kprofile_grad<KID>(r2, kp, hp);
      const double kb = uu ? 0.5 * (P[(int64_t)m * Mp + b] + P[(int64_t)b * Mp + m]) : Kbb[(int64_t)m * Bp + b];
      if (q == d) {
        s2 = fma(kb, kp, s2);
      } else {
        const double E = kb * ka.sf2 * hp;
        s1 = fma(E, dfq, s1);
        s2 = fma(E * dfq, dfq, s2);
      }
    }
    s2 = block_sum256(s2, red);
    s1 = block_sum256(s1, red);
    if (threadIdx.x == 0) {
      part[(size_t)m * (d + 1) + q] = s2;
      if (q < d) gzraw[(size_t)m * d + q] = s1;
    }
  }
}
// g_ls[s][q] = -2 inv_ls_q (sum_m part_ub + sum_m part_uu) ; g_sf2[s] = the two k' sums + sum_b dv / B ;
// g_Z[s][m][q] = inv_ls_q (2 gz_ub + 4 gz_uu): row AND column m of the symmetric Kuubar move with z_m
__global__ __launch_bounds__(256) void svgp_kbwd_reduce_batch_kernel(const double* __restrict__ part, const double* __restrict__ gzraw,
                                                                     int M, SvgpThetaS th, const double* __restrict__ dv, int B, int Bp,
                                                                     double invB, double* g_ls, double* g_sf2, double* g_Z) {
  __shared__ double red[4];
  const KernArgs& ka = th.ka[blockIdx.y];
  const int d = ka.d;
  const double* pub = part + (int64_t)blockIdx.y * 2 * M * (d + 1);
  const double* puu = pub + (int64_t)M * (d + 1);
  const double* zub = gzraw + (int64_t)blockIdx.y * 2 * M * d;
  const double* zuu = zub + (int64_t)M * d;
  dv += (int64_t)blockIdx.y * Bp;
  if (blockIdx.x == 0) {
    for (int q = 0; q <= d; ++q) {
      double s = 0.0;
      for (int m = threadIdx.x; m < M; m += 256) s += pub[(size_t)m * (d + 1) + q] + puu[(size_t)m * (d + 1) + q];
      s = block_sum256(s, red);
      if (q == d) {
        double t = 0.0;
        for (int b = threadIdx.x; b < B; b += 256) t += dv[b];
        t = block_sum256(t, red);
        if (threadIdx.x == 0) g_sf2[blockIdx.y] = s + t * invB;
      } else if (threadIdx.x == 0) {
        g_ls[(int64_t)blockIdx.y * d + q] = -2.0 * ka.inv_ls[q] * s;
      }
    }
  }
  const int64_t total = (int64_t)M * d;
  double* gz = g_Z + (int64_t)blockIdx.y * total;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    gz[e] = ka.inv_ls[(int)(e % d)] * (2.0 * zub[e] + 4.0 * zuu[e]);
}

// The reverse pass of a loss sum_s w_s ELBO_s in ONE launch: weighted sums over the S gradient slices (fixed order) for the
// parameters the samples share, and the per-sample hyper-parameter gradients scaled and packed for a single copy to the host:
//   gm = sum_s w_s g_m[s], gLS, gZ likewise ; gtheta[s] = w_s [g_sf2[s] | g_ls[s][0..d) | g_s2[s]]
struct SvgpWeights {
  double w[SVGP_MAX_S];
};
__global__ __launch_bounds__(256) void svgp_combine_kernel(SvgpWeights wt, int S, int M, int d, const double* __restrict__ g_m,
                                                           const double* __restrict__ g_LS, const double* __restrict__ g_Z,
                                                           const double* __restrict__ g_ls, const double* __restrict__ g_sf2,
                                                           const double* __restrict__ g_s2, double* __restrict__ gm, double* __restrict__ gLS,
                                                           double* __restrict__ gZ, double* __restrict__ gtheta) {
  const int64_t nm = M, nl = (int64_t)M * M, nz = (int64_t)M * d, nt = (int64_t)S * (d + 2);
  const int64_t total = nm + nl + nz + nt;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    if (e >= nm + nl + nz) {
      const int64_t t = e - (nm + nl + nz);
      const int sidx = (int)(t / (d + 2)), c = (int)(t - (int64_t)sidx * (d + 2));
      const double v = c == 0 ? g_sf2[sidx] : (c == d + 1 ? g_s2[sidx] : g_ls[(int64_t)sidx * d + c - 1]);
      gtheta[t] = wt.w[sidx] * v;
      continue;
    }
    const double* src;
    double* dst;
    int64_t n, i;
    if (e < nm) { src = g_m; dst = gm; n = nm; i = e; }
    else if (e < nm + nl) { src = g_LS; dst = gLS; n = nl; i = e - nm; }
    else { src = g_Z; dst = gZ; n = nz; i = e - nm - nl; }
    double acc = 0.0;
    for (int k = 0; k < S; ++k) acc = fma(wt.w[k], src[(int64_t)k * n + i], acc);
    dst[i] = acc;
  }
}

struct SvgpBatchWs {
  double *Kp, *Linv, *tmp, *LSp, *S1, *Q, *P, *Kub, *A, *T, *U, *Abar, *Av, *mp, *mu, *v, *dmu, *dv, *part, *kl, *splitk, *kpart, *gzraw;
  int* flags;
  size_t bytes;
};
static SvgpBatchWs carve_svgp_batch(void* ws, int Mp, int Bp, int M, int d, int S) {
  Carver c(ws);
  SvgpBatchWs w;
  const size_t mm = (size_t)Mp * Mp, mb = (size_t)Mp * Bp;
  w.LSp = c.take<double>(mm);
  w.mp = c.take<double>(Mp);
  w.kl = c.take<double>(64);
  w.Kp = c.take<double>(S * mm);
  w.Linv = c.take<double>(S * mm);
  w.tmp = c.take<double>(S * mm);
  w.S1 = c.take<double>(S * mm);
  w.Q = c.take<double>(S * mm);
  w.P = c.take<double>(S * mm);
  w.Kub = c.take<double>(S * mb);
  w.A = c.take<double>(S * mb);
  w.T = c.take<double>(S * mb);
  w.U = c.take<double>(S * mb);
  w.Abar = c.take<double>(S * mb);
  w.Av = c.take<double>(S * mb);
  w.mu = c.take<double>((size_t)S * Bp);
  w.v = c.take<double>((size_t)S * Bp);
  w.dmu = c.take<double>((size_t)S * Bp);
  w.dv = c.take<double>((size_t)S * Bp);
  w.part = c.take<double>((size_t)S * 128);
  w.splitk = c.take<double>((size_t)S * SVGP_SPLITK * mm);
  w.kpart = c.take<double>((size_t)S * 2 * M * (d + 1));
  w.gzraw = c.take<double>((size_t)S * 2 * M * d);
  w.flags = c.take<int>((size_t)S * potrf_scratch_ints(Mp));
  w.bytes = c.used();
  return w;
}


// =====================================================================================================================
// Mixture posterior predictive: the collapsed bound's predictive at S hyper-parameter samples in ONE chain of launches
// (sgp_mixture_predict) -- SURVEY section 8 f-2 "batched over S samples"; the reference recomputes everything per sample in a
// Python loop (models/bayesian_sgpr_hmc.py:198-231).  Same machinery as the batched SVGP bound above (sample index in
// blockIdx.y, two-level batched GEMMs, S factorizations per dataflow launch), PyMC3 op order per sample:
//   L = chol(Kuu + J I), A = L^-1 K_uf, B = I + A A^T / s2, L_B = chol(B), q = L_B^-1 A y,
//   A* = L^-1 K_u*, C* = L_B^-1 A*:  mean = C*^T q / s2,  var = k** - colsum(A* o A*) + colsum(C* o C*) (+ s2),
//   cov = K** - A*^T A* + C*^T C* (+ s2 I)          [models/sgpr.py:256-286 algebra]
// =====================================================================================================================
// u[s][m] (+)= sum_t A[s][m][t] y[t]   (one wave per row; rows >= Mp never launched)
__global__ __launch_bounds__(256) void mix_rowsdot_kernel(const double* __restrict__ A, int Mp, int Tp, int Tn, const double* __restrict__ y,
                                                          int accumulate, double* __restrict__ u) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Mp) return;
  A += (int64_t)blockIdx.y * Mp * Tp;
  u += (int64_t)blockIdx.y * Mp;
  double s = 0.0;
  for (int t = lane; t < Tn; t += 64) s = fma(A[(int64_t)row * Tp + t], y[t], s);
  s = wave_sum(s);
  if (lane == 0) u[row] = accumulate ? u[row] + s : s;
}
// Bm[s] = I + W[s] / s2[s]
__global__ __launch_bounds__(256) void mix_bmat_kernel(const double* __restrict__ W, SvgpThetaS th, int Mp, double* __restrict__ Bm) {
  const int64_t total = (int64_t)Mp * Mp, off = (int64_t)blockIdx.y * total;
  const double is2 = 1.0 / th.s2[blockIdx.y];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    Bm[off + e] = (r == c ? 1.0 : 0.0) + W[off + e] * is2;
  }
}
// (after chol(Kuu) and its inverse: the conditioning gate of sgp_tail.hip, cond_stats + cond_gate)
// after chol(B): a failure there is reported as M + pivot (only into a status word that is still 0)
__global__ void mix_status_kernel(int M, int* __restrict__ info, const int* __restrict__ infoB, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  if (info[s] == 0 && infoB[s] != 0) info[s] = infoB[s] < 0 ? infoB[s] : M + infoB[s];
}
// q[s] = LBinv[s] u[s]   (lower triangular, one wave per row)
__global__ __launch_bounds__(256) void mix_trmv_kernel(const double* __restrict__ Li, int Mp, const double* __restrict__ u, double* __restrict__ q) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Mp) return;
  Li += (int64_t)blockIdx.y * Mp * Mp;
  u += (int64_t)blockIdx.y * Mp;
  double s = 0.0;
  for (int j = lane; j <= row; j += 64) s = fma(Li[(int64_t)row * Mp + j], u[j], s);
  s = wave_sum(s);
  if (lane == 0) q[(int64_t)blockIdx.y * Mp + row] = s;
}
// mean[s][t] = sum_m C[m][t] q[m] / s2 ; var[s][t] = sf2 - sum A^2 + sum C^2 (+ s2)      (pred_cols_kernel per sample)
__global__ __launch_bounds__(256) void mix_pred_cols_kernel(const double* __restrict__ As, const double* __restrict__ Cm,
                                                            const double* __restrict__ q, int Mp, int Tp, int Tn, SvgpThetaS th, int pred_noise,
                                                            int64_t T, double* __restrict__ mean, double* __restrict__ var) {
  __shared__ double pm[4][64], pa[4][64], pc[4][64];
  const int64_t off = (int64_t)blockIdx.y * Mp * Tp;
  As += off; Cm += off; q += (int64_t)blockIdx.y * Mp;
  const double sf2 = th.ka[blockIdx.y].sf2, s2 = th.s2[blockIdx.y];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  double sm = 0.0, sa = 0.0, scc = 0.0;
  for (int m = w; m < Mp; m += 4) {
    const double a = As[(int64_t)m * Tp + col], c = Cm[(int64_t)m * Tp + col];
    sm = fma(c, q[m], sm);
    sa = fma(a, a, sa);
    scc = fma(c, c, scc);
  }
  pm[w][threadIdx.x & 63] = sm; pa[w][threadIdx.x & 63] = sa; pc[w][threadIdx.x & 63] = scc;
  __syncthreads();
  if (w == 0 && col < Tn) {
    const int l = threadIdx.x;
    mean[(int64_t)blockIdx.y * T + col] = (pm[0][l] + pm[1][l] + pm[2][l] + pm[3][l]) / s2;
    if (var)
      var[(int64_t)blockIdx.y * T + col] = sf2 - (pa[0][l] + pa[1][l] + pa[2][l] + pa[3][l]) + (pc[0][l] + pc[1][l] + pc[2][l] + pc[3][l]) +
                                          (pred_noise ? s2 : 0.0);
  }
}
// cov[s][t][t'] = k(xs_t, xs_t') - AtA[t][t'] + CtC[t][t'] (+ s2 on the diagonal) ; gate (Tp x Tp, optional) = cov + gate_jitter I
// padded with the identity: the input of the reference's PSD gate cholesky(cov + 1e-4 I), factored in place afterwards
template <int KID>
__global__ __launch_bounds__(256) void mix_pred_cov_kernel(const double* __restrict__ Xs, int64_t ldxs, SvgpThetaS th, const double* __restrict__ AtA,
                                                           const double* __restrict__ CtC, int Tp, int T, int pred_noise, double gate_jitter,
                                                           double* __restrict__ cov, double* __restrict__ gate) {
  const KernArgs& ka = th.ka[blockIdx.y];
  const double s2 = th.s2[blockIdx.y];
  const int64_t total = (int64_t)Tp * Tp, offp = (int64_t)blockIdx.y * total;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / Tp), j = (int)(e - (int64_t)i * Tp);
    double v = (i == j) ? 1.0 : 0.0, g = v;
    if (i < T && j < T) {
      double r2 = 0.0;
      for (int q = 0; q < ka.d; ++q) {
        const double df = (Xs[i * ldxs + q] - Xs[j * ldxs + q]) * ka.inv_ls[q];
        r2 = fma(df, df, r2);
      }
      v = ka.sf2 * kprofile<KID>(r2) - AtA[offp + e] + CtC[offp + e];
      if (i == j && pred_noise) v += s2;
      cov[(int64_t)blockIdx.y * T * T + (int64_t)i * T + j] = v;
      g = (i == j) ? v + gate_jitter : v;
    }
    if (gate) gate[offp + e] = g;
  }
}

struct MixWs {
  double *Kp, *Linv, *tmp, *W, *Bm, *LBinv, *u, *q, *Ks, *As, *Cm, *AtA, *CtC, *gate;
  int *infoB, *flags, *gflags;
  size_t bytes;
};
static int64_t mix_chunk(int64_t n) { return n < 8192 ? round_up64(n > 0 ? n : 1, 64) : 8192; }
static MixWs carve_mix(void* ws, int Mp, int64_t Cc, int64_t Tp, int S, bool want_cov, bool want_gate) {
  Carver c(ws);
  MixWs w{};
  const size_t mm = (size_t)Mp * Mp;
  w.Kp = c.take<double>(S * mm);
  w.Linv = c.take<double>(S * mm);
  w.tmp = c.take<double>(S * mm);
  w.W = c.take<double>(S * mm);
  w.Bm = c.take<double>(S * mm);
  w.LBinv = c.take<double>(S * mm);
  w.u = c.take<double>((size_t)S * Mp);
  w.q = c.take<double>((size_t)S * Mp);
  w.Ks = c.take<double>((size_t)S * Mp * Cc);
  w.As = c.take<double>((size_t)S * Mp * Cc);
  w.Cm = c.take<double>((size_t)S * Mp * Cc);
  if (want_cov) {
    w.AtA = c.take<double>((size_t)S * Tp * Tp);
    w.CtC = c.take<double>((size_t)S * Tp * Tp);
    if (want_gate) {
      w.gate = c.take<double>((size_t)S * Tp * Tp);
      w.gflags = c.take<int>((size_t)S * potrf_scratch_ints((int)Tp));
    }
  }
  w.infoB = c.take<int>(64);
  w.flags = c.take<int>((size_t)S * potrf_scratch_ints(Mp));
  w.bytes = c.used();
  return w;
}

}  // namespace sgp

using namespace sgp;

extern "C" int sgp_gauss_hermite(int n, double* x, double* w) {
  if (n <= 0 || n > 128 || !x || !w) return SGP_ERR_ARG;
  gauss_hermite_host(n, x, w);
  return SGP_OK;
}

extern "C" size_t sgp_svgp_workspace_bytes(int64_t B, int M, int d) {
  if (B <= 0 || B > (1 << 20) || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  return carve_svgp(nullptr, padded_m(M), (int)round_up64(B, 64), M, d).bytes;
}

extern "C" int sgp_svgp_elbo(const double* Xb, int64_t ldx, const double* yb, int64_t B, const double* Z, int64_t ldz,
                             const double* inv_ls, double sf2, double s2, double jitter, const double* m, const double* LS,
                             int64_t N_total, int M, int d, int kernel_id, int likelihood_id, int with_grads, double* out,
                             double* g_m, double* g_LS, double* g_Z, double* g_ls, double* g_sf2, double* g_s2, int* info,
                             void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!Xb || !yb || !Z || !inv_ls || !m || !LS || !out || !info || B <= 0 || M <= 0 || d <= 0 || ldx < d || ldz < d ||
      N_total <= 0)
    return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_MATERN52 || likelihood_id < 0 || likelihood_id > 1) return SGP_ERR_ARG;
  if (likelihood_id == 0 && !(s2 > 0.0)) return SGP_ERR_ARG;
  if (with_grads && (!g_m || !g_LS || !g_Z || !g_ls || !g_sf2 || !g_s2)) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING || B > (1 << 20)) return SGP_ERR_DIM;
  const int Mp = padded_m(M), Bp = (int)round_up64(B, 64);
  SvgpWs w = carve_svgp(ws, Mp, Bp, M, d);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const KernArgs ka = make_ka_s(inv_ls, sf2, d);
  static const GHTable gh = make_gh();
  const double invB = 1.0 / (double)B, invN = 1.0 / (double)N_total;

  svgp_forward(w, Xb, ldx, B, Z, ldz, inv_ls, sf2, jitter, m, LS, M, d, kernel_id, Mp, Bp, info, st);
  svgp_ell_kernel<<<64, 256, 0, st>>>(yb, w.mu, w.v, (int)B, s2, likelihood_id, gh, w.dmu, w.dv, w.part);
  svgp_kl_kernel<<<64, 256, 0, st>>>(m, LS, M, w.kl);
  svgp_finalize_kernel<<<1, 64, 0, st>>>(w.part, 64, w.kl, M, (int)B, (double)N_total, w.dv, out, with_grads ? g_s2 : nullptr);
  if (!with_grads) return check_launch();

  // ---- reverse pass -------------------------------------------------------------------------------
  GemmDesc u;  // U = LS T  (= LS LS^T A)
  u.A = w.LSp; u.lda = Mp; u.B = w.T; u.ldb = Bp; u.C = w.U; u.ldc = Bp;
  u.m = Mp; u.n = Bp; u.k = Mp; u.khi_mask = 1;
  gemm(u, st);
  svgp_abar_kernel<<<grid_for_s((int64_t)Mp * Bp), 256, 0, st>>>(w.A, w.U, w.mp, w.dmu, w.dv, Mp, Bp, (int)B, invB, w.Abar, w.Av);
  svgp_gm_kernel<<<(M + 3) / 4, 256, 0, st>>>(w.A, w.dmu, m, M, Bp, (int)B, invB, invN, g_m);
  GemmDesc gl;  // G = (A diag(vbar)) T^T  -> g_LS
  gl.A = w.Av; gl.lda = Bp; gl.B = w.T; gl.ldb = Bp; gl.tb = true; gl.C = w.S1; gl.ldc = Mp;
  gl.m = Mp; gl.n = Mp; gl.k = Bp;
  gemm_splitk(gl, SVGP_SPLITK, w.splitk, st);
  svgp_gls_kernel<<<grid_for_s((int64_t)M * M), 256, 0, st>>>(w.S1, Mp, LS, M, invN, g_LS);
  GemmDesc bb;  // Kubbar = L^-T Abar   (into U, no longer needed)
  bb.A = w.Linv; bb.lda = Mp; bb.ta = true; bb.B = w.Abar; bb.ldb = Bp; bb.C = w.U; bb.ldc = Bp;
  bb.m = Mp; bb.n = Bp; bb.k = Mp; bb.klo_mask = 1;
  gemm(bb, st);
  GemmDesc s1;  // S1 = Kubbar A^T ; Lbar = -tril(S1)
  s1.A = w.U; s1.lda = Bp; s1.B = w.A; s1.ldb = Bp; s1.tb = true; s1.C = w.S1; s1.ldc = Mp;
  s1.m = Mp; s1.n = Mp; s1.k = Bp;
  gemm_splitk(s1, SVGP_SPLITK, w.splitk, st);
  svgp_tril_kernel<<<grid_for_s((int64_t)Mp * Mp), 256, 0, st>>>(w.S1, Mp, -1.0, 0);
  GemmDesc q;  // Q = L^T Lbar ; Phi(Q)
  q.A = w.Kp; q.lda = Mp; q.ta = true; q.B = w.S1; q.ldb = Mp; q.C = w.Q; q.ldc = Mp;
  q.m = Mp; q.n = Mp; q.k = Mp; q.klo_mask = 3;
  gemm(q, st);
  svgp_tril_kernel<<<grid_for_s((int64_t)Mp * Mp), 256, 0, st>>>(w.Q, Mp, 1.0, 1);
  GemmDesc p1;  // tmp = L^-T Phi(Q)
  p1.A = w.Linv; p1.lda = Mp; p1.ta = true; p1.B = w.Q; p1.ldb = Mp; p1.C = w.tmp; p1.ldc = Mp;
  p1.m = Mp; p1.n = Mp; p1.k = Mp; p1.klo_mask = 3;
  gemm(p1, st);
  GemmDesc p2;  // P = tmp L^-1
  p2.A = w.tmp; p2.lda = Mp; p2.B = w.Linv; p2.ldb = Mp; p2.C = w.P; p2.ldc = Mp;
  p2.m = Mp; p2.n = Mp; p2.k = Mp; p2.klo_mask = 2;
  gemm(p2, st);
  svgp_symcrop_kernel<<<grid_for_s((int64_t)M * M), 256, 0, st>>>(w.P, Mp, M, w.Kuubar);
  // contraction of Kubbar and Kuubar with the kernel derivatives
  switch (kernel_id) {
    case SGP_KERNEL_RBF: svgp_kub_bwd_kernel<SGP_KERNEL_RBF><<<M, 256, 0, st>>>(Z, ldz, Xb, ldx, ka, w.U, Bp, (int)B, w.kpart, w.gzraw); break;
    case SGP_KERNEL_MATERN32: svgp_kub_bwd_kernel<SGP_KERNEL_MATERN32><<<M, 256, 0, st>>>(Z, ldz, Xb, ldx, ka, w.U, Bp, (int)B, w.kpart, w.gzraw); break;
    default: svgp_kub_bwd_kernel<SGP_KERNEL_MATERN52><<<M, 256, 0, st>>>(Z, ldz, Xb, ldx, ka, w.U, Bp, (int)B, w.kpart, w.gzraw); break;
  }
  svgp_kub_bwd_reduce_kernel<<<grid_for_s((int64_t)M * d, 256), 256, 0, st>>>(w.kpart, w.gzraw, M, ka, w.dv, (int)B, invB, g_ls, g_sf2, g_Z);
  const int rc = sgp_kuu_bwd(Z, ldz, inv_ls, sf2, w.Kuubar, M, d, kernel_id, g_ls, g_sf2, g_Z, w.kuu_ws, w.kuu_ws_bytes, st);
  if (rc != SGP_OK) return rc;
  return check_launch();
}

extern "C" size_t sgp_svgp_batch_workspace_bytes(int64_t B, int M, int d, int S) {
  if (B <= 0 || B > (1 << 20) || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING || S < 1 || S > SVGP_MAX_S) return 0;
  return carve_svgp_batch(nullptr, padded_m(M), (int)round_up64(B, 64), M, d, S).bytes;
}

// phase: 1 = forward (bounds, statuses, the likelihood's d/ds2), 2 = reverse from the state the forward left in `ws`, 3 = both
static int svgp_batch_impl(int phase, const double* Xb, int64_t ldx, const double* yb, int64_t B, const double* Z, int64_t ldz,
                           int S, const double* inv_ls, const double* sf2, const double* s2, double jitter, const double* m,
                           const double* LS, int64_t N_total, int M, int d, int kernel_id, int likelihood_id,
                           double* out, double* g_m, double* g_LS, double* g_Z, double* g_ls, double* g_sf2, double* g_s2,
                           int* info, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  const bool predict = phase == 4;  // latent predictive only: out = mean (S x B), g_m = variance (S x B), no labels, no likelihood
  const bool fwd = (phase & 1) != 0 || predict, with_grads = (phase & 2) != 0;
  if (!Xb || (!yb && !predict) || !Z || !inv_ls || !sf2 || !s2 || !m || !LS || B <= 0 || M <= 0 || d <= 0 || ldx < d || ldz < d || N_total <= 0)
    return SGP_ERR_ARG;
  if (fwd && (!out || !info)) return SGP_ERR_ARG;
  if (predict && !g_m) return SGP_ERR_ARG;
  if (S < 1 || S > SVGP_MAX_S) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_MATERN52 || likelihood_id < 0 || likelihood_id > 1) return SGP_ERR_ARG;
  if (with_grads && (!g_m || !g_LS || !g_Z || !g_ls || !g_sf2)) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING || B > (1 << 20)) return SGP_ERR_DIM;
  SvgpThetaS th{};
  for (int s = 0; s < S; ++s) {
    if (likelihood_id == 0 && !predict && !(s2[s] > 0.0)) return SGP_ERR_ARG;
    th.ka[s] = make_ka_s(inv_ls + (size_t)s * d, sf2[s], d);
    th.s2[s] = s2[s];
  }
  const int Mp = padded_m(M), Bp = (int)round_up64(B, 64);
  SvgpBatchWs w = carve_svgp_batch(ws, Mp, Bp, M, d, S);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  static const GHTable gh = make_gh();
  const double invB = 1.0 / (double)B, invN = 1.0 / (double)N_total;
  const int64_t mm = (int64_t)Mp * Mp, mb = (int64_t)Mp * Bp;
  const int gmm = grid_for_s(mm), gmb = grid_for_s(mb);
  auto gemm_s = [&](GemmDesc g, int64_t sa, int64_t sb, int64_t sc) {
    g.batch2 = S; g.s2A = sa; g.s2B = sb; g.s2C = sc;
    gemm(g, st);
  };
  if (fwd) {
    // ---- forward ------------------------------------------------------------------------------------------------
    svgp_prep_batch_kernel<<<gmm, 256, 0, st>>>(LS, m, M, Mp, w.LSp, w.mp, info, S);
    svgp_kl_kernel<<<64, 256, 0, st>>>(m, LS, M, w.kl);
    switch (kernel_id) {
      case SGP_KERNEL_RBF: svgp_kuu_batch_kernel<SGP_KERNEL_RBF><<<dim3(gmm, S), 256, 0, st>>>(Z, ldz, th, jitter, M, Mp, w.Kp); break;
      case SGP_KERNEL_MATERN32: svgp_kuu_batch_kernel<SGP_KERNEL_MATERN32><<<dim3(gmm, S), 256, 0, st>>>(Z, ldz, th, jitter, M, Mp, w.Kp); break;
      default: svgp_kuu_batch_kernel<SGP_KERNEL_MATERN52><<<dim3(gmm, S), 256, 0, st>>>(Z, ldz, th, jitter, M, Mp, w.Kp); break;
    }
    potrf_lower_batch(w.Kp, w.Linv, Mp, Mp, S, mm, info, w.flags, st);
    tri_inverse(w.Kp, w.Linv, w.tmp, Mp, Mp, st, S, mm);
    switch (kernel_id) {
      case SGP_KERNEL_RBF: svgp_kub_batch_kernel<SGP_KERNEL_RBF><<<dim3(gmb, S), 256, 0, st>>>(Z, ldz, Xb, ldx, th, M, Mp, (int)B, Bp, w.Kub); break;
      case SGP_KERNEL_MATERN32: svgp_kub_batch_kernel<SGP_KERNEL_MATERN32><<<dim3(gmb, S), 256, 0, st>>>(Z, ldz, Xb, ldx, th, M, Mp, (int)B, Bp, w.Kub); break;
      default: svgp_kub_batch_kernel<SGP_KERNEL_MATERN52><<<dim3(gmb, S), 256, 0, st>>>(Z, ldz, Xb, ldx, th, M, Mp, (int)B, Bp, w.Kub); break;
    }
    GemmDesc a;  // A = L^-1 Kub
    a.A = w.Linv; a.lda = Mp; a.B = w.Kub; a.ldb = Bp; a.C = w.A; a.ldc = Bp;
    a.m = Mp; a.n = Bp; a.k = Mp; a.khi_mask = 1;
    gemm_s(a, mm, mb, mb);
    GemmDesc t;  // T = LS^T A   (LS shared by the samples)
    t.A = w.LSp; t.lda = Mp; t.ta = true; t.B = w.A; t.ldb = Bp; t.C = w.T; t.ldc = Bp;
    t.m = Mp; t.n = Bp; t.k = Mp; t.klo_mask = 1;
    gemm_s(t, 0, mb, mb);
    svgp_cols_batch_kernel<<<dim3(Bp / 64, S), 256, 0, st>>>(w.A, w.T, w.mp, Mp, Bp, (int)B, th, w.mu, w.v);
    if (predict) {  // q(f*) per sample: mean and variance of the first B columns of every slice
      crop_copy(w.mu, Bp, out, B, S, (int)B, st);
      crop_copy(w.v, Bp, g_m, B, S, (int)B, st);
      return check_launch();
    }
    svgp_ell_batch_kernel<<<dim3(64, S), 256, 0, st>>>(yb, w.mu, w.v, (int)B, Bp, th, likelihood_id, gh, w.dmu, w.dv, w.part);
    svgp_finalize_batch_kernel<<<1, 64, 0, st>>>(w.part, w.kl, M, (int)B, (double)N_total, S, info, out, g_s2);
  }
  if (!with_grads) return check_launch();


  // ---- reverse (the closed-form adjoint of sgp_svgp_elbo, sample by sample in blockIdx.y) ------------------------
  GemmDesc u;  // U = LS T
  u.A = w.LSp; u.lda = Mp; u.B = w.T; u.ldb = Bp; u.C = w.U; u.ldc = Bp;
  u.m = Mp; u.n = Bp; u.k = Mp; u.khi_mask = 1;
  gemm_s(u, 0, mb, mb);
  svgp_abar_batch_kernel<<<dim3(gmb, S), 256, 0, st>>>(w.A, w.U, w.mp, w.dmu, w.dv, Mp, Bp, (int)B, invB, w.Abar, w.Av);
  svgp_gm_batch_kernel<<<dim3((M + 3) / 4, S), 256, 0, st>>>(w.A, w.dmu, m, M, Mp, Bp, (int)B, invB, invN, g_m);
  GemmDesc gl;  // G = (A diag(vbar)) T^T -> g_LS
  gl.A = w.Av; gl.lda = Bp; gl.B = w.T; gl.ldb = Bp; gl.tb = true; gl.C = w.S1; gl.ldc = Mp;
  gl.m = Mp; gl.n = Mp; gl.k = Bp;
  gl.batch2 = S; gl.s2A = mb; gl.s2B = mb; gl.s2C = mm;
  gemm_splitk(gl, SVGP_SPLITK, w.splitk, st);
  svgp_gls_batch_kernel<<<dim3(grid_for_s((int64_t)M * M), S), 256, 0, st>>>(w.S1, Mp, LS, M, invN, g_LS);
  GemmDesc bb;  // Kubbar = L^-T Abar (into U)
  bb.A = w.Linv; bb.lda = Mp; bb.ta = true; bb.B = w.Abar; bb.ldb = Bp; bb.C = w.U; bb.ldc = Bp;
  bb.m = Mp; bb.n = Bp; bb.k = Mp; bb.klo_mask = 1;
  gemm_s(bb, mm, mb, mb);
  GemmDesc s1;  // S1 = Kubbar A^T ; Lbar = -tril(S1)
  s1.A = w.U; s1.lda = Bp; s1.B = w.A; s1.ldb = Bp; s1.tb = true; s1.C = w.S1; s1.ldc = Mp;
  s1.m = Mp; s1.n = Mp; s1.k = Bp;
  s1.batch2 = S; s1.s2A = mb; s1.s2B = mb; s1.s2C = mm;
  gemm_splitk(s1, SVGP_SPLITK, w.splitk, st);
  svgp_tril_batch_kernel<<<dim3(gmm, S), 256, 0, st>>>(w.S1, Mp, -1.0, 0);
  GemmDesc q;  // Q = L^T Lbar ; Phi(Q)
  q.A = w.Kp; q.lda = Mp; q.ta = true; q.B = w.S1; q.ldb = Mp; q.C = w.Q; q.ldc = Mp;
  q.m = Mp; q.n = Mp; q.k = Mp; q.klo_mask = 3;
  gemm_s(q, mm, mm, mm);
  svgp_tril_batch_kernel<<<dim3(gmm, S), 256, 0, st>>>(w.Q, Mp, 1.0, 1);
  GemmDesc p1;  // tmp = L^-T Phi(Q)
  p1.A = w.Linv; p1.lda = Mp; p1.ta = true; p1.B = w.Q; p1.ldb = Mp; p1.C = w.tmp; p1.ldc = Mp;
  p1.m = Mp; p1.n = Mp; p1.k = Mp; p1.klo_mask = 3;
  gemm_s(p1, mm, mm, mm);
  GemmDesc p2;  // P = tmp L^-1 ; Kuubar = sym(P), read in place by the contraction kernel
  p2.A = w.tmp; p2.lda = Mp; p2.B = w.Linv; p2.ldb = Mp; p2.C = w.P; p2.ldc = Mp;
  p2.m = Mp; p2.n = Mp; p2.k = Mp; p2.klo_mask = 2;
  gemm_s(p2, mm, mm, mm);
  switch (kernel_id) {
    case SGP_KERNEL_RBF: svgp_kbwd_batch_kernel<SGP_KERNEL_RBF><<<dim3(2 * M, S), 256, 0, st>>>(Z, ldz, Xb, ldx, th, w.U, w.P, M, Mp, Bp, (int)B, w.kpart, w.gzraw); break;
    case SGP_KERNEL_MATERN32: svgp_kbwd_batch_kernel<SGP_KERNEL_MATERN32><<<dim3(2 * M, S), 256, 0, st>>>(Z, ldz, Xb, ldx, th, w.U, w.P, M, Mp, Bp, (int)B, w.kpart, w.gzraw); break;
    default: svgp_kbwd_batch_kernel<SGP_KERNEL_MATERN52><<<dim3(2 * M, S), 256, 0, st>>>(Z, ldz, Xb, ldx, th, w.U, w.P, M, Mp, Bp, (int)B, w.kpart, w.gzraw); break;
  }
  svgp_kbwd_reduce_batch_kernel<<<dim3(grid_for_s((int64_t)M * d, 256), S), 256, 0, st>>>(w.kpart, w.gzraw, M, th, w.dv, (int)B, Bp, invB, g_ls, g_sf2, g_Z);
  return check_launch();
}


extern "C" int sgp_svgp_elbo_batch(const double* Xb, int64_t ldx, const double* yb, int64_t B, const double* Z, int64_t ldz,
                                   int S, const double* inv_ls, const double* sf2, const double* s2, double jitter, const double* m,
                                   const double* LS, int64_t N_total, int M, int d, int kernel_id, int likelihood_id, int with_grads,
                                   double* out, double* g_m, double* g_LS, double* g_Z, double* g_ls, double* g_sf2, double* g_s2,
                                   int* info, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (with_grads && !g_s2) return SGP_ERR_ARG;
  return svgp_batch_impl(with_grads ? 3 : 1, Xb, ldx, yb, B, Z, ldz, S, inv_ls, sf2, s2, jitter, m, LS, N_total, M, d, kernel_id,
                         likelihood_id, out, g_m, g_LS, g_Z, g_ls, g_sf2, with_grads ? g_s2 : nullptr, info, ws, ws_bytes, stream);
}

// The reverse pass alone, from the state a sgp_svgp_elbo_batch(with_grads = 0) call with the SAME arguments left in `ws`
// (nothing else may have used the workspace in between).  Lets a caller read the bounds back -- they are final after the
// forward -- and do its host-side bookkeeping while the device runs the reverse chain.  g_s2 is the forward's output.
extern "C" int sgp_svgp_elbo_batch_reverse(const double* Xb, int64_t ldx, const double* yb, int64_t B, const double* Z, int64_t ldz,
                                           int S, const double* inv_ls, const double* sf2, const double* s2, double jitter,
                                           const double* m, const double* LS, int64_t N_total, int M, int d, int kernel_id,
                                           int likelihood_id, double* g_m, double* g_LS, double* g_Z, double* g_ls, double* g_sf2,
                                           void* ws, size_t ws_bytes, sgp_stream_t stream) {
  return svgp_batch_impl(2, Xb, ldx, yb, B, Z, ldz, S, inv_ls, sf2, s2, jitter, m, LS, N_total, M, d, kernel_id, likelihood_id,
                         nullptr, g_m, g_LS, g_Z, g_ls, g_sf2, nullptr, nullptr, ws, ws_bytes, stream);
}

// latent predictive mean / variance of q(f*) at T rows for S hyper-parameter samples (the mixture predictive of
// BayesianStochasticVariationalGP, models/bayesian_svgp.py:183-207): the forward half of the chain above with the test rows in
// place of the minibatch.  mean, var: S x T (device); info: S ints.  Workspace: sgp_svgp_batch_workspace_bytes(T, M, d, S).
extern "C" int sgp_svgp_predict_batch(const double* Xs, int64_t ldxs, int64_t T, const double* Z, int64_t ldz, int S, const double* inv_ls,
                                      const double* sf2, double jitter, const double* m, const double* LS, int M, int d, int kernel_id,
                                      double* mean, double* var, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!sf2 || S < 1 || S > SVGP_MAX_S) return SGP_ERR_ARG;
  double ones[SVGP_MAX_S];
  for (int s = 0; s < SVGP_MAX_S; ++s) ones[s] = 1.0;
  return svgp_batch_impl(4, Xs, ldxs, nullptr, T, Z, ldz, S, inv_ls, sf2, ones, jitter, m, LS, 1, M, d, kernel_id, 0, mean, var, nullptr,
                         nullptr, nullptr, nullptr, nullptr, info, ws, ws_bytes, stream);
}

// forward with the likelihood's d/ds2 (g_s2, S doubles, may be NULL): the first half of the split call
extern "C" int sgp_svgp_elbo_batch_forward(const double* Xb, int64_t ldx, const double* yb, int64_t B, const double* Z, int64_t ldz,
                                           int S, const double* inv_ls, const double* sf2, const double* s2, double jitter,
                                           const double* m, const double* LS, int64_t N_total, int M, int d, int kernel_id,
                                           int likelihood_id, double* out, double* g_s2, int* info, void* ws, size_t ws_bytes,
                                           sgp_stream_t stream) {
  return svgp_batch_impl(1, Xb, ldx, yb, B, Z, ldz, S, inv_ls, sf2, s2, jitter, m, LS, N_total, M, d, kernel_id, likelihood_id,
                         out, nullptr, nullptr, nullptr, nullptr, nullptr, g_s2, info, ws, ws_bytes, stream);
}

extern "C" int sgp_svgp_batch_combine(int S, const double* weights, int M, int d, const double* g_m, const double* g_LS,
                                      const double* g_Z, const double* g_ls, const double* g_sf2, const double* g_s2, double* gm_out,
                                      double* gLS_out, double* gZ_out, double* gtheta_out, sgp_stream_t stream) {
  if (S < 1 || S > SVGP_MAX_S || !weights || M <= 0 || d <= 0 || !g_m || !g_LS || !g_Z || !g_ls || !g_sf2 || !g_s2 || !gm_out ||
      !gLS_out || !gZ_out || !gtheta_out)
    return SGP_ERR_ARG;
  SvgpWeights wt{};
  for (int k = 0; k < S; ++k) wt.w[k] = weights[k];
  const int64_t total = (int64_t)M + (int64_t)M * M + (int64_t)M * d + (int64_t)S * (d + 2);
  svgp_combine_kernel<<<grid_for_s(total), 256, 0, (hipStream_t)stream>>>(wt, S, M, d, g_m, g_LS, g_Z, g_ls, g_sf2, g_s2, gm_out, gLS_out,
                                                                          gZ_out, gtheta_out);
  return check_launch();
}

extern "C" size_t sgp_mixture_predict_workspace_bytes(int64_t N, int64_t T, int M, int d, int S, int want_cov, int want_gate) {
  if (N < 1 || T < 1 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING || S < 1 || S > SVGP_MAX_S) return 0;
  if (want_cov && T > 8192) return 0;
  const int64_t Tp = round_up64(T, 64);
  const int64_t Cc = want_cov ? (Tp > mix_chunk(N) ? Tp : mix_chunk(N)) : (mix_chunk(N) > mix_chunk(T) ? mix_chunk(N) : mix_chunk(T));
  return carve_mix(nullptr, padded_m(M), Cc, Tp, S, want_cov != 0, want_gate != 0).bytes;
}

extern "C" int sgp_mixture_predict(const double* X, int64_t ldx, const double* y, int64_t N, const double* Xs, int64_t ldxs, int64_t T,
                                   const double* Z, int64_t ldz, int S, const double* inv_ls, const double* sf2, const double* s2,
                                   double jitter, int M, int d, int kernel_id, int pred_noise, double gate_jitter, double* mean,
                                   double* var, double* cov, int* info, int* gate_info, void* ws, size_t ws_bytes,
                                   sgp_stream_t stream) {
  if (!X || !y || !Xs || !Z || !inv_ls || !sf2 || !s2 || !mean || !info || N < 1 || T < 1 || M <= 0 || d <= 0 || ldx < d || ldxs < d || ldz < d)
    return SGP_ERR_ARG;
  if (S < 1 || S > SVGP_MAX_S || kernel_id < 0 || kernel_id > SGP_KERNEL_MATERN52) return SGP_ERR_ARG;
  if (gate_info && !cov) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING || (cov && T > 8192)) return SGP_ERR_DIM;
  SvgpThetaS th{};
  for (int s = 0; s < S; ++s) {
    if (!(s2[s] > 0.0) || !(sf2[s] > 0.0)) return SGP_ERR_ARG;
    th.ka[s] = make_ka_s(inv_ls + (size_t)s * d, sf2[s], d);
    th.s2[s] = s2[s];
  }
  const int Mp = padded_m(M);
  const int64_t Tp = round_up64(T, 64);
  const bool want_cov = cov != nullptr, want_gate = gate_info != nullptr;
  const int64_t Cc = want_cov ? (Tp > mix_chunk(N) ? Tp : mix_chunk(N)) : (mix_chunk(N) > mix_chunk(T) ? mix_chunk(N) : mix_chunk(T));
  MixWs w = carve_mix(ws, Mp, Cc, Tp, S, want_cov, want_gate);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t mm = (int64_t)Mp * Mp;
  const int gmm = grid_for_s(mm);
  auto gemm_s = [&](GemmDesc g, int64_t sa, int64_t sb, int64_t sc) {
    g.batch2 = S; g.s2A = sa; g.s2B = sb; g.s2C = sc;
    gemm(g, st);
  };
  auto kmat = [&](const double* P, int64_t ldp, int n, int np, double* out) {  // K(Z, P) for every sample: Mp x np, zero padded
    const dim3 grid(grid_for_s((int64_t)Mp * np), S);
    switch (kernel_id) {
      case SGP_KERNEL_RBF: svgp_kub_batch_kernel<SGP_KERNEL_RBF><<<grid, 256, 0, st>>>(Z, ldz, P, ldp, th, M, Mp, n, np, out); break;
      case SGP_KERNEL_MATERN32: svgp_kub_batch_kernel<SGP_KERNEL_MATERN32><<<grid, 256, 0, st>>>(Z, ldz, P, ldp, th, M, Mp, n, np, out); break;
      default: svgp_kub_batch_kernel<SGP_KERNEL_MATERN52><<<grid, 256, 0, st>>>(Z, ldz, P, ldp, th, M, Mp, n, np, out); break;
    }
  };

  // ---- train side: L, L^-1, W = A A^T, u = A y over row chunks, B, L_B^-1, q -------------------------------------------
  zero_ints(info, S, st);
  switch (kernel_id) {
    case SGP_KERNEL_RBF: svgp_kuu_batch_kernel<SGP_KERNEL_RBF><<<dim3(gmm, S), 256, 0, st>>>(Z, ldz, th, jitter, M, Mp, w.Kp); break;
    case SGP_KERNEL_MATERN32: svgp_kuu_batch_kernel<SGP_KERNEL_MATERN32><<<dim3(gmm, S), 256, 0, st>>>(Z, ldz, th, jitter, M, Mp, w.Kp); break;
    default: svgp_kuu_batch_kernel<SGP_KERNEL_MATERN52><<<dim3(gmm, S), 256, 0, st>>>(Z, ldz, th, jitter, M, Mp, w.Kp); break;
  }
  potrf_lower_batch(w.Kp, w.Linv, Mp, Mp, S, mm, info, w.flags, st);
  tri_inverse(w.Kp, w.Linv, w.tmp, Mp, Mp, st, S, mm);
  if (cond_gate_limit() > 0.0) {  // w.tmp is free again (S Mp^2 doubles >= S cond_scratch_doubles(M))
    cond_stats(w.Kp, w.Linv, Mp, (int64_t)mm, M, S, w.tmp, st);
    cond_gate(w.tmp, M, S, cond_gate_limit(), info, st);
  }
  const int64_t Nc = mix_chunk(N);
  for (int64_t t0 = 0; t0 < N; t0 += Nc) {
    const int Tn = (int)((N - t0) < Nc ? (N - t0) : Nc);
    const int Np = (int)round_up64(Tn, 64);
    const int64_t mb = (int64_t)Mp * Np;
    kmat(X + t0 * ldx, ldx, Tn, Np, w.Ks);
    GemmDesc a;  // A = L^-1 K_uf
    a.A = w.Linv; a.lda = Mp; a.B = w.Ks; a.ldb = Np; a.C = w.As; a.ldc = Np;
    a.m = Mp; a.n = Np; a.k = Mp; a.khi_mask = 1;
    gemm_s(a, mm, mb, mb);
    GemmDesc ww;  // W (+)= A A^T
    ww.A = w.As; ww.lda = Np; ww.B = w.As; ww.ldb = Np; ww.tb = true; ww.C = w.W; ww.ldc = Mp;
    ww.m = Mp; ww.n = Mp; ww.k = Np; ww.beta = t0 > 0 ? 1.0 : 0.0;
    gemm_s(ww, mb, mb, mm);
    mix_rowsdot_kernel<<<dim3(Mp / 4, S), 256, 0, st>>>(w.As, Mp, Np, Tn, y + t0, t0 > 0 ? 1 : 0, w.u);
  }
  mix_bmat_kernel<<<dim3(gmm, S), 256, 0, st>>>(w.W, th, Mp, w.Bm);
  zero_ints(w.infoB, S, st);
  potrf_lower_batch(w.Bm, w.LBinv, Mp, Mp, S, mm, w.infoB, w.flags, st);
  mix_status_kernel<<<1, 64, 0, st>>>(M, info, w.infoB, S);
  tri_inverse(w.Bm, w.LBinv, w.tmp, Mp, Mp, st, S, mm);
  mix_trmv_kernel<<<dim3(Mp / 4, S), 256, 0, st>>>(w.LBinv, Mp, w.u, w.q);

  // ---- test side ------------------------------------------------------------------------------------------------------
  const int64_t Tc = want_cov ? Tp : mix_chunk(T);
  for (int64_t t0 = 0; t0 < T; t0 += Tc) {
    const int Tn = (int)((T - t0) < Tc ? (T - t0) : Tc);
    const int Tq = (int)round_up64(Tn, 64);
    const int64_t mb = (int64_t)Mp * Tq;
    const double* xs = Xs + t0 * ldxs;
    kmat(xs, ldxs, Tn, Tq, w.Ks);
    GemmDesc a;  // A* = L^-1 K_u*
    a.A = w.Linv; a.lda = Mp; a.B = w.Ks; a.ldb = Tq; a.C = w.As; a.ldc = Tq;
    a.m = Mp; a.n = Tq; a.k = Mp; a.khi_mask = 1;
    gemm_s(a, mm, mb, mb);
    GemmDesc b;  // C* = L_B^-1 A*
    b.A = w.LBinv; b.lda = Mp; b.B = w.As; b.ldb = Tq; b.C = w.Cm; b.ldc = Tq;
    b.m = Mp; b.n = Tq; b.k = Mp; b.khi_mask = 1;
    gemm_s(b, mm, mb, mb);
    mix_pred_cols_kernel<<<dim3(Tq / 64, S), 256, 0, st>>>(w.As, w.Cm, w.q, Mp, Tq, Tn, th, pred_noise, T, mean + t0, var ? var + t0 : nullptr);
    if (want_cov) {  // one chunk: Tc = Tp
      const int64_t tt = (int64_t)Tq * Tq;
      GemmDesc x;
      x.A = w.As; x.lda = Tq; x.ta = true; x.B = w.As; x.ldb = Tq; x.C = w.AtA; x.ldc = Tq;
      x.m = Tq; x.n = Tq; x.k = Mp;
      gemm_s(x, mb, mb, tt);
      GemmDesc yv;
      yv.A = w.Cm; yv.lda = Tq; yv.ta = true; yv.B = w.Cm; yv.ldb = Tq; yv.C = w.CtC; yv.ldc = Tq;
      yv.m = Tq; yv.n = Tq; yv.k = Mp;
      gemm_s(yv, mb, mb, tt);
      const dim3 gc(grid_for_s(tt), S);
      switch (kernel_id) {
        case SGP_KERNEL_RBF: mix_pred_cov_kernel<SGP_KERNEL_RBF><<<gc, 256, 0, st>>>(xs, ldxs, th, w.AtA, w.CtC, Tq, Tn, pred_noise, gate_jitter, cov, w.gate); break;
        case SGP_KERNEL_MATERN32: mix_pred_cov_kernel<SGP_KERNEL_MATERN32><<<gc, 256, 0, st>>>(xs, ldxs, th, w.AtA, w.CtC, Tq, Tn, pred_noise, gate_jitter, cov, w.gate); break;
        default: mix_pred_cov_kernel<SGP_KERNEL_MATERN52><<<gc, 256, 0, st>>>(xs, ldxs, th, w.AtA, w.CtC, Tq, Tn, pred_noise, gate_jitter, cov, w.gate); break;
      }
      if (want_gate) zero_ints(gate_info, S, st);
      if (want_gate) potrf_lower_batch(w.gate, nullptr, Tq, Tq, S, tt, gate_info, w.gflags, st);  // the reference's PSD gate, S at a time
    }
  }
  return check_launch();
}

extern "C" int sgp_svgp_predict(const double* Xs, int64_t ldxs, int64_t T, const double* Z, int64_t ldz, const double* inv_ls,
                                double sf2, double jitter, const double* m, const double* LS, int M, int d, int kernel_id,
                                double* mean, double* var, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!Xs || !Z || !inv_ls || !m || !LS || !mean || !var || !info || T <= 0 || M <= 0 || d <= 0 || ldxs < d || ldz < d)
    return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_MATERN52) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING || T > (1 << 20)) return SGP_ERR_DIM;
  const int Mp = padded_m(M), Bp = (int)round_up64(T, 64);
  SvgpWs w = carve_svgp(ws, Mp, Bp, M, d);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  svgp_forward(w, Xs, ldxs, T, Z, ldz, inv_ls, sf2, jitter, m, LS, M, d, kernel_id, Mp, Bp, info, st);
  crop_copy(w.mu, 1, mean, 1, (int)T, 1, st);
  crop_copy(w.v, 1, var, 1, (int)T, 1, st);
  return check_launch();
}
