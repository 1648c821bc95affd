// The O(M^3) tail of the collapsed bound, its adjoints, the Kuu block and the predictive.
//
//   L = chol(Kuu) ; W = L^-1 Phi L^-T ; B = I + W/s2 ; LB = chol(B) ; q = LB^-1 L^-1 b
//   F = -[ N/2 log 2pi + N/2 log s2 + sum log diag LB + (yy/s2 - q.q/s2^2)/2 + (kappa - tr W)/(2 s2) ]
//
// This is the op order of pymc3's MarginalSparse VFE logp (reference models/bayesian_sgpr_hmc.py:66,71)
// applied to the streamed sufficient statistics, and equals gpytorch's
// ExactMarginalLogLikelihood + InducingPointKernelAddedLossTerm on the reference's SparseGPR
// (models/sgpr.py:37,114,125) times N.  Everything runs on padded Mp x Mp matrices (Mp = 128-multiple;
// padding rows are identity for Kuu / zero for Phi so factors stay exact) through the MFMA GEMM,
// blocked Cholesky and recursive triangular inverse of sgp_dense.hip; no host round trip.
#include "sgp_dense.hpp"
#include "sgp_ctx.hpp"
#include "sgp_composite.hpp"

namespace sgp {

// ---------------------------------------------------------------------------------------------
// Kuu
// ---------------------------------------------------------------------------------------------
template <int KID>
__global__ __launch_bounds__(256) void kuu_kernel(const double* __restrict__ Z, int64_t ldz, KernArgs ka, double jitter,
                                                  int M, double* __restrict__ Kuu) {
  const int64_t total = (int64_t)M * M;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / M), j = (int)(e - (int64_t)i * M);
    double r2 = 0.0;
    for (int q = 0; q < ka.d; ++q) {
      const double df = (Z[i * ldz + q] - Z[j * ldz + q]) * ka.inv_ls[q];
      r2 = fma(df, df, r2);
    }
    double v = ka.sf2 * kprofile<KID>(r2);
    if (i == j) v += jitter;
    Kuu[e] = v;
  }
}

// per-row partials of  sum_{m,m'} Kuubar[m][m'] dKuu[m][m']/d(.)  (Kuubar treated as symmetric)
//   part[m][0..d)  = sum_m' E (z~_mj - z~_m'j)^2         (E = Kuubar sf2 dk'/dr2)
//   part[m][d]     = sum_m' Kuubar k'
//   gz[m][j]       = sum_m' E (z~_mj - z~_m'j)
template <int KID>
__global__ __launch_bounds__(256) void kuu_bwd_kernel(const double* __restrict__ Z, int64_t ldz, KernArgs ka,
                                                      const double* __restrict__ Kb, int M,
                                                      double* __restrict__ part, double* __restrict__ gzraw) {
  __shared__ double red[4][2 * SGP_MAX_DIM + 1];
  const int m = blockIdx.x;
  const int d = ka.d;
  constexpr int MAXC = SGP_MAX_INDUCING / 256;
  double E[MAXC], KK[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int mp = c * 256 + threadIdx.x;
    E[c] = 0.0;
    KK[c] = 0.0;
    if (mp < M) {
      double r2 = 0.0;
      for (int q = 0; q < d; ++q) {
        const double df = (Z[m * ldz + q] - Z[mp * ldz + q]) * ka.inv_ls[q];
        r2 = fma(df, df, r2);
      }
      double kp, hp;
      kprofile_grad<KID>(r2, kp, hp);
      const double kb = Kb[(int64_t)m * M + mp];
      E[c] = kb * ka.sf2 * hp;
      KK[c] = kb * kp;
    }
  }
  // all 2 d + 1 sums of the row behind ONE barrier (a block_sum256 each -- 74 barriers at d = 18 -- was most of the kernel's
  // 38 us at C3): wave sums into LDS, then one thread per sum adds the four waves in a fixed order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) s += KK[c];
  s = wave_sum(s);
  if (lane == 0) red[wave][2 * d] = s;
  for (int q = 0; q < d; ++q) {
    double s2 = 0.0, s1 = 0.0;
    const double zm = Z[m * ldz + q];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int mp = c * 256 + threadIdx.x;
      if (mp < M) {
        const double df = (zm - Z[mp * ldz + q]) * ka.inv_ls[q];
        s1 = fma(E[c], df, s1);
        s2 = fma(E[c] * df, df, s2);
      }
    }
    s2 = wave_sum(s2);
    s1 = wave_sum(s1);
    if (lane == 0) {
      red[wave][q] = s2;
      red[wave][d + q] = s1;
    }
  }
  __syncthreads();
  if (threadIdx.x <= 2 * d) {
    const int t = threadIdx.x;
    const double v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if (t < d) part[(size_t)m * (d + 1) + t] = v;
    else if (t < 2 * d) gzraw[(size_t)m * d + (t - d)] = v;
    else part[(size_t)m * (d + 1) + d] = v;
  }
}
// The same sums when dF/dZ is not wanted (every leapfrog of a sampler): only the totals over (m, m') are needed, so the 2 d + 1 wave
// reductions PER ROW of the kernel above (most of its 28 us at C3) become d + 1 per workgroup: workgroup <-> rows blockIdx, blockIdx +
// grid, ..., thread <-> columns, one pass, every sum carried in registers.  part[workgroup][0 .. d] in the layout of part[m][.] above:
// kuu_bwd_reduce_kernel adds the workgroups' rows instead of the matrix rows.  Fixed assignment, fixed order: the same bits on every
// rank and every call (not the bits of the per-row kernel: the terms are grouped differently).
template <int KID>
__global__ __launch_bounds__(256) void kuu_bwd_total_kernel(const double* __restrict__ Z, int64_t ldz, KernArgs ka,
                                                            const double* __restrict__ Kb, int M, double* __restrict__ part) {
  __shared__ double red[4][SGP_MAX_DIM + 1];
  const int d = ka.d;
  double s2[SGP_MAX_DIM];
#pragma unroll
  for (int q = 0; q < SGP_MAX_DIM; ++q) s2[q] = 0.0;
  double sk = 0.0;
  for (int m = blockIdx.x; m < M; m += gridDim.x) {
    double zm[SGP_MAX_DIM];
#pragma unroll
    for (int q = 0; q < SGP_MAX_DIM; ++q) zm[q] = q < d ? Z[m * ldz + q] : 0.0;
    for (int mp = threadIdx.x; mp < M; mp += 256) {
      double df[SGP_MAX_DIM];
      double r2 = 0.0;
#pragma unroll
      for (int q = 0; q < SGP_MAX_DIM; ++q) {
        if (q < d) {
          df[q] = (zm[q] - Z[mp * ldz + q]) * ka.inv_ls[q];
          r2 = fma(df[q], df[q], r2);
        }
      }
      double kp, hp;
      kprofile_grad<KID>(r2, kp, hp);
      const double kb = Kb[(int64_t)m * M + mp];
      const double E = kb * ka.sf2 * hp;
      sk = fma(kb, kp, sk);
#pragma unroll
      for (int q = 0; q < SGP_MAX_DIM; ++q) {
        if (q < d) s2[q] = fma(E * df[q], df[q], s2[q]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sk = wave_sum(sk);
  if (lane == 0) red[wave][d] = sk;
#pragma unroll
  for (int q = 0; q < SGP_MAX_DIM; ++q) {
    if (q < d) {
      const double a = wave_sum(s2[q]);
      if (lane == 0) red[wave][q] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x <= d) {
    const int t = threadIdx.x;
    part[(size_t)blockIdx.x * (d + 1) + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  }
}
__global__ __launch_bounds__(256) void kuu_bwd_reduce_kernel(const double* __restrict__ part, const double* __restrict__ gzraw,
                                                             int M, KernArgs ka, double* g_ls, double* g_sf2, double* g_Z) {
  const int d = ka.d;
  if (blockIdx.x == 0) {
    // wave w sums parameters w, w + 4, ...: lanes stride over the rows, one wave reduction each, no barrier; fixed order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = wave; q <= d; q += 4) {
      double s = 0.0;
      for (int m = lane; m < M; m += 64) s += part[(size_t)m * (d + 1) + q];
      s = wave_sum(s);
      if (lane == 0) {
        if (q == d) *g_sf2 += s;
        else g_ls[q] += -2.0 * ka.inv_ls[q] * s;
      }
    }
  }
  if (g_Z) {
    const int64_t total = (int64_t)M * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
      const int q = (int)(e % d);
      // d r2[m][m'] / d z_mq = 2 diff inv_ls ; row m and column m of the symmetric Kuubar both contribute
      g_Z[e] += 4.0 * ka.inv_ls[q] * gzraw[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// small kernels of the tail
// ---------------------------------------------------------------------------------------------
// Bm = I + W / s2 (full matrix) ; sc[SC_TRW] = tr W
enum { SC_TRW = 0, SC_LOGDET = 1, SC_QQ = 2, SC_TRSP = 3, SC_BA = 4, SC_APA = 5, SC_N = 8 };

__global__ __launch_bounds__(256) void make_B_kernel(const double* __restrict__ W, int Mp, double inv_s2, double* __restrict__ Bm,
                                                     double* __restrict__ trW, double* __restrict__ zero_me, int nB,
                                                     const double* __restrict__ Li, const double* __restrict__ x, double* __restrict__ y) {
  if ((int)blockIdx.x >= nB) {  // y = L^-1 x rides along (gemv_rows_kernel's loop and order: one wave per row, four rows per block)
    const int row = ((int)blockIdx.x - nB) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= Mp) return;
    double s = 0.0;
    for (int j = lane; j < Mp; j += 64) s = fma(Li[(int64_t)row * Mp + j], x[j], s);
    s = wave_sum(s);
    if (lane == 0) y[row] = s;
    return;
  }
  const int64_t total = (int64_t)Mp * Mp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)nB * 256) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    // symmetrise: the two triangles of L^-1 Phi L^-T differ by rounding only
    const double w = 0.5 * (W[e] + W[(int64_t)c * Mp + r]);
    Bm[e] = (r == c ? 1.0 : 0.0) + w * inv_s2;
    if (zero_me) zero_me[e] = 0.0;  // the buffer L_B^-1 grows in (potrf_lower's own clearing launch, folded in)
  }
  if ((int)blockIdx.x == nB - 1) {  // tr W rides along (was a launch of its own)
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < Mp; i += 256) s += W[(int64_t)i * Mp + i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) *trW = s;
  }
}
// One launch ahead of the latency-critical chain: scalars <- 0, status word <- 0 (unless it already carries the
// status of sgp_kuu_factor), tile flags of the second factorization <- 0, bp <- b zero-padded.
__global__ __launch_bounds__(256) void tail_prep_kernel(int* info, double* __restrict__ sc, int* __restrict__ flags, int nflags,
                                                        const double* __restrict__ b, int M, int Mp, double* __restrict__ bp) {
  const int e0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  if (e0 == 0 && info) *info = 0;
  if (e0 < SC_N) sc[e0] = 0.0;
  for (int e = e0; e < nflags; e += stride) flags[e] = 0;
  for (int e = e0; e < Mp; e += stride) bp[e] = e < M ? b[e] : 0.0;
}
// After chol(B): sum log diag(LB)^2, q.q and the dataflow launch's time-out word, in one block.
__device__ __forceinline__ void potrf_scalars(const double* __restrict__ LB, const double* __restrict__ q, int Mp,
                                              const int* abort_flag, int* info, double* red, double& logdet, double& qq) {
  double s = 0.0, t = 0.0;
  for (int i = threadIdx.x; i < Mp; i += 256) {
    s += log(LB[(int64_t)i * Mp + i]);
    t = fma(q[i], q[i], t);
  }
  logdet = 2.0 * block_sum256(s, red);
  qq = block_sum256(t, red);
  if (threadIdx.x == 0 && *abort_flag != 0) *info = SGP_INFO_TIMEOUT;
}
// The three scalars of s2bar that need a pass over an M x M matrix, row by row (round 5: five launches before -- frob_partial, sum256,
// dot, gemv, dot; now a job of adjoint_mid_kernel): one wave per row i writes
//   rows3[i]          = sum_j Binv[i][j] W[i][j]      -> tr(B^-1 W)     rows3[Mp + i] = (W alpha)_i alpha_i -> alpha^T W alpha
//   rows3[2 Mp + i]   = u_i alpha_i                   -> u . alpha
// finalize_bound_kernel adds each array up in a fixed order.

// LB != null: log det B, q.q and the time-out word are computed here (LB == null: read from sc[SC_LOGDET] / sc[SC_QQ], a caller that has them).
struct FinalizeArgs {
  const double* sc;
  const double* yy;
  const double* kappa;
  double s2, Nd;
  int with_adj;
  const double* LB;
  const double* q;
  int Mp;
  const int* abort_flag;
  int* info;
  double* out;
  const double* rows3;
};
__device__ __forceinline__ void finalize_bound_body(const FinalizeArgs& a, double* red) {
  const double* __restrict__ sc = a.sc;
  const double* __restrict__ yy = a.yy;
  const double* __restrict__ kappa = a.kappa;
  const double s2 = a.s2, Nd = a.Nd;
  const int with_adj = a.with_adj, Mp = a.Mp;
  const double* __restrict__ LB = a.LB;
  const double* __restrict__ q = a.q;
  const int* abort_flag = a.abort_flag;
  int* info = a.info;
  double* __restrict__ out = a.out;
  const double* __restrict__ rows3 = a.rows3;
  double logdetB = 0.0, qq = 0.0;
  double trSP = 0.0, aPa = 0.0, ba = 0.0;
  if (with_adj) {  // the row sums of adjoint_rows_kernel, each added up in one fixed order
    double s0 = 0.0, s1 = 0.0, s2_ = 0.0;
    for (int i = threadIdx.x; i < Mp; i += 256) {
      s0 += rows3[i];
      s1 += rows3[Mp + i];
      s2_ += rows3[2 * Mp + i];
    }
    trSP = block_sum256(s0, red);
    aPa = block_sum256(s1, red);
    ba = block_sum256(s2_, red);
  }
  if (LB) {
    potrf_scalars(LB, q, Mp, abort_flag, info, red, logdetB, qq);
  } else {
    logdetB = sc[SC_LOGDET];
    qq = sc[SC_QQ];
  }
  if (threadIdx.x != 0) return;
  const double LOG2PI = 1.8378770664093453;
  const double trW = sc[SC_TRW];
  const double quad = *yy / s2 - qq / (s2 * s2);
  const double logmarg = -(0.5 * Nd * LOG2PI + 0.5 * Nd * log(s2) + 0.5 * logdetB + 0.5 * quad);
  const double trace_term = (*kappa - trW) / (2.0 * s2);
  out[SGP_OUT_F] = logmarg - trace_term;
  out[SGP_OUT_LOGMARG] = logmarg;
  out[SGP_OUT_TRACE] = trace_term;
  out[SGP_OUT_LOGDETB] = logdetB;
  out[SGP_OUT_QUAD] = quad;
  out[SGP_OUT_TRW] = trW;
  if (with_adj) {
    const double s22 = s2 * s2;
    out[SGP_OUT_S2BAR] = -0.5 * (-trSP / s22 + Nd / s2 - *yy / s22 + 2.0 * ba / (s22 * s2) - aPa / (s22 * s22)
                                 - *kappa / s22 + trW / s22);
    out[SGP_OUT_KAPPABAR] = -1.0 / (2.0 * s2);
  } else {
    out[SGP_OUT_S2BAR] = 0.0;
    out[SGP_OUT_KAPPABAR] = 0.0;
  }
}
__global__ __launch_bounds__(256) void finalize_bound_kernel(FinalizeArgs a) {
  __shared__ double red[4];
  finalize_bound_body(a, red);
}

// Adjoints in the whitened basis (everything between L^-T ... L^-1 is formed from B, B^-1 and g = B^-1 u, whose
// entries are O(1) however ill-conditioned Kuu is):
//     2 s2 Phibar = L^-T C L^-1      C = I - B^-1 - g g^T / s2^2
//    -2 Kuubar    = L^-T S L^-1      S = B + B^-1 - 2 I + g g^T / s2^2 = W / s2 - I + B^-1 + g g^T / s2^2
// The textbook form (Kuu^-1 - Sigma^-1 - alpha alpha^T) subtracts matrices of size cond(Kuu) to get an O(1) result:
// with cond(Kuu) ~ 1e8 (inducing inputs closer than the lengthscale) its gradients were off by 1e-2 relative, this
// form by 1e-9 (tests/studies/logp_noise.py, profiles/r02_logp_noise.json) -- same number of M^3 products.
// CS = [C | S], two Mp x Mp matrices back to back (one batched GEMM pair sandwiches both).
// ONE launch (round 5) for the three jobs behind g = B^-1 u that need nothing but W, B^-1, g, u and L^-1 -- they were three launches, two
// of them on the critical path between the sandwiches' GEMMs (profiles/r05_v3_c3_timeline.txt): blocks [0, nC) form [C | S]
// (whitened_cs_kernel's job), the next Mp / 16 the three row-wise scalars of s2bar (adjoint_rows_kernel's: a wave per row), the last
// Mp / 64 t = L^-T g (gemv_cols_kernel's: a block per 64 columns, sixteen waves over the rows, partial sums added in wave order).
// Every sum keeps its order: the same bits as the three launches.
__global__ __launch_bounds__(1024) void adjoint_mid_kernel(const double* __restrict__ W, const double* __restrict__ Binv,
                                                           const double* __restrict__ g, const double* __restrict__ u,
                                                           const double* __restrict__ Li, int Mp, double s2, int nC,
                                                           double* __restrict__ CS, double* __restrict__ rows3, double* __restrict__ t) {
  __shared__ double part[16][64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (b < nC) {
    const int64_t total = (int64_t)Mp * Mp;
    const double is2 = 1.0 / s2, is22 = 1.0 / (s2 * s2);
    for (int64_t e = (int64_t)b * 1024 + threadIdx.x; e < total; e += (int64_t)nC * 1024) {
      const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
      const int64_t pt = (int64_t)c * Mp + r;
      const double w = 0.5 * (W[e] + W[pt]);
      const double bi = 0.5 * (Binv[e] + Binv[pt]);
      const double gg = g[r] * g[c] * is22;
      const double id = (r == c) ? 1.0 : 0.0;
      CS[e] = id - bi - gg;
      CS[total + e] = w * is2 - id + bi + gg;
    }
    return;
  }
  if (b < nC + Mp / 16) {
    const int row = (b - nC) * 16 + wv;
    double f = 0.0, wa = 0.0;
    for (int j = lane; j < Mp; j += 64) {
      const double w = W[(int64_t)row * Mp + j];
      f = fma(Binv[(int64_t)row * Mp + j], w, f);
      wa = fma(w, g[j], wa);
    }
    f = wave_sum(f);
    wa = wave_sum(wa);
    if (lane == 0) {
      const double a = g[row];
      rows3[row] = f;
      rows3[Mp + row] = wa * a;
      rows3[2 * Mp + row] = u[row] * a;
    }
    return;
  }
  {
    const int col = (b - nC - Mp / 16) * 64 + lane;
    double s0 = 0.0, s1 = 0.0, s2_ = 0.0, s3 = 0.0;
    int j = wv;
    for (; j + 48 < Mp; j += 64) {
      s0 = fma(Li[(int64_t)j * Mp + col], g[j], s0);
      s1 = fma(Li[(int64_t)(j + 16) * Mp + col], g[j + 16], s1);
      s2_ = fma(Li[(int64_t)(j + 32) * Mp + col], g[j + 32], s2_);
      s3 = fma(Li[(int64_t)(j + 48) * Mp + col], g[j + 48], s3);
    }
    for (; j < Mp; j += 16) s0 = fma(Li[(int64_t)j * Mp + col], g[j], s0);
    part[wv][lane] = (s0 + s1) + (s2_ + s3);
    __syncthreads();
    if (wv == 0) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += part[k][lane];
      t[col] = acc;
    }
  }
}
// Phibar = sym(R0) / (2 s2), Kuubar = -sym(R1) / 2 (cropped to M x M, ld M), bbar = t / s2^2 with t = L^-T g
// (+ one more block: the evaluation's final scalars -- finalize_bound_kernel's job, which needs nothing this launch writes)
__global__ __launch_bounds__(256) void adjoint_out_kernel(const double* __restrict__ R, const double* __restrict__ t,
                                                          int Mp, int M, double s2, double* __restrict__ Phibar,
                                                          double* __restrict__ Kuubar, double* __restrict__ bbar, int nA, FinalizeArgs fin) {
  if ((int)blockIdx.x == nA) {
    __shared__ double red[4];
    finalize_bound_body(fin, red);
    return;
  }
  const int64_t total = (int64_t)M * M;
  const int64_t mm = (int64_t)Mp * Mp;
  const double h = 0.25 / s2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)nA * 256) {
    const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
    const int64_t p = (int64_t)r * Mp + c, pt = (int64_t)c * Mp + r;
    Phibar[e] = h * (R[p] + R[pt]);
    Kuubar[e] = -0.25 * (R[mm + p] + R[mm + pt]);
  }
  const double is22 = 1.0 / (s2 * s2);
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < M; i += 256) bbar[i] = t[i] * is22;
}

// ---------------------------------------------------------------------------------------------
// predictive helpers
// ---------------------------------------------------------------------------------------------
// Ks[m][t] = sf2 k'(z_m, xs_t) for m < M, t < T (zero in the padding); Mp x Tp, ld Tp
template <int KID>
__global__ __launch_bounds__(256) void kus_kernel(const double* __restrict__ Z, int64_t ldz, const double* __restrict__ Xs,
                                                  int64_t ldxs, KernArgs ka, int M, int Mp, int T, int Tp,
                                                  double* __restrict__ Ks) {
  const int64_t total = (int64_t)Mp * Tp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / Tp), t = (int)(e - (int64_t)m * Tp);
    double v = 0.0;
    if (m < M && t < T) {
      double r2 = 0.0;
      for (int q = 0; q < ka.d; ++q) {
        const double df = (Z[m * ldz + q] - Xs[t * ldxs + q]) * ka.inv_ls[q];
        r2 = fma(df, df, r2);
      }
      v = ka.sf2 * kprofile<KID>(r2);
    }
    Ks[e] = v;
  }
}
// mean[t] = sum_m C[m][t] q[m] / s2 ; var[t] = sf2 - sum As^2 + sum C^2 (+ s2)
__global__ __launch_bounds__(256) void pred_cols_kernel(const double* __restrict__ As, const double* __restrict__ Cm,
                                                        const double* __restrict__ q, int Mp, int Tp, int T, double sf2,
                                                        double s2, int pred_noise, double* __restrict__ mean,
                                                        double* __restrict__ var) {
  __shared__ double pm[4][64], pa[4][64], pc[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  double sm = 0.0, sa = 0.0, scc = 0.0;
  for (int m = w; m < Mp; m += 4) {
    const double a = As[(int64_t)m * Tp + col], c = Cm[(int64_t)m * Tp + col];
    sm = fma(c, q[m], sm);
    sa = fma(a, a, sa);
    scc = fma(c, c, scc);
  }
  pm[w][threadIdx.x & 63] = sm;
  pa[w][threadIdx.x & 63] = sa;
  pc[w][threadIdx.x & 63] = scc;
  __syncthreads();
  if (w == 0 && col < T) {
    const int l = threadIdx.x;
    mean[col] = (pm[0][l] + pm[1][l] + pm[2][l] + pm[3][l]) / s2;
    if (var)
      var[col] = sf2 - (pa[0][l] + pa[1][l] + pa[2][l] + pa[3][l]) + (pc[0][l] + pc[1][l] + pc[2][l] + pc[3][l]) +
                 (pred_noise ? s2 : 0.0);
  }
}
// cov[t][t'] = k(xs_t, xs_t') - AtA[t][t'] + CtC[t][t'] (+ s2 on the diagonal)
template <int KID>
__global__ __launch_bounds__(256) void pred_cov_kernel(const double* __restrict__ Xs, int64_t ldxs, KernArgs ka,
                                                       const double* __restrict__ AtA, const double* __restrict__ CtC,
                                                       int Tp, int T, double s2, int pred_noise, double* __restrict__ cov) {
  const int64_t total = (int64_t)T * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int a = (int)(e / T), b = (int)(e - (int64_t)a * T);
    double r2 = 0.0;
    for (int q = 0; q < ka.d; ++q) {
      const double df = (Xs[a * ldxs + q] - Xs[b * ldxs + q]) * ka.inv_ls[q];
      r2 = fma(df, df, r2);
    }
    const int64_t p = (int64_t)a * Tp + b, pt = (int64_t)b * Tp + a;
    double v = ka.sf2 * kprofile<KID>(r2) - 0.5 * (AtA[p] + AtA[pt]) + 0.5 * (CtC[p] + CtC[pt]);
    if (a == b && pred_noise) v += s2;
    cov[e] = v;
  }
}

// cov (already holding K**) += -AtA + CtC (symmetrised) (+ s2 on the diagonal)
__global__ __launch_bounds__(256) void pred_cov_add_kernel(const double* __restrict__ AtA, const double* __restrict__ CtC, int Tp,
                                                           int T, double s2, int pred_noise, double* __restrict__ cov) {
  const int64_t total = (int64_t)T * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int a = (int)(e / T), b = (int)(e - (int64_t)a * T);
    const int64_t p = (int64_t)a * Tp + b, pt = (int64_t)b * Tp + a;
    double v = cov[e] - 0.5 * (AtA[p] + AtA[pt]) + 0.5 * (CtC[p] + CtC[pt]);
    if (a == b && pred_noise) v += s2;
    cov[e] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// exchange format of the statistics: Phi is symmetric, only its lower triangle crosses xGMI
// ---------------------------------------------------------------------------------------------
// tri = [ Phi[i][j], j <= i, row by row (M (M + 1) / 2) | b (M) | yy | kappa ]
__global__ __launch_bounds__(256) void stats_pack_kernel(const double* __restrict__ stats, int M, double* __restrict__ tri) {
  const int64_t mm = (int64_t)M * M, nt = (int64_t)M * (M + 1) / 2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < mm + M + 2; e += (int64_t)gridDim.x * 256) {
    if (e >= mm) {
      tri[nt + (e - mm)] = stats[e];
      continue;
    }
    const int i = (int)(e / M), j = (int)(e - (int64_t)i * M);
    if (j <= i) tri[(int64_t)i * (i + 1) / 2 + j] = stats[e];
  }
}
__global__ __launch_bounds__(256) void stats_unpack_kernel(const double* __restrict__ tri, int M, double* __restrict__ stats) {
  const int64_t mm = (int64_t)M * M, nt = (int64_t)M * (M + 1) / 2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < mm + M + 2; e += (int64_t)gridDim.x * 256) {
    if (e >= mm) {
      stats[e] = tri[nt + (e - mm)];
      continue;
    }
    const int i = (int)(e / M), j = (int)(e - (int64_t)i * M);
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    stats[e] = tri[(int64_t)hi * (hi + 1) / 2 + lo];
  }
}

// ---------------------------------------------------------------------------------------------
// whitened pass 1 (PyMC3 op order): A = L^-1 K_uf materialised in row chunks, W = A A^T, u = A y
// ---------------------------------------------------------------------------------------------
// up[m] (+)= sum_t As[m][t] y[t]  (one wave per row m; As is Mp x Tp, ld Tp; y has T valid entries)
__global__ __launch_bounds__(256) void rows_dot_y_kernel(const double* __restrict__ As, int Mp, int Tp, int T,
                                                         const double* __restrict__ y, int accumulate, double* __restrict__ up) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= Mp) return;
  double s0 = 0.0, s1 = 0.0;
  int t = lane;
  for (; t + 64 < T; t += 128) {
    s0 = fma(As[(int64_t)m * Tp + t], y[t], s0);
    s1 = fma(As[(int64_t)m * Tp + t + 64], y[t + 64], s1);
  }
  for (; t < T; t += 64) s0 = fma(As[(int64_t)m * Tp + t], y[t], s0);
  const double s = wave_sum(s0 + s1);
  if (lane == 0) up[m] = accumulate ? up[m] + s : s;
}
__global__ __launch_bounds__(256) void yy_kappa_kernel(const double* __restrict__ y, int64_t N, double kdiag,
                                                       double* __restrict__ yy, double* __restrict__ kappa) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < N; i += 256) s = fma(y[i], y[i], s);
  s = block_sum256(s, red);
  if (threadIdx.x == 0) {
    *yy = s;
    *kappa = (double)N * kdiag;
  }
}

static KernArgs make_ka(const double* inv_ls, double sf2, int d) {
  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = (inv_ls && j < d) ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;
  return ka;
}
static int grid_for(int64_t total, int cap = 2048) {
  int64_t g = (total + 255) / 256;
  if (g < 1) g = 1;
  return (int)(g < cap ? g : cap);
}

struct BoundWs {
  double *M0, *M1, *M2, *M3, *M4, *M5, *M6, *M7;  // M0 / M4 are the first halves of the two-matrix buffers CS / TT
  double *bp, *u, *q, *alpha, *t1, *sc, *partial, *rows3;
  int *flags, *flagsB;  // potrf tile-ready flags: chol(Kuu) (cleared by potrf_lower), chol(B) (cleared by tail_prep_kernel)
  size_t bytes;
};
static BoundWs carve_bound(void* ws, int Mp, int with_adj) {
  Carver c(ws);
  BoundWs w;
  const size_t mm = (size_t)Mp * Mp;
  (void)with_adj;
  w.M0 = c.take<double>(2 * mm);
  w.M1 = c.take<double>(mm);
  w.M2 = c.take<double>(mm);
  w.M3 = c.take<double>(mm);
  w.M4 = c.take<double>(2 * mm);
  w.M5 = c.take<double>(mm);
  w.M6 = c.take<double>(mm);
  w.M7 = c.take<double>(mm);
  w.bp = c.take<double>(Mp);
  w.u = c.take<double>(Mp);
  w.q = c.take<double>(Mp);
  w.alpha = c.take<double>(Mp);
  w.t1 = c.take<double>(Mp);
  w.sc = c.take<double>(SC_N);
  w.partial = c.take<double>(256);
  w.rows3 = c.take<double>((size_t)3 * Mp);
  w.flags = c.take<int>(potrf_scratch_ints(Mp));
  w.flagsB = c.take<int>(potrf_scratch_ints(Mp));
  w.bytes = c.used();
  return w;
}

}  // namespace sgp

using namespace sgp;

extern "C" int sgp_abi_version(void) { return SGP_ABI_VERSION; }
extern "C" void sgp_set_cu_budget(int n) { set_cu_budget(n); }

extern "C" const char* sgp_status_string(int status) {
  switch (status) {
    case SGP_OK: return "ok";
    case SGP_ERR_ARG: return "invalid argument (null pointer, non-positive size or unknown kernel_id)";
    case SGP_ERR_DIM: return "dimension out of range (d > SGP_MAX_DIM or M > SGP_MAX_INDUCING)";
    case SGP_ERR_WORKSPACE: return "workspace missing or too small";
    case SGP_ERR_LAUNCH: return "HIP launch failed";
    default: return status > 0 ? "matrix not positive definite (LAPACK-style pivot index)" : "unknown status";
  }
}

extern "C" size_t sgp_stats_packed_len(int M) {
  return M > 0 && M <= SGP_MAX_INDUCING ? (size_t)M * (M + 1) / 2 + M + 2 : 0;
}
extern "C" int sgp_stats_pack_lower(const double* stats, int M, double* tri, sgp_stream_t stream) {
  if (!stats || !tri || M <= 0) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  stats_pack_kernel<<<grid_for((int64_t)M * M + M + 2), 256, 0, (hipStream_t)stream>>>(stats, M, tri);
  return check_launch();
}
extern "C" int sgp_stats_unpack_lower(const double* tri, int M, double* stats, sgp_stream_t stream) {
  if (!stats || !tri || M <= 0) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  stats_unpack_kernel<<<grid_for((int64_t)M * M + M + 2), 256, 0, (hipStream_t)stream>>>(tri, M, stats);
  return check_launch();
}

extern "C" int sgp_kuu(const double* Z, int64_t ldz, const double* inv_ls, double sf2, double jitter, int M, int d,
                       int kernel_id, double* Kuu, sgp_stream_t stream) {
  if (!Z || !inv_ls || !Kuu || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  hipStream_t st = (hipStream_t)stream;
  if (kernel_id == SGP_KERNEL_COMPOSITE) {
    CompSpec cs;
    if (comp_parse(inv_ls, d, &cs) != SGP_OK) return SGP_ERR_ARG;
    comp_kmatrix(Z, ldz, M, Z, ldz, M, cs, d, M, M, jitter, Kuu, st);
    return check_launch();
  }
  const KernArgs ka = make_ka(inv_ls, sf2, d);
  const int g = grid_for((int64_t)M * M);
  switch (kernel_id) {
    case SGP_KERNEL_RBF: kuu_kernel<SGP_KERNEL_RBF><<<g, 256, 0, st>>>(Z, ldz, ka, jitter, M, Kuu); break;
    case SGP_KERNEL_MATERN32: kuu_kernel<SGP_KERNEL_MATERN32><<<g, 256, 0, st>>>(Z, ldz, ka, jitter, M, Kuu); break;
    default: kuu_kernel<SGP_KERNEL_MATERN52><<<g, 256, 0, st>>>(Z, ldz, ka, jitter, M, Kuu); break;
  }
  return check_launch();
}

extern "C" size_t sgp_kuu_bwd_workspace_bytes(int M, int d) {
  if (M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  Carver c(nullptr);
  c.take<double>((size_t)M * (d + 1));
  c.take<double>((size_t)M * d);
  const size_t comp = d <= COMP_MAX_DIM ? comp_kuu_bwd_workspace_bytes(M, d) : 0;  // one size for every kernel_id
  return c.used() > comp ? c.used() : comp;
}

extern "C" int sgp_kuu_bwd(const double* Z, int64_t ldz, const double* inv_ls, double sf2, const double* Kuubar, int M,
                           int d, int kernel_id, double* g_ls, double* g_sf2, double* g_Z, void* ws, size_t ws_bytes,
                           sgp_stream_t stream) {
  if (!Z || !inv_ls || !Kuubar || !g_ls || !g_sf2 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (!ws || ws_bytes < sgp_kuu_bwd_workspace_bytes(M, d)) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (kernel_id == SGP_KERNEL_COMPOSITE) {  // ADDS into the SGP_COMP_LEN-double block in g_ls (and g_Z)
    CompSpec cs;
    if (comp_parse(inv_ls, d, &cs) != SGP_OK) return SGP_ERR_ARG;
    return comp_kuu_bwd(Z, ldz, cs, Kuubar, M, d, g_ls, g_Z, ws, ws_bytes, st);
  }
  Carver c(ws);
  double* part = c.take<double>((size_t)M * (d + 1));
  double* gzraw = c.take<double>((size_t)M * d);
  const KernArgs ka = make_ka(inv_ls, sf2, d);
  if (!g_Z) {  // totals only: a quarter of the launch time (see kuu_bwd_total_kernel)
    const int G = M < 1024 ? M : 1024;
    switch (kernel_id) {
      case SGP_KERNEL_RBF: kuu_bwd_total_kernel<SGP_KERNEL_RBF><<<G, 256, 0, st>>>(Z, ldz, ka, Kuubar, M, part); break;
      case SGP_KERNEL_MATERN32: kuu_bwd_total_kernel<SGP_KERNEL_MATERN32><<<G, 256, 0, st>>>(Z, ldz, ka, Kuubar, M, part); break;
      default: kuu_bwd_total_kernel<SGP_KERNEL_MATERN52><<<G, 256, 0, st>>>(Z, ldz, ka, Kuubar, M, part); break;
    }
    kuu_bwd_reduce_kernel<<<1, 256, 0, st>>>(part, nullptr, G, ka, g_ls, g_sf2, nullptr);
    return check_launch();
  }
  switch (kernel_id) {
    case SGP_KERNEL_RBF: kuu_bwd_kernel<SGP_KERNEL_RBF><<<M, 256, 0, st>>>(Z, ldz, ka, Kuubar, M, part, gzraw); break;
    case SGP_KERNEL_MATERN32: kuu_bwd_kernel<SGP_KERNEL_MATERN32><<<M, 256, 0, st>>>(Z, ldz, ka, Kuubar, M, part, gzraw); break;
    default: kuu_bwd_kernel<SGP_KERNEL_MATERN52><<<M, 256, 0, st>>>(Z, ldz, ka, Kuubar, M, part, gzraw); break;
  }
  kuu_bwd_reduce_kernel<<<grid_for((int64_t)M * d, 256), 256, 0, st>>>(part, gzraw, M, ka, g_ls, g_sf2, g_Z);
  return check_launch();
}

extern "C" size_t sgp_bound_workspace_bytes(int M, int with_adjoints) {
  (void)with_adjoints;  // one size for both modes: `factors` needs G as well
  if (M <= 0 || M > SGP_MAX_INDUCING) return 0;
  return carve_bound(nullptr, padded_m(M), 1).bytes;
}
extern "C" size_t sgp_bound_factors_len(int M) { return M > 0 ? (size_t)2 * M * M + M : 0; }

extern "C" size_t sgp_kuu_factor_len(int M) {
  if (M <= 0 || M > SGP_MAX_INDUCING) return 0;
  const size_t Mp = padded_m(M);
  return Mp * Mp;  // L^-1, padded
}
extern "C" size_t sgp_kuu_factor_workspace_bytes(int M) {
  if (M <= 0 || M > SGP_MAX_INDUCING) return 0;
  const size_t Mp = padded_m(M);
  Carver c(nullptr);
  c.take<double>(Mp * Mp);
  c.take<double>(Mp * Mp);
  c.take<int>(potrf_scratch_ints((int)Mp));
  return c.used();
}
// Conditioning gate of the explicit-inverse path.  Everything downstream of sgp_kuu_factor multiplies by the explicit L^-1;
// unlike LAPACK's substitution that is not backward stable, and once cond(K_uu + J I) passes ~1e13 the bound it produces is
// noise (measured at the CO2 model's M = 480, profiles/r03_co2_m480_chol_ab.json: cond 1e15, F = 7776 / 9440 for theta 1e-7
// apart where LAPACK gives 1706.7726 / 1706.7731 -- a spurious spike that traps a Markov chain).  The factor itself gives a
// condition estimate that can only UNDERSHOOT (round 4; round 3 used trace(K) for lambda_max, which overshoots by up to M and
// refused well-posed problems with a large amplitude, e.g. M sf2 >= 1e7 with one near-duplicate inducing pair):
//     lambda_max(K) = ||L||_2^2 >= max_j ||L e_j||^2   (a column of L)
//     lambda_max(K) >= 1^T K 1 / M = ||L^T 1||^2 / M   (the Rayleigh quotient of the constant vector: the mean row sum of K, within
//                                                       a small factor of lambda_max for a positive kernel matrix)
//     1 / lambda_min(K) = ||L^-1||_2^2 >= max_i ||e_i^T L^-1||^2   (a row of the explicit inverse the chain forms anyway; the
//                                                       smallest PIVOT alone overestimates lambda_min by 24 x on the CO2 chain below)
// so  est = max(first two) x (third) <= cond(K)  always.  Measured against the eigenvalues (numpy): 7.1e13 for cond 8.6e13 on a
// long 1-D chain (300 points 0.17 apart, lengthscale 3, amplitude 2e6: refused), 7.2e10 for 9.0e10 on 1000 well-spread 2-D points
// with one near-duplicate pair and amplitude 2e4 (accepted; round 3's estimate said 2e13), 5.2e6 for 1.4e7 on a benign chain.
// Above the limit the matrix is reported as numerically not positive definite at the row of L^-1 that carries the estimate
// (info = row + 1), so samplers see a zero-density region (a divergence, as PyMC3 treats a failed factorization) instead
// of a finite, meaningless density.  The single-launch path (M <= 128, substitution solves) is not gated: it tracks LAPACK
// (DESIGN section 4a).
double sgp::cond_gate_limit() { return cur_ctx().cond_limit; }
// Scratch of one matrix (cond_scratch_doubles(M) doubles; nb = ceil(M / 64), Mc = 64 nb):
//   colN[br][j], colS[br][j]  (2 nb Mc)  partial sums over the 64 rows of row block br of L[i][j]^2 and L[i][j]   (tiles br >= j / 64)
//   rowN[i]                   (Mc)       ||e_i^T L^-1||^2
// Every partial is written by exactly one thread and summed in a fixed order: the same bits on every rank.  (The first version walked
// whole columns per thread and whole matrices per workgroup: 88 + 57 + 312 us at M = 1024, rocprofv3 -- on the K_uu chain, which IS
// the critical path at C3; these take ~5 us each.)
// column partials of L for tile (br, bc), br >= bc
__device__ __forceinline__ void cond_coltile_body(const double* __restrict__ Ls, int64_t ld, int M, int bc, int br, int nb,
                                                  double* __restrict__ sc, double (*a2)[64], double (*a1)[64]) {
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6, j = bc * 64 + c;
  const int Mc = nb * 64;
  double s2 = 0.0, s1 = 0.0;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int i = br * 64 + r + 4 * k;
    if (i < M && j < M && i >= j) { const double v = Ls[(int64_t)i * ld + j]; s2 = fma(v, v, s2); s1 += v; }
  }
  a2[r][c] = s2;
  a1[r][c] = s1;
  __syncthreads();
  if (r == 0) {
    sc[(int64_t)br * Mc + j] = (a2[0][c] + a2[1][c]) + (a2[2][c] + a2[3][c]);
    sc[(int64_t)(nb + br) * Mc + j] = (a1[0][c] + a1[1][c]) + (a1[2][c] + a1[3][c]);
  }
}
// one wave per row of the lower-triangular Li: out[i] = sum_{j <= i} Li[i][j]^2   (block = 4 rows)
__device__ __forceinline__ void tri_rowsq_body(const double* __restrict__ Li, int64_t ld, int M, int blk, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, i = blk * 4 + (threadIdx.x >> 6);
  if (i >= M) return;
  const double* row = Li + (int64_t)i * ld;
  double s = 0.0;
  for (int j = lane; j <= i; j += 64) { const double v = row[j]; s = fma(v, v, s); }
  s = wave_sum(s);
  if (lane == 0) out[i] = s;
}
// ONE launch for both halves of the estimate's inputs: blocks [0, nb (nb + 1) / 2) take the lower-triangle tiles of L, the rest four
// rows of L^-1 each; blockIdx.y = matrix
__global__ __launch_bounds__(256) void cond_stats_kernel(const double* __restrict__ L, const double* __restrict__ Linv, int64_t ld,
                                                         int64_t stride, int M, int nb, int64_t sstride, double* __restrict__ scratch) {
  __shared__ double a2[4][64], a1[4][64];
  const int ntile = nb * (nb + 1) / 2;
  double* sc = scratch + (int64_t)blockIdx.y * sstride;
  if ((int)blockIdx.x < ntile) {
    int br = (int)((sqrtf(8.0f * blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while ((br + 1) * (br + 2) / 2 <= (int)blockIdx.x) ++br;
    while (br * (br + 1) / 2 > (int)blockIdx.x) --br;
    const int bc = blockIdx.x - br * (br + 1) / 2;
    cond_coltile_body(L + (int64_t)blockIdx.y * stride, ld, M, bc, br, nb, sc, a2, a1);
  } else {
    tri_rowsq_body(Linv + (int64_t)blockIdx.y * stride, ld, M, blockIdx.x - ntile, sc + (int64_t)2 * nb * nb * 64);
  }
}
__global__ __launch_bounds__(256) void tri_rowsq_kernel(const double* __restrict__ Li, int64_t ld, int M, double* __restrict__ out) {
  tri_rowsq_body(Li, ld, M, blockIdx.x, out);
}
size_t sgp::cond_scratch_doubles(int M) {
  const size_t nb = (size_t)(M + 63) / 64, Mc = nb * 64;
  return 2 * nb * Mc + Mc;
}
void sgp::cond_stats(const double* L, const double* Linv, int64_t ld, int64_t stride, int M, int S, double* scratch, hipStream_t st) {
  const int nb = (M + 63) / 64;
  const int64_t ss = (int64_t)cond_scratch_doubles(M);
  cond_stats_kernel<<<dim3(nb * (nb + 1) / 2 + (M + 3) / 4, S), 256, 0, st>>>(L, Linv, ld, stride, M, nb, ss, scratch);
}
// est = lambda_max estimate (columns of L) x 1 / lambda_min estimate (rows of L^-1) > limit  ->  info = that row + 1.  One workgroup per
// matrix (blockIdx.x = sample); `info` is only written while it is still 0.
__global__ __launch_bounds__(256) void cond_gate_kernel(const double* __restrict__ scratch, int64_t sstride, int M, double limit,
                                                        int* __restrict__ info) {
  __shared__ double red[12];
  __shared__ int redi[4];
  double lam, inv_min;
  int at;
  cond_estimate_block(scratch + (int64_t)blockIdx.x * sstride, M, red, redi, lam, inv_min, at);
  if (threadIdx.x == 0 && info[blockIdx.x] == 0 && !(lam * inv_min <= limit)) info[blockIdx.x] = at + 1;  // (NaN trips as well)
}
void sgp::cond_gate(const double* scratch, int M, int S, double limit, int* info, hipStream_t st) {
  cond_gate_kernel<<<S, 256, 0, st>>>(scratch, (int64_t)cond_scratch_doubles(M), M, limit, info);
}

__global__ __launch_bounds__(256) void kuu_factor_prep_kernel(const double* __restrict__ K, int M, int Mp, double* __restrict__ L,
                                                              double* __restrict__ Linv, int* __restrict__ flags, int nflags,
                                                              int* __restrict__ info) {
  const int64_t total = (int64_t)Mp * Mp, e0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  for (int64_t e = e0; e < total; e += stride) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    L[e] = (r < M && c < M) ? K[(int64_t)r * M + c] : (r == c ? 1.0 : 0.0);
    Linv[e] = 0.0;
  }
  for (int64_t e = e0; e < nflags; e += stride) flags[e] = 0;
  if (e0 == 0) *info = 0;
}
// What follows the factorization, in ONE single-workgroup launch (round 5; it was potrf_timeout + cond_gate + tri_rowsq's consumer
// ordered_sum + a 4-byte copy of the status word: five launches on the K_uu chain, ~30 us of C3's critical path with their gaps):
//   the dataflow launch gave up           -> info = SGP_INFO_TIMEOUT
//   the condition estimate is over limit  -> info = row + 1 (while info is still 0; limit <= 0: no gate)
//   trace_out (optional)                  -> [tr(K_uu^-1) | the M row sums of squares of L^-1]: the same bits sgp_kuu_inverse_trace produces
//                                            (tri_rowsq_body's rows, ordered_sum_kernel's fixed thread <-> row mapping and tree)
__global__ __launch_bounds__(256) void kuu_post_kernel(const double* __restrict__ scratch, int M, double limit, const int* __restrict__ abort_flag,
                                                       int* __restrict__ info, double* __restrict__ trace_out) {
  __shared__ double red[12];
  __shared__ int redi[4];
  const int nb = (M + 63) / 64;
  const double* rown = scratch + (int64_t)2 * nb * nb * 64;
  const bool aborted = *abort_flag != 0;
  if (limit > 0.0 && !aborted) {
    double lam, inv_min;
    int at;
    cond_estimate_block(scratch, M, red, redi, lam, inv_min, at);
    if (threadIdx.x == 0 && *info == 0 && !(lam * inv_min <= limit)) *info = at + 1;  // (NaN trips as well)
    __syncthreads();
  }
  if (aborted && threadIdx.x == 0) *info = SGP_INFO_TIMEOUT;
  if (trace_out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) {
      const double v = rown[i];
      trace_out[1 + i] = v;
      s += v;
    }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) trace_out[0] = s;
  }
}
// L^-1 of chol(Kuu), padded: the part of the tail that does not depend on the streamed statistics, so a
// caller can run it on a second stream underneath pass 1.  Five launches: prep, the factorization (which forms the whole inverse:
// sgp_potrf_chain.hpp), the column / row partials, and kuu_post_kernel (M <= 64 or a CU budget of 1: + tri_inverse's).
static int kuu_factor_impl(const double* Kuu, int M, double* Linv_out, int* info, double* trace_out, void* ws, size_t ws_bytes,
                           sgp_stream_t stream) {
  if (!Kuu || !Linv_out || !info || M <= 0) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (!ws || ws_bytes < sgp_kuu_factor_workspace_bytes(M)) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int Mp = padded_m(M);
  Carver c(ws);
  double* L = c.take<double>((size_t)Mp * Mp);
  double* tmp = c.take<double>((size_t)Mp * Mp);
  int* flags = c.take<int>(potrf_scratch_ints(Mp));
  // one launch instead of four: L <- K_uu identity-padded, L^-1's buffer <- 0, the factorization's flags <- 0, status word <- 0
  kuu_factor_prep_kernel<<<grid_for((int64_t)Mp * Mp), 256, 0, st>>>(Kuu, M, Mp, L, Linv_out, flags, (int)potrf_flag_ints(Mp), info);
  if (!potrf_lower(L, Linv_out, Mp, Mp, info, 0, flags, st, nullptr, nullptr, /*caller_managed=*/true, /*prepped=*/3))
    tri_inverse(L, Linv_out, tmp, Mp, Mp, st);
  const double limit = cond_gate_limit();
  // `tmp` is free again: the partials of L's columns and of L^-1's rows live at its start
  if (limit > 0.0 || trace_out) cond_stats(L, Linv_out, Mp, 0, M, 1, tmp, st);
  kuu_post_kernel<<<1, 256, 0, st>>>(tmp, M, limit, potrf_abort_flag(flags, Mp), info, trace_out);
  return check_launch();
}
extern "C" int sgp_kuu_factor(const double* Kuu, int M, double* Linv_out, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  return kuu_factor_impl(Kuu, M, Linv_out, info, nullptr, ws, ws_bytes, stream);
}
extern "C" int sgp_kuu_factor_ex(const double* Kuu, int M, double* Linv_out, int* info, double* trace_out, void* ws, size_t ws_bytes,
                                 sgp_stream_t stream) {
  return kuu_factor_impl(Kuu, M, Linv_out, info, trace_out, ws, ws_bytes, stream);
}
// ---- guard of the streaming evaluation order -----------------------------------------------------------------------------------
// Phi = K_uf K_fu carries a rounding of ~2^-53 max_i Phi_ii per entry however it was summed (fp64 or integer cores: the result is a
// double), and W = L^-1 Phi L^-T amplifies it by 1 / lambda_k(K_uu) in the k-th eigen-direction: to first order
//     |dF| ~ (1 / 2 s2) sum_k |dW_kk| ~ 2^-53 max_i Phi_ii tr(K_uu^-1) / s2,          tr(K_uu^-1) = ||L^-1||_F^2.
// Calibrated against the PyMC3-order CPU oracle over lengthscales 0.2 .. 20 x noise 0.01 .. 3 at N = 200 000, M = 512 (profiles/
// r04_theta_sweep_streaming.jsonl): the estimate tracks the measured |dF| / N within a factor of 5 over ten orders of magnitude (it
// overshoots for exactly duplicated inducing rows, whose eigen-directions carry no error).  The caller (core.py) re-evaluates in the
// whitened (PyMC3) order when estimate / N exceeds its tolerance.  Both kernels sum in a FIXED order: every rank of a sharded
// evaluation holds the same L^-1 and the same all-reduced Phi, so every rank gets the same bits and takes the same decision.
// row sums of squares by one wave per row, then ONE workgroup adds them with a fixed thread <-> row mapping and a fixed tree: the
// same bits on every rank
__global__ __launch_bounds__(256) void ordered_sum_kernel(const double* __restrict__ v, int n, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  s = block_sum256(s, red);
  if (threadIdx.x == 0) out[0] = s;
}
extern "C" size_t sgp_kuu_inverse_trace_len(void) { return 1 + SGP_MAX_INDUCING; }
extern "C" int sgp_kuu_inverse_trace(const double* kuu_linv, int M, double* trace_out, sgp_stream_t stream) {
  if (!kuu_linv || !trace_out || M <= 0) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  hipStream_t st = (hipStream_t)stream;
  tri_rowsq_kernel<<<(M + 3) / 4, 256, 0, st>>>(kuu_linv, padded_m(M), M, trace_out + 1);
  ordered_sum_kernel<<<1, 256, 0, st>>>(trace_out + 1, M, trace_out);
  return check_launch();
}
__global__ __launch_bounds__(256) void streaming_estimate_kernel(const double* __restrict__ Phi, int M, const double* __restrict__ tr,
                                                                 double s2, double n, double* __restrict__ est) {
  __shared__ double red[4];
  double mx = 0.0;
  for (int i = threadIdx.x; i < M; i += 256) mx = fmax(mx, Phi[(int64_t)i * M + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) est[0] = 0x1p-53 * fmax(fmax(red[0], red[1]), fmax(red[2], red[3])) * tr[0] / (s2 * n);
}
extern "C" int sgp_streaming_error_estimate(const double* stats, const double* trace_inv, double s2, int64_t N, int M, double* est,
                                            sgp_stream_t stream) {
  if (!stats || !trace_inv || !est || M <= 0 || N < 0 || !(s2 > 0.0)) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  streaming_estimate_kernel<<<1, 256, 0, (hipStream_t)stream>>>(stats, M, trace_inv, s2, N > 0 ? (double)N : 1.0, est);
  return check_launch();
}

// the same estimate with max_i Phi_ii replaced by its upper bound N sf2^2 (a kernel profile is <= 1): what a caller that is NOT
// running the streaming order can still compute, to decide when to try it again
__global__ void streaming_bound_kernel(const double* __restrict__ tr, double sf2, double s2, double* __restrict__ est) {
  est[0] = 0x1p-53 * sf2 * sf2 * tr[0] / s2;
}
extern "C" int sgp_streaming_error_bound(const double* trace_inv, double sf2, double s2, double* est, sgp_stream_t stream) {
  if (!trace_inv || !est || !(s2 > 0.0) || !(sf2 > 0.0)) return SGP_ERR_ARG;
  streaming_bound_kernel<<<1, 1, 0, (hipStream_t)stream>>>(trace_inv, sf2, s2, est);
  return check_launch();
}

// ABI 3: estimate and upper bound in one call, from whatever the evaluation order has of Phi's diagonal:
//   est[0] = 2^-53 max_i diag[i * stride] tr(K_uu^-1) / (s2 N)   (diag = Phi with stride M + 1, or the extended order's phi_diag with stride 1;
//                                                                  diag = NULL: est[0] = est[1], all a whitened-order evaluation can say)
//   est[1] = 2^-53 sf2^2 tr(K_uu^-1) / s2                         (max_i Phi_ii <= N sf2^2)
__global__ __launch_bounds__(256) void streaming_report_kernel(const double* __restrict__ diag, int64_t stride, int M,
                                                               const double* __restrict__ tr, double sf2, double s2, double n,
                                                               double* __restrict__ est) {
  __shared__ double red[4];
  double mx = 0.0;
  if (diag)
    for (int i = threadIdx.x; i < M; i += 256) mx = fmax(mx, diag[(int64_t)i * stride]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ub = 0x1p-53 * sf2 * sf2 * tr[0] / s2;
    est[1] = ub;
    est[0] = diag ? 0x1p-53 * fmax(fmax(red[0], red[1]), fmax(red[2], red[3])) * tr[0] / (s2 * n) : ub;
  }
}
extern "C" int sgp_streaming_error_report(const double* diag, int64_t stride, const double* trace_inv, double sf2, double s2, int64_t N,
                                          int M, double* est, sgp_stream_t stream) {
  if (!trace_inv || !est || M <= 0 || N < 0 || !(s2 > 0.0) || !(sf2 > 0.0) || (diag && stride <= 0)) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  streaming_report_kernel<<<1, 256, 0, (hipStream_t)stream>>>(diag, stride, M, trace_inv, sf2, s2, N > 0 ? (double)N : 1.0, est);
  return check_launch();
}

extern "C" void sgp_set_cond_limit(double limit) { default_ctx().cond_limit = limit >= 0.0 ? limit : 1e13; }  // deprecated shim

// whitened: Phi / b already are W = A A^T and u = A y with A = L^-1 K_uf (sgp_suffstats_fwd_whitened); kuu_linv required
static int bound_impl(const double* Kuu, const double* Phi, const double* b, const double* yy,
                      const double* kappa, double s2, int64_t N, int M, int with_adjoints, double* out,
                      double* Phibar, double* bbar, double* Kuubar, double* factors,
                      const double* kuu_linv, int* info, void* ws, size_t ws_bytes,
                      sgp_stream_t stream, bool whitened, double* Cw_out = nullptr) {
  if ((!Kuu && !kuu_linv) || !Phi || !b || !yy || !kappa || !out || !info || M <= 0 || N < 0 || !(s2 > 0.0)) return SGP_ERR_ARG;
  if (whitened && !kuu_linv) return SGP_ERR_ARG;
  if (with_adjoints && (!Phibar || !bbar || !Kuubar)) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  const int Mp = padded_m(M);
  BoundWs w = carve_bound(ws, Mp, 1);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ld = Mp;
  const size_t mm = (size_t)Mp * Mp;
  const bool need_G = with_adjoints || factors;

  // status word: zeroed here, except when L^-1 comes from sgp_kuu_factor -- then *info already holds that call's
  // status and (like every factorization here) chol(B) only reports into it while it is still 0
  const int nflags = (int)potrf_flag_ints(Mp);
  tail_prep_kernel<<<(nflags + 255) / 256, 256, 0, st>>>(kuu_linv ? nullptr : info, w.sc, w.flagsB, nflags, b, M, Mp, w.bp);

  // L = chol(Kuu) in M0, L^-1 in M1 -- or L^-1 handed over by sgp_kuu_factor (read-only from here on)
  if (kuu_linv) {
    w.M1 = const_cast<double*>(kuu_linv);
  } else {
    pad_copy(Kuu, M, M, M, w.M0, ld, Mp, Mp, 1.0, st);
    if (!potrf_lower(w.M0, w.M1, ld, Mp, info, 0, w.flags, st)) tri_inverse(w.M0, w.M1, w.M2, ld, Mp, st);
  }

  // W = L^-1 Phi L^-T in M5 (V in M4); Phi is used in place when it needs no padding
  if (whitened) {
    if (M == Mp) w.M5 = const_cast<double*>(Phi);
    else pad_copy(Phi, M, M, M, w.M5, ld, Mp, Mp, 0.0, st);
    w.u = w.bp;  // tail_prep_kernel has padded u into bp
  } else {
  if (M == Mp) w.M3 = const_cast<double*>(Phi);
  else pad_copy(Phi, M, M, M, w.M3, ld, Mp, Mp, 0.0, st);
  {
    GemmDesc g;
    g.A = w.M1; g.lda = ld; g.B = w.M3; g.ldb = ld; g.C = w.M4; g.ldc = ld;
    g.m = Mp; g.n = Mp; g.k = Mp; g.khi_mask = 1;
    g.lower_only = true;  // W's lower tiles (below) read V's lower tiles only
    gemm(g, st);
    GemmDesc h;
    h.A = w.M4; h.lda = ld; h.B = w.M1; h.ldb = ld; h.tb = true; h.C = w.M5; h.ldc = ld;
    h.m = Mp; h.n = Mp; h.k = Mp; h.khi_mask = 2;
    h.lower_only = true;  // W is symmetric: the tiles above the diagonal (the long k ranges of this mask) are the mirrored lower ones
    h.mirror = true;
    gemm(h, st);
  }

  }

  // B = I + W/s2 in M6 -> LB ; q = LB^-1 u rides along with the factorization; LB^-1 (M7) only when G is wanted
  {  // (streaming order: u = L^-1 b in the same launch -- ld = Mp here)
    const int nB = grid_for((int64_t)mm);
    make_B_kernel<<<nB + (whitened ? 0 : Mp / 4), 256, 0, st>>>(w.M5, Mp, 1.0 / s2, w.M6, w.sc + SC_TRW, need_G ? w.M7 : nullptr, nB, w.M1,
                                                                w.bp, w.u);
  }
  const bool lb_inverted = potrf_lower(w.M6, need_G ? w.M7 : nullptr, ld, Mp, info, M, w.flagsB, st, w.u, w.q, /*caller_managed=*/true, /*prepped=*/2);
  const int* abort_flag = potrf_abort_flag(w.flagsB, Mp);
  // (log det B, q.q and the time-out word are finalize_bound_kernel's, from L_B and q, which nothing below overwrites: the launch that
  // used to take them here, ahead of tri_inverse(), is gone)
  if (need_G && !lb_inverted) tri_inverse(w.M6, w.M7, w.M2, ld, Mp, st);

  if (factors) {
    // what sgp_predict needs: L^-1, LB^-1 and q.  The product G = LB^-1 L^-1 is deliberately NOT formed: on ill-conditioned
    // K_uu its entries are of size sqrt(cond) and cancel in G k_u*; the predictive applies the two factors one after the other
    crop_copy(w.M1, ld, factors, M, M, M, st);
    crop_copy(w.M7, ld, factors + (size_t)M * M, M, M, M, st);
    crop_copy(w.q, 1, factors + (size_t)2 * M * M, 1, M, 1, st);
  }
  if (with_adjoints) {
    // B^-1 = LB^-T LB^-1 in M2 ; g = B^-1 u = LB^-T q in alpha
    GemmDesc s;
    s.A = w.M7; s.lda = ld; s.ta = true; s.B = w.M7; s.ldb = ld; s.C = w.M2; s.ldc = ld;
    s.m = Mp; s.n = Mp; s.k = Mp; s.klo_mask = 3;
    gemm(s, st);
    gemv(w.M7, ld, Mp, true, w.q, w.alpha, st);
    // [C | S] in M0.., [C | S] L^-1 in M4.., L^-T [C | S] L^-1 back in M0..: two batched products for both sandwiches
    double* CS = w.M0;
    double* TT = w.M4;
    {  // [C | S], the row-wise scalars of s2bar and t = L^-T g: one launch (ld = Mp)
      const int nC = grid_for((int64_t)mm / 4);
      adjoint_mid_kernel<<<nC + Mp / 16 + Mp / 64, 1024, 0, st>>>(w.M5, w.M2, w.alpha, w.u, w.M1, Mp, s2, nC, CS, w.rows3, w.t1);
    }
    if (Cw_out) crop_copy(CS, ld, Cw_out, M, M, M, st);  // C itself, before the sandwich (factored pass 2)
    GemmDesc t1;
    t1.A = CS; t1.lda = ld; t1.sA = (int64_t)mm; t1.B = w.M1; t1.ldb = ld; t1.sB = 0; t1.C = TT; t1.ldc = ld; t1.sC = (int64_t)mm;
    t1.m = Mp; t1.n = Mp; t1.k = Mp; t1.batch = 2; t1.klo_mask = 2;
    gemm(t1, st);
    GemmDesc t2;
    t2.A = w.M1; t2.lda = ld; t2.ta = true; t2.sA = 0; t2.B = TT; t2.ldb = ld; t2.sB = (int64_t)mm; t2.C = CS; t2.ldc = ld; t2.sC = (int64_t)mm;
    t2.m = Mp; t2.n = Mp; t2.k = Mp; t2.batch = 2; t2.klo_mask = 1;
    gemm(t2, st);
    // scalars for s2bar (slots keep their names): tr(B^-1 W) = tr(Sigma^-1 Phi), u.g = b.alpha, g^T W g = alpha^T Phi alpha
  }
  const FinalizeArgs fin{w.sc, yy, kappa, s2, (double)N, with_adjoints, w.M6, w.q, Mp, abort_flag, info, out, w.rows3};
  if (with_adjoints) {  // the last block of the launch writes the evaluation's scalars
    const int nA = grid_for((int64_t)M * M);
    adjoint_out_kernel<<<nA + 1, 256, 0, st>>>(w.M0, w.t1, Mp, M, s2, Phibar, Kuubar, bbar, nA, fin);
  } else {
    finalize_bound_kernel<<<1, 256, 0, st>>>(fin);
  }
  return check_launch();
}

extern "C" int sgp_bound_from_stats(const double* Kuu, const double* Phi, const double* b, const double* yy,
                                    const double* kappa, double s2, int64_t N, int M, int with_adjoints, double* out,
                                    double* Phibar, double* bbar, double* Kuubar, double* factors,
                                    const double* kuu_linv, int* info, void* ws, size_t ws_bytes,
                                    sgp_stream_t stream) {
  return bound_impl(Kuu, Phi, b, yy, kappa, s2, N, M, with_adjoints, out, Phibar, bbar, Kuubar, factors, kuu_linv, info, ws,
                    ws_bytes, stream, false);
}
extern "C" int sgp_bound_from_whitened_stats(const double* W, const double* u, const double* yy, const double* kappa, double s2,
                                             int64_t N, int M, int with_adjoints, double* out, double* Phibar, double* bbar,
                                             double* Kuubar, double* factors, const double* kuu_linv, int* info, void* ws,
                                             size_t ws_bytes, sgp_stream_t stream) {
  return bound_impl(nullptr, W, u, yy, kappa, s2, N, M, with_adjoints, out, Phibar, bbar, Kuubar, factors, kuu_linv, info, ws,
                    ws_bytes, stream, true);
}

// as sgp_bound_from_whitened_stats, plus C = I - B^-1 - g g^T / s2^2 (M x M) for the factored pass 2
// (sgp_suffstats_bwd_factored): 2 s2 Phibar = L^-T C L^-1 without ever forming Phibar's cond(K_uu)-sized entries
extern "C" int sgp_bound_from_whitened_stats_ex(const double* W, const double* u, const double* yy, const double* kappa, double s2,
                                                int64_t N, int M, int with_adjoints, double* out, double* Phibar, double* bbar,
                                                double* Kuubar, double* factors, const double* kuu_linv, int* info, double* Cw_out,
                                                void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (Cw_out && !with_adjoints) return SGP_ERR_ARG;
  return bound_impl(nullptr, W, u, yy, kappa, s2, N, M, with_adjoints, out, Phibar, bbar, Kuubar, factors, kuu_linv, info, ws,
                    ws_bytes, stream, true, Cw_out);
}

// ---- whitened pass 1 -------------------------------------------------------------------------------------------
constexpr int64_t WH_CHUNK = 32768;  // data rows of K_uf / A materialised at a time
static int64_t wh_chunk(int64_t N) {
  const int64_t np = round_up64(N > 0 ? N : 1, 64);
  return np < WH_CHUNK ? np : WH_CHUNK;
}
// k-slices of W += A A^T per chunk: its lower 64 x 64 tiles alone (136 at M = 1024) leave half of the 256 CUs idle over a 32768-long
// contraction, so the contraction is cut until there are ~2 workgroups per CU (round 4; partial tiles summed in slice order)
static int wh_slices(int Mp, int64_t Tc) {
  const int nb = Mp / 64, tiles = nb * (nb + 1) / 2;
  int S = 1;
  while (S < 8 && tiles * S < 512 && Tc % (2 * S * 16) == 0 && Tc / (2 * S) >= 1024) S *= 2;
  return S;
}
extern "C" size_t sgp_suffstats_whitened_workspace_bytes(int64_t N, int M, int d) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  const size_t Mp = padded_m(M), Tc = (size_t)wh_chunk(N);
  Carver c(nullptr);
  c.take<double>(Mp * Tc);
  c.take<double>(Mp * Tc);
  c.take<double>(Mp * Mp);
  c.take<double>(Mp);
  c.take<double>((size_t)wh_slices((int)Mp, (int64_t)Tc) * Mp * Mp);
  return c.used();
}
extern "C" int sgp_suffstats_fwd_whitened(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                          const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                          const double* kuu_linv, double* W, double* u, double* yy, double* kappa, void* ws,
                                          size_t ws_bytes, sgp_stream_t stream) {
  if (!Z || !inv_ls || !kuu_linv || !W || !u || !yy || !kappa || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  CompSpec cs{};
  double kdiag = sf2;
  if (kernel_id == SGP_KERNEL_COMPOSITE) {
    if (comp_parse(inv_ls, d, &cs) != SGP_OK) return SGP_ERR_ARG;
    kdiag = cs.kdiag;
  }
  if (!ws || ws_bytes < sgp_suffstats_whitened_workspace_bytes(N, M, d)) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int Mp = padded_m(M);
  const int64_t Tc = wh_chunk(N);
  Carver c(ws);
  double* Ks = c.take<double>((size_t)Mp * Tc);
  double* As = c.take<double>((size_t)Mp * Tc);
  double* Wp = c.take<double>((size_t)Mp * Mp);
  double* up = c.take<double>(Mp);
  const int S = wh_slices(Mp, Tc);
  double* parts = c.take<double>((size_t)S * Mp * Mp);
  if (S > 1) fill_zero(parts, (size_t)S * Mp * Mp, st);  // the tiles above the diagonal are never written; the reduction reads them
  const KernArgs ka = make_ka(kernel_id == SGP_KERNEL_COMPOSITE ? nullptr : inv_ls, sf2, d);
  if (N == 0) {
    fill_zero(Wp, (size_t)Mp * Mp, st);
    fill_zero(up, Mp, st);
  }
  for (int64_t t0 = 0; t0 < N; t0 += Tc) {
    const int Tn = (int)((N - t0) < Tc ? (N - t0) : Tc);
    const int Tp = (int)round_up64(Tn, 64);
    const double* xs = X + t0 * ldx;
    const int g = grid_for((int64_t)Mp * Tp);
    switch (kernel_id) {  // K_uf chunk, M x T layout, zero in the padding
      case SGP_KERNEL_RBF: kus_kernel<SGP_KERNEL_RBF><<<g, 256, 0, st>>>(Z, ldz, xs, ldx, ka, M, Mp, Tn, Tp, Ks); break;
      case SGP_KERNEL_MATERN32: kus_kernel<SGP_KERNEL_MATERN32><<<g, 256, 0, st>>>(Z, ldz, xs, ldx, ka, M, Mp, Tn, Tp, Ks); break;
      case SGP_KERNEL_COMPOSITE: comp_kmatrix(Z, ldz, M, xs, ldx, Tn, cs, d, Mp, Tp, 0.0, Ks, st); break;
      default: kus_kernel<SGP_KERNEL_MATERN52><<<g, 256, 0, st>>>(Z, ldz, xs, ldx, ka, M, Mp, Tn, Tp, Ks); break;
    }
    GemmDesc a;  // A = L^-1 K_uf
    a.A = kuu_linv; a.lda = Mp; a.B = Ks; a.ldb = Tp; a.C = As; a.ldc = Tp;
    a.m = Mp; a.n = Tp; a.k = Mp; a.khi_mask = 1;
    gemm(a, st);
    GemmDesc w;  // W (+)= A A^T, lower tiles only (mirrored below)
    w.A = As; w.lda = Tp; w.B = As; w.ldb = Tp; w.tb = true; w.C = Wp; w.ldc = Mp;
    w.m = Mp; w.n = Mp; w.k = Tp; w.beta = t0 > 0 ? 1.0 : 0.0; w.lower_only = true;
    gemm_splitk(w, S, parts, st);  // (a last, shorter chunk whose length S 16 does not divide takes the plain product)
    rows_dot_y_kernel<<<Mp / 4, 256, 0, st>>>(As, Mp, Tp, Tn, y + t0, t0 > 0 ? 1 : 0, up);
  }
  mirror_lower(Wp, Mp, Mp, st);
  crop_copy(Wp, Mp, W, M, M, M, st);
  crop_copy(up, 1, u, 1, M, 1, st);
  yy_kappa_kernel<<<1, 256, 0, st>>>(y, N, kdiag, yy, kappa);
  return check_launch();
}

extern "C" size_t sgp_predict_workspace_bytes(int64_t T, int M, int d, int want_cov) {
  if (T <= 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  const size_t Mp = padded_m(M);
  const size_t Tc = want_cov ? (size_t)round_up64(T, 64) : (size_t)(T < 16384 ? round_up64(T, 64) : 16384);
  Carver c(nullptr);
  c.take<double>(Mp * Mp);
  c.take<double>(Mp * Mp);
  c.take<double>(Mp);
  c.take<double>(Mp * Tc);
  c.take<double>(Mp * Tc);
  c.take<double>(Mp * Tc);
  if (want_cov) {
    c.take<double>(Tc * Tc);
    c.take<double>(Tc * Tc);
  }
  return c.used();
}

extern "C" int sgp_predict(const double* Xs, int64_t ldxs, int64_t T, const double* Z, int64_t ldz, const double* inv_ls,
                           double sf2, double s2, const double* factors, int M, int d, int kernel_id, int pred_noise,
                           double* mean, double* var, double* cov, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!Xs || !Z || !inv_ls || !factors || !mean || T <= 0 || M <= 0 || d <= 0 || ldxs < d || ldz < d) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  CompSpec cs{};
  if (kernel_id == SGP_KERNEL_COMPOSITE) {
    if (comp_parse(inv_ls, d, &cs) != SGP_OK) return SGP_ERR_ARG;
    sf2 = cs.kdiag;  // prior variance k(x, x)
  }
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  const int want_cov = cov != nullptr;
  if (want_cov && T > 32768) return SGP_ERR_DIM;
  if (!ws || ws_bytes < sgp_predict_workspace_bytes(T, M, d, want_cov)) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int Mp = padded_m(M);
  const int64_t Tc = want_cov ? round_up64(T, 64) : (T < 16384 ? round_up64(T, 64) : 16384);
  Carver c(ws);
  double* Li = c.take<double>((size_t)Mp * Mp);
  double* G = c.take<double>((size_t)Mp * Mp);
  double* q = c.take<double>(Mp);
  double* Ks = c.take<double>((size_t)Mp * Tc);
  double* As = c.take<double>((size_t)Mp * Tc);
  double* Cm = c.take<double>((size_t)Mp * Tc);
  double *AtA = nullptr, *CtC = nullptr;
  if (want_cov) {
    AtA = c.take<double>((size_t)Tc * Tc);
    CtC = c.take<double>((size_t)Tc * Tc);
  }
  const KernArgs ka = make_ka(inv_ls, sf2, d);
  pad_copy(factors, M, M, M, Li, Mp, Mp, Mp, 1.0, st);
  pad_copy(factors + (size_t)M * M, M, M, M, G, Mp, Mp, Mp, 1.0, st);
  pad_copy(factors + (size_t)2 * M * M, 1, M, 1, q, 1, Mp, 1, 0.0, st);

  for (int64_t t0 = 0; t0 < T; t0 += Tc) {
    const int Tn = (int)((T - t0) < Tc ? (T - t0) : Tc);
    const int Tp = (int)round_up64(Tn, 64);
    const double* xs = Xs + t0 * ldxs;
    const int g = grid_for((int64_t)Mp * Tp);
    switch (kernel_id) {
      case SGP_KERNEL_RBF: kus_kernel<SGP_KERNEL_RBF><<<g, 256, 0, st>>>(Z, ldz, xs, ldxs, ka, M, Mp, Tn, Tp, Ks); break;
      case SGP_KERNEL_MATERN32: kus_kernel<SGP_KERNEL_MATERN32><<<g, 256, 0, st>>>(Z, ldz, xs, ldxs, ka, M, Mp, Tn, Tp, Ks); break;
      case SGP_KERNEL_COMPOSITE: comp_kmatrix(Z, ldz, M, xs, ldxs, Tn, cs, d, Mp, Tp, 0.0, Ks, st); break;
      default: kus_kernel<SGP_KERNEL_MATERN52><<<g, 256, 0, st>>>(Z, ldz, xs, ldxs, ka, M, Mp, Tn, Tp, Ks); break;
    }
    GemmDesc a;
    a.A = Li; a.lda = Mp; a.B = Ks; a.ldb = Tp; a.C = As; a.ldc = Tp;
    a.m = Mp; a.n = Tp; a.k = Mp; a.khi_mask = 1;
    gemm(a, st);
    GemmDesc b;  // C = LB^-1 (L^-1 K_u*): the second slot of `factors` holds LB^-1
    b.A = G; b.lda = Mp; b.B = As; b.ldb = Tp; b.C = Cm; b.ldc = Tp;
    b.m = Mp; b.n = Tp; b.k = Mp; b.khi_mask = 1;
    gemm(b, st);
    pred_cols_kernel<<<Tp / 64, 256, 0, st>>>(As, Cm, q, Mp, Tp, Tn, sf2, s2, pred_noise, mean + t0, var ? var + t0 : nullptr);
    if (want_cov) {
      GemmDesc x;
      x.A = As; x.lda = Tp; x.ta = true; x.B = As; x.ldb = Tp; x.C = AtA; x.ldc = Tp;
      x.m = Tp; x.n = Tp; x.k = Mp;
      gemm(x, st);
      GemmDesc y;
      y.A = Cm; y.lda = Tp; y.ta = true; y.B = Cm; y.ldb = Tp; y.C = CtC; y.ldc = Tp;
      y.m = Tp; y.n = Tp; y.k = Mp;
      gemm(y, st);
      const int gc = grid_for((int64_t)Tn * Tn);
      switch (kernel_id) {
        case SGP_KERNEL_COMPOSITE:
          comp_kmatrix(xs, ldxs, Tn, xs, ldxs, Tn, cs, d, Tn, Tn, 0.0, cov, st);
          pred_cov_add_kernel<<<gc, 256, 0, st>>>(AtA, CtC, Tp, Tn, s2, pred_noise, cov);
          break;
        case SGP_KERNEL_RBF: pred_cov_kernel<SGP_KERNEL_RBF><<<gc, 256, 0, st>>>(xs, ldxs, ka, AtA, CtC, Tp, Tn, s2, pred_noise, cov); break;
        case SGP_KERNEL_MATERN32: pred_cov_kernel<SGP_KERNEL_MATERN32><<<gc, 256, 0, st>>>(xs, ldxs, ka, AtA, CtC, Tp, Tn, s2, pred_noise, cov); break;
        default: pred_cov_kernel<SGP_KERNEL_MATERN52><<<gc, 256, 0, st>>>(xs, ldxs, ka, AtA, CtC, Tp, Tn, s2, pred_noise, cov); break;
      }
    }
  }
  return check_launch();
}
