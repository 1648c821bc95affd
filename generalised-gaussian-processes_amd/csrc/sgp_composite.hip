// Materialised path for SGP_KERNEL_COMPOSITE (see sgp_composite.hpp).  Same outputs as the streaming kernels of
// sgp_suffstats_fwd.hip / sgp_suffstats_bwd.hip -- [Phi | b | yy | kappa] and the gradient of F through K_uf and
// K_uu -- for covariances that are sums of products of isotropic factors.  Row chunks bound the scratch; every
// cross-workgroup sum goes through partial arrays reduced in a fixed order (no fp64 atomics), like the fast path.
#include "sgp_composite.hpp"
#include "sgp_dense.hpp"

namespace sgp {

constexpr int64_t COMP_CHUNK_ROWS = 65536;  // rows of K_fu materialised at a time
constexpr int COMP_BLK_ROWS = 1024;         // rows one workgroup of the gradient kernel contracts

// out[i][j] = k(a_i, b_j) for i < na, j < nb (+ jitter on i == j), zero in the padding; rows_p x cols_p, ld cols_p
__global__ __launch_bounds__(256) void comp_k_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                                     int64_t ldb, CompSpec cs, int d, int64_t na, int nb, int64_t rows_p,
                                                     int cols_p, double jitter, double* __restrict__ out) {
  const int64_t total = rows_p * cols_p;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / cols_p;
    const int j = (int)(e - i * cols_p);
    double v = 0.0;
    if (i < na && j < nb) {
      v = comp_value(cs, A + i * lda, B + (int64_t)j * ldb, d);
      if (i == j) v += jitter;
    }
    out[e] = v;
  }
}

// bp[m] (+)= sum_i K[i][m] y[i]   (block <-> 64 columns, its 16 waves split the rows, partials added in wave order)
__global__ __launch_bounds__(1024) void comp_colsum_kernel(const double* __restrict__ K, int64_t ldk, const double* __restrict__ y,
                                                           int64_t rows, int Mp, int accumulate, double* __restrict__ bp) {
  __shared__ double part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = blockIdx.x * 64 + lane;
  double s = 0.0;
  for (int64_t i = w; i < rows; i += 16) s = fma(K[i * ldk + m], y[i], s);
  part[w][lane] = s;
  __syncthreads();
  if (w == 0) {
    double t = accumulate ? bp[m] : 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][lane];
    bp[m] = t;
  }
}

__global__ __launch_bounds__(256) void comp_scalars_kernel(const double* __restrict__ y, int64_t N, double kdiag,
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

static int grid_for_c(int64_t total, int cap = 4096) {
  int64_t g = (total + 255) / 256;
  if (g < 1) g = 1;
  return (int)(g < cap ? g : cap);
}

void comp_kmatrix(const double* A, int64_t lda, int64_t na, const double* B, int64_t ldb, int nb, const CompSpec& cs, int d,
                  int64_t rows_p, int cols_p, double jitter, double* out, hipStream_t st) {
  comp_k_kernel<<<grid_for_c(rows_p * cols_p), 256, 0, st>>>(A, lda, B, ldb, cs, d, na, nb, rows_p, cols_p, jitter, out);
}

static int64_t chunk_rows(int64_t N) {
  const int64_t np = round_up64(N > 0 ? N : 1, 64);
  return np < COMP_CHUNK_ROWS ? np : COMP_CHUNK_ROWS;
}

size_t comp_fwd_workspace_bytes(int64_t N, int M) {
  const size_t Mp = padded_m(M);
  Carver c(nullptr);
  c.take<double>((size_t)chunk_rows(N) * Mp);
  c.take<double>(Mp * Mp);
  c.take<double>(Mp);
  return c.used();
}

int comp_suffstats_fwd(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const CompSpec& cs,
                       int64_t N, int M, int d, double* Phi, double* b, double* yy, double* kappa, void* ws, size_t ws_bytes,
                       hipStream_t st) {
  if (!ws || ws_bytes < comp_fwd_workspace_bytes(N, M)) return SGP_ERR_WORKSPACE;
  const int Mp = padded_m(M);
  const int64_t Rc = chunk_rows(N);
  Carver c(ws);
  double* Kc = c.take<double>((size_t)Rc * Mp);
  double* Php = c.take<double>((size_t)Mp * Mp);
  double* bp = c.take<double>(Mp);
  fill_zero(Php, (size_t)Mp * Mp, st);
  fill_zero(bp, Mp, st);
  for (int64_t r0 = 0; r0 < N; r0 += Rc) {
    const int64_t rn = (N - r0) < Rc ? (N - r0) : Rc;
    const int64_t rp = round_up64(rn, 64);
    comp_k_kernel<<<grid_for_c(rp * Mp), 256, 0, st>>>(X + r0 * ldx, ldx, Z, ldz, cs, d, rn, M, rp, Mp, 0.0, Kc);
    GemmDesc g;  // Phi += Kc^T Kc
    g.A = Kc; g.lda = Mp; g.ta = true; g.B = Kc; g.ldb = Mp; g.C = Php; g.ldc = Mp;
    g.m = Mp; g.n = Mp; g.k = (int)rp; g.beta = 1.0;
    gemm(g, st);
    comp_colsum_kernel<<<Mp / 64, 1024, 0, st>>>(Kc, Mp, y + r0, rn, Mp, 1, bp);
  }
  crop_copy(Php, Mp, Phi, M, M, M, st);
  crop_copy(bp, 1, b, 1, M, 1, st);
  comp_scalars_kernel<<<1, 256, 0, st>>>(y, N, cs.kdiag, yy, kappa);
  return check_launch();
}

// ---------------------------------------------------------------------------------------------
// gradient contraction:  sum_{i, m} Kbar[i][m] d k(a_i, b_m) / d(.)   with  Kbar = C (+ y_i bb_m)
//   workgroup <-> blk_rows rows; thread <-> (row lane rl, column lane cl): RL x MC = 256, MC = 64 / 128 / 256 columns
//   per pass so that small inducing sets still fill the block (with one workgroup per 1024 rows and a thread per
//   column, the CO2 workload -- N = 634, M = 64 -- ran 634 evaluations of the derivative in sequence on 64 lanes).
//   gppart[blk][SGP_COMP_LEN]  parameter-block partials ; gzpart[blk * RL + rl][M][d]  partials of the derivative
//   in b_m (one slice per row lane: the reduce kernel adds the slices in order, no atomics).
// ---------------------------------------------------------------------------------------------
struct GradGeom {
  int MC, RL, blk_rows;
};
static GradGeom grad_geom(int64_t rows, int M) {
  GradGeom g;
  g.MC = M <= 64 ? 64 : (M <= 128 ? 128 : 256);
  g.RL = 256 / g.MC;
  int64_t br = (rows + 1023) / 1024;  // ~1024 workgroups at most
  if (br < 2 * g.RL) br = 2 * g.RL;
  {  // between one and two rounds of the 256 CUs (N = 634: 317 workgroups of 2 rows): one full round instead
    const int64_t nb = (rows + br - 1) / br;
    if (nb > 256 && nb <= 512) br = (rows + 255) / 256;
  }
  br = (br + g.RL - 1) / g.RL * g.RL;
  if (br > COMP_BLK_ROWS) br = COMP_BLK_ROWS;
  g.blk_rows = (int)br;
  return g;
}

__global__ __launch_bounds__(256) void comp_grad_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ yv,
                                                        const double* __restrict__ B, int64_t ldb, CompSpec cs,
                                                        const double* __restrict__ C, int64_t ldc, const double* __restrict__ bb,
                                                        int64_t nrows, int M, int d, int64_t blk0, int MC, int RL, int blk_rows,
                                                        double* __restrict__ gppart, double* __restrict__ gzpart) {
  __shared__ double red[4 * SGP_COMP_LEN];
  const int64_t r0 = (int64_t)blockIdx.x * blk_rows;
  const int64_t r1 = (r0 + blk_rows) < nrows ? (r0 + blk_rows) : nrows;
  const int64_t blk = blk0 + blockIdx.x;
  const int cl = threadIdx.x % MC, rl = threadIdx.x / MC;
  double acc[SGP_COMP_LEN];
#pragma unroll
  for (int p = 0; p < SGP_COMP_LEN; ++p) acc[p] = 0.0;
  for (int m = cl; m < M; m += MC) {
    double bv[COMP_MAX_DIM], gz[COMP_MAX_DIM];
    for (int j = 0; j < d; ++j) {
      bv[j] = B[(int64_t)m * ldb + j];
      gz[j] = 0.0;
    }
    const double bbm = bb ? bb[m] : 0.0;
    for (int64_t i = r0 + rl; i < r1; i += RL) {
      double gpar[SGP_COMP_LEN], dkdb[COMP_MAX_DIM];
#pragma unroll
      for (int p = 0; p < SGP_COMP_LEN; ++p) gpar[p] = 0.0;
      comp_grad(cs, A + i * lda, bv, d, gpar, dkdb);
      const double kb = C[i * ldc + m] + (yv ? yv[i] * bbm : 0.0);
#pragma unroll
      for (int p = 0; p < SGP_COMP_LEN; ++p) acc[p] = fma(kb, gpar[p], acc[p]);
      for (int j = 0; j < d; ++j) gz[j] = fma(kb, dkdb[j], gz[j]);
    }
    if (gzpart)
      for (int j = 0; j < d; ++j) gzpart[((blk * RL + rl) * M + m) * d + j] = gz[j];
  }
  // all SGP_COMP_LEN sums behind ONE barrier (a block_sum256 per parameter -- 66 barriers -- cost more than the block's
  // few rows of derivatives: 120 us for the kernel at N = 634, M = 480)
#pragma unroll
  for (int p = 0; p < SGP_COMP_LEN; ++p) {
    const double s = wave_sum(acc[p]);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * SGP_COMP_LEN + p] = s;
  }
  __syncthreads();
  if (threadIdx.x < SGP_COMP_LEN) {
    const int p = threadIdx.x;
    gppart[blk * SGP_COMP_LEN + p] = (red[p] + red[SGP_COMP_LEN + p]) + (red[2 * SGP_COMP_LEN + p] + red[3 * SGP_COMP_LEN + p]);
  }
}

// g_blk[p] (+)= scale_p * sum_blk gppart[blk][p] (+ amp_extra on the amplitude slots) ; g_Z[m][j] (+)= scale_z * sum_blk gzpart
__global__ __launch_bounds__(256) void comp_grad_reduce_kernel(const double* __restrict__ gppart, const double* __restrict__ gzpart,
                                                               int64_t nblk, int64_t nzslice, int M, int d, CompSpec cs, double amp_extra,
                                                               double scale_z, int accumulate, double* __restrict__ g_blk,
                                                               double* __restrict__ g_Z) {
  // Fixed thread <-> partial mapping and fixed trees: deterministic.  (One thread per parameter walking all the partials one
  // dependent load after the other took 75 us for 317 blocks.)
  __shared__ double zred[4][64];
  if (blockIdx.x == 0) {
    // wave w sums parameters w, w + 4, ...: lanes stride over the partials, one wave reduction each, no barrier
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave; p < SGP_COMP_LEN; p += 4) {
      double s = 0.0;
      for (int64_t k = lane; k < nblk; k += 64) s += gppart[k * SGP_COMP_LEN + p];
      s = wave_sum(s);
      if (lane == 0) {
        bool is_amp = false;
        for (int t = 0; t < cs.nterms; ++t) is_amp = is_amp || (p == 1 + 8 * t);
        if (is_amp) s += amp_extra;
        g_blk[p] = accumulate ? g_blk[p] + s : s;
      }
    }
  }
  if (g_Z) {
    // 64 elements per pass of a block, four slice lanes each
    const int64_t total = (int64_t)M * d;
    const int el = threadIdx.x & 63, kl = threadIdx.x >> 6;
    for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < total; e0 += (int64_t)gridDim.x * 64) {
      const int64_t e = e0 + el;
      double s = 0.0;
      if (e < total)
        for (int64_t k = kl; k < nzslice; k += 4) s += gzpart[k * total + e];
      zred[kl][el] = s;
      __syncthreads();
      if (kl == 0 && e < total) {
        const double v = ((zred[0][el] + zred[1][el]) + (zred[2][el] + zred[3][el])) * scale_z;
        g_Z[e] = accumulate ? g_Z[e] + v : v;
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void comp_pad2_kernel(const double* __restrict__ src, int M, int Mp, double scale,
                                                        double* __restrict__ dst) {
  const int64_t total = (int64_t)Mp * Mp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / Mp), j = (int)(e - (int64_t)i * Mp);
    dst[e] = (i < M && j < M) ? scale * src[(int64_t)i * M + j] : 0.0;
  }
}

static size_t grad_blocks_max(int64_t N, const GradGeom& gg) {  // blocks never straddle row chunks: one extra per chunk
  const int64_t Rc = chunk_rows(N);
  return (size_t)((N + gg.blk_rows - 1) / gg.blk_rows + (N + Rc - 1) / Rc + 1);
}

size_t comp_bwd_workspace_bytes(int64_t N, int M, int d) {
  const size_t Mp = padded_m(M);
  const GradGeom gg = grad_geom(N, M);
  const size_t nblk = grad_blocks_max(N, gg);
  Carver c(nullptr);
  c.take<double>((size_t)chunk_rows(N) * Mp);
  c.take<double>((size_t)chunk_rows(N) * Mp);
  c.take<double>(Mp * Mp);
  c.take<double>(nblk * SGP_COMP_LEN);
  c.take<double>(nblk * gg.RL * (size_t)M * d);
  return c.used();
}

// Phibar == nullptr: the factored form (Linv, Cw, s2), see sgp_composite.hpp
static int comp_bwd_impl(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const CompSpec& cs,
                         const double* Phibar, const double* Linv, const double* Cw, double s2, const double* bbar,
                         double kappabar, int64_t N, int M, int d, double* g_blk, double* g_Z, void* ws, size_t ws_bytes,
                         hipStream_t st) {
  const bool factored = Phibar == nullptr;
  if (!ws || ws_bytes < (factored ? comp_bwd_factored_workspace_bytes(N, M, d) : comp_bwd_workspace_bytes(N, M, d))) return SGP_ERR_WORKSPACE;
  const int Mp = padded_m(M);
  const int64_t Rc = chunk_rows(N);
  const GradGeom gg = grad_geom(N, M);
  const size_t nblk_max = grad_blocks_max(N, gg);
  Carver c(ws);
  double* Kc = c.take<double>((size_t)Rc * Mp);
  double* Cc = c.take<double>((size_t)Rc * Mp);
  double* P2 = c.take<double>((size_t)Mp * Mp);
  double* gpp = c.take<double>(nblk_max * SGP_COMP_LEN);
  double* gzp = c.take<double>(nblk_max * gg.RL * (size_t)M * d);
  double* T1 = factored ? c.take<double>((size_t)Rc * Mp) : nullptr;
  comp_pad2_kernel<<<grid_for_c((int64_t)Mp * Mp), 256, 0, st>>>(factored ? Cw : Phibar, M, Mp, factored ? 1.0 / s2 : 2.0, P2);
  int64_t blk0 = 0;
  for (int64_t r0 = 0; r0 < N; r0 += Rc) {
    const int64_t rn = (N - r0) < Rc ? (N - r0) : Rc;
    const int64_t rp = round_up64(rn, 64);
    comp_k_kernel<<<grid_for_c(rp * Mp), 256, 0, st>>>(X + r0 * ldx, ldx, Z, ldz, cs, d, rn, M, rp, Mp, 0.0, Kc);
    if (factored) {
      GemmDesc g1;  // T1 = Kc L^-T      (rows a_n^T = (L^-1 k_n)^T)
      g1.A = Kc; g1.lda = Mp; g1.B = Linv; g1.ldb = Mp; g1.tb = true; g1.C = T1; g1.ldc = Mp;
      g1.m = (int)rp; g1.n = Mp; g1.k = Mp;
      gemm(g1, st);
      GemmDesc g2;  // Kc <- T1 (Cw / s2)   (the chunk of K_fu is not needed again: comp_grad_kernel re-evaluates the kernel)
      g2.A = T1; g2.lda = Mp; g2.B = P2; g2.ldb = Mp; g2.C = Kc; g2.ldc = Mp;
      g2.m = (int)rp; g2.n = Mp; g2.k = Mp;
      gemm(g2, st);
      GemmDesc g3;  // Cc = Kc L^-1
      g3.A = Kc; g3.lda = Mp; g3.B = Linv; g3.ldb = Mp; g3.C = Cc; g3.ldc = Mp;
      g3.m = (int)rp; g3.n = Mp; g3.k = Mp;
      gemm(g3, st);
    } else {
      GemmDesc g;  // Cc = Kc (2 Phibar)
      g.A = Kc; g.lda = Mp; g.B = P2; g.ldb = Mp; g.C = Cc; g.ldc = Mp;
      g.m = (int)rp; g.n = Mp; g.k = Mp;
      gemm(g, st);
    }
    const int nb = (int)((rn + gg.blk_rows - 1) / gg.blk_rows);
    comp_grad_kernel<<<nb, 256, 0, st>>>(X + r0 * ldx, ldx, y + r0, Z, ldz, cs, Cc, Mp, bbar, rn, M, d, blk0, gg.MC, gg.RL,
                                         gg.blk_rows, gpp, g_Z ? gzp : nullptr);
    blk0 += nb;
  }
  comp_grad_reduce_kernel<<<grid_for_c((int64_t)M * d, 256), 256, 0, st>>>(gpp, g_Z ? gzp : nullptr, blk0, blk0 * gg.RL, M, d, cs,
                                                                            kappabar * (double)N, 1.0, 0, g_blk, g_Z);
  return check_launch();
}

size_t comp_bwd_factored_workspace_bytes(int64_t N, int M, int d) {
  return comp_bwd_workspace_bytes(N, M, d) + round_up64((int64_t)chunk_rows(N) * padded_m(M) * 8, 256) + 256;
}

int comp_suffstats_bwd(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const CompSpec& cs,
                       const double* Phibar, const double* bbar, double kappabar, int64_t N, int M, int d, double* g_blk,
                       double* g_Z, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!Phibar) return SGP_ERR_ARG;
  return comp_bwd_impl(X, ldx, y, Z, ldz, cs, Phibar, nullptr, nullptr, 1.0, bbar, kappabar, N, M, d, g_blk, g_Z, ws, ws_bytes, st);
}

int comp_suffstats_bwd_factored(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const CompSpec& cs,
                                const double* Linv, const double* Cw, double s2, const double* bbar, double kappabar, int64_t N,
                                int M, int d, double* g_blk, double* g_Z, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!Linv || !Cw || !(s2 > 0.0)) return SGP_ERR_ARG;
  return comp_bwd_impl(X, ldx, y, Z, ldz, cs, nullptr, Linv, Cw, s2, bbar, kappabar, N, M, d, g_blk, g_Z, ws, ws_bytes, st);
}

size_t comp_kuu_bwd_workspace_bytes(int M, int d) {
  const GradGeom gg = grad_geom(M, M);
  const size_t nblk = (size_t)((M + gg.blk_rows - 1) / gg.blk_rows);
  Carver c(nullptr);
  c.take<double>(nblk * SGP_COMP_LEN);
  c.take<double>(nblk * gg.RL * (size_t)M * d);
  return c.used();
}

int comp_kuu_bwd(const double* Z, int64_t ldz, const CompSpec& cs, const double* Kuubar, int M, int d, double* g_blk,
                 double* g_Z, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!ws || ws_bytes < comp_kuu_bwd_workspace_bytes(M, d)) return SGP_ERR_WORKSPACE;
  const GradGeom gg = grad_geom(M, M);
  const int nb = (M + gg.blk_rows - 1) / gg.blk_rows;
  Carver c(ws);
  double* gpp = c.take<double>((size_t)nb * SGP_COMP_LEN);
  double* gzp = c.take<double>((size_t)nb * gg.RL * M * d);
  comp_grad_kernel<<<nb, 256, 0, st>>>(Z, ldz, nullptr, Z, ldz, cs, Kuubar, M, nullptr, M, M, d, 0, gg.MC, gg.RL, gg.blk_rows, gpp,
                                       g_Z ? gzp : nullptr);
  // k(z_i, z_m) depends on z_m through both arguments; with a symmetric Kuubar the two halves are equal
  comp_grad_reduce_kernel<<<grid_for_c((int64_t)M * d, 256), 256, 0, st>>>(gpp, g_Z ? gzp : nullptr, nb, (int64_t)nb * gg.RL, M, d,
                                                                            cs, 0.0, 2.0, 1, g_blk, g_Z);
  return check_launch();
}

}  // namespace sgp
