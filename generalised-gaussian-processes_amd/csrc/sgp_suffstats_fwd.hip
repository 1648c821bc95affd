// Streaming pass 1: fused kernel-matrix assembly + SYRK on the fp64 matrix cores.
//
//   Phi = Kuf Kuf^T (M x M),  b = Kuf y,  yy = y^T y,  kappa = sum_n k(x_n, x_n)
//
// replaces the materialised N x M kernel matrix and the N M^2 contraction of
// InducingPointKernel / ExactMarginalLogLikelihood (reference models/sgpr.py:37,125) and of
// pm.gp.MarginalSparse (reference models/bayesian_sgpr_hmc.py:71).  Kuf never touches HBM:
//
//   * a prologue writes X~ = X * inv_ls (row-major, padded to DP columns / whole chunks) so the
//     main kernel reads every data row through the SCALAR cache (the row index is wave-uniform);
//   * each 256-thread workgroup owns one 128 x 128 tile of the lower triangle of Phi and one
//     contiguous range of 16-row data chunks.  A thread keeps one scaled inducing row z~ in
//     registers, generates k'(x_n, z_m) = profile(|x~_n - z~_m|^2) for its row against the 16 data
//     rows of a chunk and drops them into an LDS tile laid out [n][row] (row stride 272 doubles so
//     the MFMA operand reads of the two 16-lane halves fall in disjoint bank halves);
//   * the four waves each hold a 64 x 64 accumulator block (16 MFMA tiles = 128 VGPRs) and feed
//     v_mfma_f64_16x16x4_f64 straight from that LDS tile: both operands of the SYRK are the same
//     K tile, lane l supplying K[row0 + (l&15)][n0 + (l>>4)];
//   * LDS is double buffered: generation of chunk c+1 (VALU) is issued in the same loop body as the
//     MFMAs of chunk c, one barrier per chunk; two workgroups fit per CU (69.6 KB LDS, <=256 VGPRs);
//   * partial tiles go to a per-split slab and a second, deterministic kernel sums the splits in a
//     fixed order, applies sf2^2 and mirrors the lower triangle (no fp64 atomics, bit-reproducible).
#include "sgp_common.hpp"

namespace sgp {

constexpr int TILE = 128;          // Phi tile edge per workgroup
constexpr int NB = 16;             // data rows per chunk
constexpr int KROW = 2 * TILE + 16;  // LDS row stride in doubles (272: +128 B shift per n)
constexpr int TARGET_WGS = 512;    // 2 workgroups per CU x 256 CUs

static inline int dp_for(int d) {
  const int opts[] = {2, 4, 8, 16, 24, 32};
  for (int o : opts)
    if (d <= o) return o;
  return -1;
}

struct FwdPlan {
  int Mp, ntr, ntiles, DP;
  int64_t nchunks, Npad;
  int nsplit, cps;
};

static FwdPlan make_plan(int64_t N, int M, int d) {
  FwdPlan p;
  p.Mp = padded_m(M);
  p.ntr = p.Mp / TILE;
  p.ntiles = p.ntr * (p.ntr + 1) / 2;
  p.DP = dp_for(d);
  p.nchunks = (N + NB - 1) / NB;
  p.Npad = p.nchunks * NB;
  int64_t want = (TARGET_WGS + p.ntiles - 1) / p.ntiles;
  int64_t ns = p.nchunks < want ? p.nchunks : want;
  if (ns < 1) ns = 1;
  p.cps = (int)((p.nchunks + ns - 1) / ns);
  if (p.cps < 1) p.cps = 1;
  p.nsplit = p.nchunks > 0 ? (int)((p.nchunks + p.cps - 1) / p.cps) : 1;
  return p;
}

struct FwdWs {
  double *Xs, *ys, *Zs, *slab, *bpart, *yypart;
  size_t bytes;
};
static FwdWs carve_fwd(void* ws, const FwdPlan& p) {
  Carver c(ws);
  FwdWs w;
  w.Xs = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.DP);
  w.ys = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1));
  w.Zs = c.take<double>((size_t)p.Mp * p.DP);
  w.slab = c.take<double>((size_t)p.nsplit * p.ntiles * TILE * TILE);
  w.bpart = c.take<double>((size_t)p.nsplit * p.Mp);
  w.yypart = c.take<double>(256);
  w.bytes = c.used();
  return w;
}

// ---------------------------------------------------------------------------------------------
// prologue kernels
// ---------------------------------------------------------------------------------------------
// out[r][j] = in[r][j] * inv_ls[j] for r < rows, j < d ; zero padding elsewhere.
__global__ void scale_rows_kernel(const double* __restrict__ in, int64_t ld, int64_t rows, int64_t rows_pad,
                                  int DP, KernArgs ka, double* __restrict__ out) {
  const int64_t total = rows_pad * DP;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / DP;
    const int j = (int)(i - r * DP);
    double v = 0.0;
    if (r < rows && j < ka.d) v = in[r * ld + j] * ka.inv_ls[j];
    out[i] = v;
  }
}

// ys = y zero-padded to Npad ; yypart[block] = partial sum of y^2 (fixed grid of 256 blocks).
__global__ __launch_bounds__(256) void prep_y_kernel(const double* __restrict__ y, int64_t N, int64_t Npad,
                                                     double* __restrict__ ys, double* __restrict__ yypart) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < Npad; i += (int64_t)gridDim.x * 256) {
    const double v = i < N ? y[i] : 0.0;
    ys[i] = v;
    s = fma(v, v, s);
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) yypart[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------------------
template <int DP, int KID, bool DIAG>
__device__ __forceinline__ void fwd_tile(double (*Ks)[NB][KROW], const double* __restrict__ Xs,
                                         const double* __restrict__ ys, const double* __restrict__ Zs,
                                         int64_t N, int M, int Mp, int64_t c0, int64_t c1, int I0, int J0,
                                         double* __restrict__ out, double* __restrict__ bout) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  // generation role: off-diagonal tiles -> thread t owns LDS row t (I rows 0..127, J rows 128..255)
  // and all 16 data rows of a chunk; diagonal tiles -> only 128 distinct rows, the two thread
  // halves split the 16 data rows between them.
  const int lrow = DIAG ? (tid & 127) : tid;
  const int zrow = (lrow < TILE) ? I0 + lrow : J0 + lrow - TILE;
  const int nbeg = DIAG ? (wave >> 1) * (NB / 2) : 0;
  constexpr int NCNT = DIAG ? NB / 2 : NB;
  const double zmask = zrow < M ? 1.0 : 0.0;
  constexpr int boff = DIAG ? 0 : TILE;  // where the B-operand rows live in the LDS tile

  double zr[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) zr[j] = Zs[(size_t)zrow * DP + j];

  double bacc = 0.0;

  // branch-free generation of this thread's NCNT elements of chunk c into LDS buffer `buf`
  auto gen = [&](int64_t c, int buf) {
    const int64_t nbase = c * NB;
#pragma unroll 4
    for (int i = 0; i < NCNT; ++i) {
      const int n = nbeg + i;
      const double* __restrict__ xr = Xs + (nbase + n) * DP;  // wave-uniform -> scalar loads
      double r2 = 0.0;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const double df = xr[j] - zr[j];
        r2 = fma(df, df, r2);
      }
      const double msk = (nbase + n) < N ? zmask : 0.0;
      const double kv = kprofile<KID>(r2) * msk;
      Ks[buf][n][lrow] = kv;
      if constexpr (DIAG) bacc = fma(kv, ys[nbase + n], bacc);
    }
  };

  d4 acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};

  const bool skip_mfma = DIAG && wi == 0 && wj == 1;  // strictly-upper 64x64 block of a diagonal tile

  if (c0 < c1) {
    gen(c0, 0);
    __syncthreads();
    for (int64_t c = c0; c < c1; ++c) {
      const int buf = (int)((c - c0) & 1);
      if (c + 1 < c1) gen(c + 1, buf ^ 1);
      if (!skip_mfma) {
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {
          const double* kr = &Ks[buf][ks * 4 + l4][0];
          double a[4], bq[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) a[u] = kr[wi * 64 + u * 16 + l15];
#pragma unroll
          for (int v = 0; v < 4; ++v) bq[v] = kr[boff + wj * 64 + v * 16 + l15];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = mfma16(a[u], bq[v], acc[u][v]);
        }
      }
      __syncthreads();
    }
  }

  // epilogue: accumulators -> slab[split][tile][128][128]
  if (!skip_mfma) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wi * 64 + u * 16 + l4 + 4 * r;
          const int col = wj * 64 + v * 16 + l15;
          out[row * TILE + col] = acc[u][v][r];
        }
  }
  if constexpr (DIAG) {
    double* scratch = &Ks[0][0][0];
    scratch[tid] = bacc;
    __syncthreads();
    if (tid < TILE) bout[I0 + tid] = scratch[tid] + scratch[tid + TILE];
  }
}

template <int DP, int KID>
__global__ __launch_bounds__(256, (DP <= 8 ? 2 : 1)) void suffstats_fwd_kernel(
    const double* __restrict__ Xs, const double* __restrict__ ys, const double* __restrict__ Zs,
    int64_t N, int M, int Mp, int64_t nchunks, int cps, int ntiles,
    double* __restrict__ slab, double* __restrict__ bpart) {
  __shared__ double Ks[2][NB][KROW];

  // lower-triangular tile index -> (ti, tj), tj <= ti
  const int t = blockIdx.x;
  int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while (ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - ti * (ti + 1) / 2;
  const int split = blockIdx.y;
  const int64_t c0 = (int64_t)split * cps;
  const int64_t c1 = (c0 + cps < nchunks) ? c0 + cps : nchunks;
  double* out = slab + ((size_t)split * ntiles + t) * (TILE * TILE);
  double* bout = bpart + (size_t)split * Mp;
  if (ti == tj)
    fwd_tile<DP, KID, true>(Ks, Xs, ys, Zs, N, M, Mp, c0, c1, ti * TILE, tj * TILE, out, bout);
  else
    fwd_tile<DP, KID, false>(Ks, Xs, ys, Zs, N, M, Mp, c0, c1, ti * TILE, tj * TILE, out, bout);
}

// ---------------------------------------------------------------------------------------------
// deterministic split reduction + symmetrisation:  Phi = sf2^2 * sum_s slab[s]
// one block per 32 x 32 sub-block of the lower triangle (ti32 >= tj32)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_phi_kernel(const double* __restrict__ slab, int nsplit, int ntiles,
                                                         int M, double scale, double* __restrict__ Phi) {
  __shared__ double tile[32][33];
  const int t = blockIdx.x;
  int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  while (bi * (bi + 1) / 2 > t) --bi;
  const int bj = t - bi * (bi + 1) / 2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty 0..7
  const int Ti = (bi * 32) / TILE, Tj = (bj * 32) / TILE;
  const int tileidx = Ti * (Ti + 1) / 2 + Tj;
  const double* base = slab + (size_t)tileidx * (TILE * TILE);
  const size_t sstride = (size_t)ntiles * (TILE * TILE);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int lr = ty + 8 * k;  // row inside the 32x32 block
    const int gi = bi * 32 + lr, gj = bj * 32 + tx;
    double s = 0.0;
    if (gi >= gj) {  // computed part of the slab (lower triangle incl. diagonal)
      const size_t off = (size_t)(gi - Ti * TILE) * TILE + (gj - Tj * TILE);
      for (int sp = 0; sp < nsplit; ++sp) s += base[sp * sstride + off];
    }
    tile[lr][tx] = s * scale;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int lr = ty + 8 * k;
    const int gi = bi * 32 + lr, gj = bj * 32 + tx;
    if (gi < M && gj < M && gi >= gj) Phi[(size_t)gi * M + gj] = tile[lr][tx];
    // mirrored element: Phi[gj'][gi'] with roles swapped so the store is row-contiguous
    const int mi = bj * 32 + lr, mj = bi * 32 + tx;  // (row, col) in the upper triangle
    if (mi < M && mj < M && mj > mi) Phi[(size_t)mi * M + mj] = tile[tx][lr];
  }
}

__global__ __launch_bounds__(256) void finalize_stats_kernel(const double* __restrict__ bpart, int nsplit, int Mp, int M,
                                                             const double* __restrict__ yypart, int nyy, double sf2,
                                                             double kappa_val, double* __restrict__ b,
                                                             double* __restrict__ yy, double* __restrict__ kappa) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m < M) {
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += bpart[(size_t)sp * Mp + m];
    b[m] = s * sf2;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nyy; ++i) s += yypart[i];
    *yy = s;
    *kappa = kappa_val;
  }
}

__global__ void zero_kernel(double* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

template <int DP>
static void launch_fwd(int kid, dim3 grid, hipStream_t st, const FwdWs& w, int64_t N, int M, const FwdPlan& p) {
  switch (kid) {
    case SGP_KERNEL_RBF:
      suffstats_fwd_kernel<DP, SGP_KERNEL_RBF><<<grid, 256, 0, st>>>(w.Xs, w.ys, w.Zs, N, M, p.Mp, p.nchunks, p.cps, p.ntiles, w.slab, w.bpart);
      break;
    case SGP_KERNEL_MATERN32:
      suffstats_fwd_kernel<DP, SGP_KERNEL_MATERN32><<<grid, 256, 0, st>>>(w.Xs, w.ys, w.Zs, N, M, p.Mp, p.nchunks, p.cps, p.ntiles, w.slab, w.bpart);
      break;
    default:
      suffstats_fwd_kernel<DP, SGP_KERNEL_MATERN52><<<grid, 256, 0, st>>>(w.Xs, w.ys, w.Zs, N, M, p.Mp, p.nchunks, p.cps, p.ntiles, w.slab, w.bpart);
      break;
  }
}

}  // namespace sgp

using namespace sgp;

extern "C" size_t sgp_suffstats_workspace_bytes(int64_t N, int M, int d) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  FwdPlan p = make_plan(N, M, d);
  return carve_fwd(nullptr, p).bytes;
}

extern "C" int sgp_suffstats_fwd(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                 const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                 double* Phi, double* b, double* yy, double* kappa, void* ws, size_t ws_bytes,
                                 sgp_stream_t stream) {
  if (!Z || !inv_ls || !Phi || !b || !yy || !kappa || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_MATERN52) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  FwdPlan p = make_plan(N, M, d);
  FwdWs w = carve_fwd(ws, p);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;

  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;

  if (N > 0) {
    const int64_t tot = p.Npad * p.DP;
    int gx = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    scale_rows_kernel<<<gx, 256, 0, st>>>(X, ldx, N, p.Npad, p.DP, ka, w.Xs);
  }
  scale_rows_kernel<<<(p.Mp * p.DP + 255) / 256, 256, 0, st>>>(Z, ldz, M, p.Mp, p.DP, ka, w.Zs);
  prep_y_kernel<<<256, 256, 0, st>>>(y, N, p.Npad, w.ys, w.yypart);

  if (N > 0) {
    dim3 grid(p.ntiles, p.nsplit);
    switch (p.DP) {
      case 2: launch_fwd<2>(kernel_id, grid, st, w, N, M, p); break;
      case 4: launch_fwd<4>(kernel_id, grid, st, w, N, M, p); break;
      case 8: launch_fwd<8>(kernel_id, grid, st, w, N, M, p); break;
      case 16: launch_fwd<16>(kernel_id, grid, st, w, N, M, p); break;
      case 24: launch_fwd<24>(kernel_id, grid, st, w, N, M, p); break;
      default: launch_fwd<32>(kernel_id, grid, st, w, N, M, p); break;
    }
  } else {
    const size_t n = (size_t)p.nsplit * p.ntiles * TILE * TILE;
    zero_kernel<<<256, 256, 0, st>>>(w.slab, n);
    zero_kernel<<<8, 256, 0, st>>>(w.bpart, (size_t)p.nsplit * p.Mp);
  }
  const int nb32 = p.Mp / 32;
  reduce_phi_kernel<<<nb32 * (nb32 + 1) / 2, 256, 0, st>>>(w.slab, p.nsplit, p.ntiles, M, sf2 * sf2, Phi);
  finalize_stats_kernel<<<(M + 255) / 256, 256, 0, st>>>(w.bpart, p.nsplit, p.Mp, M, w.yypart, 256, sf2,
                                                         sf2 * (double)N, b, yy, kappa);
  return check_launch();
}
