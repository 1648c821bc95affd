// Streaming pass 2: gradients of the collapsed bound through Kuf (never materialised).
//
//   Kbar_uf = 2 Phibar Kuf + bbar y^T                      (M x N, the second N M^2 GEMM)
//   g_theta = sum_{m,n} Kbar_uf[m,n] * dKuf[m,n]/dtheta     theta in { lengthscale_j, sf2, Z[m,j] }
//
// This is what torch autograd / Theano reverse mode do for the reference on every Adam step and
// every NUTS leapfrog (loss.backward() -- reference models/sgpr.py:129; pm.NUTS logp_dlogp --
// reference models/bayesian_sgpr_hmc.py:73-78), restated as one fused kernel:
//
//   * workgroup (mb, split) owns the 128 inducing rows of block mb and a range of 128-row data blocks;
//   * per data block it runs C[128 m x 128 n] = sum_m' Phibar[m][m'] k'(z_m', x_n) on the fp64 matrix
//     cores: the A operand streams 16 x 128 slabs of the (symmetric, padded) Phibar from L2 through
//     LDS, the B operand is generated on the fly exactly like pass 1 with the roles of X and Z
//     swapped (thread keeps x~_n in registers, z~_m' arrives through the scalar cache);
//   * the epilogue passes C through LDS (re-using the main-loop buffers) so that every thread owns ONE
//     inducing row again: it regenerates k', dk'/dr2 and the per-dimension differences against
//     wave-uniform data rows and keeps its dF/dZ, dF/dl, dF/dsf2 sums in registers -- no atomics;
//   * blockIdx.x enumerates mb fastest: with round-robin XCD dispatch all workgroups of one mb share
//     one XCD and its L2 keeps that block's 1 MB slab of Phibar resident (speed only).
//   * per-split partials are summed in a fixed order by a second kernel (bit-reproducible).
#include "sgp_common.hpp"

namespace sgp {

constexpr int BT = 128;            // tile edge (inducing rows x data rows)
constexpr int BK = 16;             // m' chunk
constexpr int BROW = BT + 16;      // LDS row stride (144 doubles: +128 B bank shift per k)
constexpr int BWD_TARGET_WGS = 512;

static inline int bwd_dp_for(int d) {
  const int opts[] = {2, 4, 8, 16, 24, 32};
  for (int o : opts)
    if (d <= o) return o;
  return -1;
}

struct BwdPlan {
  int Mp, nmb, DP, nsplit, bps;
  int64_t nblocks, Npad;
};
static BwdPlan make_bwd_plan(int64_t N, int M, int d) {
  BwdPlan p;
  p.Mp = padded_m(M);
  p.nmb = p.Mp / BT;
  p.DP = bwd_dp_for(d);
  p.nblocks = (N + BT - 1) / BT;
  p.Npad = p.nblocks * BT;
  int64_t want = (BWD_TARGET_WGS + p.nmb - 1) / p.nmb;
  int64_t ns = p.nblocks < want ? p.nblocks : want;
  if (ns < 1) ns = 1;
  p.bps = (int)((p.nblocks + ns - 1) / ns);
  if (p.bps < 1) p.bps = 1;
  p.nsplit = p.nblocks > 0 ? (int)((p.nblocks + p.bps - 1) / p.bps) : 1;
  return p;
}
struct BwdWs {
  double *Xs, *ys, *Zs, *Pb, *bb, *gzpart, *glpart;
  size_t bytes;
};
static BwdWs carve_bwd(void* ws, const BwdPlan& p) {
  Carver c(ws);
  BwdWs w;
  w.Xs = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.DP);
  w.ys = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1));
  w.Zs = c.take<double>((size_t)p.Mp * p.DP);
  w.Pb = c.take<double>((size_t)p.Mp * p.Mp);
  w.bb = c.take<double>((size_t)p.Mp);
  w.gzpart = c.take<double>((size_t)p.nsplit * p.Mp * p.DP);
  w.glpart = c.take<double>((size_t)p.nsplit * p.nmb * (p.DP + 1));
  w.bytes = c.used();
  return w;
}

__global__ void bwd_scale_rows_kernel(const double* __restrict__ in, int64_t ld, int64_t rows, int64_t rows_pad,
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
__global__ void bwd_pad_vec_kernel(const double* __restrict__ in, int64_t n, int64_t npad, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = i < n ? in[i] : 0.0;
}
// Pb (Mp x Mp) <- symmetrised, zero-padded Phibar (M x M)
__global__ void bwd_pad_sym_kernel(const double* __restrict__ P, int M, int Mp, double* __restrict__ out) {
  const int64_t total = (int64_t)Mp * Mp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    double v = 0.0;
    if (r < M && c < M) v = 0.5 * (P[(int64_t)r * M + c] + P[(int64_t)c * M + r]);
    out[e] = v;
  }
}

constexpr int BSM = 2 * 2 * BK * BROW;  // doubles of LDS shared by the main loop and the epilogue (9216)
constexpr int CS = 129;                  // row stride of the epilogue's C image  Ct[64 n][128 m]
static_assert(64 * CS <= BSM, "epilogue image must fit in the main-loop buffers");

template <int DP, int KID>
__global__ __launch_bounds__(256, (DP <= 8 ? 2 : 1)) void suffstats_bwd_kernel(
    const double* __restrict__ Xs, const double* __restrict__ ys, const double* __restrict__ Zs,
    const double* __restrict__ Pb, const double* __restrict__ bb, double sf2,
    int64_t N, int M, int Mp, int nmb, int64_t nblocks, int bps, int want_gz,
    double* __restrict__ gzpart, double* __restrict__ glpart) {
  __shared__ double smem[BSM];
  double (*Pt)[BK][BROW] = reinterpret_cast<double (*)[BK][BROW]>(smem);                  // [2][BK][BROW]
  double (*Kt)[BK][BROW] = reinterpret_cast<double (*)[BK][BROW]>(smem + 2 * BK * BROW);  // [2][BK][BROW]
  double (*Ct)[CS] = reinterpret_cast<double (*)[CS]>(smem);                              // [64][CS], epilogue only

  const int mb = blockIdx.x % nmb;
  const int split = blockIdx.x / nmb;
  const int64_t nb0 = (int64_t)split * bps;
  const int64_t nb1 = (nb0 + bps < nblocks) ? nb0 + bps : nblocks;
  const int m0 = mb * BT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int half = wave >> 1;                        // wave-uniform: tid >> 7
  const int nloc = tid & 127;                        // main loop: data row this thread generates for
  const int mh = half * (BK / 2);                    // ... and which half of the m' chunk
  const int prow = tid >> 4, pcol = (tid & 15) * 8;  // main loop: staging role for the Phibar slab
  const int erow = tid & 127;                        // epilogue: inducing row this thread contracts
  const int nchunks = Mp / BK;

  // epilogue state: this thread's inducing row and its raw gradient sums (scaled by the reduce kernel)
  double zrow[DP], gl[DP], gz[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) {
    zrow[j] = Zs[(size_t)(m0 + erow) * DP + j];
    gl[j] = 0.0;
    gz[j] = 0.0;
  }
  double gs = 0.0;
  const double bbm = bb[m0 + erow];
  const double mmask = (m0 + erow) < M ? 1.0 : 0.0;

  for (int64_t nb = nb0; nb < nb1; ++nb) {
    const int64_t n0 = nb * BT;
    double xr[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) xr[j] = Xs[(n0 + nloc) * DP + j];
    const double nmask = (n0 + nloc) < N ? 1.0 : 0.0;

    double pv[8];
    auto fetchP = [&](int ch) {
      const double* s = Pb + (int64_t)(ch * BK + prow) * Mp + m0 + pcol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const d2 x = *reinterpret_cast<const d2*>(s + 2 * e);
        pv[2 * e] = x[0];
        pv[2 * e + 1] = x[1];
      }
    };
    auto stashP = [&](int buf) {
#pragma unroll
      for (int e = 0; e < 8; ++e) Pt[buf][prow][pcol + e] = pv[e];
    };
    auto gen = [&](int ch, int buf) {
#pragma unroll 4
      for (int i = 0; i < BK / 2; ++i) {
        const int mp = ch * BK + mh + i;                       // wave-uniform inducing row
        const double* __restrict__ zr = Zs + (size_t)mp * DP;  // -> scalar loads
        double r2 = 0.0;
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const double df = xr[j] - zr[j];
          r2 = fma(df, df, r2);
        }
        const double msk = mp < M ? nmask : 0.0;
        Kt[buf][mh + i][nloc] = kprofile<KID>(r2) * msk;
      }
    };

    d4 acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};

    fetchP(0);
    stashP(0);
    gen(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
      const int buf = ch & 1;
      if (ch + 1 < nchunks) fetchP(ch + 1);
      if (ch + 1 < nchunks) gen(ch + 1, buf ^ 1);
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        const double* pr = &Pt[buf][ks * 4 + l4][0];
        const double* kr = &Kt[buf][ks * 4 + l4][0];
        double a[4], bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = pr[wi * 64 + u * 16 + l15];
#pragma unroll
        for (int v = 0; v < 4; ++v) bq[v] = kr[wj * 64 + v * 16 + l15];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] = mfma16(a[u], bq[v], acc[u][v]);
      }
      if (ch + 1 < nchunks) stashP(buf ^ 1);
      __syncthreads();
    }

    // ---- epilogue: C = Phibar K' goes through LDS in two 64-column halves; thread (erow, half)
    //      then walks 32 wave-uniform data rows, so x~_n and y_n arrive through the scalar cache ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (wj == h) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[v * 16 + l15][wi * 64 + u * 16 + l4 + 4 * r] = acc[u][v][r];
      }
      __syncthreads();
#pragma unroll 2
      for (int i = 0; i < 32; ++i) {
        const int nl = half * 32 + i;                 // wave-uniform column inside this half
        const int64_t n = n0 + h * 64 + nl;
        const double* __restrict__ xq = Xs + n * DP;  // -> scalar loads
        double df[DP];
        double r2 = 0.0;
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          df[j] = zrow[j] - xq[j];
          r2 = fma(df[j], df[j], r2);
        }
        double kp, hp;
        kprofile_grad<KID>(r2, kp, hp);
        const double msk = n < N ? mmask : 0.0;
        const double kbar = (2.0 * sf2 * Ct[nl][erow] + bbm * ys[n]) * msk;  // dF/dK[m][n]
        gs = fma(kbar, kp, gs);
        const double E = kbar * sf2 * hp;                                    // dF/d r2[m][n]
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const double t = E * df[j];
          gz[j] += t;
          gl[j] = fma(t, df[j], gl[j]);
        }
      }
      __syncthreads();
    }
  }

  // ---- write per-(split, mb) partials ------------------------------------------------------------
  double* scratch = smem;
  if (want_gz) {
    // the two thread halves hold the same inducing rows: combine in a fixed order
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      __syncthreads();
      if (half == 1) scratch[erow] = gz[j];
      __syncthreads();
      if (half == 0) gzpart[((size_t)split * Mp + m0 + erow) * DP + j] = gz[j] + scratch[erow];
    }
  }
  auto block_total = [&](double v, int slot) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    if (tid == 0) glpart[((size_t)split * nmb + mb) * (DP + 1) + slot] = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  };
#pragma unroll
  for (int j = 0; j < DP; ++j) block_total(gl[j], j);
  block_total(gs, DP);
}

// g_ls[j] = -2 inv_ls_j * sum_parts SL[j] ; g_sf2 = sum_parts SS + kappabar * N ;
// g_Z[m][j] = 2 inv_ls_j * sum_splits SZ[m][j]
__global__ __launch_bounds__(256) void bwd_reduce_kernel(const double* __restrict__ gzpart, const double* __restrict__ glpart,
                                                         int nsplit, int nmb, int Mp, int M, int DP, KernArgs ka,
                                                         double kappa_term, double* __restrict__ g_ls,
                                                         double* __restrict__ g_sf2, double* __restrict__ g_Z) {
  const int d = ka.d;
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x <= d) {
      const int j = threadIdx.x == d ? DP : threadIdx.x;
      double s = 0.0;
      for (int p = 0; p < nsplit * nmb; ++p) s += glpart[(size_t)p * (DP + 1) + j];
      if ((int)threadIdx.x == d) *g_sf2 = s + kappa_term;
      else g_ls[threadIdx.x] = -2.0 * ka.inv_ls[threadIdx.x] * s;
    }
  }
  if (g_Z) {
    const int64_t total = (int64_t)M * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
      const int m = (int)(e / d), j = (int)(e - (int64_t)m * d);
      double s = 0.0;
      for (int sp = 0; sp < nsplit; ++sp) s += gzpart[((size_t)sp * Mp + m) * DP + j];
      g_Z[e] = 2.0 * ka.inv_ls[j] * s;
    }
  }
}

template <int DP>
static void launch_bwd(int kid, int grid, hipStream_t st, const BwdWs& w, double sf2, int64_t N, int M,
                       const BwdPlan& p, int want_gz) {
#define SGP_BWD_ARGS w.Xs, w.ys, w.Zs, w.Pb, w.bb, sf2, N, M, p.Mp, p.nmb, p.nblocks, p.bps, want_gz, w.gzpart, w.glpart
  switch (kid) {
    case SGP_KERNEL_RBF: suffstats_bwd_kernel<DP, SGP_KERNEL_RBF><<<grid, 256, 0, st>>>(SGP_BWD_ARGS); break;
    case SGP_KERNEL_MATERN32: suffstats_bwd_kernel<DP, SGP_KERNEL_MATERN32><<<grid, 256, 0, st>>>(SGP_BWD_ARGS); break;
    default: suffstats_bwd_kernel<DP, SGP_KERNEL_MATERN52><<<grid, 256, 0, st>>>(SGP_BWD_ARGS); break;
  }
#undef SGP_BWD_ARGS
}

}  // namespace sgp

using namespace sgp;

extern "C" size_t sgp_suffstats_bwd_workspace_bytes(int64_t N, int M, int d) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  BwdPlan p = make_bwd_plan(N, M, d);
  return carve_bwd(nullptr, p).bytes;
}

extern "C" int sgp_suffstats_bwd(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                 const double* inv_ls, double sf2, const double* Phibar, const double* bbar,
                                 double kappabar, int64_t N, int M, int d, int kernel_id,
                                 double* g_ls, double* g_sf2, double* g_Z, void* ws, size_t ws_bytes,
                                 sgp_stream_t stream) {
  if (!Z || !inv_ls || !Phibar || !bbar || !g_ls || !g_sf2 || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_MATERN52) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  BwdPlan p = make_bwd_plan(N, M, d);
  BwdWs w = carve_bwd(ws, p);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;

  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;

  if (N > 0) {
    const int64_t tot = p.Npad * p.DP;
    const int gx = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    bwd_scale_rows_kernel<<<gx, 256, 0, st>>>(X, ldx, N, p.Npad, p.DP, ka, w.Xs);
    bwd_pad_vec_kernel<<<(int)((p.Npad + 255) / 256 < 2048 ? (p.Npad + 255) / 256 : 2048), 256, 0, st>>>(y, N, p.Npad, w.ys);
  }
  bwd_scale_rows_kernel<<<(p.Mp * p.DP + 255) / 256, 256, 0, st>>>(Z, ldz, M, p.Mp, p.DP, ka, w.Zs);
  bwd_pad_sym_kernel<<<2048, 256, 0, st>>>(Phibar, M, p.Mp, w.Pb);
  bwd_pad_vec_kernel<<<(p.Mp + 255) / 256, 256, 0, st>>>(bbar, M, p.Mp, w.bb);

  const int want_gz = g_Z != nullptr;
  const int grid = p.nmb * p.nsplit;
  switch (p.DP) {
    case 2: launch_bwd<2>(kernel_id, grid, st, w, sf2, N, M, p, want_gz); break;
    case 4: launch_bwd<4>(kernel_id, grid, st, w, sf2, N, M, p, want_gz); break;
    case 8: launch_bwd<8>(kernel_id, grid, st, w, sf2, N, M, p, want_gz); break;
    case 16: launch_bwd<16>(kernel_id, grid, st, w, sf2, N, M, p, want_gz); break;
    case 24: launch_bwd<24>(kernel_id, grid, st, w, sf2, N, M, p, want_gz); break;
    default: launch_bwd<32>(kernel_id, grid, st, w, sf2, N, M, p, want_gz); break;
  }
  const int64_t tot = (int64_t)M * d;
  const int rg = (int)((tot + 255) / 256 < 1024 ? (tot + 255) / 256 : 1024);
  bwd_reduce_kernel<<<rg < 1 ? 1 : rg, 256, 0, st>>>(w.gzpart, w.glpart, p.nsplit, p.nmb, p.Mp, M, p.DP, ka,
                                                    kappabar * (double)N, g_ls, g_sf2, g_Z);
  return check_launch();
}
