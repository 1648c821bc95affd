// Streaming pass 2: gradients of the collapsed bound through Kuf.
//
//   Kbar_uf = 2 Phibar Kuf + bbar y^T                      (M x N, the second N M^2 GEMM)
//   g_theta = sum_{m,n} Kbar_uf[m,n] * dKuf[m,n]/dtheta     theta in { lengthscale_j, sf2, Z[m,j] }
//
// This is what torch autograd / Theano reverse mode do for the reference on every Adam step and
// every NUTS leapfrog (loss.backward() -- reference models/sgpr.py:129; pm.NUTS logp_dlogp --
// reference models/bayesian_sgpr_hmc.py:73-78), restated as one GEMM-with-epilogue kernel over the
// SAME materialised K'_fu [Npad x Mp] that pass 1 assembled (handed over by the caller, or
// re-assembled here when it is not):
//
//   * workgroup (mb, split) owns the 128 inducing columns of block mb and a range of 128-row data blocks;
//   * per data block it runs C^T[128 n x 128 m] = sum_m' K'_fu[n][m'] Phibar[m'][m] on the fp64 matrix
//     cores: 16-deep slabs of K'_fu and of the (symmetrised, padded) Phibar go global -> registers -> LDS,
//     double buffered; 4 waves x 16 MFMA tiles as in pass 1;
//   * epilogue: C^T passes through LDS (re-using the main-loop buffers) in two 64-row halves so that every
//     thread owns ONE inducing column m again (z~_m in registers) and walks 32 wave-uniform data rows
//     (x~_n, y_n through the scalar cache): it recomputes k', dk'/dr2 for its own (n, m) -- once per
//     element, no redundancy -- forms dF/dK = 2 sf2 C + bbar_m y_n, dF/dr2 = dF/dK sf2 dk'/dr2 and keeps
//     its dF/dZ, dF/dl, dF/dsf2 sums in registers -- no atomics, per-split partials summed in a fixed order;
//   * workgroup ids are laid out so that the column blocks of one split share an XCD (round-robin dispatch):
//     its L2 then serves the re-reads of every K'_fu row block (speed only, never correctness).
#include "sgp_common.hpp"
#include "sgp_stream.hpp"
#include "sgp_composite.hpp"
#include "sgp_dense.hpp"
#include <cstdlib>

namespace sgp {

constexpr int BK = 16;           // m' chunk
constexpr int ALD = BK + 2;      // LDS row stride of the K'_fu slab  At[128 n][18]: 18 r + k hits every bank pair once
                                 // per half-wave operand read (17 left a 2-way conflict: SQ_LDS_BANK_CONFLICT 32 %)
constexpr int BROW = TILE + 16;  // LDS row stride of the Phibar slab Bt[16 m'][144]
constexpr int A_DBL = TILE * ALD;   // 2304
constexpr int B_DBL = BK * BROW;    // 2304
constexpr int BSM = 2 * (A_DBL + B_DBL);  // 9216 doubles = 73.7 KB: two workgroups per CU
constexpr int CS = 129;                   // row stride of the epilogue image Ct[64 n][128 m]
static_assert(64 * CS <= BSM, "epilogue image must fit in the main-loop buffers");

__global__ void bwd_pad_vec_kernel(const double* __restrict__ in, int64_t n, int64_t npad, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = i < n ? in[i] : 0.0;
}
// (Pb <- symmetrised, zero-padded Phibar and the padded bbar: a job of the prologue launch since round 5, sgp_stream.hpp: PadSymJob)
// Pb (Mp x Mp) <- scale * P (Mp x Mp in, rows / columns >= M zeroed): no symmetrisation (the factored mode's triangular L^-1)
__global__ void bwd_pad_plain_kernel(const double* __restrict__ P, int M, int Mp, double scale, double* __restrict__ out) {
  const int64_t total = (int64_t)Mp * Mp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    out[e] = (r < M && c < M) ? scale * P[e] : 0.0;
  }
}
// out (Mp x Mp) <- scale * P (M x M in), zero padding
__global__ void bwd_pad_small_kernel(const double* __restrict__ P, int M, int Mp, double scale, double* __restrict__ out) {
  const int64_t total = (int64_t)Mp * Mp;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
    out[e] = (r < M && c < M) ? scale * P[(int64_t)r * M + c] : 0.0;
  }
}

// dk'/dr2 from k' itself (and r2 for the Matern profiles): no second exp
template <int KID>
__device__ __forceinline__ double hprime_from_k(double kp, double r2) {
  if constexpr (KID == SGP_KERNEL_RBF) {
    return -0.5 * kp;
  } else if constexpr (KID == SGP_KERNEL_MATERN32) {
    const double a = 1.7320508075688772 * sqrt(r2);
    return -1.5 * kp / (1.0 + a);
  } else {
    const double a = 2.23606797749979 * sqrt(r2);
    return -(5.0 / 6.0) * (1.0 + a) * kp / (1.0 + a + a * a * (1.0 / 3.0));
  }
}

// GZ: the derivative with respect to the inducing inputs is wanted (Adam steps); HMC leapfrogs keep Z fixed and skip
// its 2 DP accumulate operations per element (GZ == (want_gz != 0))
//
// d > 8 (C3: d = 18): the one-pass epilogue keeps z~, df and up to 2 DP running sums per thread -- beside the 128
// accumulator registers that leaves room for ONE workgroup per CU, i.e. one wave per SIMD in the GEMM.  For the RBF
// profile dk'/dr2 = -k'/2 and k' itself sits in the materialised K'_fu the main loop has just streamed, so the
// epilogue needs neither r2 nor exp(): pass A turns the C^T image in LDS into dF/dr2 in place (one coalesced load of k'
// per element), pass B walks the dimensions eight at a time with the DP <= 8 register budget -> two workgroups per CU
// for every d.  The Matern profiles need r2 and keep the one-pass epilogue (one workgroup per CU when d > 8).
// KP: the streamed matrix IS K'_fu (false in the factored mode of sgp_suffstats_bwd_factored, where it is K'_fu L^-T Cw / s2 and
// the epilogue must recompute k')
template <int DP, int KID, bool KP>
constexpr bool kbar_two_pass() {
#ifdef SGP_AB_KBAR_TWO_PASS_ALL  // A/B (tools/ab_build.sh): the two-pass epilogue for every d
  return KP && KID == SGP_KERNEL_RBF;
#else
  return KP && DP > 8 && KID == SGP_KERNEL_RBF;
#endif
}

// rows of the one-pass epilogue in flight per thread (A/B: -DSGP_AB_KBAR_UNROLL4, tools/ab_build.sh -- four rows: 33.58 / 33.61 / 33.74
// against 33.50 / 33.39 / 33.52 ms for two, three alternations on one box, profiles/r03_ab_kbar_unroll4.txt: no gain, two stays)
#ifdef SGP_AB_KBAR_UNROLL4
#define SGP_KBAR_EPI_UNROLL _Pragma("unroll 4")
#else
#define SGP_KBAR_EPI_UNROLL _Pragma("unroll 2")
#endif

template <int DP, int KID, bool GZ, bool KP = true>
__global__ __launch_bounds__(256, ((DP <= 8 || kbar_two_pass<DP, KID, KP>()) ? 2 : 1)) void kbar_contract_kernel(
    const double* __restrict__ Kfu, const double* __restrict__ Xs, const double* __restrict__ ys,
    const double* __restrict__ Zs, const double* __restrict__ Pb, const double* __restrict__ bb, double sf2,
    int64_t row0, int64_t nblocks, SplitMap bmap, int64_t N, int M, int Mp, int nmb, int want_gz, int accumulate,
    double* __restrict__ gacc, double* __restrict__ gzpart, double* __restrict__ glpart) {
  __shared__ double smem[BSM];
  double (*At)[TILE][ALD] = reinterpret_cast<double (*)[TILE][ALD]>(smem);               // [2][128][18]
  double (*Bt)[BK][BROW] = reinterpret_cast<double (*)[BK][BROW]>(smem + 2 * A_DBL);     // [2][16][144]
  double (*Ct)[CS] = reinterpret_cast<double (*)[CS]>(smem);                             // [64][129], epilogue only

  // id -> (xcd, column block, split group): the nmb column blocks of one split share id % 8, i.e. one XCD
  // under round-robin dispatch, whose L2 then serves the nmb re-reads of every K'_fu row block
  const int xcd = blockIdx.x & 7;
  const int jj = blockIdx.x >> 3;
  const int mb = jj % nmb;
  const int split = (jj / nmb) * 8 + xcd;
  int64_t nb0, nb1;
  split_range(bmap, split, nblocks, nb0, nb1);
  const int m0 = mb * TILE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int half = wave >> 1;                        // wave-uniform: tid >> 7
  // staging role for the Phibar slab (16 x 128): row tid / 16, four 16-byte pieces 32 doubles apart, so that the 16 lanes
  // of a row write 256 contiguous bytes per ds_write_b128 (8 contiguous doubles per thread made every such store 4-way
  // bank-conflicted: SQ_LDS_BANK_CONFLICT was 44 % of the kernel's LDS cycles)
  const int prow = tid >> 4, pcol = (tid & 15) * 2;
  const int erow = tid & 127;                        // epilogue: inducing column this thread contracts
  const int nchunks = Mp / BK;

  // staging role for the K'_fu slab (128 x 16): quad q = tid + 256 i -> row q / 8, columns 2 (q % 8) ..+1
  int arow[4], acol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i;
    arow[i] = q >> 3;
    acol[i] = (q & 7) * 2;
  }

  // Raw gradient sums of this thread's inducing column (scaled by the reduce kernel).  They live in a
  // per-workgroup global scratch line (L2 resident, touched once per 128-row block) instead of registers:
  // with 128 accumulator VGPRs and two prefetch stages the main loop has no room for 2 DP + 1 more.
  double* gmine = gacc + (size_t)blockIdx.x * (2 * DP + 1) * 256 + tid;
  const double bbm = bb[m0 + erow];
  const double mmask = (m0 + erow) < M ? 1.0 : 0.0;

  // per-thread element offsets (32-bit) relative to wave-uniform slab bases: one VGPR per address
  int aoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = arow[i] * Mp + acol[i];
  const int poff = prow * Mp + pcol;

  for (int64_t nb = nb0; nb < nb1; ++nb) {
    const int64_t r0 = nb * TILE;  // first row of this data block inside Kfu
    const double* Ablk = Kfu + r0 * Mp;

    // K'_fu comes from HBM: two register stages (loads of slab ch+2 are issued before the MFMAs of slab
    // ch and written to LDS after the MFMAs of slab ch+1).  Phibar is L2/MALL resident: one stage.
    d2 avA[4], avB[4], pv[4];
    auto fetchA = [&](int ch, d2 (&av)[4]) {
      const double* ab = Ablk + ch * BK;  // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const d2*>(ab + aoff[i]);
    };
    auto fetchP = [&](int ch) {
      const double* pb = Pb + (int64_t)ch * BK * Mp + m0;  // wave-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e) pv[e] = *reinterpret_cast<const d2*>(pb + poff + 32 * e);
    };
    auto stashA = [&](int buf, const d2 (&av)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<d2*>(&At[buf][arow[i]][acol[i]]) = av[i];  // 16-B aligned: 144-B rows
    };
    auto stashP = [&](int buf) {
#pragma unroll
      for (int e = 0; e < 4; ++e) *reinterpret_cast<d2*>(&Bt[buf][prow][pcol + 32 * e]) = pv[e];
    };

    d4 acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};

    auto mfma_slab = [&](int buf) {
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        double a[4], bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = At[buf][wi * 64 + u * 16 + l15][ks * 4 + l4];
#pragma unroll
        for (int v = 0; v < 4; ++v) bq[v] = Bt[buf][ks * 4 + l4][wj * 64 + v * 16 + l15];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][v] = mfma16(a[u], bq[v], acc[u][v]);
      }
    };

    fetchA(0, avA);
    fetchP(0);
    stashA(0, avA);
    stashP(0);
    fetchA(1, avB);  // nchunks = Mp / 16 is a multiple of 8
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {
      if (ch + 2 < nchunks) fetchA(ch + 2, avA);
      fetchP(ch + 1);
      mfma_slab(0);
      stashA(1, avB);
      stashP(1);
      __syncthreads();
      if (ch + 3 < nchunks) fetchA(ch + 3, avB);
      if (ch + 2 < nchunks) fetchP(ch + 2);
      mfma_slab(1);
      if (ch + 2 < nchunks) {
        stashA(0, avA);
        stashP(0);
      }
      __syncthreads();
    }

    // ---- epilogue: two 64-row halves of C^T through LDS, thread <-> inducing column ----------------
    const bool first = nb == nb0;
    if constexpr (kbar_two_pass<DP, KID, KP>()) {
      double gs = first ? 0.0 : gmine[(size_t)(2 * DP) * 256];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (wi == h) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
              for (int r = 0; r < 4; ++r) Ct[u * 16 + l4 + 4 * r][wj * 64 + v * 16 + l15] = acc[u][v][r];
        }
        __syncthreads();
        // pass A: C^T -> dF/dr2 in place (every thread rewrites only the 32 entries it reads); k' is zero in the padding
        const int64_t nbase = row0 + r0 + h * 64 + half * 32;  // first of this thread's 32 (wave-uniform) data rows
        const double* __restrict__ kcol = Ablk + (int64_t)(h * 64 + half * 32) * Mp + m0 + erow;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
          const double kp = kcol[(int64_t)i * Mp];
          const double kbar = 2.0 * sf2 * Ct[half * 32 + i][erow] + bbm * ys[nbase + i];  // dF/dK[m][n]
          gs = fma(kbar, kp, gs);
          Ct[half * 32 + i][erow] = -0.5 * sf2 * kbar * kp;                                // dF/d r2[m][n]
        }
        // pass B: the dimensions eight at a time
        constexpr int DG = DP < 8 ? DP : 8;
#pragma unroll 1
        for (int j0 = 0; j0 < DP; j0 += DG) {
          double zg[DG], gl[DG], gz[DG];
          const bool fresh = first && h == 0;
#pragma unroll
          for (int k = 0; k < DG; ++k) {
            zg[k] = Zs[(size_t)(m0 + erow) * DP + j0 + k];
            gl[k] = fresh ? 0.0 : gmine[(size_t)(j0 + k) * 256];
            gz[k] = (fresh || !GZ) ? 0.0 : gmine[(size_t)(DP + j0 + k) * 256];
          }
#pragma unroll 2
          for (int i = 0; i < 32; ++i) {
            const double* __restrict__ xq = Xs + (nbase + i) * DP + j0;  // -> scalar loads
            const double E = Ct[half * 32 + i][erow];
#pragma unroll
            for (int k = 0; k < DG; ++k) {
              const double df = zg[k] - xq[k];
              const double t = E * df;
              if constexpr (GZ) gz[k] += t;
              gl[k] = fma(t, df, gl[k]);
            }
          }
#pragma unroll
          for (int k = 0; k < DG; ++k) {
            gmine[(size_t)(j0 + k) * 256] = gl[k];
            if constexpr (GZ) gmine[(size_t)(DP + j0 + k) * 256] = gz[k];
          }
        }
        __syncthreads();
      }
      gmine[(size_t)(2 * DP) * 256] = gs;
    } else {
    double zrow[DP], gl[DP], gz[DP], gs;
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      zrow[j] = Zs[(size_t)(m0 + erow) * DP + j];
      gl[j] = first ? 0.0 : gmine[(size_t)j * 256];
      gz[j] = (first || !want_gz) ? 0.0 : gmine[(size_t)(DP + j) * 256];
    }
    gs = first ? 0.0 : gmine[(size_t)(2 * DP) * 256];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (wi == h) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[u * 16 + l4 + 4 * r][wj * 64 + v * 16 + l15] = acc[u][v][r];
      }
      __syncthreads();
#if defined(SGP_AB_KBAR_NO_EPI_MATH)  // A/B (tools/ab_build.sh): the epilogue's LDS traffic and barriers without its arithmetic
      for (int i = 0; i < 32; ++i) gs += Ct[half * 32 + i][erow];
      if (false)
#endif
      SGP_KBAR_EPI_UNROLL
      for (int i = 0; i < 32; ++i) {
        const int nl = half * 32 + i;                       // wave-uniform row inside this half
        const int64_t n = row0 + r0 + h * 64 + nl;          // global (padded) data row
        const double* __restrict__ xq = Xs + n * DP;        // -> scalar loads
        double df[DP];
        double r2 = 0.0;
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          df[j] = zrow[j] - xq[j];
          r2 = fma(df[j], df[j], r2);
        }
        // k' and dk'/dr2 are recomputed (one exp, ~22 fp64 VALU ops) rather than re-read from K'_fu:
        // cheaper than a dependent 8-byte-per-lane global load in this loop
        double kp, hp;
        kprofile_grad<KID>(r2, kp, hp);
        const double msk = n < N ? mmask : 0.0;
        const double kbar = (2.0 * sf2 * Ct[nl][erow] + bbm * ys[n]) * msk;  // dF/dK[m][n]
        gs = fma(kbar, kp, gs);
        const double E = kbar * sf2 * hp;                                    // dF/d r2[m][n]
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const double t = E * df[j];
          if constexpr (GZ) gz[j] += t;
          gl[j] = fma(t, df[j], gl[j]);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      gmine[(size_t)j * 256] = gl[j];
      if (want_gz) gmine[(size_t)(DP + j) * 256] = gz[j];
    }
    gmine[(size_t)(2 * DP) * 256] = gs;
    }
  }

  // ---- per-(split, mb) partials ----------------------------------------------------------------------
  double gl[DP], gz[DP], gs;
  {
    const bool any = nb0 < nb1;  // this thread's own stores: no fence needed to read them back
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      gl[j] = any ? gmine[(size_t)j * 256] : 0.0;
      gz[j] = (any && want_gz) ? gmine[(size_t)(DP + j) * 256] : 0.0;
    }
    gs = any ? gmine[(size_t)(2 * DP) * 256] : 0.0;
  }
  double* scratch = smem;
  if (want_gz) {
#pragma unroll
    for (int j = 0; j < DP; ++j) {  // the two thread halves hold the same columns: combine in a fixed order
      __syncthreads();
      if (half == 1) scratch[erow] = gz[j];
      __syncthreads();
      if (half == 0) {
        double* dst = gzpart + ((size_t)split * Mp + m0 + erow) * DP + j;
        const double v = gz[j] + scratch[erow];
        *dst = accumulate ? *dst + v : v;
      }
    }
  }
  auto block_total = [&](double v, int slot) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    if (tid == 0) {
      double* dst = glpart + ((size_t)split * nmb + mb) * (DP + 1) + slot;
      const double t = scratch[0] + scratch[1] + scratch[2] + scratch[3];
      *dst = accumulate ? *dst + t : t;
    }
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
    // wave w sums parameters w, w + 4, ...: lanes stride over the (split, column block) partials, one wave reduction each --
    // a fixed thread <-> partial mapping and a fixed tree (deterministic), no barriers
    const int np = nsplit * nmb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = wave; q <= d; q += 4) {
      const int j = q == d ? DP : q;
      double s = 0.0;
      for (int p = lane; p < np; p += 64) s += glpart[(size_t)p * (DP + 1) + j];
      s = wave_sum(s);
      if (lane == 0) {
        if (q == d) *g_sf2 = s + kappa_term;
        else g_ls[q] = -2.0 * ka.inv_ls[q] * s;
      }
    }
  }
  if (g_Z) {
    const int64_t total = (int64_t)M * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
      const int m = (int)(e / d), j = (int)(e - (int64_t)m * d);
      double s = 0.0;
#pragma unroll 8
      for (int sp = 0; sp < nsplit; ++sp) s += gzpart[((size_t)sp * Mp + m) * DP + j];
      g_Z[e] = 2.0 * ka.inv_ls[j] * s;
    }
  }
}

struct BwdWs {
  double *Xs, *ys, *Zs, *Pb, *bb, *gzpart, *glpart, *gacc, *bpart, *yypart, *Kfu;
  size_t bytes;
};
static BwdWs carve_bwd(void* ws, const StreamPlan& p, bool need_kfu) {
  Carver c(ws);
  BwdWs w;
  w.Xs = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.DP);
  w.ys = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1));
  w.Zs = c.take<double>((size_t)p.Mp * p.DP);
  w.Pb = c.take<double>((size_t)p.Mp * p.Mp);
  w.bb = c.take<double>((size_t)p.Mp);
  w.gzpart = c.take<double>((size_t)p.nsplit_b * p.Mp * p.DP);
  w.glpart = c.take<double>((size_t)p.nsplit_b * p.nmb * (p.DP + 1));
  w.gacc = c.take<double>((size_t)p.nsplit_b * p.nmb * (2 * p.DP + 1) * 256);
  w.bpart = c.take<double>((size_t)bpart_rows(p) * p.Mp);
  w.yypart = c.take<double>(256);
  w.Kfu = need_kfu ? c.take<double>((size_t)(p.sc_rows > 0 ? p.sc_rows : 1) * p.Mp) : nullptr;
  w.bytes = c.used();
  return w;
}

template <int DP, bool KP = true>
static void launch_bwd(int kid, int grid, hipStream_t st, const double* Kfu, const BwdWs& w, double sf2, int64_t row0,
                       int64_t nblocks, int bps, int64_t N, int M, const StreamPlan& p, int want_gz, int accumulate) {
  const SplitMap bmap{{p.taper_b[0], p.taper_b[1], p.taper_b[2], p.taper_b[3]}, bps};
#define SGP_BWD_ARGS Kfu, w.Xs, w.ys, w.Zs, w.Pb, w.bb, sf2, row0, nblocks, bmap, N, M, p.Mp, p.nmb, want_gz, accumulate, w.gacc, w.gzpart, w.glpart
  switch (kid) {
#define SGP_BWD_LAUNCH(K) \
  do { \
    if (want_gz) kbar_contract_kernel<DP, K, true, KP><<<grid, 256, 0, st>>>(SGP_BWD_ARGS); \
    else kbar_contract_kernel<DP, K, false, KP><<<grid, 256, 0, st>>>(SGP_BWD_ARGS); \
  } while (0)
    case SGP_KERNEL_RBF: SGP_BWD_LAUNCH(SGP_KERNEL_RBF); break;
    case SGP_KERNEL_MATERN32: SGP_BWD_LAUNCH(SGP_KERNEL_MATERN32); break;
    default: SGP_BWD_LAUNCH(SGP_KERNEL_MATERN52); break;
  }
#undef SGP_BWD_LAUNCH
#undef SGP_BWD_ARGS
}

}  // namespace sgp

using namespace sgp;

static size_t bwd_workspace_bytes(int64_t N, int M, int d, bool library_kfu) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  StreamPlan p = make_stream_plan(N, M, d);
  const size_t fast = carve_bwd(nullptr, p, library_kfu).bytes;
  const size_t comp = d <= COMP_MAX_DIM ? comp_bwd_workspace_bytes(N, M, d) : 0;  // one size for every kernel_id
  return fast > comp ? fast : comp;
}
extern "C" size_t sgp_suffstats_bwd_workspace_bytes(int64_t N, int M, int d) { return bwd_workspace_bytes(N, M, d, true); }
extern "C" size_t sgp_suffstats_bwd_workspace_bytes_ex(int64_t N, int M, int d, int caller_owns_kfu) {
  return bwd_workspace_bytes(N, M, d, caller_owns_kfu == 0);
}

extern "C" int sgp_suffstats_bwd(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                 const double* inv_ls, double sf2, const double* Phibar, const double* bbar,
                                 double kappabar, const double* Kfu_in, int64_t N, int M, int d, int kernel_id,
                                 double* g_ls, double* g_sf2, double* g_Z, void* ws, size_t ws_bytes,
                                 sgp_stream_t stream) {
  if (!Z || !inv_ls || !Phibar || !bbar || !g_ls || !g_sf2 || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (kernel_id == SGP_KERNEL_COMPOSITE) {  // g_ls receives the SGP_COMP_LEN-double gradient block, g_sf2 = 0
    CompSpec cs;
    if (comp_parse(inv_ls, d, &cs) != SGP_OK) return SGP_ERR_ARG;
    fill_zero(g_sf2, 1, (hipStream_t)stream);
    return comp_suffstats_bwd(X, ldx, y, Z, ldz, cs, Phibar, bbar, kappabar, N, M, d, g_ls, g_Z, ws, ws_bytes, (hipStream_t)stream);
  }
  StreamPlan p = make_stream_plan(N, M, d);
  // with a caller-owned K'_fu the whole row range is one super-chunk; the split count stays what the
  // workspace query sized (it only shrinks when the caller's range has fewer 128-row blocks)
  if (Kfu_in) p.sc_rows = p.Npad;
  BwdWs w = carve_bwd(ws, p, Kfu_in == nullptr);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;

  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;

  {  // one launch: the scaled rows of both passes' prologue and pass 2's padded, symmetrised Phibar / bbar
    PadSymJob pad;
    pad.P = Phibar; pad.out = w.Pb; pad.vec = bbar; pad.vout = w.bb;
    stream_prologue(p, ka, X, ldx, y, Z, ldz, N, M, w.Xs, w.ys, w.Zs, w.yypart, st, pad);
  }

  const int want_gz = g_Z != nullptr;
  const int grid = p.nmb * p.nsplit_b;
  auto launch = [&](const double* Kfu, int64_t row0, int64_t nblocks, int accumulate) {
    int bps = (int)((nblocks + p.nsplit_b - 1) / p.nsplit_b);
    if (bps < 1) bps = 1;
    switch (p.DP) {
      case 2: launch_bwd<2>(kernel_id, grid, st, Kfu, w, sf2, row0, nblocks, bps, N, M, p, want_gz, accumulate); break;
      case 4: launch_bwd<4>(kernel_id, grid, st, Kfu, w, sf2, row0, nblocks, bps, N, M, p, want_gz, accumulate); break;
      case 8: launch_bwd<8>(kernel_id, grid, st, Kfu, w, sf2, row0, nblocks, bps, N, M, p, want_gz, accumulate); break;
      case 16: launch_bwd<16>(kernel_id, grid, st, Kfu, w, sf2, row0, nblocks, bps, N, M, p, want_gz, accumulate); break;
      case 24: launch_bwd<24>(kernel_id, grid, st, Kfu, w, sf2, row0, nblocks, bps, N, M, p, want_gz, accumulate); break;
      default: launch_bwd<32>(kernel_id, grid, st, Kfu, w, sf2, row0, nblocks, bps, N, M, p, want_gz, accumulate); break;
    }
  };
  if (p.Npad == 0) launch(nullptr, 0, 0, 0);  // empty shard: writes zero partials
  for (int64_t r0 = 0; r0 < p.Npad; r0 += p.sc_rows) {
    const int64_t rows = (p.Npad - r0) < p.sc_rows ? (p.Npad - r0) : p.sc_rows;
    const double* Kfu = Kfu_in ? Kfu_in + r0 * p.Mp : nullptr;
    if (!Kfu_in) {
      stream_assemble(p, kernel_id, w.Xs, w.ys, w.Zs, r0, rows, N, M, w.Kfu, w.bpart, st);
      Kfu = w.Kfu;
    }
    timing_begin(TIMING_KBAR, st);
    launch(Kfu, r0, rows / TILE, r0 > 0 ? 1 : 0);
    timing_end(TIMING_KBAR, st);
  }
  const int64_t tot = (int64_t)M * d;
  const int rg = (int)((tot + 255) / 256 < 1024 ? (tot + 255) / 256 : 1024);
  bwd_reduce_kernel<<<rg < 1 ? 1 : rg, 256, 0, st>>>(w.gzpart, w.glpart, p.nsplit_b, p.nmb, p.Mp, M, p.DP, ka,
                                                    kappabar * (double)N, g_ls, g_sf2, g_Z);
  return check_launch();
}

// ---- factored pass 2 (the whitened evaluation order, ill-conditioned K_uu) -----------------------------------------------
// 2 Phibar = L^-T (Cw / s2) L^-1 is never formed: its entries are of size cond(K_uu) and cancel in Phibar K_uf (1e-2 .. 1e-1
// relative error on the gradients of the CO2 model at cond 3e9 .. 5e10, tests/studies/logp_noise.py).  Instead
//     Kbar_fu = ((K_fu L^-T)(Cw / s2)) L^-1 + y bbar^T
// one factor after the other: two plain GEMMs on the materialised K'_fu (these shards are small: N M <= 2^20 in
// CollapsedBound), the third product inside kbar_contract_kernel (Pb = L^-1 / 2, epilogue recomputing k').
// SGP_BWD_FULLY_FACTORED=1 keeps the three-product chain (A/B, accuracy studies); read once
static int bwd_fully_factored() {
  static const int v = getenv("SGP_BWD_FULLY_FACTORED") ? atoi(getenv("SGP_BWD_FULLY_FACTORED")) : 0;
  return v;
}
// caller_t: T1 = K'_fu L^-T arrives from pass 1 (sgp_suffstats_fwd_whitened_rows kept it): neither K'_fu nor T1 live in the workspace
static size_t bwd_factored_workspace_bytes(int64_t N, int M, int d, bool caller_t) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  StreamPlan p = make_stream_plan(N, M, d);
  p.sc_rows = p.Npad;
  const int nt = (caller_t ? 0 : 1) + (bwd_fully_factored() ? 1 : 0);
  const size_t fast = carve_bwd(nullptr, p, !caller_t).bytes + nt * round_up64((int64_t)(p.Npad > 0 ? p.Npad : 1) * p.Mp * 8 + 256, 256) +
                      3 * round_up64((int64_t)p.Mp * p.Mp * 8 + 256, 256);
  const size_t comp = d <= COMP_MAX_DIM ? comp_bwd_factored_workspace_bytes(N, M, d) : 0;
  return fast > comp ? fast : comp;
}
extern "C" size_t sgp_suffstats_bwd_factored_workspace_bytes(int64_t N, int M, int d) { return bwd_factored_workspace_bytes(N, M, d, false); }
extern "C" size_t sgp_suffstats_bwd_factored_workspace_bytes_ex(int64_t N, int M, int d, int caller_owns_t) {
  return bwd_factored_workspace_bytes(N, M, d, caller_owns_t != 0);
}

static int bwd_factored(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz, const double* inv_ls, double sf2,
                        const double* kuu_linv, const double* Cw, double s2, const double* bbar, double kappabar, int64_t N, int M,
                        int d, int kernel_id, const double* T_in, double* g_ls, double* g_sf2, double* g_Z, void* ws, size_t ws_bytes,
                        sgp_stream_t stream) {
  if (!Z || !inv_ls || !kuu_linv || !Cw || !bbar || !g_ls || !g_sf2 || N < 0 || M <= 0 || d <= 0 || ldz < d || !(s2 > 0.0))
    return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  if (T_in && kernel_id == SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  hipStream_t st = (hipStream_t)stream;
  if (!ws || ws_bytes < bwd_factored_workspace_bytes(N, M, d, T_in != nullptr)) return SGP_ERR_WORKSPACE;
  if (kernel_id == SGP_KERNEL_COMPOSITE) {
    CompSpec cs;
    if (comp_parse(inv_ls, d, &cs) != SGP_OK) return SGP_ERR_ARG;
    fill_zero(g_sf2, 1, st);
    return comp_suffstats_bwd_factored(X, ldx, y, Z, ldz, cs, kuu_linv, Cw, s2, bbar, kappabar, N, M, d, g_ls, g_Z, ws, ws_bytes, st);
  }
  const int fully = bwd_fully_factored();
  StreamPlan p = make_stream_plan(N, M, d);
  p.sc_rows = p.Npad;  // one super-chunk: the whole K'_fu is materialised
  BwdWs w = carve_bwd(ws, p, T_in == nullptr);
  Carver c(static_cast<char*>(ws) + round_up64((int64_t)w.bytes, 256));
  const size_t rows = (size_t)(p.Npad > 0 ? p.Npad : 1);
  double* T1 = T_in ? nullptr : c.take<double>(rows * p.Mp);
  double* T2 = fully ? c.take<double>(rows * p.Mp) : nullptr;
  double* P2 = c.take<double>((size_t)p.Mp * p.Mp);
  double* Q0 = c.take<double>((size_t)p.Mp * p.Mp);
  double* R = c.take<double>((size_t)p.Mp * p.Mp);
  // Round 4: the last two factors are multiplied FIRST, Q = (Cw / s2)(L^-1 / 2) -- an M^3 product -- so that ONE N M^2 product
  // (inside kbar_contract_kernel) is left behind T1 = K'_fu L^-T instead of two: what must stay factored is L^-T ... L^-1 around the
  // whitened core (entries of size cond(K_uu) in the explicit Phibar); Q's are of size sqrt(cond), the same size the last factor
  // of the fully factored chain has anyway.

  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;
  stream_prologue(p, ka, X, ldx, y, Z, ldz, N, M, w.Xs, w.ys, w.Zs, w.yypart, st);
  bwd_pad_plain_kernel<<<2048, 256, 0, st>>>(kuu_linv, M, p.Mp, 0.5, fully ? w.Pb : Q0);  // the kernel forms 2 sf2 (T Pb)
  bwd_pad_small_kernel<<<2048, 256, 0, st>>>(Cw, M, p.Mp, 1.0 / s2, P2);
  if (!fully) {
    GemmDesc q;  // Pb = (Cw / s2)(L^-1 / 2): L^-1 lower triangular -> column block c of the product needs k >= its start
    q.A = P2; q.lda = p.Mp; q.B = Q0; q.ldb = p.Mp; q.C = w.Pb; q.ldc = p.Mp;
    q.m = p.Mp; q.n = p.Mp; q.k = p.Mp; q.klo_mask = 2;
    gemm(q, st);
  }
  bwd_pad_vec_kernel<<<(p.Mp + 255) / 256, 256, 0, st>>>(bbar, M, p.Mp, w.bb);
  const int want_gz = g_Z != nullptr;
  const int grid = p.nmb * p.nsplit_b;
  const double* T1c = T_in ? T_in : T1;
  if (p.Npad > 0) {
    if (!T_in) {
      stream_assemble(p, kernel_id, w.Xs, w.ys, w.Zs, 0, p.Npad, N, M, w.Kfu, w.bpart, st);
      // T1 = K'_fu R, R = L^-T as a row-major operand -- the same product, by the same kernel, that pass 1 keeps for the _ex entry point
      // (kuu_linv is Mp x Mp: its padding rows hold an identity block that K'_fu's zero columns never meet)
      transpose_square(kuu_linv, p.Mp, R, st);
      GemmDesc g1;
      g1.A = w.Kfu; g1.lda = p.Mp; g1.B = R; g1.ldb = p.Mp; g1.C = T1; g1.ldc = p.Mp;
      g1.m = (int)p.Npad; g1.n = p.Mp; g1.k = p.Mp;
      g1.khi_mask = 2;  // R is upper triangular: column block c needs k < its end only -- half the product (round 4)
      gemm(g1, st);
    }
    if (fully) {
      GemmDesc g2;  // T2 = T1 (Cw / s2)
      g2.A = T1c; g2.lda = p.Mp; g2.B = P2; g2.ldb = p.Mp; g2.C = T2; g2.ldc = p.Mp;
      g2.m = (int)p.Npad; g2.n = p.Mp; g2.k = p.Mp;
      gemm(g2, st);
    }
  }
  const double* Tin = fully ? T2 : T1c;
  {
    const int64_t nblocks = p.Npad / TILE;
    int bps = (int)((nblocks + p.nsplit_b - 1) / p.nsplit_b);
    if (bps < 1) bps = 1;
    switch (p.DP) {
      case 2: launch_bwd<2, false>(kernel_id, grid, st, Tin, w, sf2, 0, nblocks, bps, N, M, p, want_gz, 0); break;
      case 4: launch_bwd<4, false>(kernel_id, grid, st, Tin, w, sf2, 0, nblocks, bps, N, M, p, want_gz, 0); break;
      case 8: launch_bwd<8, false>(kernel_id, grid, st, Tin, w, sf2, 0, nblocks, bps, N, M, p, want_gz, 0); break;
      case 16: launch_bwd<16, false>(kernel_id, grid, st, Tin, w, sf2, 0, nblocks, bps, N, M, p, want_gz, 0); break;
      case 24: launch_bwd<24, false>(kernel_id, grid, st, Tin, w, sf2, 0, nblocks, bps, N, M, p, want_gz, 0); break;
      default: launch_bwd<32, false>(kernel_id, grid, st, Tin, w, sf2, 0, nblocks, bps, N, M, p, want_gz, 0); break;
    }
  }
  const int64_t tot = (int64_t)M * d;
  const int rg = (int)((tot + 255) / 256 < 1024 ? (tot + 255) / 256 : 1024);
  bwd_reduce_kernel<<<rg < 1 ? 1 : rg, 256, 0, st>>>(w.gzpart, w.glpart, p.nsplit_b, p.nmb, p.Mp, M, p.DP, ka,
                                                    kappabar * (double)N, g_ls, g_sf2, g_Z);
  return check_launch();
}

extern "C" int sgp_suffstats_bwd_factored(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                          const double* inv_ls, double sf2, const double* kuu_linv, const double* Cw, double s2,
                                          const double* bbar, double kappabar, int64_t N, int M, int d, int kernel_id,
                                          double* g_ls, double* g_sf2, double* g_Z, void* ws, size_t ws_bytes,
                                          sgp_stream_t stream) {
  return bwd_factored(X, ldx, y, Z, ldz, inv_ls, sf2, kuu_linv, Cw, s2, bbar, kappabar, N, M, d, kernel_id, nullptr, g_ls, g_sf2, g_Z, ws,
                      ws_bytes, stream);
}
extern "C" int sgp_suffstats_bwd_factored_ex(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                             const double* inv_ls, double sf2, const double* kuu_linv, const double* Cw, double s2,
                                             const double* bbar, double kappabar, int64_t N, int M, int d, int kernel_id,
                                             const double* T_in, double* g_ls, double* g_sf2, double* g_Z, void* ws, size_t ws_bytes,
                                             sgp_stream_t stream) {
  return bwd_factored(X, ldx, y, Z, ldz, inv_ls, sf2, kuu_linv, Cw, s2, bbar, kappabar, N, M, d, kernel_id, T_in, g_ls, g_sf2, g_Z, ws,
                      ws_bytes, stream);
}
