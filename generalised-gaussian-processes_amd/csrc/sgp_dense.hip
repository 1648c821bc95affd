// M x M dense back end on gfx950: fp64 MFMA GEMM, blocked Cholesky, triangular inverse.
// Replaces what the reference reaches through torch.linalg / LAPACK / Theano for the two M x M
// factorizations and solves of the collapsed bound (psd_safe_cholesky, triangular_solve inside
// gpytorch's InducingPointKernel / ExactMarginalLogLikelihood -- reference models/sgpr.py:37,125 --
// and cholesky / solve_lower inside pymc3 MarginalSparse -- reference models/bayesian_sgpr_hmc.py:71).
//
// Design: everything O(M^3) is phrased as 64 x 64-tiled GEMMs on v_mfma_f64_16x16x4_f64 so the only
// latency-bound pieces are the 64 x 64 diagonal blocks, which one workgroup factors and inverts in LDS.
//   potrf  : ONE launch, 64 x 64 tile dataflow (tile owners wait on ready flags; diagonal tiles factored in registers,
//            off-diagonal tiles = MFMA rank-64 updates + a 16-column-panel triangular solve); optional L^-1 rhs row
//   trtri  : recursive doubling on the inverted diagonal blocks, Inv21 = -Inv22 (L21 Inv11), batched GEMMs
// Triangular structure is exploited by clipping each output tile's k-range (GemmDesc::klo/khi masks).
#include "sgp_dense.hpp"

namespace sgp {

constexpr int GT = 64;    // GEMM tile edge
constexpr int GK = 16;    // k-chunk
constexpr int GLD = GT + 16;  // LDS row stride (80 doubles: +128 B bank shift per k)

struct GemmP {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc, sA, sB, sC;
  int m, n, k;
  double alpha, beta;
  int klo_mask, khi_mask, lower_only;
};

// Operand tiles in LDS: S[k][col'] with 64-double rows and col' = (col + rot(k)) & 63, rot(k) = 4 (k >> 2) + 16 (k & 1).
//  * MFMA operand reads (ds_read_b64, 32-lane groups, 64 banks): a group reads 16 columns of k-rows kr and kr + 1; the
//    16-double shift between odd and even rows puts them on the two halves of the bank row -> conflict-free.
//  * stores (32 banks, 16- / 8-lane groups): the four k-quads a group writes are shifted by 4 doubles each ->
//    conflict-free (an unswizzled [16][80] image made every ds_write_b64 4-way: the stores of one k-chunk then
//    occupied the LDS for half as long as its MFMAs run).
__device__ __forceinline__ int tile_rot(int k) { return 4 * (k >> 2) + 16 * (k & 1); }

// op(A) tile -> S; "K-contiguous" source (A not transposed / B transposed): thread = (row, k-quad), 32 B along k;
// "MN-contiguous" source (A transposed / B not transposed): thread = (k, column pair), 2 x 16 B along the row.
// t = thread index inside its 256-thread group.
template <bool KCONTIG>
__device__ __forceinline__ void tile_fetch(const double* P, int64_t ld, int r0, int k0, int t, d2 (&v)[2]) {
  if constexpr (KCONTIG) {
    const int row = t >> 2, kq = (t & 3) * 4;
    const double* s = P + (int64_t)(r0 + row) * ld + k0 + kq;
    v[0] = *reinterpret_cast<const d2*>(s);
    v[1] = *reinterpret_cast<const d2*>(s + 2);
  } else {
    const int kk = t >> 4, c2 = (t & 15) * 2;
    const double* s = P + (int64_t)(k0 + kk) * ld + r0 + c2;
    v[0] = *reinterpret_cast<const d2*>(s);
    v[1] = *reinterpret_cast<const d2*>(s + 32);
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void tile_stash(double (*S)[GT], int t, const d2 (&v)[2]) {
  if constexpr (KCONTIG) {
    const int row = t >> 2, kq = (t & 3) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) S[kq + e][(row + tile_rot(kq + e)) & 63] = v[e >> 1][e & 1];
  } else {
    const int kk = t >> 4, c2 = (t & 15) * 2, rot = tile_rot(kk);
    *reinterpret_cast<d2*>(&S[kk][(c2 + rot) & 63]) = v[0];
    *reinterpret_cast<d2*>(&S[kk][(c2 + 32 + rot) & 63]) = v[1];
  }
}

// One 64 x 64 output tile per workgroup of 8 waves.  The two 4-wave groups take alternate 16-deep k-chunks, so every
// SIMD holds two waves; each group double-buffers its operand tiles in LDS (the stores of chunk j + 1 are issued
// before the MFMAs of chunk j, one barrier per chunk) and keeps two chunks of global loads in flight in registers.
// The groups' partial tiles are added in a fixed order (group 0 + group 1) through LDS: results do not depend on
// timing.  History on one box, 1024^3: one group, single buffer 64 us; two groups 52 us; this version see DESIGN.md.
struct GemmShared {
  double As[2][2][GK][GT];  // [group][stage]
  double Bs[2][2][GK][GT];
};
static_assert(sizeof(GemmShared) >= sizeof(double) * GT * GT, "partial tile is exchanged through the operand tiles");

template <bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm64_kernel(GemmP p) {
  __shared__ GemmShared sh;
  const int bj = blockIdx.x, bi = blockIdx.y;
  if (p.lower_only && bj > bi) return;
  const int64_t bz = blockIdx.z;
  const double* A = p.A + bz * p.sA;
  const double* B = p.B + bz * p.sB;
  double* C = p.C + bz * p.sC;

  int klo = 0, khi = p.k;
  if (p.klo_mask & 1) klo = max(klo, bi * GT);
  if (p.klo_mask & 2) klo = max(klo, bj * GT);
  if (p.khi_mask & 1) khi = min(khi, (bi + 1) * GT);
  if (p.khi_mask & 2) khi = min(khi, (bj + 1) * GT);

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int r0 = bi * GT, c0 = bj * GT;
  // column of this lane's operands in k-row ks * 4 + l4, before the 4 ks part of the rotation
  const int ca = wi * 32 + l15 + 16 * (l4 & 1), cb = wj * 32 + l15 + 16 * (l4 & 1);

  d4 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};

  if (klo < khi) {
    // chunk j of this group starts at klo + (2 j + grp) GK; both groups run the same number of barriers
    const int nchunk = (khi - klo + GK - 1) / GK, niter = (nchunk + 1) / 2;
    d2 ra0[2], rb0[2], ra1[2], rb1[2];
    auto fetch = [&](int j, d2 (&ra)[2], d2 (&rb)[2]) {
      const int c = 2 * j + grp;
      if (c < nchunk) {
        tile_fetch<!TA>(A, p.lda, r0, klo + c * GK, tid, ra);
        tile_fetch<TB>(B, p.ldb, c0, klo + c * GK, tid, rb);
      }
    };
    auto stash = [&](int j, int stage, const d2 (&ra)[2], const d2 (&rb)[2]) {
      if (2 * j + grp < nchunk) {
        tile_stash<!TA>(sh.As[grp][stage], tid, ra);
        tile_stash<TB>(sh.Bs[grp][stage], tid, rb);
      }
    };
    // r holds chunk j + 1 on entry and chunk j + 3 on exit
    auto body = [&](int j, int stage, d2 (&ra)[2], d2 (&rb)[2]) {
      const bool live = 2 * j + grp < nchunk;
      double a0[GK / 4], a1[GK / 4], b0[GK / 4], b1[GK / 4];
      if (live) {  // operands of the whole chunk first: the MFMAs start as soon as the barrier falls
        const double (*As)[GT] = sh.As[grp][stage];
        const double (*Bs)[GT] = sh.Bs[grp][stage];
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
          const int kr = ks * 4 + l4;
          a0[ks] = As[kr][(ca + 4 * ks) & 63];
          a1[ks] = As[kr][(ca + 16 + 4 * ks) & 63];
          b0[ks] = Bs[kr][(cb + 4 * ks) & 63];
          b1[ks] = Bs[kr][(cb + 16 + 4 * ks) & 63];
        }
      }
      auto mac = [&](int ks) {
        acc[0][0] = mfma16(a0[ks], b0[ks], acc[0][0]);
        acc[0][1] = mfma16(a0[ks], b1[ks], acc[0][1]);
        acc[1][0] = mfma16(a1[ks], b0[ks], acc[1][0]);
        acc[1][1] = mfma16(a1[ks], b1[ks], acc[1][1]);
      };
      if (live) mac(0);
      stash(j + 1, stage ^ 1, ra, rb);  // LDS stores and global loads of later chunks issue under the MFMAs
      if (live) mac(1);
      fetch(j + 3, ra, rb);
      if (live) {
        mac(2);
        mac(3);
      }
      __syncthreads();
    };
    fetch(0, ra0, rb0);
    fetch(1, ra1, rb1);
    stash(0, 0, ra0, rb0);
    fetch(2, ra0, rb0);
    __syncthreads();
    for (int j = 0; j < niter; j += 2) {
      body(j, 0, ra1, rb1);
      if (j + 1 < niter) body(j + 1, 1, ra0, rb0);
    }
    // group 1's partial tile -> LDS -> added by group 0
    double* red = &sh.As[0][0][0][0];
    if (grp == 1) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((u * 2 + v) * 4 + r) * 256 + tid] = acc[u][v][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u][v][r] += red[((u * 2 + v) * 4 + r) * 256 + tid];
    }
  }
  if (grp != 0) return;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + wi * 32 + u * 16 + l4 + 4 * r;
        const int col = c0 + wj * 32 + v * 16 + l15;
        double* dst = C + (int64_t)row * p.ldc + col;
        const double val = p.alpha * acc[u][v][r];
        *dst = (p.beta == 0.0) ? val : fma(p.beta, *dst, val);
      }
}

// C = alpha * sum over S k-slices (+ beta C): for products with a few output tiles and a long contraction (the
// M x M x B products of the SVGP reverse pass: 16 tiles, k = 4096) -- S workgroups per tile write partial tiles, a
// second launch adds them in slice order.  scratch: S * m * n doubles.  No k-range masks, no batch.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const double* __restrict__ part, int S, int m, int n, double* __restrict__ C,
                                                            int64_t ldc, double alpha, double beta) {
  const int64_t total = (int64_t)m * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e - (int64_t)r * n);
    double s = 0.0;
    for (int z = 0; z < S; ++z) s += part[(int64_t)z * total + e];
    double* dst = C + (int64_t)r * ldc + c;
    *dst = (beta == 0.0) ? alpha * s : fma(beta, *dst, alpha * s);
  }
}
void gemm_splitk(const GemmDesc& g, int S, double* scratch, hipStream_t st) {
  if (g.m <= 0 || g.n <= 0 || S <= 1 || g.k % (S * GK) != 0) {
    gemm(g, st);
    return;
  }
  const int kc = g.k / S;
  GemmDesc p = g;
  p.k = kc;
  p.batch = S;
  p.sA = g.ta ? (int64_t)kc * g.lda : kc;  // op(A) is m x k: the k index runs along rows of A when transposed
  p.sB = g.tb ? kc : (int64_t)kc * g.ldb;
  p.C = scratch;
  p.ldc = g.n;
  p.sC = (int64_t)g.m * g.n;
  p.alpha = 1.0;
  p.beta = 0.0;
  gemm(p, st);
  const int64_t total = (int64_t)g.m * g.n;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  splitk_reduce_kernel<<<blocks, 256, 0, st>>>(scratch, S, g.m, g.n, g.C, g.ldc, g.alpha, g.beta);
}

void gemm(const GemmDesc& g, hipStream_t st) {
  if (g.m <= 0 || g.n <= 0 || g.batch <= 0) return;
  GemmP p{g.A, g.B, g.C, g.lda, g.ldb, g.ldc, g.sA, g.sB, g.sC, g.m, g.n, g.k,
          g.alpha, g.beta, g.klo_mask, g.khi_mask, g.lower_only ? 1 : 0};
  dim3 grid(g.n / GT, g.m / GT, g.batch);
  if (!g.ta && !g.tb) gemm64_kernel<false, false><<<grid, 512, 0, st>>>(p);
  else if (!g.ta && g.tb) gemm64_kernel<false, true><<<grid, 512, 0, st>>>(p);
  else if (g.ta && !g.tb) gemm64_kernel<true, false><<<grid, 512, 0, st>>>(p);
  else gemm64_kernel<true, true><<<grid, 512, 0, st>>>(p);
}

// ---------------------------------------------------------------------------------------------
// 64 x 64 diagonal block: Cholesky + inverse of the factor, one workgroup, all in LDS
// ---------------------------------------------------------------------------------------------
constexpr int DB = 64;
constexpr int DLD = DB + 1;

// Inv[o+s .. o+2s)[o .. o+s) = -Inv22 * (S21 * Inv11) for `npairs` pairs at o = 0, 2s, ...
// The intermediate T = S21 Inv11 is parked, transposed, in the strictly-upper block of S that a lower-triangular
// matrix leaves unused (T[i][j] at S[o + j][o + s + i]): two LDS tiles instead of three.
__device__ __forceinline__ void inv_combine(double (*S)[DLD], double (*Inv)[DLD], int s, int npairs) {
  const int tid = threadIdx.x;
  const int per = s * s;
  for (int e = tid; e < npairs * per; e += 256) {
    const int pr = e / per, r = e - pr * per;
    const int i = r / s, j = r - i * s;
    const int o = pr * 2 * s;
    double acc = 0.0;
#pragma unroll 8
    for (int q = j; q < s; ++q) acc = fma(S[o + s + i][o + q], Inv[o + q][o + j], acc);  // Inv11 lower: q >= j
    S[o + j][o + s + i] = acc;
  }
  __syncthreads();
  for (int e = tid; e < npairs * per; e += 256) {
    const int pr = e / per, r = e - pr * per;
    const int i = r / s, j = r - i * s;
    const int o = pr * 2 * s;
    double acc = 0.0;
#pragma unroll 8
    for (int q = 0; q <= i; ++q) acc = fma(Inv[o + s + i][o + s + q], S[o + j][o + s + q], acc);  // Inv22 lower: q <= i
    Inv[o + s + i][o + j] = -acc;
  }
  __syncthreads();
}

// Inv = S^-1 for a 64 x 64 lower-triangular S in LDS (Inv must be zero on entry; the strictly-upper part of S is
// scratch): four 16 x 16 diagonal blocks by forward substitution, then two doubling steps 16 -> 32 -> 64.
__device__ __forceinline__ void block_inverse64(double (*S)[DLD], double (*Inv)[DLD]) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int b16 = (tid >> 4) * 16, c = tid & 15;
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int q = 0; q < r; ++q) s = fma(-S[b16 + r][b16 + q], x[q], s);
      x[r] = s / S[b16 + r][b16 + r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Inv[b16 + r][b16 + c] = x[r];
  }
  __syncthreads();
  inv_combine(S, Inv, 16, 2);
  inv_combine(S, Inv, 32, 1);
}

// inverses of the 64 x 64 diagonal blocks of an already-factored L (one workgroup per block)
__global__ __launch_bounds__(256) void tri_diag_inv_kernel(const double* L, double* Linv, int64_t ld) {
  __shared__ double S[DB][DLD];
  __shared__ double Inv[DB][DLD];
  const int tid = threadIdx.x, k0 = blockIdx.x * DB;
  for (int e = tid; e < DB * DB; e += 256) {
    const int i = e >> 6, j = e & 63;
    S[i][j] = (j <= i) ? L[(int64_t)(k0 + i) * ld + k0 + j] : 0.0;
    Inv[i][j] = 0.0;
  }
  __syncthreads();
  block_inverse64(S, Inv);
  for (int e = tid; e < DB * DB; e += 256) {
    const int r = e >> 6, c = e & 63;
    Linv[(int64_t)(k0 + r) * ld + k0 + c] = Inv[r][c];
  }
}
void tri_diag_inverse(const double* L, double* Linv, int64_t ld, int Mp, hipStream_t st) {
  tri_diag_inv_kernel<<<Mp / DB, 256, 0, st>>>(L, Linv, ld);
}

constexpr int PLD = 18;  // LDS row stride of a 64 x 16 panel (16-byte aligned rows)

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) off the serial pivot chain: hardware seed (v_rsq_f64, ~2^-26 relative) + two Newton steps
// y <- y + y (1/2 - d y^2 / 2); seven dependent VALU ops instead of the sqrt + divide expansions (~40).
__device__ __forceinline__ double rsqrt_newton(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = fma(y, fma(-h * y, y, 0.5), y);
  y = fma(y, fma(-h * y, y, 0.5), y);
  return y;
}

// 16 consecutive doubles from LDS (16-byte aligned; wave-uniform or per-lane address)
__device__ __forceinline__ void lds_row16(double (&v)[16], const double* p) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const d2 t = *reinterpret_cast<const d2*>(p + 2 * e);
    v[2 * e] = t[0];
    v[2 * e + 1] = t[1];
  }
}
// Cholesky of a 64 x 64 block held as  thread (i = tid & 63, g = tid >> 6) <-> a[k] = A[i][16 g + k].  Wave pb factors
// the 16-column panel pb in registers; on return a[] holds L (garbage above the diagonal), Sp the panels and Dinv the
// inverses of the four 16 x 16 diagonal blocks.  prog[0..3] must be 0 on entry (the caller's barrier covers it).
//
// The pivot chain is the critical path.  Inside a panel the columns are formed left-looking:
//     t  = L[j][j-1]                     one v_readlane pair from lane j (the column finished a moment ago)
//     d  = p_j - t^2 ;  rs = 1/sqrt(d)   p_j = A[j][j] - sum_{k<jj-1} L[j][k]^2, fetched by a readlane
//     l  = (p_i - l_prev t) rs           every lane: its entry of column j
// and p for the NEXT pivot -- this lane's finished entries dotted with row j + 1 of the panel (LDS at a wave-uniform
// address for the entries stored two or more pivots ago, one more readlane for the newest) -- has no dependence on
// the current rsqrt.  Wall-clock stamps: a 16-pivot panel takes 1.6-2.4 us = 250-350 cycles per pivot although the
// dependent chain is ~90: the lone wave is ISSUE-bound (~45 instructions per pivot at 5-7 cycles each).
//
// Between panels there is no barrier and no rank-16 update (that cost the next panel's wave 1.2 us before it could
// start): the waves to the right of the chain wave follow it column by column.  The chain wave stores every finished
// column also transposed (Lt[pb][jj][row]) and then bumps prog[pb]; a follower polls prog[pb], reads its own entry of
// the column and the 16 entries of its panel's rows (one contiguous, wave-uniform segment of Lt) and applies the
// rank-1 update to its 16 columns.  When the chain wave finishes column 15 the next panel's wave is one rank-1 update
// away from starting its own chain.  LDS executes one wave's operations in order, so column-then-counter stores and
// counter-then-column loads need only compiler barriers.  The last panel's block inverse is computed by wave 0
// (idle by then), row by row behind the chain.
__device__ __forceinline__ int prog_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Dinv[blk] row rr for column c = lane (lanes < 16), given the finished row 16 blk + rr of the panel in `row`
__device__ __forceinline__ void dinv_row(double (&y)[16], const double (&row)[16], int rr, int c, double rdiag) {
  double s0 = (rr == c) ? 1.0 : 0.0, s1 = 0.0;  // two accumulators: half the dependent-FMA latency
#pragma unroll
  for (int q = 0; q < 16; q += 2) {
    if (q < rr) s0 = fma(-row[q], y[q], s0);
    if (q + 1 < rr) s1 = fma(-row[q + 1], y[q + 1], s1);
  }
  y[rr] = (s0 + s1) * rdiag;
}

__device__ __forceinline__ void diag_factor64_fast(double (&a)[16], double (*Sp)[DB][PLD], double (*Lt)[16][DB], int* prog,
                                                   double* rdiag3, double (*Dinv)[16][17], int* bad, int i, int g) {
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    if (g == pb) {
      const int base = 16 * pb;
      double rsv[16];  // 1 / L[j][j] of this panel's pivots (wave-uniform)
      double p = a[0], lprev = 0.0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = base + jj;
        // p for pivot jj + 1 without its k = jj term: independent of this pivot's chain (row j + 1 of the panel was
        // completed up to column jj - 1 by the previous iterations' stores, same wave: LDS keeps program order)
        double pnext = 0.0;
        if (jj < 15) {
          double s0 = a[jj + 1], s1 = 0.0;
          const double* row = &Sp[pb][j + 1][0];
          const int nl = jj > 0 ? jj - 1 : 0;  // k < jj - 1 from LDS (stored at least two pivots ago) ...
#pragma unroll
          for (int k = 0; k + 1 < nl; k += 2) {
            const d2 r2 = *reinterpret_cast<const d2*>(row + k);
            s0 = fma(-a[k], r2[0], s0);
            s1 = fma(-a[k + 1], r2[1], s1);
          }
          if (nl & 1) s0 = fma(-a[nl - 1], row[nl - 1], s0);
          if (jj > 0) s1 = fma(-a[jj - 1], readlane_f64(lprev, j + 1), s1);  // ... k = jj - 1 straight from lane j + 1
          pnext = s0 + s1;
        }
        double d, afull;
        if (jj == 0) {
          d = readlane_f64(p, j);
          afull = p;
        } else {
          const double t = readlane_f64(lprev, j);
          const double pj = readlane_f64(p, j);
          d = fma(-t, t, pj);
          afull = fma(-lprev, t, p);
        }
        if (!(d > 0.0)) {  // non-positive or NaN pivot: LAPACK-style report, continue on a unit pivot
          if (i == 0 && *bad == 0) *bad = j + 1;
          d = 1.0;
        }
        rsv[jj] = rsqrt_newton(d);
        const double l = afull * rsv[jj];  // L[i][j] (row j itself: d / sqrt(d) = sqrt(d))
        a[jj] = l;
        Sp[pb][i][jj] = (i >= j) ? l : 0.0;
        Lt[pb][jj][i] = l;
        if (pb == 3) rdiag3[jj] = rsv[jj];
        asm volatile("" ::: "memory");
        __hip_atomic_store(&prog[pb], jj + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lprev = l;
        p = pnext;
      }
      // Inverse of the 16 x 16 diagonal block just factored (what the triangular solves of the tiles below this one
      // start from), while the next panel's wave runs its chain: lane c < 16 <-> column c, forward substitution with
      // the reciprocal pivots kept from the chain.  (Panel 3: wave 0 does it, see below.)
      if (pb < 3 && i < 16) {
        double y[16], lrow[16], lnext[16];
        lds_row16(lrow, &Sp[pb][16 * pb][0]);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          if (rr < 15) lds_row16(lnext, &Sp[pb][16 * pb + rr + 1][0]);
          __builtin_amdgcn_sched_barrier(0);
          dinv_row(y, lrow, rr, i, rsv[rr]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 16; ++q) lrow[q] = lnext[q];
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) Dinv[pb][rr][i] = y[rr];
      }
    } else if (g > pb) {
      // follow the chain wave: rank-1 update of this wave's 16 columns c = 16 g + k by every column as it appears:
      //   a[i][c] -= L[i][j] L[c][j]
#pragma unroll 1
      for (int jj = 0; jj < 16; ++jj) {
        while (prog_load(&prog[pb]) <= jj) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const double own = Lt[pb][jj][i];
        double lc[16];
        lds_row16(lc, &Lt[pb][jj][16 * g]);  // wave-uniform segment: broadcast reads
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = fma(-own, lc[k], a[k]);
      }
    } else if (pb == 3 && g == 0) {
      // block inverse of the last panel, one row behind wave 3's chain
      if (i < 16) {
        double y[16], lrow[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          // counter and row prefix are read back to back (LDS keeps the order: a row read after a counter value > rr is
          // complete); only when the chain is not there yet is the pair repeated
          double rd;
          for (;;) {
            const int done = prog_load(&prog[3]);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; 2 * e < rr; ++e) {
              const d2 t2 = *reinterpret_cast<const d2*>(&Sp[3][48 + rr][2 * e]);
              lrow[2 * e] = t2[0];
              lrow[2 * e + 1] = t2[1];
            }
            rd = rdiag3[rr];
            asm volatile("" ::: "memory");
            if (done > rr) break;
            __builtin_amdgcn_s_sleep(1);
          }
          dinv_row(y, lrow, rr, i, rd);
          Dinv[3][rr][i] = y[rr];  // stored at once: keeps the arithmetic inside the loop (it was being sunk below it)
        }
      }
    }
  }
  __syncthreads();
}


// ---------------------------------------------------------------------------------------------
// Whole factorization in ONE launch: tile dataflow.
//   The lower triangle is cut into 64 x 64 tiles, numbered column by column; workgroup w owns items w, w + G, ...
//   and handles them in that order.  Tile (i, j), i > j:
//       acc = sum_{p<j} L(i,p) L(j,p)^T          each term as soon as its two operand tiles are published (MFMA)
//       T   = A(i,j) - acc
//       L(i,j) = T L(j,j)^-T                     once L(j,j) is published; matrix cores, 16-column panels
//       publish: store, release fence, ready[tile] = 1
//   The item of tile (j+1, j) carries on with the diagonal tile (j+1, j+1) (see the kernel): update by X X^T from
//   LDS, chol() in registers (diag_factor64_fast), publish.  Tile (0, 0) is an item of its own; the items of the
//   other diagonal tiles are empty.
//   Every dependency of an item has a smaller number, so with all G <= 256 workgroups resident (one per CU; if
//   another kernel holds CUs they trickle in as it retires -- nothing they wait for depends on them) the
//   smallest unfinished item can always run: no deadlock.  Look-ahead is implicit: off the critical path
//   (chol -> solve of the next row block -> its X X^T -> chol) everything is done early.
//   Flags are agent-scope atomics behind release fences (L2 is per XCD on gfx950).  A spin that exceeds
//   DF_SPIN_LIMIT polls raises the abort flag and reports info = POTRF_TIMEOUT instead of hanging.
// ---------------------------------------------------------------------------------------------
constexpr int DF_MAX_WG = 256;
constexpr int DF_SPIN_LIMIT = 1 << 24;
constexpr int POTRF_TIMEOUT = SGP_INFO_TIMEOUT;
constexpr int TLD = DB + 2;  // LDS stride of the T / X tile (16-byte aligned rows)

struct DfShared {
  union {
    struct { double As[GK][GLD]; double Bs[GK][GLD]; } mac;  // operand chunks of the rank-64 updates
    double Sp[4][DB][PLD];                                    // panels of L(j,j) (factor / solve phase)
  };
  double Ts[DB][TLD];
  double Dinv[4][16][17];  // inverses of the four 16 x 16 diagonal blocks of L(j,j) (odd stride: MFMA operand reads)
  double Lt[4][16][DB];    // diagonal tiles: finished columns, transposed, for the waves following the pivot chain
  int prog[4];             // ... and how many columns of each panel are finished
  double rdiag3[16];       // reciprocal pivots of the last panel (for wave 0's block inverse behind the chain)
  int bad;
  int dead;
};

__device__ __forceinline__ int df_flag_load(const int* f) {
  return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Visibility of a published tile (L2 is per XCD on gfx950, write-back, not coherent across XCDs for ordinary lines):
//   producer: ordinary stores, then an agent-scope RELEASE fence (L2 write-back) before the flag is raised;
//   consumer: the flag is an agent-scope atomic; the tile is then read with ordinary loads.  No acquire-side cache
//   invalidate is needed: every L2 starts the launch clean, a line of tile X enters the L2 of another XCD only
//   through a consumer's first read, which happens after X was published (written back), and X never changes again.
// One lane per workgroup polls (540 waves hammering a handful of cache lines delayed the very store they were waiting
// for: occasional 10x slower launches); the others wait at the barrier.  Flags sit DF_FLAG_STRIDE ints apart, one cache
// line each, so the polls spread over the L2 channels.  Returns false when the launch has been aborted.
constexpr int DF_FLAG_STRIDE = 32;
__device__ __forceinline__ bool df_wait(const int* flag, int* abort_flag, int* dead) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (df_flag_load(flag) == 0) {
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      if ((spins & 255) == 0 && df_flag_load(abort_flag) != 0) { *dead = 1; break; }
      if (spins > DF_SPIN_LIMIT) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *dead = 1;
        break;
      }
    }
  }
  __syncthreads();
  return *dead == 0;
}

// acc += P(64 x 64) Q(64 x 64)^T, both row-major with the contraction index contiguous (tiles of L)
__device__ __forceinline__ void df_mac(const double* P, const double* Q, int64_t ld, DfShared& sh, d4 (&acc)[2][2]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int row = t >> 2, kq = (t & 3) * 4;
  double va[4][4], vb[4][4];  // the whole of both tiles: one round trip to L2, then four LDS chunks
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double* sa = P + (int64_t)row * ld + c * GK + kq;
    const double* sb = Q + (int64_t)row * ld + c * GK + kq;
    const d2 a0 = *reinterpret_cast<const d2*>(sa), a1 = *reinterpret_cast<const d2*>(sa + 2);
    const d2 b0 = *reinterpret_cast<const d2*>(sb), b1 = *reinterpret_cast<const d2*>(sb + 2);
    va[c][0] = a0[0]; va[c][1] = a0[1]; va[c][2] = a1[0]; va[c][3] = a1[1];
    vb[c][0] = b0[0]; vb[c][1] = b0[1]; vb[c][2] = b1[0]; vb[c][3] = b1[1];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sh.mac.As[kq + e][row] = va[c][e];
      sh.mac.Bs[kq + e][row] = vb[c][e];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      const int kr = ks * 4 + l4;
      const double a0 = sh.mac.As[kr][wi * 32 + l15], a1 = sh.mac.As[kr][wi * 32 + 16 + l15];
      const double b0 = sh.mac.Bs[kr][wj * 32 + l15], b1 = sh.mac.Bs[kr][wj * 32 + 16 + l15];
      acc[0][0] = mfma16(a0, b0, acc[0][0]);
      acc[0][1] = mfma16(a0, b1, acc[0][1]);
      acc[1][0] = mfma16(a1, b0, acc[1][0]);
      acc[1][1] = mfma16(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
}

// Forward substitution of one right-hand side, carried along by the factorization (work item `ntile`, so every tile
// it waits for has a smaller number): block jb of sol needs the published row jb of L,
//     r = rhs_jb - sum_{p<jb} L(jb,p) sol_p        thread (row, 16-column slice) partial dot products, summed through LDS
//     sol_jb = L(jb,jb)^-1 r                       one wave, lane <-> row, 64 steps of readlane + fma
// All but the last block are done while later block columns are still being factored.
__device__ __forceinline__ void df_solve_rhs(const double* A, int64_t ld, int nb, const int* ready, int* abort_flag,
                                             const double* rhs, double* sol, DfShared& sh) {
  const int tid = threadIdx.x, row = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* qs = &sh.Ts[0][0];          // the solution so far (<= 4096 doubles fit the T tile)
  double* red = &sh.Dinv[0][0][0];    // 4 x 64 partial sums
  auto tile_no = [&](int ti, int tj) { return tj * nb - (tj * (tj - 1)) / 2 + (ti - tj); };
  for (int jb = 0; jb < nb; ++jb) {
    double part = 0.0;
    for (int p = 0; p < jb; ++p) {
      if (!df_wait(ready + tile_no(jb, p) * DF_FLAG_STRIDE, abort_flag, &sh.dead)) return;
      const double* src = A + ((int64_t)jb * DB + row) * ld + (int64_t)p * DB + 16 * g;
      double lv[16], qv[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const d2 t2 = *reinterpret_cast<const d2*>(src + 2 * k);
        lv[2 * k] = t2[0];
        lv[2 * k + 1] = t2[1];
      }
      lds_row16(qv, qs + p * DB + 16 * g);
#pragma unroll
      for (int k = 0; k < 16; ++k) part = fma(lv[k], qv[k], part);
    }
    red[g * DB + row] = part;
    if (!df_wait(ready + tile_no(jb, jb) * DF_FLAG_STRIDE, abort_flag, &sh.dead)) return;
    __syncthreads();
    if (g == 0) {
      double rr = rhs[jb * DB + row] - (red[row] + red[DB + row] + red[2 * DB + row] + red[3 * DB + row]);
      const double* lsrc = A + ((int64_t)jb * DB + row) * ld + (int64_t)jb * DB;
      double lrow[DB];
#pragma unroll
      for (int k = 0; k < DB / 2; ++k) {
        const d2 t2 = *reinterpret_cast<const d2*>(lsrc + 2 * k);
        lrow[2 * k] = t2[0];
        lrow[2 * k + 1] = t2[1];
      }
      const double dinv = 1.0 / lsrc[row];  // own diagonal entry
      double mine = 0.0;
#pragma unroll
      for (int c = 0; c < DB; ++c) {
        const double xc = readlane_f64(rr, c) * readlane_f64(dinv, c);
        if (row == c) mine = xc;
        rr = fma(-lrow[c], xc, rr);  // rows <= c go stale, never read again
      }
      qs[jb * DB + row] = mine;
      sol[jb * DB + row] = mine;
    }
    __syncthreads();
  }
}

// acc += X X^T for the 64 x 64 tile X held in LDS as Xs[row][col] (the sub-diagonal tile its owner has just solved):
// the last rank-64 update of the next diagonal tile without a round trip through global memory.
__device__ __forceinline__ void df_mac_lds(const double (*Xs)[TLD], d4 (&acc)[2][2], int wi, int wj, int l15, int l4) {
#pragma unroll
  for (int ks = 0; ks < DB / 4; ++ks) {
    const int k = 4 * ks + l4;
    const double a0 = Xs[wi * 32 + l15][k], a1 = Xs[wi * 32 + 16 + l15][k];
    const double b0 = Xs[wj * 32 + l15][k], b1 = Xs[wj * 32 + 16 + l15][k];
    acc[0][0] = mfma16(a0, b0, acc[0][0]);
    acc[0][1] = mfma16(a0, b1, acc[0][1]);
    acc[1][0] = mfma16(a1, b0, acc[1][0]);
    acc[1][1] = mfma16(a1, b1, acc[1][1]);
  }
}

// Work items = tiles in column-major order, with one fusion on the critical path: the item of the sub-diagonal tile
// (j+1, j) also owns the diagonal tile (j+1, j+1) (whose own item is skipped).  That workgroup solves
// X = T L(j,j)^-T, publishes X and -- still holding X in LDS -- applies X X^T to the diagonal tile it has already
// updated with every earlier column, factors it and publishes it: the hand-over (publish, poll, reload from L2) between
// the two critical tiles of a step is gone.  dinv_g: nb x 4 x 16 x 16 doubles of scratch, the inverses of the 16 x 16
// diagonal blocks of every L(j,j), written by the diagonal tile's owner before it raises the tile's flag.
__global__ __launch_bounds__(256) void potrf_dataflow_kernel(double* A, int64_t ld, int nb, int* ready, double* dinv_g, int* info,
                                                              int info_base, const double* rhs, double* sol, double* Linv) {
  __shared__ DfShared sh;
  const int tid = threadIdx.x, r = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int ntile = nb * (nb + 1) / 2;
  int* abort_flag = ready + ntile * DF_FLAG_STRIDE;
  if (tid == 0) sh.dead = 0;
  __syncthreads();
  const int nitem = ntile + (rhs ? 1 : 0);
  auto tile_no = [&](int ti, int tj) { return tj * nb - (tj * (tj - 1)) / 2 + (ti - tj); };
  auto zero_acc = [&](d4 (&acc)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};
  };
  // acc (MFMA layout) -> sh.Ts[row][col]
  auto acc_to_ts = [&](const d4 (&acc)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) sh.Ts[wi * 32 + u * 16 + l4 + 4 * q][wj * 32 + v * 16 + l15] = acc[u][v][q];
  };
  // publish tile number tn: stores written back (release) -> barrier -> flag
  auto publish = [&](int tn) __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(ready + tn * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // Diagonal tile jd: x[] = T (row r, columns 16 g ..) -> factor, store, report, 16 x 16 block inverses -> dinv_g,
  // publish, then (off the critical path) the 64 x 64 block inverse -> Linv.  sh.Ts / sh.Sp are free on entry.
  auto diag_finish = [&](int jd, double (&x)[16]) __attribute__((always_inline)) {
    double* Ajj = A + (int64_t)jd * DB * (ld + 1);
    diag_factor64_fast(x, sh.Sp, sh.Lt, sh.prog, sh.rdiag3, sh.Dinv, &sh.bad, r, g);
    double* dst = Ajj + (int64_t)r * ld + 16 * g;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      *reinterpret_cast<d2*>(dst + 2 * k) = d2{(16 * g + 2 * k <= r) ? x[2 * k] : 0.0, (16 * g + 2 * k + 1 <= r) ? x[2 * k + 1] : 0.0};
    {
      double* dg = dinv_g + (size_t)jd * 1024;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = tid + 256 * e;  // (blk, row, col) = (idx >> 8, (idx >> 4) & 15, idx & 15)
        dg[idx] = sh.Dinv[idx >> 8][(idx >> 4) & 15][idx & 15];
      }
    }
    if (tid == 0 && sh.bad != 0 && *info == 0) *info = info_base + jd * DB + sh.bad;
    publish(tile_no(jd, jd));
    if (Linv) {
      // inverse of this diagonal block -> Linv, level 0 of tri_inverse().  The T tile holds S, the panel / operand
      // region holds the inverse.
      static_assert(sizeof(sh.Ts) >= sizeof(double) * DB * DLD && sizeof(sh.Sp) >= sizeof(double) * DB * DLD, "LDS reuse");
      double (*S)[DLD] = reinterpret_cast<double (*)[DLD]>(&sh.Ts[0][0]);
      double (*Inv)[DLD] = reinterpret_cast<double (*)[DLD]>(&sh.Sp[0][0][0]);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        S[r][16 * g + k] = (16 * g + k <= r) ? x[k] : 0.0;
        Inv[r][16 * g + k] = 0.0;
      }
      __syncthreads();
      block_inverse64(S, Inv);
      double* ldst = Linv + ((int64_t)jd * DB + r) * ld + (int64_t)jd * DB + 16 * g;
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<d2*>(ldst + 2 * k) = d2{Inv[r][16 * g + 2 * k], Inv[r][16 * g + 2 * k + 1]};
      __syncthreads();
    }
  };

  int j = 0, start = 0;  // column of the current tile and number of the first tile of that column
  for (int t = blockIdx.x; t < nitem; t += gridDim.x) {
    if (t == ntile) {  // the last work item: sol = L^-1 rhs, 64 entries at a time, trailing the factorization
      df_solve_rhs(A, ld, nb, ready, abort_flag, rhs, sol, sh);
      break;
    }
    while (t >= start + (nb - j)) { start += nb - j; ++j; }
    const int i = j + (t - start);
    if (i == j && j > 0) continue;  // done by the owner of tile (j, j - 1)
    double* Aij = A + (int64_t)i * DB * ld + (int64_t)j * DB;

    if (i == j) {  // tile (0, 0): nothing to wait for
      double x[16];
      const double* src = Aij + (int64_t)r * ld + 16 * g;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const d2 v = *reinterpret_cast<const d2*>(src + 2 * k);
        x[2 * k] = v[0];
        x[2 * k + 1] = v[1];
      }
      if (tid < 4) sh.prog[tid] = 0;
      if (tid == 0) sh.bad = 0;
      __syncthreads();
      diag_finish(0, x);
      continue;
    }

    {  // the mirrored tile is the strictly-upper part of the result: zero
      double* U = A + (int64_t)j * DB * ld + (int64_t)i * DB;
      for (int e = tid; e < DB * DB / 2; e += 256) {
        const int rr = e >> 5, cc = (e & 31) * 2;
        *reinterpret_cast<d2*>(U + (int64_t)rr * ld + cc) = d2{0.0, 0.0};
      }
    }
    const bool head = (i == j + 1);  // this item continues with the diagonal tile (i, i)

    d4 acc[2][2], accd[2][2];
    zero_acc(acc);
    zero_acc(accd);
    for (int p = 0; p < j; ++p) {
      if (!df_wait(ready + tile_no(i, p) * DF_FLAG_STRIDE, abort_flag, &sh.dead)) return;
      if (!df_wait(ready + tile_no(j, p) * DF_FLAG_STRIDE, abort_flag, &sh.dead)) return;
      const double* Lip = A + (int64_t)i * DB * ld + (int64_t)p * DB;
      df_mac(Lip, A + (int64_t)j * DB * ld + (int64_t)p * DB, ld, sh, acc);
      if (head) df_mac(Lip, Lip, ld, sh, accd);  // the diagonal tile's updates by the columns before j
    }
    acc_to_ts(acc);
    if (tid < 4) sh.prog[tid] = 0;
    if (tid == 0) sh.bad = 0;
    __syncthreads();

    // X = T L(j,j)^-T on the matrix cores, one wave per 16 rows of T and no barrier between the panels.
    // Y = T[rows]^T is kept as four 16 x 16 blocks in the MFMA accumulator layout (component s of block pb, lane l:
    // T[16 g + (l & 15)][16 pb + 4 s + (l >> 4)]), which is exactly the B operand of k-step s, so
    //     X_pb^T = Dinv_pb Y_pb                  (Dinv_pb: inverse of the 16 x 16 diagonal block pb of L(j,j))
    //     Y_q   -= L[q][pb] X_pb^T   for q > pb
    // chain from block to block in registers; only the A operands (Dinv, L) come from LDS.
    d4 yb[4];
    {
      const double* src = Aij + (int64_t)(16 * g + l15) * ld + l4;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) yb[pb][sq] = src[16 * pb + 4 * sq] - sh.Ts[16 * g + l15][16 * pb + 4 * sq + l4];
    }
    double xd[16];  // head: row r, columns 16 g .. of A(i,i), fetched before the wait
    if (head) {
      const double* src = A + (int64_t)i * DB * (ld + 1) + (int64_t)r * ld + 16 * g;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const d2 v = *reinterpret_cast<const d2*>(src + 2 * k);
        xd[2 * k] = v[0];
        xd[2 * k + 1] = v[1];
      }
    }
    if (!df_wait(ready + tile_no(j, j) * DF_FLAG_STRIDE, abort_flag, &sh.dead)) return;
    {  // L(j,j) -> Sp panels, block inverses -> Dinv
      const double* src = A + (int64_t)j * DB * (ld + 1) + (int64_t)r * ld + 16 * g;
      const double* dg = dinv_g + (size_t)j * 1024;
      double v[16], dv[4];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const d2 t2 = *reinterpret_cast<const d2*>(src + 2 * k);
        v[2 * k] = t2[0];
        v[2 * k + 1] = t2[1];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) dv[e] = dg[tid + 256 * e];
#pragma unroll
      for (int k = 0; k < 16; ++k) sh.Sp[g][r][k] = v[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = tid + 256 * e;
        sh.Dinv[idx >> 8][(idx >> 4) & 15][idx & 15] = dv[e];
      }
    }
    __syncthreads();
    d4 xb[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      d4 xa = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) xa = mfma16(sh.Dinv[pb][l15][4 * sq + l4], yb[pb][sq], xa);
      xb[pb] = xa;
#pragma unroll
      for (int q = pb + 1; q < 4; ++q)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) yb[q] = mfma16(-sh.Sp[pb][16 * q + l15][4 * sq + l4], xa[sq], yb[q]);
    }
    {
      double* dst = Aij + (int64_t)(16 * g + l15) * ld + l4;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
          dst[16 * pb + 4 * sq] = xb[pb][sq];
          if (head) sh.Ts[16 * g + l15][16 * pb + 4 * sq + l4] = xb[pb][sq];  // X stays on chip for the update below
        }
    }
    if (!head) {
      publish(t);
    } else {
      // X is published only after the update below: its stores drain while the MFMAs run (raising the flag first
      // meant waiting 1.2-1.9 us for the write-back on the critical path; the other tiles of column i need X much later)
      __syncthreads();
      df_mac_lds(sh.Ts, accd, wi, wj, l15, l4);
      publish(t);  // (barrier: everybody is done reading X)
      acc_to_ts(accd);
      if (tid < 4) sh.prog[tid] = 0;
      if (tid == 0) sh.bad = 0;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; ++k) xd[k] -= sh.Ts[r][16 * g + k];
      __syncthreads();  // Ts / the operand region become scratch of the factorization
      diag_finish(i, xd);
    }
  }
}

// Flags / status words are cleared by a plain kernel, never hipMemsetAsync: with the Kuu chain replayed from a hipGraph
// by a helper host thread while the main thread enqueues its own memsets, the runtime's memset nodes were seen to
// leave stale words behind on ROCm 7.2 (a spurious abort flag, info = 0x0c0c0c0c).
__global__ void zero_ints_kernel(int* p, int n) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) p[e] = 0;
}
void zero_ints(int* p, int n, hipStream_t st) {
  const int blocks = (n + 255) / 256;
  zero_ints_kernel<<<blocks < 64 ? blocks : 64, 256, 0, st>>>(p, n);
}
__global__ void potrf_timeout_kernel(const int* abort_flag, int* info) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *abort_flag != 0) *info = POTRF_TIMEOUT;
}

size_t potrf_flag_ints(int Mp) {
  const size_t nb = Mp / DB;
  return (nb * (nb + 1) / 2 + 1) * DF_FLAG_STRIDE;  // one cache line per tile flag + the abort flag
}
size_t potrf_scratch_ints(int Mp) {
  const size_t nb = Mp / DB;
  return potrf_flag_ints(Mp) + nb * 1024 * 2;  // + the 16 x 16 block inverses of every diagonal tile (doubles)
}

const int* potrf_abort_flag(const int* scratch, int Mp) {
  const int nb = Mp / DB;
  return scratch + (size_t)(nb * (nb + 1) / 2) * DF_FLAG_STRIDE;
}

void potrf_lower(double* A, double* Linv, int64_t ld, int Mp, int* info, int info_base, int* scratch, hipStream_t st,
                 const double* rhs, double* sol, bool caller_managed) {
  const int nb = Mp / DB;
  const int ntile = nb * (nb + 1) / 2;
  const int nitem = ntile + (rhs ? 1 : 0);
  if (!caller_managed) zero_ints(scratch, (int)potrf_flag_ints(Mp), st);
  if (Linv) fill_zero(Linv, (size_t)Mp * ld, st);  // level 0 of tri_inverse(): diagonal-block inverses (written by the
                                                   // diagonal tile owners inside the launch), zero elsewhere
  potrf_dataflow_kernel<<<nitem < DF_MAX_WG ? nitem : DF_MAX_WG, 256, 0, st>>>(
      A, ld, nb, scratch, reinterpret_cast<double*>(scratch + potrf_flag_ints(Mp)), info, info_base, rhs, sol, Linv);
  if (!caller_managed) potrf_timeout_kernel<<<1, 64, 0, st>>>(scratch + ntile * DF_FLAG_STRIDE, info);
}

void tri_inverse(const double* L, double* Linv, double* tmp, int64_t ld, int Mp, hipStream_t st) {
  for (int s = DB; s < Mp; s *= 2) {
    const int np = Mp / (2 * s);
    const int rem = Mp - np * 2 * s;
    const int64_t pstride = (int64_t)2 * s * (ld + 1);
    auto level = [&](int64_t o, int batch, int n2) {
      // T = L21 * Inv11   (n2 x s) = (n2 x s)(s x s), Inv11 lower-triangular -> k >= column tile start
      GemmDesc a;
      a.A = L + o * (ld + 1) + (int64_t)s * ld; a.lda = ld; a.sA = pstride;
      a.B = Linv + o * (ld + 1); a.ldb = ld; a.sB = pstride;
      a.C = tmp + o * (ld + 1) + (int64_t)s * ld; a.ldc = ld; a.sC = pstride;
      a.m = n2; a.n = s; a.k = s; a.batch = batch; a.klo_mask = 2;
      gemm(a, st);
      // Inv21 = -Inv22 * T   (n2 x s) = (n2 x n2)(n2 x s), Inv22 lower-triangular -> k <= row tile end
      GemmDesc b;
      b.A = Linv + o * (ld + 1) + (int64_t)s * (ld + 1); b.lda = ld; b.sA = pstride;
      b.B = tmp + o * (ld + 1) + (int64_t)s * ld; b.ldb = ld; b.sB = pstride;
      b.C = Linv + o * (ld + 1) + (int64_t)s * ld; b.ldc = ld; b.sC = pstride;
      b.m = n2; b.n = s; b.k = n2; b.batch = batch; b.alpha = -1.0; b.khi_mask = 1;
      gemm(b, st);
    };
    if (np > 0) level(0, np, s);
    if (rem > s) level((int64_t)np * 2 * s, 1, rem - s);
  }
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemv_rows_kernel(const double* __restrict__ A, int64_t ld, int Mp,
                                                        const double* __restrict__ x, double* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= Mp) return;
  double s = 0.0;
  for (int j = lane; j < Mp; j += 64) s = fma(A[(int64_t)row * ld + j], x[j], s);
  s = wave_sum(s);
  if (lane == 0) y[row] = s;
}
__global__ __launch_bounds__(1024) void gemv_cols_kernel(const double* __restrict__ A, int64_t ld, int Mp,
                                                         const double* __restrict__ x, double* __restrict__ y) {
  // y[i] = sum_j A[j][i] x[j]; a block owns 64 columns, its 16 waves split the rows (4 waves took 80 us at
  // Mp = 1024: 256 dependent steps per wave on 16 CUs); partial sums are added in wave order
  __shared__ double part[16][64];
  const int lane = threadIdx.x & 63, col = blockIdx.x * 64 + lane;
  const int w = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int j = w;
  for (; j + 48 < Mp; j += 64) {
    s0 = fma(A[(int64_t)j * ld + col], x[j], s0);
    s1 = fma(A[(int64_t)(j + 16) * ld + col], x[j + 16], s1);
    s2 = fma(A[(int64_t)(j + 32) * ld + col], x[j + 32], s2);
    s3 = fma(A[(int64_t)(j + 48) * ld + col], x[j + 48], s3);
  }
  for (; j < Mp; j += 16) s0 = fma(A[(int64_t)j * ld + col], x[j], s0);
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][lane];
    y[col] = t;
  }
}
void gemv(const double* A, int64_t ld, int Mp, bool trans, const double* x, double* y, hipStream_t st) {
  if (!trans) gemv_rows_kernel<<<Mp / 4, 256, 0, st>>>(A, ld, Mp, x, y);
  else gemv_cols_kernel<<<Mp / 64, 1024, 0, st>>>(A, ld, Mp, x, y);
}

__global__ void pad_copy_kernel(const double* __restrict__ src, int64_t lds, int rs, int cs, double* __restrict__ dst,
                                int64_t ldd, int rows, int cols, double diag_pad) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols), c = (int)(e - (int64_t)r * cols);
    double v = 0.0;
    if (r < rs && c < cs) v = src[(int64_t)r * lds + c];
    else if (r == c) v = diag_pad;
    dst[(int64_t)r * ldd + c] = v;
  }
}
void pad_copy(const double* src, int64_t lds, int rs, int cs, double* dst, int64_t ldd, int rows, int cols,
              double diag_pad, hipStream_t st) {
  const int64_t total = (int64_t)rows * cols;
  const int g = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  pad_copy_kernel<<<g, 256, 0, st>>>(src, lds, rs, cs, dst, ldd, rows, cols, diag_pad);
}
__global__ void crop_copy_kernel(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd, int rs, int cs) {
  const int64_t total = (int64_t)rs * cs;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / cs), c = (int)(e - (int64_t)r * cs);
    dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c];
  }
}
void crop_copy(const double* src, int64_t lds, double* dst, int64_t ldd, int rs, int cs, hipStream_t st) {
  const int64_t total = (int64_t)rs * cs;
  if (total <= 0) return;
  const int g = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  crop_copy_kernel<<<g, 256, 0, st>>>(src, lds, dst, ldd, rs, cs);
}
__global__ void fill_zero_kernel(double* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0;
}
void fill_zero(double* p, size_t n, hipStream_t st) {
  if (n == 0) return;
  const int g = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  fill_zero_kernel<<<g, 256, 0, st>>>(p, n);
}
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* A, int64_t ld, int Mp) {
  __shared__ double t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) t[ty + 8 * k][tx] = A[(int64_t)(bi * 32 + ty + 8 * k) * ld + bj * 32 + tx];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = bj * 32 + ty + 8 * k, c = bi * 32 + tx;  // upper-triangle destination
    if (c > r) A[(int64_t)r * ld + c] = t[tx][ty + 8 * k];
  }
}
void mirror_lower(double* A, int64_t ld, int Mp, hipStream_t st) {
  mirror_lower_kernel<<<dim3(Mp / 32, Mp / 32), 256, 0, st>>>(A, ld, Mp);
}

// ---------------------------------------------------------------------------------------------
// public M x M entry points (sgp.h)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logdiag_kernel(const double* __restrict__ L, int64_t ld, int M, double* out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < M; i += 256) s += log(L[(int64_t)i * ld + i]);
  s = block_sum256(s, red);
  if (threadIdx.x == 0) *out = s;
}

}  // namespace sgp

using namespace sgp;

extern "C" size_t sgp_chol_workspace_bytes(int M) {
  if (M <= 0 || M > SGP_MAX_INDUCING) return 0;
  const size_t Mp = padded_m(M);
  Carver c(nullptr);
  c.take<double>(Mp * Mp);
  c.take<double>(Mp * Mp);
  c.take<int>(potrf_scratch_ints((int)Mp));
  return c.used();
}

extern "C" int sgp_chol_lower(double* A, int64_t lda, int M, int* info, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!A || !info || M <= 0 || lda < M) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (!ws || ws_bytes < sgp_chol_workspace_bytes(M)) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int Mp = padded_m(M);
  Carver c(ws);
  double* Ap = c.take<double>((size_t)Mp * Mp);
  double* Li = c.take<double>((size_t)Mp * Mp);
  int* flags = c.take<int>(potrf_scratch_ints(Mp));
  zero_ints(info, 1, st);
  pad_copy(A, lda, M, M, Ap, Mp, Mp, Mp, 1.0, st);
  potrf_lower(Ap, Li, Mp, Mp, info, 0, flags, st);
  crop_copy(Ap, Mp, A, lda, M, M, st);
  return check_launch();
}

extern "C" size_t sgp_trsm_workspace_bytes(int M, int k) {
  if (M <= 0 || k <= 0 || M > SGP_MAX_INDUCING) return 0;
  const size_t Mp = padded_m(M), kp = round_up(k, 64);
  Carver c(nullptr);
  c.take<double>(Mp * Mp);  // L padded
  c.take<double>(Mp * Mp);  // Linv
  c.take<double>(Mp * Mp);  // tmp
  c.take<double>(Mp * kp);  // B padded
  c.take<double>(Mp * kp);  // result
  return c.used();
}

// B <- L^-1 B or L^-T B through the explicit blocked inverse (same machinery as the bound's tail).
extern "C" int sgp_trsm_lower(const double* L, int64_t ldl, double* B, int64_t ldb, int trans, int M, int k,
                              void* ws, size_t ws_bytes, sgp_stream_t stream) {
  if (!L || !B || M <= 0 || k <= 0 || ldl < M || ldb < k) return SGP_ERR_ARG;
  if (M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (!ws || ws_bytes < sgp_trsm_workspace_bytes(M, k)) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int Mp = padded_m(M), kp = round_up(k, 64);
  Carver c(ws);
  double* Lp = c.take<double>((size_t)Mp * Mp);
  double* Li = c.take<double>((size_t)Mp * Mp);
  double* tmp = c.take<double>((size_t)Mp * Mp);
  double* Bp = c.take<double>((size_t)Mp * kp);
  double* R = c.take<double>((size_t)Mp * kp);
  pad_copy(L, ldl, M, M, Lp, Mp, Mp, Mp, 1.0, st);
  fill_zero(Li, (size_t)Mp * Mp, st);
  tri_diag_inverse(Lp, Li, Mp, Mp, st);
  tri_inverse(Lp, Li, tmp, Mp, Mp, st);
  pad_copy(B, ldb, M, k, Bp, kp, Mp, kp, 0.0, st);
  GemmDesc g;
  g.A = Li; g.lda = Mp; g.ta = (trans != 0);
  g.B = Bp; g.ldb = kp;
  g.C = R; g.ldc = kp;
  g.m = Mp; g.n = kp; g.k = Mp;
  if (!trans) g.khi_mask = 1; else g.klo_mask = 1;
  gemm(g, st);
  crop_copy(R, kp, B, ldb, M, k, st);
  return check_launch();
}

extern "C" int sgp_logdiag_sum(const double* L, int64_t ldl, int M, double* out, sgp_stream_t stream) {
  if (!L || !out || M <= 0 || ldl < M) return SGP_ERR_ARG;
  logdiag_kernel<<<1, 256, 0, (hipStream_t)stream>>>(L, ldl, M, out);
  return check_launch();
}
