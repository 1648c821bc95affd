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
#include <type_traits>
#include "sgp_dense.hpp"
#include "sgp_ctx.hpp"
#include "sgp_potrf.hpp"
#include "sgp_potrf_chain.hpp"
#include <cstdlib>

namespace sgp {

constexpr int GT = 64;    // GEMM tile edge

struct GemmP {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc, sA, sB, sC;
  int m, n, k;
  double alpha, beta;
  int klo_mask, khi_mask, lower_only;
  int batch;
  int64_t s2A, s2B, s2C;
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
// T = tile edge (64, or 32: the same image with its first 32 "columns" in use -- the rotation still runs modulo 64).
template <bool KCONTIG, int T>
__device__ __forceinline__ void tile_fetch(const double* P, int64_t ld, int r0, int k0, int t, d2 (&v)[2]) {
  if constexpr (KCONTIG) {
    const int row = t >> 2, kq = (t & 3) * 4;
    if (T == 64 || row < T) {
      const double* s = P + (int64_t)(r0 + row) * ld + k0 + kq;
      v[0] = *reinterpret_cast<const d2*>(s);
      v[1] = *reinterpret_cast<const d2*>(s + 2);
    }
  } else {
    const int kk = t >> 4, c2 = (t & 15) * 2;
    const double* s = P + (int64_t)(k0 + kk) * ld + r0 + c2;
    v[0] = *reinterpret_cast<const d2*>(s);
    if constexpr (T == 64) v[1] = *reinterpret_cast<const d2*>(s + 32);
  }
}
template <bool KCONTIG, int T>
__device__ __forceinline__ void tile_stash(double (*S)[GT], int t, const d2 (&v)[2]) {
  if constexpr (KCONTIG) {
    const int row = t >> 2, kq = (t & 3) * 4;
    if (T == 64 || row < T) {
#pragma unroll
      for (int e = 0; e < 4; ++e) S[kq + e][(row + tile_rot(kq + e)) & 63] = v[e >> 1][e & 1];
    }
  } else {
    const int kk = t >> 4, c2 = (t & 15) * 2, rot = tile_rot(kk);
    *reinterpret_cast<d2*>(&S[kk][(c2 + rot) & 63]) = v[0];
    if constexpr (T == 64) *reinterpret_cast<d2*>(&S[kk][(c2 + 32 + rot) & 63]) = v[1];
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

//
// T = 32 (round 4): the same kernel on 32 x 32 output tiles, one MFMA tile per wave -- for the M <= 512 products of the tail, whose
// 64 x 64 tiling fills a quarter of the chip (512^3: 64 workgroups, 21 us, all of it MFMA time of the 64 CUs in use; C3 runs sixteen
// such launches on its critical path).  Every output element sees the same k-order and the same group split: bit-identical results.
template <bool TA, bool TB, int T = 64>
__global__ __launch_bounds__(512) void gemm64_kernel(GemmP p) {
  __shared__ GemmShared sh;
  constexpr int NT = T / 32;  // 16 x 16 MFMA tiles per wave and dimension
  const int bj = blockIdx.x, bi = blockIdx.y;
  if (p.lower_only && bj > bi) return;
  const int64_t bo = (int)blockIdx.z / p.batch, bz = (int)blockIdx.z - bo * p.batch;  // (outer, inner) batch index
  const double* A = p.A + bz * p.sA + bo * p.s2A;
  const double* B = p.B + bz * p.sB + bo * p.s2B;
  double* C = p.C + bz * p.sC + bo * p.s2C;

  int klo = 0, khi = p.k;
  if (p.klo_mask & 1) klo = max(klo, bi * T) / GK * GK;  // (a 32-row tile may start inside a 64-block of the triangular operand:
  if (p.klo_mask & 2) klo = max(klo, bj * T) / GK * GK;  //  32 is a multiple of the chunk depth, the division is a no-op kept for clarity)
  if (p.khi_mask & 1) khi = min(khi, (bi + 1) * T);
  if (p.khi_mask & 2) khi = min(khi, (bj + 1) * T);

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int r0 = bi * T, c0 = bj * T;
  // column of this lane's operands in k-row ks * 4 + l4, before the 4 ks part of the rotation
  const int ca = wi * (T / 2) + l15 + 16 * (l4 & 1), cb = wj * (T / 2) + l15 + 16 * (l4 & 1);

  d4 acc[NT][NT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int v = 0; v < NT; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};

  if (klo < khi) {
    // This group's k-chunks are klo + (2 q + grp) GK, q = 0, 1, ... (the groups alternate: both run the same number of barriers).
    // A step takes NS of them: one at T = 64; TWO at T = 32 (round 5), whose 32-column operand image leaves the other half of the
    // 64-column LDS rows free -- half as many barriers for the latency-bound M <= 512 products (13 - 17 us each at 512^3, a fifth of
    // it MFMA time: profiles/r05_v3pre_c3_timeline.txt).  Four register sets keep four steps of global loads in flight (two before).
    // The chunks of a group are still accumulated in increasing order: the same bits as before, and as the other tiling.
    constexpr int NS = T == 32 ? 2 : 1;
    const int nchunk = (khi - klo + GK - 1) / GK, niter = ((nchunk + 1) / 2 + NS - 1) / NS;
    d2 ra0[2], rb0[2], ra1[2], rb1[2], ra2[2], rb2[2], ra3[2], rb3[2];
    auto chunk_of = [&](int j, int sub) { return 2 * (NS * j + sub) + grp; };
    auto fetch = [&](int j, d2 (&ra)[2], d2 (&rb)[2]) {
      if constexpr (NS == 1) {
        const int c = chunk_of(j, 0);
        if (c < nchunk) {
          tile_fetch<!TA, T>(A, p.lda, r0, klo + c * GK, tid, ra);
          tile_fetch<TB, T>(B, p.ldb, c0, klo + c * GK, tid, rb);
        }
      } else {
        // K-contiguous operand: thread = (row, k-quad) over 64 "rows" -- rows 32 .. 63 are rows 0 .. 31 of the step's second chunk;
        // MN-contiguous operand: thread = (k, column pair), its second 16 bytes are the second chunk's.  Either way the registers and
        // the LDS image (column + 32) are those of a 64-wide tile: tile_stash<.., 64> stores both chunks.
        const int c0k = chunk_of(j, 0), c1k = chunk_of(j, 1);
        auto one = [&](auto kc, const double* P, int64_t ld, int base, d2 (&v)[2]) {
          if constexpr (decltype(kc)::value) {
            const int row = tid >> 2, kq = (tid & 3) * 4, c = row < 32 ? c0k : c1k;
            if (c < nchunk) {
              const double* sp = P + (int64_t)(base + (row & 31)) * ld + klo + c * GK + kq;
              v[0] = *reinterpret_cast<const d2*>(sp);
              v[1] = *reinterpret_cast<const d2*>(sp + 2);
            }
          } else {
            const int kk = tid >> 4, c2 = (tid & 15) * 2;
            if (c0k < nchunk) v[0] = *reinterpret_cast<const d2*>(P + (int64_t)(klo + c0k * GK + kk) * ld + base + c2);
            if (c1k < nchunk) v[1] = *reinterpret_cast<const d2*>(P + (int64_t)(klo + c1k * GK + kk) * ld + base + c2);
          }
        };
        one(std::integral_constant<bool, !TA>{}, A, p.lda, r0, ra);
        one(std::integral_constant<bool, TB>{}, B, p.ldb, c0, rb);
      }
    };
    auto stash = [&](int j, int stage, const d2 (&ra)[2], const d2 (&rb)[2]) {
      if (chunk_of(j, 0) < nchunk) {  // (a missing second chunk leaves stale columns behind that nobody reads)
        tile_stash<!TA, NS == 2 ? 64 : T>(sh.As[grp][stage], tid, ra);
        tile_stash<TB, NS == 2 ? 64 : T>(sh.Bs[grp][stage], tid, rb);
      }
    };
    // r holds step j + 1 on entry and step j + 5 on exit
    auto body = [&](int j, int stage, d2 (&ra)[2], d2 (&rb)[2]) {
      const bool live = chunk_of(j, 0) < nchunk, live2 = NS == 2 && chunk_of(j, 1) < nchunk;
      double a0[NS * GK / 4], a1[GK / 4], b0[NS * GK / 4], b1[GK / 4];
      if (live) {  // operands of the whole step first: the MFMAs start as soon as the barrier falls
        const double (*As)[GT] = sh.As[grp][stage];
        const double (*Bs)[GT] = sh.Bs[grp][stage];
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
          const int kr = ks * 4 + l4;
          a0[ks] = As[kr][(ca + 4 * ks) & 63];
          b0[ks] = Bs[kr][(cb + 4 * ks) & 63];
          if constexpr (NT == 2) {
            a1[ks] = As[kr][(ca + 16 + 4 * ks) & 63];
            b1[ks] = Bs[kr][(cb + 16 + 4 * ks) & 63];
          }
          if constexpr (NS == 2) {
            a0[GK / 4 + ks] = As[kr][(ca + 32 + 4 * ks) & 63];
            b0[GK / 4 + ks] = Bs[kr][(cb + 32 + 4 * ks) & 63];
          }
        }
      }
      auto mac = [&](int ks) {
        acc[0][0] = mfma16(a0[ks], b0[ks], acc[0][0]);
        if constexpr (NT == 2) {
          acc[0][1] = mfma16(a0[ks], b1[ks], acc[0][1]);
          acc[1][0] = mfma16(a1[ks], b0[ks], acc[1][0]);
          acc[1][1] = mfma16(a1[ks], b1[ks], acc[1][1]);
        }
      };
      if (live) mac(0);
      stash(j + 1, stage ^ 1, ra, rb);  // LDS stores and global loads of later steps issue under the MFMAs
      if (live) mac(1);
      fetch(j + 5, ra, rb);
      if (live) {
        mac(2);
        mac(3);
      }
      if constexpr (NS == 2) {
        if (live2) {
          mac(4);
          mac(5);
          mac(6);
          mac(7);
        }
      }
      __syncthreads();
    };
    fetch(0, ra0, rb0);
    fetch(1, ra1, rb1);
    fetch(2, ra2, rb2);
    fetch(3, ra3, rb3);
    stash(0, 0, ra0, rb0);
    fetch(4, ra0, rb0);
    __syncthreads();
    for (int j = 0; j < niter; j += 4) {  // step c lives in register set c mod 4, LDS stage c mod 2
      body(j, 0, ra1, rb1);
      if (j + 1 < niter) body(j + 1, 1, ra2, rb2);
      if (j + 2 < niter) body(j + 2, 0, ra3, rb3);
      if (j + 3 < niter) body(j + 3, 1, ra0, rb0);
    }
    // group 1's partial tile -> LDS -> added by group 0
    double* red = &sh.As[0][0][0][0];
    if (grp == 1) {
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int v = 0; v < NT; ++v)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((u * 2 + v) * 4 + r) * 256 + tid] = acc[u][v][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int v = 0; v < NT; ++v)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u][v][r] += red[((u * 2 + v) * 4 + r) * 256 + tid];
    }
  }
  if (grp != 0) return;
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int v = 0; v < NT; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + wi * (T / 2) + u * 16 + l4 + 4 * r;
        const int col = c0 + wj * (T / 2) + v * 16 + l15;
        double* dst = C + (int64_t)row * p.ldc + col;
        const double val = p.alpha * acc[u][v][r];
        const double res = (p.beta == 0.0) ? val : fma(p.beta, *dst, val);
        *dst = res;
        if (p.lower_only == 2 && bi != bj) C[(int64_t)col * p.ldc + row] = res;  // the mirrored tile of a symmetric product
      }
}

// C = alpha * sum over S k-slices (+ beta C): for products with a few output tiles and a long contraction (the
// M x M x B products of the SVGP reverse pass: 16 tiles, k = 4096) -- S workgroups per tile write partial tiles, a
// second launch adds them in slice order.  scratch: S * m * n doubles.  No k-range masks, no batch.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const double* __restrict__ part, int S, int m, int n, double* __restrict__ C,
                                                            int64_t ldc, double alpha, double beta, int64_t s2C) {
  const int64_t total = (int64_t)m * n;
  part += (int64_t)blockIdx.y * S * total;  // outer batch (GemmDesc::batch2)
  C += (int64_t)blockIdx.y * s2C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e - (int64_t)r * n);
    double s = 0.0;
    for (int z = 0; z < S; ++z) s += part[(int64_t)z * total + e];
    double* dst = C + (int64_t)r * ldc + c;
    *dst = (beta == 0.0) ? alpha * s : fma(beta, *dst, alpha * s);
  }
}
void gemm_splitk(const GemmDesc& g, int S, double* scratch, hipStream_t st) {
  if (g.m <= 0 || g.n <= 0 || S <= 1 || g.k % (S * GK) != 0 || g.batch != 1) {
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
  p.s2C = (int64_t)S * g.m * g.n;  // outer batch: every problem has its own S partial tiles (s2A / s2B as given)
  p.alpha = 1.0;
  p.beta = 0.0;
  gemm(p, st);
  const int64_t total = (int64_t)g.m * g.n;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  splitk_reduce_kernel<<<dim3(blocks, g.batch2), 256, 0, st>>>(scratch, S, g.m, g.n, g.C, g.ldc, g.alpha, g.beta, g.s2C);
}

// out[k][c] = in[c][k] (n x n, ld n, n a multiple of 32)
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, int n, double* __restrict__ out) {
  __shared__ double t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) t[ty + 8 * k][tx] = in[(size_t)(by + ty + 8 * k) * n + bx + tx];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) out[(size_t)(bx + ty + 8 * k) * n + by + tx] = t[tx][ty + 8 * k];
}
void transpose_square(const double* in, int n, double* out, hipStream_t st) {
  transpose_kernel<<<dim3(n / 32, n / 32), 256, 0, st>>>(in, n, out);
}

// ---- tall products: C[m x n] = alpha A[m x k] B[k x n], m >> n, k (round 4) -------------------------------------------------------
// The 64 x 64 kernel above is built for the latency of M x M products (eight waves share one tile, one LDS operand read per MFMA):
// on T = K'_fu L^-T (10^6 x 1024 x 1024, clipped to the triangle) it ran at 34 TFLOP/s.  This is the main loop of pass 2
// (kbar_contract_kernel, sgp_suffstats_bwd.hip) with a plain store behind it: a workgroup of four waves owns a 128 x 128 tile, every
// wave 64 x 64 of it in 128 accumulator registers (one operand read per two MFMAs); 16-deep slabs of A (k-contiguous rows, from HBM:
// two register stages) and of B (L2 resident: one) go global -> registers -> LDS, double buffered, two workgroups per CU.
// tri: 0 = full k range, 1 = B upper triangular (column block cb needs k < its end).
constexpr int TT = 128, TBK = 16, TALD = TBK + 2, TBROW = TT + 16;
constexpr int TA_DBL = TT * TALD, TB_DBL = TBK * TBROW;
__global__ __launch_bounds__(256, 2) void gemm_tall_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                                           int64_t ldb, double* __restrict__ C, int64_t ldc, int nrb, int ncb, int k,
                                                           int tri, double alpha, int rb_per_wg) {
  __shared__ double smem[2 * (TA_DBL + TB_DBL)];
  double (*At)[TT][TALD] = reinterpret_cast<double (*)[TT][TALD]>(smem);
  double (*Bt)[TBK][TBROW] = reinterpret_cast<double (*)[TBK][TBROW]>(smem + 2 * TA_DBL);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int prow = tid >> 4, pcol = (tid & 15) * 2;
  // per-thread element offsets (32-bit: the host checks 128 ld < 2^31) relative to wave-uniform bases: one VGPR per address
  int arow[4], acol[4], aoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i;
    arow[i] = q >> 3;
    acol[i] = (q & 7) * 2;
    aoff[i] = arow[i] * (int)lda + acol[i];
  }
  const int poff = prow * (int)ldb + pcol;
  const int coff = (wi * 64 + l4) * (int)ldc + wj * 64 + l15;

  // a workgroup walks its row blocks and, inside one, every column block (widest k range first): the same work per workgroup whatever
  // the triangle, the A rows of a block come back from this XCD's own L2 for the later column blocks, and a finished tile's stores
  // drain under the next tile's main loop (one tile per workgroup: 23.6 ms at C5, the short tiles all start-up)
  const int rb_end = min(nrb, ((int)blockIdx.x + 1) * rb_per_wg);
#pragma unroll 1
  for (int rb = (int)blockIdx.x * rb_per_wg; rb < rb_end; ++rb) {
    const double* Ablk = A + (int64_t)rb * TT * lda;
#pragma unroll 1
    for (int cbi = 0; cbi < ncb; ++cbi) {
      const int cb = ncb - 1 - cbi;
      const int m0 = cb * TT;
      const int khi = tri == 1 ? min(k, m0 + TT) : k;
      const int nchunks = khi / TBK;  // a multiple of 8: k and the tile edge are multiples of 128

      d2 avA[4], avB[4], pv[4];
      auto fetchA = [&](int ch, d2 (&av)[4]) {
        const double* ab = Ablk + (int64_t)ch * TBK;
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const d2*>(ab + aoff[i]);
      };
      auto fetchP = [&](int ch) {
        const double* pb = B + (int64_t)ch * TBK * ldb + m0;
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = *reinterpret_cast<const d2*>(pb + poff + 32 * e);
      };
      auto stashA = [&](int buf, const d2 (&av)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<d2*>(&At[buf][arow[i]][acol[i]]) = av[i];
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
        for (int ks = 0; ks < TBK / 4; ++ks) {
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
      if (nchunks > 0) {
        fetchA(0, avA);
        fetchP(0);
        stashA(0, avA);
        stashP(0);
        fetchA(1, avB);
        __syncthreads();
#pragma unroll 1
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
      }
      double* Cblk = C + (int64_t)rb * TT * ldc + m0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            Cblk[coff + (u * 16 + 4 * r) * (int)ldc + v * 16] = alpha * acc[u][v][r];
    }
  }
}
// SGP_GEMM_TALL=0 switches the kernel off (A/B); rows from which it is used
static int gemm_tall_min_rows() {
  static const int v = getenv("SGP_GEMM_TALL") ? atoi(getenv("SGP_GEMM_TALL")) : 16384;  // >= 128 workgroups of >= 1 row block
  return v;
}
static bool gemm_tall(const GemmDesc& g, hipStream_t st) {
  const int minr = gemm_tall_min_rows();
  if (minr <= 0 || g.ta || g.tb || g.batch != 1 || g.batch2 != 1 || g.beta != 0.0 || g.lower_only) return false;
  if (g.m < minr || g.m % TT || g.n % TT || g.k % TT || g.k <= 0) return false;
  if (g.lda >= (1 << 23) || g.ldb >= (1 << 23) || g.ldc >= (1 << 23)) return false;  // 32-bit offsets inside a 128-row block
  int tri = 0;
  if (g.klo_mask == 0 && g.khi_mask == 2 && g.n == g.k) tri = 1;
  else if (g.klo_mask != 0 || g.khi_mask != 0) return false;
  const int nrb = g.m / TT, ncb = g.n / TT;
  // row blocks per workgroup: ~4 rounds of the 512 resident workgroups when there are that many row blocks, one block each otherwise
  static const int rounds = getenv("SGP_GEMM_TALL_ROUNDS") ? atoi(getenv("SGP_GEMM_TALL_ROUNDS")) : 4;  // tuning knob
  int per = (nrb + 512 * rounds - 1) / (512 * rounds);
  if (per < 1) per = 1;
  const int grid = (nrb + per - 1) / per;
  gemm_tall_kernel<<<grid, 256, 0, st>>>(g.A, g.lda, g.B, g.ldb, g.C, g.ldc, nrb, ncb, g.k, tri, g.alpha, per);
  return true;
}

// 64 x 64 tilings with at most this many tiles x 2 take the 32 x 32 tiles: the CU count (SGP_GEMM_SMALL_TILES=0 switches them off: A/B)
static int gemm_small_tile_threshold() {
  static const int v = getenv("SGP_GEMM_SMALL_TILES") ? atoi(getenv("SGP_GEMM_SMALL_TILES")) : 256;
  return v;
}
void gemm(const GemmDesc& g, hipStream_t st) {
  if (g.m <= 0 || g.n <= 0 || g.batch <= 0) return;
  GemmP p{g.A, g.B, g.C, g.lda, g.ldb, g.ldc, g.sA, g.sB, g.sC, g.m, g.n, g.k,
          g.alpha, g.beta, g.klo_mask, g.khi_mask, g.lower_only ? (g.mirror ? 2 : 1) : 0, g.batch, g.s2A, g.s2B, g.s2C};
  if (g.batch2 <= 0) return;
  if (gemm_tall(g, st)) return;
  // fewer 64 x 64 tiles than half the CUs: four times as many 32 x 32 tiles instead (bit-identical results)
  const int64_t tiles64 = (int64_t)(g.n / GT) * (g.m / GT) * g.batch * g.batch2;
  if (tiles64 * 2 <= gemm_small_tile_threshold()) {
    dim3 grid(g.n / 32, g.m / 32, g.batch * g.batch2);
    if (!g.ta && !g.tb) gemm64_kernel<false, false, 32><<<grid, 512, 0, st>>>(p);
    else if (!g.ta && g.tb) gemm64_kernel<false, true, 32><<<grid, 512, 0, st>>>(p);
    else if (g.ta && !g.tb) gemm64_kernel<true, false, 32><<<grid, 512, 0, st>>>(p);
    else gemm64_kernel<true, true, 32><<<grid, 512, 0, st>>>(p);
    return;
  }
  dim3 grid(g.n / GT, g.m / GT, g.batch * g.batch2);
  if (!g.ta && !g.tb) gemm64_kernel<false, false><<<grid, 512, 0, st>>>(p);
  else if (!g.ta && g.tb) gemm64_kernel<false, true><<<grid, 512, 0, st>>>(p);
  else if (g.ta && !g.tb) gemm64_kernel<true, false><<<grid, 512, 0, st>>>(p);
  else gemm64_kernel<true, true><<<grid, 512, 0, st>>>(p);
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

// ticketed: the items are drawn from the scratch's claim counter (sgp_potrf_items.hpp: CF_TICKET, zeroed with the flags) instead of dealt
__global__ __launch_bounds__(256) void potrf_dataflow_kernel(double* A, int64_t ld, int nb, int* ready, double* dinv_g, int* info,
                                                              int info_base, const double* rhs, double* sol, double* Linv, int ticketed) {
  __shared__ DfShared sh;
  potrf_dataflow_body(A, ld, nb, ready, dinv_g, info, info_base, rhs, sol, Linv, sh, blockIdx.x, gridDim.x,
                      ticketed ? ready + (size_t)ch_flag_base(CF_TICKET, nb) * DF_FLAG_STRIDE : nullptr);
}

// Round 5: the chain-workgroup factorization (sgp_potrf_chain.hpp).  Workgroup 0 (eight waves) is the chain workgroup; the others
// keep four waves (the upper four leave at once: the hardware barrier counts live waves only).
union ChainSharedAll {
  ChShared ch;
  DfShared df;
  __device__ ChainSharedAll() {}
};
__global__ __launch_bounds__(CH_THREADS) void potrf_chain_kernel(double* A, int64_t ld, int nb, int* scratch, int* info, int info_base,
                                                                const double* rhs, double* sol, double* Linv, int mode) {
  __shared__ ChainSharedAll sh;
  const ChScratch sc = ch_scratch(scratch, nb);
  if (blockIdx.x == 0) {
    if (threadIdx.x < 256) chain_d_role(A, ld, nb, sc, info, info_base, sh.ch);
    else chain_s_role(A, ld, nb, sc, sh.ch, mode);
    return;
  }
  if (threadIdx.x >= 256) return;
  chain_outside(A, ld, nb, sc, rhs, sol, Linv, sh.df, (int)blockIdx.x - 1, (int)gridDim.x - 1, mode);
}
#ifdef SGP_CH_TRACE
// (trace build only, tools/potrf_trace_check.py) the wait / raise log of the last chain-workgroup launch: counts[CH_TRACE_WG], then the entries
extern "C" __attribute__((visibility("default"))) int sgp_debug_potrf_trace(int* host_counts, int* host_entries, int clear) {
  int st = 0;
  if (host_counts) st |= (int)hipMemcpyFromSymbol(host_counts, HIP_SYMBOL(g_ch_trace_n), sizeof(int) * CH_TRACE_WG);
  if (host_entries) st |= (int)hipMemcpyFromSymbol(host_entries, HIP_SYMBOL(g_ch_trace), sizeof(int) * CH_TRACE_WG * CH_TRACE_LEN);
  if (clear) {
    static int zeros[CH_TRACE_WG];
    st |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ch_trace_n), zeros, sizeof(zeros));
  }
  return st;
}
#endif

// S factorizations side by side: blockIdx.y selects the matrix, its flags / block inverses and its status word
__global__ __launch_bounds__(256) void potrf_dataflow_batch_kernel(double* A, int64_t ld, int nb, int* scratch, int64_t scratch_ints,
                                                                    int64_t flag_ints, int* info, double* Linv, int64_t stride, int ticketed) {
  __shared__ DfShared sh;
  const int64_t s = blockIdx.y;
  int* ready = scratch + s * scratch_ints;
  potrf_dataflow_body(A + s * stride, ld, nb, ready, reinterpret_cast<double*>(ready + flag_ints), info + s, 0, nullptr, nullptr,
                      Linv ? Linv + s * stride : nullptr, sh, blockIdx.x, gridDim.x,
                      ticketed ? ready + (size_t)ch_flag_base(CF_TICKET, nb) * DF_FLAG_STRIDE : nullptr);
}
__global__ void potrf_timeout_batch_kernel(const int* scratch, int64_t scratch_ints, int abort_off, int* info, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S && scratch[s * scratch_ints + abort_off] != 0) info[s] = POTRF_TIMEOUT;
}

#ifdef SGP_POTRF_STAMPS
extern "C" __attribute__((visibility("default"))) int sgp_debug_potrf_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_potrf_stamps), sizeof(unsigned long long) * 64 * 16);
}
#endif

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
  // one cache line per tile flag + the abort flag + (chain-workgroup kernel) per column: two prep flags and four panel counters
  return ch_flag_slots((int)nb) * DF_FLAG_STRIDE;
}
size_t potrf_scratch_ints(int Mp) {
  const size_t nb = Mp / DB;
  // + the 16 x 16 block inverses of every diagonal tile + (chain-workgroup kernel) the prep items' partial sums US / UD (doubles)
  // + the transposed blocks of L^-1 (chain-workgroup kernel with an inverse wanted)
  return potrf_flag_ints(Mp) + ch_scratch_doubles((int)nb) * 2;
}

const int* potrf_abort_flag(const int* scratch, int Mp) {
  const int nb = Mp / DB;
  return scratch + (size_t)(nb * (nb + 1) / 2) * DF_FLAG_STRIDE;
}

// Workgroups of the dataflow launch: one per CU at most (112 KB of LDS each), so that every workgroup of the launch is
// resident and the smallest unfinished item always has an owner that can run (forward progress, see sgp_potrf.hpp).  The CU
// count is the device's, not a constant: on a partitioned (CPX / DPX) device a launch of 256 could not be co-resident.
// A caller that enqueues on a CU-masked stream (hipExtStreamCreateWithCUMask: CollapsedBound reserves a few CUs for the K_uu
// chain beside pass 1 on small shards) tells this host thread how many CUs its launches can occupy: sgp_set_cu_budget(n),
// 0 = the whole device.  Per host thread, because the side chain is enqueued by a helper thread.
// (round 4: the budget is an option of the call's context, SGP_OPT_CU_BUDGET; the deprecated per-thread setter overrides it)
static thread_local int t_cu_budget = 0;
void set_cu_budget(int n) { t_cu_budget = n > 0 ? n : 0; }
static int cu_budget() { return t_cu_budget > 0 ? t_cu_budget : cur_ctx().cu_budget; }
static int df_max_workgroups() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = DF_MAX_WG;
    n = cus < DF_MAX_WG ? cus : DF_MAX_WG;
  }
  return (cu_budget() > 0 && cu_budget() < n) ? cu_budget() : n;
}

// CUs a launch of the calling host thread can occupy: the device's count, or the caller's budget for a CU-masked stream.
// Used by every kernel whose workgroups wait for each other (all of them must be resident at once).
int available_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = DF_MAX_WG;  // no device visible (CPU-only build check): the shape-only answer
    cus = n;
  }
  return (cu_budget() > 0 && cu_budget() < cus) ? cu_budget() : cus;
}

bool potrf_lower(double* A, double* Linv, int64_t ld, int Mp, int* info, int info_base, int* scratch, hipStream_t st,
                 const double* rhs, double* sol, bool caller_managed, int prepped) {
  const int nb = Mp / DB;
  const int ntile = nb * (nb + 1) / 2;
  const int nitem = ntile + (rhs ? 1 : 0);
  const int max_wg = df_max_workgroups();
  if (!caller_managed && !(prepped & 1)) zero_ints(scratch, (int)potrf_flag_ints(Mp), st);
  if (Linv && !(prepped & 2)) fill_zero(Linv, (size_t)Mp * ld, st);  // level 0 of tri_inverse(): diagonal-block inverses (written by the
                                                   // diagonal tile owners inside the launch), zero elsewhere
  // the chain-workgroup kernel from two block columns on, when at least one other workgroup can be resident beside it
  static const int use_chain = getenv("SGP_POTRF_CHAIN") ? atoi(getenv("SGP_POTRF_CHAIN")) : 1;  // 0: the round-1 dataflow kernel (A/B); 2: the chain kernel for every nb >= 2
  // launch modes of the chain kernel (sgp_potrf_chain.hpp: CH_MODE_*), read once
  static const int env_mode = ((getenv("SGP_POTRF_ACQUIRE") && atoi(getenv("SGP_POTRF_ACQUIRE"))) ? CH_MODE_ACQUIRE : 0) |
                              ((getenv("SGP_POTRF_LIGHT") && !atoi(getenv("SGP_POTRF_LIGHT"))) ? CH_MODE_NOLIGHT : 0) |
                              ((getenv("SGP_POTRF_TICKET") && atoi(getenv("SGP_POTRF_TICKET"))) ? CH_MODE_TICKET : 0);
  // the ticketed claim where the caller says the GPU is shared (SGP_OPT_SHARED_DEVICE of the call's context; SGP_POTRF_TICKET=1 forces it)
  int mode = env_mode | (cur_ctx().shared_device ? CH_MODE_TICKET : 0);
  // The acquire-free consumer side (and the light same-XCD hand-overs) rest on "one writer per cache line, no reader before its flag":
  // tiles and scratch blocks must not share 128-byte lines.  The scratch blocks never do (potrf_scratch_ints: whole lines per flag, 8 KB+
  // per block, `scratch` itself checked here); a caller's matrix with an odd leading dimension or base address does -- such a call takes
  // the acquire mode, which needs neither.
  if ((ld & 15) != 0 || (reinterpret_cast<uintptr_t>(A) & 127) != 0 || (reinterpret_cast<uintptr_t>(scratch) & 127) != 0 ||
      (Linv && (reinterpret_cast<uintptr_t>(Linv) & 127) != 0))
    mode |= CH_MODE_ACQUIRE | CH_MODE_NOLIGHT;
  // Without an inverse and up to four block columns the round-1 kernel is the faster one (M = 256: 92 against 107 us per call,
  // profiles/r05_v8_potrf_bench_ab.jsonl; with the inverse the chain kernel wins at every size: 75 against 122 us) -- VERDICT r5 next-4a
  const bool small_plain = !Linv && nb <= 4 && use_chain != 2;
  if (use_chain && nb >= 2 && max_wg >= 2 && !small_plain) {
    const int nout_items = ch_tile_items(nb) + (Linv ? ch_inv_items(nb) : 0) + (rhs ? 1 : 0);  // tiles (tile (c+2, c) twice), blocks of L^-1, rhs
    const int nout = nout_items < max_wg - 1 ? nout_items : max_wg - 1;
    potrf_chain_kernel<<<1 + nout, CH_THREADS, 0, st>>>(A, ld, nb, scratch, info, info_base, rhs, sol, Linv, mode);
    if (!caller_managed) potrf_timeout_kernel<<<1, 64, 0, st>>>(scratch + ntile * DF_FLAG_STRIDE, info);
    return Linv != nullptr;  // the whole of L^-1, not only its diagonal blocks
  }
  potrf_dataflow_kernel<<<nitem < max_wg ? nitem : max_wg, 256, 0, st>>>(
      A, ld, nb, scratch, reinterpret_cast<double*>(scratch + potrf_flag_ints(Mp)), info, info_base, rhs, sol, Linv, (mode & CH_MODE_TICKET) ? 1 : 0);
  if (!caller_managed) potrf_timeout_kernel<<<1, 64, 0, st>>>(scratch + ntile * DF_FLAG_STRIDE, info);
  return false;
}

void potrf_lower_batch(double* A, double* Linv, int64_t ld, int Mp, int S, int64_t stride, int* info, int* scratch, hipStream_t st) {
  const int nb = Mp / DB;
  const int ntile = nb * (nb + 1) / 2;
  const int64_t sints = (int64_t)potrf_scratch_ints(Mp), fints = (int64_t)potrf_flag_ints(Mp);
  // every workgroup of the launch must be resident: the S groups share the device's CUs
  int per = df_max_workgroups() / S;
  if (per < 1) per = 1;
  zero_ints(scratch, (int)(sints * S), st);
  if (Linv) fill_zero(Linv, (size_t)stride * (S - 1) + (size_t)Mp * ld, st);
  potrf_dataflow_batch_kernel<<<dim3(ntile < per ? ntile : per, S), 256, 0, st>>>(A, ld, nb, scratch, sints, fints, info, Linv, stride,
                                                                                cur_ctx().shared_device ? 1 : 0);
  potrf_timeout_batch_kernel<<<1, 64, 0, st>>>(scratch, sints, ntile * DF_FLAG_STRIDE, info, S);
}

void tri_inverse(const double* L, double* Linv, double* tmp, int64_t ld, int Mp, hipStream_t st, int nbatch, int64_t stride) {
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
      a.batch2 = nbatch; a.s2A = a.s2B = a.s2C = stride;
      gemm(a, st);
      // Inv21 = -Inv22 * T   (n2 x s) = (n2 x n2)(n2 x s), Inv22 lower-triangular -> k <= row tile end
      GemmDesc b;
      b.A = Linv + o * (ld + 1) + (int64_t)s * (ld + 1); b.lda = ld; b.sA = pstride;
      b.B = tmp + o * (ld + 1) + (int64_t)s * ld; b.ldb = ld; b.sB = pstride;
      b.C = Linv + o * (ld + 1) + (int64_t)s * ld; b.ldc = ld; b.sC = pstride;
      b.m = n2; b.n = s; b.k = n2; b.batch = batch; b.alpha = -1.0; b.khi_mask = 1;
      b.batch2 = nbatch; b.s2A = b.s2B = b.s2C = stride;
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
  potrf_lower(Ap, nullptr, Mp, Mp, info, 0, flags, st);  // (no inverse wanted: the launch then carries no L^-1 items)
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
