// Streaming pass 1: kernel-matrix assembly + SYRK on the fp64 matrix cores.
//
//   Phi = Kuf Kuf^T (M x M),  b = Kuf y,  yy = y^T y,  kappa = sum_n k(x_n, x_n)
//
// replaces the N x M kernel matrix and the N M^2 contraction of InducingPointKernel /
// ExactMarginalLogLikelihood (reference models/sgpr.py:37,125) and of pm.gp.MarginalSparse
// (reference models/bayesian_sgpr_hmc.py:71).
//
// Measured on MI355X (profiles/r01_fp64_rates_microbench.txt): v_mfma_f64_16x16x4_f64 issues every
// 64 cycles per SIMD (77 TFLOP/s chip-wide) and fp64 VALU work does NOT overlap with it -- the two
// share the fp64 datapath (MFMA + n v_fma_f64 costs 64 + ~5n cycles).  Every exp() spent re-generating
// a kernel value inside the contraction is therefore paid at full price, so the value is generated
// exactly once:
//
//   1. `kfu_assemble_kernel` (HBM-write bound, "kernel assembly"): thread <-> inducing column m keeps
//      z~_m in registers, walks 256 wave-uniform data rows (x~_n, y_n arrive through the scalar cache)
//      and writes k'(x_n, z_m) = profile(|x~_n - z~_m|^2) into K'_fu [Npad x Mp] with 512-byte
//      coalesced stores; the b = K^T y partial of its rows falls out of the same loop.
//   2. `syrk_tile_kernel` (fp64-MFMA bound, "contraction"): workgroup <-> (128 x 128 tile of the lower
//      triangle of Phi, split of the row range).  16-row chunks of K'_fu go global -> registers -> LDS
//      (double buffered, [n][col] with a 272-double stride so the two 16-lane halves of an operand read
//      hit disjoint bank halves); 4 waves x 16 MFMA tiles (64 x 64 per wave) accumulate in 128 VGPRs.
//      Both SYRK operands are the same K' rows: lane l supplies K'[n0 + (l>>4)][col0 + (l&15)].
//      Workgroup ids are laid out so that all tiles of one split land on one XCD (round-robin dispatch):
//      its L2 then serves the 8-9 re-reads of every K' row block (speed only, never correctness).
//   3. `reduce_phi_kernel`: sums the per-split slabs in a fixed order, applies sf2^2 and mirrors the
//      lower triangle -- no fp64 atomics anywhere, results are bit-reproducible.
#include "sgp_common.hpp"
#include "sgp_stream.hpp"
#include "sgp_dense.hpp"
#include "sgp_composite.hpp"
#include "sgp_ctx.hpp"
#include <cstdlib>

namespace sgp {

// (every switch and every piece of state below lives in the call's context: sgp_ctx.hpp)
static_assert(TIMING_SLOTS == CTX_TIMING_SLOTS, "timing slots");
size_t stream_kfu_budget() { return cur_ctx().kfu_budget; }

// (a failed event call only leaves the optional timing slot unused: sgp_timing_last_ms then reports SGP_ERR_ARG)
void timing_begin(int slot, hipStream_t st) {
  Ctx& c = cur_ctx();
  if (!c.timing) return;
  if (!c.ev_ready) {
    bool ok = true;
    for (int s = 0; s < TIMING_SLOTS; ++s)
      for (int k = 0; k < 2; ++k) ok = (hipEventCreate(&c.ev[s][k]) == hipSuccess) && ok;
    if (!ok) return;
    c.ev_ready = 1;
  }
  if (hipEventRecord(c.ev[slot][0], st) != hipSuccess) c.ev_used[slot] = 0;
}
void timing_end(int slot, hipStream_t st) {
  Ctx& c = cur_ctx();
  if (!c.timing || !c.ev_ready) return;
  c.ev_used[slot] = hipEventRecord(c.ev[slot][1], st) == hipSuccess ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// prologue kernels
// ---------------------------------------------------------------------------------------------
// out[r][j] = in[r][j] * inv_ls[j] for r < rows, j < d ; zero padding elsewhere.
// The prologue of a pass over the rows in ONE launch (round 5; three launches before: 10 us less on every pass of a C3-sized evaluation):
// blocks [0, gx): X / lengthscale into the padded Xs;  [gx, gx + gz): the same for Z;  the last 256: y padded + its sum of squares in
// 256 fixed partial sums (the block partition of each job is what its own launch had, so every number is bit for bit what it was).
__device__ __forceinline__ void scale_rows_job(const double* __restrict__ in, int64_t ld, int64_t rows, int64_t rows_pad, int DP,
                                               const KernArgs& ka, double* __restrict__ out, int blk, int nblk) {
  const int64_t total = rows_pad * DP;
  for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < total; i += (int64_t)nblk * 256) {
    const int64_t r = i / DP;
    const int j = (int)(i - r * DP);
    double v = 0.0;
    if (r < rows && j < ka.d) v = in[r * ld + j] * ka.inv_ls[j];
    out[i] = v;
  }
}
__global__ __launch_bounds__(256) void stream_prologue_kernel(const double* __restrict__ X, int64_t ldx, int64_t N, int64_t Npad,
                                                              const double* __restrict__ Z, int64_t ldz, int M, int Mp, int DP, KernArgs ka,
                                                              double* __restrict__ Xs, double* __restrict__ Zs,
                                                              const double* __restrict__ y, double* __restrict__ ys,
                                                              double* __restrict__ yypart, int gx, int gz, PadSymJob pad, int gp) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  if (b >= gx + gz + 256) {  // pass 2's padded, symmetrised Phibar (+ bbar): the same element <-> thread mapping as its own launch had
    const int pb = b - gx - gz - 256;
    const int64_t total = (int64_t)Mp * Mp;
    for (int64_t e = (int64_t)pb * 256 + threadIdx.x; e < total; e += (int64_t)gp * 256) {
      const int r = (int)(e / Mp), c = (int)(e - (int64_t)r * Mp);
      double v = 0.0;
      if (r < M && c < M) v = 0.5 * (pad.P[(int64_t)r * M + c] + pad.P[(int64_t)c * M + r]);
      pad.out[e] = v;
    }
    if (pad.vout && pb == 0)
      for (int i = threadIdx.x; i < Mp; i += 256) pad.vout[i] = i < M ? pad.vec[i] : 0.0;
    return;
  }
  if (b < gx) {
    scale_rows_job(X, ldx, N, Npad, DP, ka, Xs, b, gx);
  } else if (b < gx + gz) {
    scale_rows_job(Z, ldz, M, Mp, DP, ka, Zs, b - gx, gz);
  } else {
    const int yb = b - gx - gz;
    double s = 0.0;
    for (int64_t i = (int64_t)yb * 256 + threadIdx.x; i < Npad; i += (int64_t)256 * 256) {
      const double v = i < N ? y[i] : 0.0;
      ys[i] = v;
      s = fma(v, v, s);
    }
    s = block_sum256(s, red);
    if (threadIdx.x == 0) yypart[yb] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// 1. kernel assembly:  Kfu[n][m] = k'(x_n, z_m)   (no sf2; zero in the padding)
//    grid = (rows / ASM_ROWS, ceil(Mp / 256)): a workgroup owns 256 columns x 256 rows; every wave stores
//    512 contiguous bytes per data row, a workgroup 2 KB.  bpart[rowblock][m] = sum_n Kfu[n][m] y[n].
// ---------------------------------------------------------------------------------------------
template <int DP, int KID, int ROWS>
__global__ __launch_bounds__(256) void kfu_assemble_kernel(const double* __restrict__ Xs, const double* __restrict__ ys,
                                                           const double* __restrict__ Zs, int64_t row0, int64_t N, int M,
                                                           int Mp, double* __restrict__ Kfu, double* __restrict__ bpart) {
  constexpr int ASM_ROWS = ROWS;  // rows of this workgroup (sgp::ASM_ROWS, or a quarter of it on small shards: StreamPlan::asm_sub)
  // the block's 256 scaled data rows and targets are staged once in LDS (coalesced) and then read as
  // wave-wide broadcasts: measured 80 % of wave time parked on per-row scalar loads before (SQ_WAIT_ANY)
  __shared__ double xs[ASM_ROWS][DP];
  __shared__ double ysh[ASM_ROWS];
  __shared__ double etab[EXP_TAB_N];
  const int64_t rbase = (int64_t)blockIdx.x * ASM_ROWS;  // row inside this super-chunk's Kfu
  {
    const double* src = Xs + (row0 + rbase) * DP;
    double* dst = &xs[0][0];
    for (int e = threadIdx.x; e < ASM_ROWS * DP; e += 256) dst[e] = src[e];
    if ((int)threadIdx.x < ASM_ROWS) ysh[threadIdx.x] = ys[row0 + rbase + threadIdx.x];
    sgp_exp_tab_load(etab);
  }
  __syncthreads();
  const int m = blockIdx.y * 256 + threadIdx.x;
  if (m >= Mp) return;                                   // Mp is a multiple of 128: whole waves drop out
  const double zmask = m < M ? 1.0 : 0.0;

  double zr[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) zr[j] = Zs[(size_t)m * DP + j];

  double bacc = 0.0;
#pragma unroll 4
  for (int i = 0; i < ASM_ROWS; ++i) {
    const int64_t n = row0 + rbase + i;                    // global data row, wave-uniform
    double r2 = 0.0;
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const double df = xs[i][j] - zr[j];
      r2 = fma(df, df, r2);
    }
    const double msk = n < N ? zmask : 0.0;
    const double kv = kprofile_tab<KID>(r2, etab) * msk;
    __builtin_nontemporal_store(kv, &Kfu[(rbase + i) * Mp + m]);  // streamed once (1.80 vs 1.90 ms with plain stores)
    bacc = fma(kv, ysh[i], bacc);
  }
  bpart[((row0 + rbase) / ASM_ROWS) * Mp + m] = bacc;
}

// ---------------------------------------------------------------------------------------------
// 2. contraction: slab[split][tile] (+)= sum over this split's chunks of K'_I^T K'_J
// ---------------------------------------------------------------------------------------------
// Diagonal 128 x 128 tiles: only the 36 MFMA tiles (16 x 16) on or below the diagonal are needed.  A 64 x 64 block
// per wave would leave one wave idle and the other three at full work -- and an idle wave's SIMD share is not
// recovered (the co-resident workgroup's wave there just waits at its own barrier), so a diagonal tile cost as much as
// a full one.  Instead the 36 tiles are dealt 9 to a wave (row-major walk of the triangle): 1.31 ms instead of 1.72 ms
// per full-size diagonal workgroup, 16.3 vs 17.3 ms for the launch (per-workgroup stamps, same box).
// Tile t of wave w is (DT_R[w][t], DT_C[w][t]) in units of 16; its operands are v[r], v[c] with v[b] = K'[k][16 b + ..]
// (both operands come from the same 128 columns); wave w needs v[0 .. DT_NV[w] - 1].
__device__ constexpr int DT_R[4][9] = {{0, 1, 1, 2, 2, 2, 3, 3, 3}, {3, 4, 4, 4, 4, 4, 5, 5, 5}, {5, 5, 5, 6, 6, 6, 6, 6, 6}, {6, 7, 7, 7, 7, 7, 7, 7, 7}};
__device__ constexpr int DT_C[4][9] = {{0, 0, 1, 0, 1, 2, 0, 1, 2}, {3, 0, 1, 2, 3, 4, 0, 1, 2}, {3, 4, 5, 0, 1, 2, 3, 4, 5}, {6, 0, 1, 2, 3, 4, 5, 6, 7}};
__device__ constexpr int DT_NV[4] = {4, 6, 7, 8};

template <int W>
__device__ __forceinline__ void diag_chunk(const double (*K)[KROW], d4 (&acc)[4][4], int l15, int l4) {
#pragma unroll
  for (int ks = 0; ks < NB / 4; ++ks) {
    const double* kr = &K[ks * 4 + l4][0];
    double v[8];
#pragma unroll
    for (int b = 0; b < DT_NV[W]; ++b) v[b] = kr[16 * b + l15];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t >> 2][t & 3] = mfma16(v[DT_R[W][t]], v[DT_C[W][t]], acc[t >> 2][t & 3]);
  }
}
template <int W>
__device__ __forceinline__ void diag_store(const d4 (&acc)[4][4], double* __restrict__ out, int accumulate, int l15, int l4) {
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* dst = out + (16 * DT_R[W][t] + l4 + 4 * r) * TILE + 16 * DT_C[W][t] + l15;
      *dst = accumulate ? *dst + acc[t >> 2][t & 3][r] : acc[t >> 2][t & 3][r];
    }
}

// NW = waves per workgroup: 4 (each wave a 64 x 64 block, 2 workgroups = 2 waves per SIMD) or
//                           8 (each wave a 64 x 32 block, 2 workgroups = 4 waves per SIMD, <= 128 VGPRs)
// GLDS: the chunks travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: one wave instruction = one contiguous
//       1 KB row segment, no VGPR staging, no ds_write) instead of through two register stages.
template <bool DIAG, int NW, bool GLDS, bool SKIP>
__device__ __forceinline__ void syrk_tile(double (*Ks)[NB][KROW], const double* __restrict__ Kfu, int Mp, int64_t c0,
                                          int64_t c1, int I0, int J0, int accumulate, double* __restrict__ out) {
  constexpr int skip_upper = SKIP ? 1 : 0;
  constexpr int NT = NW * 64;
  constexpr int VB = NW == 4 ? 4 : 2;     // 16-column MFMA tiles per wave (rows: always 4 = 64 rows)
  constexpr int WCOLS = VB * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = NW == 4 ? wave >> 1 : wave >> 2, wj = NW == 4 ? (wave & 1) : (wave & 3);
  const int l15 = lane & 15, l4 = lane >> 4;
  constexpr int boff = DIAG ? 0 : TILE;   // where the B-operand columns live in the LDS tile
  constexpr int NQ = (DIAG ? 1024 : 2048) / NT;  // 16-byte quads this thread moves per chunk

  // staging role: quad q = tid + NT i -> LDS row n = q / 128 (off-diagonal) or q / 64 (diagonal), 64 consecutive threads
  // cover one contiguous 1 KB row segment of K'_fu.  Row n = row0 + RSTEP i and the column does not depend on i, so ONE
  // 32-bit lane offset serves all NQ loads (scalar base per i) and ONE LDS address all NQ stores (immediate offsets): the
  // 64-bit address pair and LDS address per quad this replaced had the kernel spill two registers, with a scratch reload and
  // s_waitcnt vmcnt(0) at the top of the loop.  (Same-box A/B: 65.1-65.4 TF before, 65.2 after -- no measurable difference.)
  constexpr int RSTEP = DIAG ? NT / 64 : NT / 128;  // LDS / chunk rows between consecutive quads of a thread
  const int srow0 = DIAG ? (tid >> 6) : (tid >> 7);
  const int rq = tid & 127;
  const int scol0 = DIAG ? (tid & 63) * 2 : (rq < 64 ? rq * 2 : TILE + (rq - 64) * 2);
  const unsigned goff0 = (unsigned)(srow0 * Mp + (DIAG ? I0 + scol0 : (rq < 64 ? I0 + rq * 2 : J0 + (rq - 64) * 2))) * 8u;  // bytes
  // Two register stages: the loads of chunk c+2 are issued before the MFMAs of chunk c and consumed
  // (written to LDS) after the MFMAs of chunk c+1 -- two MFMA phases of latency cover.
  d2 stA[NQ], stB[NQ];
  auto fetch = [&](int64_t c, d2 (&st)[NQ]) {
    const char* base = reinterpret_cast<const char*>(Kfu + c * NB * Mp);  // wave-uniform
#pragma unroll
    for (int i = 0; i < NQ; ++i) st[i] = *reinterpret_cast<const d2*>(base + (size_t)i * RSTEP * Mp * 8 + goff0);
  };
  auto stash = [&](int buf, const d2 (&st)[NQ]) {
    double* dst = &Ks[buf][srow0][scol0];
#pragma unroll
    for (int i = 0; i < NQ; ++i) *reinterpret_cast<d2*>(dst + i * RSTEP * KROW) = st[i];
  };

  d4 acc[4][VB];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < VB; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};

  // diagonal tiles, SKIP (default): the balanced 9-tiles-per-wave walk above (4-wave workgroups) or idle waves for the
  // strictly-upper blocks (8-wave workgroups); !SKIP computes the full tile (A/B knob SGP_SYRK_SKIP_UPPER=0)
  constexpr bool BALANCED = DIAG && NW == 4 && SKIP;
  const bool idle = DIAG && !BALANCED && skip_upper && (wj * WCOLS >= wi * 64 + 64);
  auto mfma_chunk = [&](int buf) {
    // While it issues MFMAs a wave outranks the co-resident workgroup's wave on its SIMD (which is then staging or
    // about to): 16.4-16.5 vs 16.7-16.9 ms for the launch, three alternations on one box.  (The same two lines in
    // pass 2 cost 0.8 %: there the partner's epilogue VALU work is what gets starved.)
    __builtin_amdgcn_s_setprio(1);
    if constexpr (BALANCED) {
      switch (wave) {
        case 0: diag_chunk<0>(Ks[buf], acc, l15, l4); break;
        case 1: diag_chunk<1>(Ks[buf], acc, l15, l4); break;
        case 2: diag_chunk<2>(Ks[buf], acc, l15, l4); break;
        default: diag_chunk<3>(Ks[buf], acc, l15, l4); break;
      }
      __builtin_amdgcn_s_setprio(0);
      return;
    }
    if (idle) {
      __builtin_amdgcn_s_setprio(0);
      return;
    }
    // (A/B, same box, three alternations: reading the operands of k-step ks + 1 before the MFMAs of k-step ks issue --
    // explicit double-buffered operand registers behind sched_barriers -- changes nothing, 65.2-65.5 TF either way: with two
    // waves per SIMD the matrix pipe is fed by the partner wave whenever this one waits for LDS.)
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks) {
      const double* kr = &Ks[buf][ks * 4 + l4][0];
      double a[4], bq[VB];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = kr[wi * 64 + u * 16 + l15];
#pragma unroll
      for (int v = 0; v < VB; ++v) bq[v] = kr[boff + wj * WCOLS + v * 16 + l15];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < VB; ++v) acc[u][v] = mfma16(a[u], bq[v], acc[u][v]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  if constexpr (GLDS) {
    // wave-uniform LDS bases of this thread's NQ row segments (lane l lands at base + 16 l bytes)
    auto dma = [&](int64_t c, int buf) {
      const char* base = reinterpret_cast<const char*>(Kfu + c * NB * Mp);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int qb = wave * 64 + NT * i;  // first quad of this wave's segment
        const int row = DIAG ? (qb >> 6) : (qb >> 7);
        const int col0 = DIAG ? 0 : (((qb & 127) < 64) ? 0 : TILE);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)i * RSTEP * Mp * 8 + goff0),
                                         (__attribute__((address_space(3))) void*)&Ks[buf][row][col0], 16, 0, 0);
      }
    };
    if (c0 < c1) {
      dma(c0, 0);
      __syncthreads();  // (emits vmcnt(0)): chunk c0 has landed for every wave
      for (int64_t c = c0; c < c1; ++c) {
        const int buf = (int)((c - c0) & 1);
        if (c + 1 < c1) dma(c + 1, buf ^ 1);  // that buffer was last read before the barrier that ended iteration c-1
        mfma_chunk(buf);
        __syncthreads();  // my DMA of chunk c+1 has landed (vmcnt(0)) and everybody is done reading `buf`
      }
    }
  } else {
    // (A/B: writing chunk c+1 to LDS at the top of iteration c -- one register stage, ds_writes hidden under the
    // MFMAs -- is 4.7 % slower: the loads then have one MFMA phase instead of two to land.)
    if (c0 < c1) {
      fetch(c0, stA);
      stash(0, stA);
      if (c0 + 1 < c1) fetch(c0 + 1, stB);
      __syncthreads();
      // invariant at the top of iteration c: LDS buffer (c-c0)&1 holds chunk c; chunk c+1 is in flight / in
      // stage B (even trips) or stage A (odd trips)
      int64_t c = c0;
#ifdef SGP_AB_DIAG_DUMMY_ASM
      // A/B cost model (tools/ab_build.sh -DSGP_AB_DIAG_DUMMY_ASM): what would the contraction cost if the diagonal-tile
      // workgroups GENERATED their K' column block (8 exp() per thread and chunk) and wrote it out for the off-diagonal tiles
      // (8 doubles per thread and chunk), i.e. if kernel assembly lived inside this launch?  The values written back are the
      // ones just loaded, so results do not change.
      double sink = 0.0;
      auto dummy = [&](int64_t cc, const d2 (&st)[NQ]) {
        if constexpr (DIAG) {
          char* base = const_cast<char*>(reinterpret_cast<const char*>(Kfu + cc * NB * Mp));
#pragma unroll
          for (int i = 0; i < NQ; ++i) {
            sink += sgp_exp(-st[i][0]) + sgp_exp(-st[i][1]);
            double* dst = reinterpret_cast<double*>(base + (size_t)i * RSTEP * Mp * 8 + goff0);
            __builtin_nontemporal_store(st[i][0], dst);
            __builtin_nontemporal_store(st[i][1], dst + 1);
          }
        }
      };
#define SGP_DUMMY(cc, st) dummy(cc, st)
#else
#define SGP_DUMMY(cc, st)
#endif
      for (; c + 1 < c1; c += 2) {
        if (c + 2 < c1) fetch(c + 2, stA);
        mfma_chunk(0);
        SGP_DUMMY(c + 1, stB);
        stash(1, stB);  // chunk c+1
        __syncthreads();
        if (c + 3 < c1) fetch(c + 3, stB);
        mfma_chunk(1);
        if (c + 2 < c1) { SGP_DUMMY(c + 2, stA); }
        if (c + 2 < c1) stash(0, stA);  // chunk c+2
        __syncthreads();
      }
#ifdef SGP_AB_DIAG_DUMMY_ASM
      if (sink == 1.2345e300) out[0] = sink;  // keeps the exps alive
#endif
      if (c < c1) mfma_chunk(0);  // odd chunk count: the last chunk sits in buffer 0
    }
  
}

  if constexpr (BALANCED) {
    switch (wave) {
      case 0: diag_store<0>(acc, out, accumulate, l15, l4); break;
      case 1: diag_store<1>(acc, out, accumulate, l15, l4); break;
      case 2: diag_store<2>(acc, out, accumulate, l15, l4); break;
      default: diag_store<3>(acc, out, accumulate, l15, l4); break;
    }
    return;
  }
  {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < VB; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wi * 64 + u * 16 + l4 + 4 * r;
          const int col = wj * WCOLS + v * 16 + l15;
          double* dst = out + row * TILE + col;
          *dst = accumulate ? *dst + acc[u][v][r] : acc[u][v][r];
        }
  }
}

template <int NW, bool GLDS, bool SKIP>
__global__ __launch_bounds__(NW * 64, NW / 2) void syrk_tile_kernel(const double* __restrict__ Kfu, int Mp, int64_t nchunks,
                                                                    SplitMap smap, int ntiles, int accumulate,
                                                                    double* __restrict__ slab, int nsplit) {
  __shared__ double Ks[2][NB][KROW];
  // id -> (xcd, tile, split group): all tiles of a split share id % 8, i.e. one XCD under round-robin dispatch.  When nsplit is
  // not a multiple of 8 (the one-round head block: 14 splits x 36 tiles = 504 workgroups) the last nsplit % 8 splits are dealt
  // to the XCDs as one run of (split, tile) tasks cut into 8 equal pieces: every XCD gets the same number of workgroups and
  // the tiles of at most two splits (grid = 8 (nsplit / 8 * ntiles + ceil((nsplit % 8) ntiles / 8)), see syrk_grid()).
  const int id = blockIdx.x;
  const int xcd = id & 7;
  const int jj = id >> 3;
  const int full = (nsplit >> 3) * ntiles;
  int t, split;
  if (jj < full) {
    t = jj % ntiles;
    split = (jj / ntiles) * 8 + xcd;
  } else {
    const int rem = (nsplit & 7) * ntiles, per = (rem + 7) >> 3;
    const int u = xcd * per + (jj - full);
    if (jj - full >= per || u >= rem) return;
    t = u % ntiles;
    split = (nsplit & ~7) + u / ntiles;
  }
  int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  while (ti * (ti + 1) / 2 > t) --ti;
  const int tj = t - ti * (ti + 1) / 2;
  int64_t c0, c1;
  split_range(smap, split, nchunks, c0, c1);
  double* out = slab + ((size_t)split * ntiles + t) * (TILE * TILE);
  if (ti == tj)
    syrk_tile<true, NW, GLDS, SKIP>(Ks, Kfu, Mp, c0, c1, ti * TILE, tj * TILE, accumulate, out);
  else
    syrk_tile<false, NW, GLDS, SKIP>(Ks, Kfu, Mp, c0, c1, ti * TILE, tj * TILE, accumulate, out);
}

// ---------------------------------------------------------------------------------------------
// 3. deterministic split reduction + symmetrisation:  Phi = sf2^2 * sum_s slab[s]
//    one block per 32 x 32 sub-block of the lower triangle
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
    const int lr = ty + 8 * k;
    const int gi = bi * 32 + lr, gj = bj * 32 + tx;
    double s = 0.0;
    if (gi >= gj) {  // computed part of the slab (lower triangle incl. diagonal)
      const size_t off = (size_t)(gi - Ti * TILE) * TILE + (gj - Tj * TILE);
#pragma unroll 8
      for (int sp = 0; sp < nsplit; ++sp) s += base[sp * sstride + off];  // fixed order; 8 loads in flight
    }
    tile[lr][tx] = s * scale;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int lr = ty + 8 * k;
    const int gi = bi * 32 + lr, gj = bj * 32 + tx;
    if (gi < M && gj < M && gi >= gj) Phi[(size_t)gi * M + gj] = tile[lr][tx];
    const int mi = bj * 32 + lr, mj = bi * 32 + tx;  // mirrored element, row-contiguous store
    if (mi < M && mj < M && mj > mi) Phi[(size_t)mi * M + mj] = tile[tx][lr];
  }
}

// stage 1 of the b reduction: tmp[g][m] = sum of bpart rows g, g + G, g + 2G, ...  (grid = (Mp / 64, G);
// a block owns 64 columns, its 4 waves interleave the rows and combine through LDS in a fixed order)
__global__ __launch_bounds__(256) void bpart_stage1_kernel(const double* __restrict__ bpart, int64_t nrb, int Mp, int G,
                                                           double* __restrict__ tmp) {
  __shared__ double part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6, g = blockIdx.y;
  double s = 0.0;
  for (int64_t rb = g + (int64_t)w * G; rb < nrb; rb += 4 * (int64_t)G) s += bpart[rb * Mp + col];
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0) {
    const int l = threadIdx.x;
    tmp[(size_t)g * Mp + col] = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
  }
}

__global__ __launch_bounds__(256) void finalize_stats_kernel(const double* __restrict__ btmp, int G, int Mp, int M,
                                                             const double* __restrict__ yypart, int nyy, double sf2,
                                                             double kappa_val, double* __restrict__ b,
                                                             double* __restrict__ yy, double* __restrict__ kappa) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m < M) {
    double s = 0.0;
#pragma unroll 8
    for (int g = 0; g < G; ++g) s += btmp[(size_t)g * Mp + m];  // same order of additions; the loads of eight steps in flight
    b[m] = s * sf2;
  }
  if (blockIdx.x == 0) {  // fixed-order block reduction (one thread walking the 256 partials took 30 us)
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nyy; i += 256) s += yypart[i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) {
      *yy = s;
      *kappa = kappa_val;
    }
  }
}

template <int DP, int ROWS>
static void launch_assemble_rows(int kid, dim3 grid, hipStream_t st, const double* Xs, const double* ys, const double* Zs,
                                 int64_t row0, int64_t N, int M, int Mp, double* Kfu, double* bpart) {
  switch (kid) {
    case SGP_KERNEL_RBF: kfu_assemble_kernel<DP, SGP_KERNEL_RBF, ROWS><<<grid, 256, 0, st>>>(Xs, ys, Zs, row0, N, M, Mp, Kfu, bpart); break;
    case SGP_KERNEL_MATERN32: kfu_assemble_kernel<DP, SGP_KERNEL_MATERN32, ROWS><<<grid, 256, 0, st>>>(Xs, ys, Zs, row0, N, M, Mp, Kfu, bpart); break;
    default: kfu_assemble_kernel<DP, SGP_KERNEL_MATERN52, ROWS><<<grid, 256, 0, st>>>(Xs, ys, Zs, row0, N, M, Mp, Kfu, bpart); break;
  }
}
template <int DP>
static void launch_assemble(int kid, int sub, dim3 grid, hipStream_t st, const double* Xs, const double* ys, const double* Zs,
                            int64_t row0, int64_t N, int M, int Mp, double* Kfu, double* bpart) {
  if (sub == 16) launch_assemble_rows<DP, ASM_ROWS / 16>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Kfu, bpart);
  else if (sub == 8) launch_assemble_rows<DP, ASM_ROWS / 8>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Kfu, bpart);
  else if (sub == 4) launch_assemble_rows<DP, ASM_ROWS / 4>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Kfu, bpart);
  else launch_assemble_rows<DP, ASM_ROWS>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Kfu, bpart);
}

// Assemble rows [row0, row0 + rows) of K'_fu (rows a multiple of ASM_ROWS) into Kfu (which starts at row0).
void stream_assemble(const StreamPlan& p, int kid, const double* Xs, const double* ys, const double* Zs, int64_t row0,
                     int64_t rows, int64_t N, int M, double* Kfu, double* bpart, hipStream_t st) {
  dim3 grid((unsigned)(rows / ASM_ROWS * p.asm_sub), (p.Mp + 255) / 256);
  switch (p.DP) {
    case 2: launch_assemble<2>(kid, p.asm_sub, grid, st, Xs, ys, Zs, row0, N, M, p.Mp, Kfu, bpart); break;
    case 4: launch_assemble<4>(kid, p.asm_sub, grid, st, Xs, ys, Zs, row0, N, M, p.Mp, Kfu, bpart); break;
    case 8: launch_assemble<8>(kid, p.asm_sub, grid, st, Xs, ys, Zs, row0, N, M, p.Mp, Kfu, bpart); break;
    case 16: launch_assemble<16>(kid, p.asm_sub, grid, st, Xs, ys, Zs, row0, N, M, p.Mp, Kfu, bpart); break;
    case 24: launch_assemble<24>(kid, p.asm_sub, grid, st, Xs, ys, Zs, row0, N, M, p.Mp, Kfu, bpart); break;
    default: launch_assemble<32>(kid, p.asm_sub, grid, st, Xs, ys, Zs, row0, N, M, p.Mp, Kfu, bpart); break;
  }
}

void stream_prologue(const StreamPlan& p, const KernArgs& ka, const double* X, int64_t ldx, const double* y,
                     const double* Z, int64_t ldz, int64_t N, int M, double* Xs, double* ys, double* Zs, double* yypart,
                     hipStream_t st, const PadSymJob& pad) {
  int gx = 0;
  if (N > 0) {
    const int64_t tot = p.Npad * p.DP;
    gx = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
  }
  const int gz = (p.Mp * p.DP + 255) / 256;
  int gp = 0;
  if (pad.P) {
    const int64_t blocks = ((int64_t)p.Mp * p.Mp + 255) / 256;
    gp = (int)(blocks < 2048 ? blocks : 2048);
  }
  stream_prologue_kernel<<<gx + gz + 256 + gp, 256, 0, st>>>(X, ldx, N, p.Npad, Z, ldz, M, p.Mp, p.DP, ka, Xs, Zs, y, ys, yypart, gx, gz, pad, gp);
}

static int syrk_grid(int nsplit, int ntiles) { return 8 * ((nsplit >> 3) * ntiles + (((nsplit & 7) * ntiles + 7) >> 3)); }

// ---- kernel assembly beside the contraction: an A/B knob, OFF by default (measured: a loss) ------------------------------
// Assembly is HBM-write bound, the contraction matrix-core bound, and the contraction needs all of K'_fu only at its very
// end -- so VERDICT r2 asked for the assembly of row block r + 1 on a second stream beside the contraction of block r.  With
// the whole K'_fu resident, mode 1 runs pass 1 as
//     main stream:  prologue, assembly(head rows), contraction(head: ONE round of resident workgroups), contraction(tail)
//     side stream:                                 assembly(tail rows)  -- beside the head contraction
// (head = 14 splits x 36 tiles at M = 1024, its splits as long as the tail's full-size ones; the tail launch waits for the
// side stream's event).  Same-box alternations at C5 (profiles/r03_overlap_ab.jsonl): one block 18.93 / 19.00 ms per
// evaluation, head + tail enqueued serially (mode 2) 19.47, overlapped (mode 1) 19.71-19.83.  Two effects, both losses:
// (i) a lone round of workgroups costs 2.2 ms for 1/9 of the rows (nothing fills its ragged end; the tail's rounds cost 1.6),
// (ii) there is no overlap to be had between these two kernels as launched: the contraction's two workgroups take a CU's whole
// register file and the assembly's eight take all of its wave slots, so whichever stream gets a CU first owns it until its
// workgroups retire -- [head contraction || tail assembly] took 3.6 ms, the sum of the two alone.  Real overlap would need the
// assembly INSIDE the contraction's launch (producer workgroups in its grid); its ceiling is the 0.8 ms by which the assembly's
// HBM time exceeds its fp64 VALU time, because fp64 VALU and MFMA share the datapath.  Kept for A/B.
constexpr int HEAD_SPLITS_MAX = 24;
static bool side_stream_ready(Ctx& c) {
  if (c.side) return true;
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return false;
  if (hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, hi) != hipSuccess) { c.side = nullptr; return false; }
  if (hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming) != hipSuccess) {
    (void)hipStreamDestroy(c.side);
    c.side = nullptr;
    return false;
  }
  return true;
}
// chunks of the head block (0 = no head): one round of resident workgroups whose splits are as long as the tail's full-size ones
static void head_block(const StreamPlan& p, int64_t nchunks, int* head_ns, int64_t* head_chunks) {
  *head_ns = 0;
  *head_chunks = 0;
  const int g_asm_overlap = cur_ctx().asm_overlap;
  int hs = RESIDENT_WGS / p.ntiles;
  if (hs > HEAD_SPLITS_MAX) hs = HEAD_SPLITS_MAX;
  const int64_t w = 8 * (int64_t)p.taper[0] + 4 * p.taper[1] + 2 * p.taper[2] + p.taper[3];  // tapered plans only (big shards)
  if (g_asm_overlap == 0 || hs < 8 || w == 0 || p.sc_rows != p.Npad) return;
  // a full-size tail split has 8 nchunks_tail / (8 w) chunks; head split = the same length: nh = hs c, c = (nchunks - nh) / w
  int64_t c = nchunks / (w + hs);
  int64_t nh = (hs * c) / (ASM_ROWS / NB) * (ASM_ROWS / NB);  // whole assembly row blocks
  if (c < 64 || nh <= 0 || nh * 4 > nchunks) return;
  *head_ns = hs;
  *head_chunks = nh;
}

// ---- which matrix cores contract (value-only evaluations) ------------------------------------------------------------------
// 0: fp64 (syrk_tile_kernel), 1 (default): the integer cores (sgp_suffstats_i8.hip) where they win -- enough work, and a K'_fu
// nobody keeps -- 2: the integer cores whenever the call allows it (tests).  SGP_CONTRACTION / sgp_set_contraction.
// default rule: rows x Mp^2 >= 2^32 (profiles/r03_i8_boundary.jsonl: 8192 x 1024, 30000 x 512, 65536 x 256, 1M x 128 win by 5-36 %;
// 16384 x 256, 20000 x 384, 200000 x 100, 4096 x 512 lose -- too few 128 x 64 tiles x 16384-row splits to fill 256 CUs)
constexpr double I8_MIN_WORK = 4294967296.0;

static int contraction_mode() { return cur_ctx().contraction; }
// the rule itself (also sgp_ctx_contraction_would_use_i8): Npad rows of the shard, padded M
static bool i8_rule(int mode, int64_t Npad, int Mp) {
  return Npad > 0 && (mode == 2 || (mode == 1 && (double)Npad * Mp * Mp >= I8_MIN_WORK));
}

constexpr int BRED_G = 64;  // row groups of the two-stage b reduction
struct FwdWs {
  double *Xs, *ys, *Zs, *Kfu, *slab, *bpart, *btmp, *yypart;
  uint8_t* Q;  // digit planes of one super-chunk when the caller owns K'_fu (otherwise they live in Kfu)
  size_t bytes;
};
// qrows: rows of one super-chunk of digit planes = the plan's sc_rows BEFORE a caller-owned K'_fu turns it into Npad
static FwdWs carve_fwd(void* ws, const StreamPlan& p, bool need_kfu, int64_t qrows) {
  Carver c(ws);
  FwdWs w;
  w.Xs = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.DP);
  w.ys = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1));
  w.Zs = c.take<double>((size_t)p.Mp * p.DP);
  int nslab = p.nsplit + HEAD_SPLITS_MAX;  // + the head block's splits
  if (qrows > 0 && i8_nsplit(qrows, p.Mp) > nslab) nslab = i8_nsplit(qrows, p.Mp);  // the int8 contraction's own split count
  w.slab = c.take<double>((size_t)nslab * p.ntiles * TILE * TILE);
  w.bpart = c.take<double>((size_t)bpart_rows(p) * p.Mp);
  w.btmp = c.take<double>((size_t)BRED_G * p.Mp);
  w.yypart = c.take<double>(256);
  w.Kfu = need_kfu ? c.take<double>((size_t)(p.sc_rows > 0 ? p.sc_rows : 1) * p.Mp) : nullptr;
  // (with a caller-owned K'_fu the planes need a super-chunk of their own -- only where the contraction rule can pick the integer path)
  const bool own_q = !need_kfu && i8_rule(contraction_mode(), p.Npad, p.Mp);
  w.Q = need_kfu ? reinterpret_cast<uint8_t*>(w.Kfu) : (own_q ? c.take<uint8_t>((size_t)(qrows > 0 ? qrows : 1) * p.Mp * 7) : nullptr);
  w.bytes = c.used();
  return w;
}

// the fp64 contraction of `nchunks` 16-row chunks of a row-major [rows x Mp] matrix into the per-split slabs (options of the context)
static void launch_syrk(const Ctx& cx, const double* K, int Mp, int64_t nchunks, const SplitMap& smap, int ntiles, int nsplit, int accum,
                        double* slab, hipStream_t st) {
  const int skip_upper = cx.syrk_skip_upper;  // 0 = full diagonal tiles (A/B knob SGP_SYRK_SKIP_UPPER, read at context creation)
  const int nwaves = cx.syrk_waves, glds = cx.syrk_glds;
  const int g = syrk_grid(nsplit, ntiles);
#define SGP_SYRK_LAUNCH(NWV, GL, SK) \
  syrk_tile_kernel<NWV, GL, SK><<<g, NWV * 64, 0, st>>>(K, Mp, nchunks, smap, ntiles, accum, slab, nsplit)
  if (glds) {
    if (skip_upper) SGP_SYRK_LAUNCH(4, true, true); else SGP_SYRK_LAUNCH(4, true, false);
  } else if (nwaves == 8) {
    if (skip_upper) SGP_SYRK_LAUNCH(8, false, true); else SGP_SYRK_LAUNCH(8, false, false);
  } else {
    if (skip_upper) SGP_SYRK_LAUNCH(4, false, true); else SGP_SYRK_LAUNCH(4, false, false);
  }
#undef SGP_SYRK_LAUNCH
}

// ---- whitened pass 1 in the streaming layout (round 4) ---------------------------------------------------------------------
// bpart[rowblock][m] = sum over the block's ASM_ROWS rows of T[n][m] y[n] -- the partials kfu_assemble_kernel leaves for K'^T y, here for
// T^T y (same layout, same fixed-order reduction behind it); T starts at row0, grid = (rows / ASM_ROWS, ceil(Mp / 256))
__global__ __launch_bounds__(256) void tpart_kernel(const double* __restrict__ T, const double* __restrict__ ys, int64_t row0, int Mp,
                                                    double* __restrict__ bpart) {
  __shared__ double ysh[ASM_ROWS];
  const int64_t rbase = (int64_t)blockIdx.x * ASM_ROWS;
  ysh[threadIdx.x] = ys[row0 + rbase + threadIdx.x];
  __syncthreads();
  const int m = blockIdx.y * 256 + threadIdx.x;
  if (m >= Mp) return;
  const double* src = T + (size_t)rbase * Mp + m;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 2
  for (int i = 0; i < ASM_ROWS; i += 4) {
    a0 = fma(__builtin_nontemporal_load(src + (size_t)i * Mp), ysh[i], a0);
    a1 = fma(__builtin_nontemporal_load(src + (size_t)(i + 1) * Mp), ysh[i + 1], a1);
    a2 = fma(__builtin_nontemporal_load(src + (size_t)(i + 2) * Mp), ysh[i + 2], a2);
    a3 = fma(__builtin_nontemporal_load(src + (size_t)(i + 3) * Mp), ysh[i + 3], a3);
  }
  bpart[((row0 + rbase) / ASM_ROWS) * Mp + m] = (a0 + a1) + (a2 + a3);
}

// ---- the extended streaming order (round 4): Phi to 2^-61, the triple product in double-double -------------------------------------
// Phi as an unevaluated sum hi + lo: the splits' (slab, slab_lo) tiles added in double-double in a fixed order, both triangles written,
// Mp x Mp (ld Mp), no amplitude.  One block per 32 x 32 sub-block of the lower triangle, as reduce_phi_kernel.
__global__ __launch_bounds__(256) void reduce_phi_dd_kernel(const double* __restrict__ slab, const double* __restrict__ slab_lo, int nsplit,
                                                            int ntiles, int Mp, double* __restrict__ Ph, double* __restrict__ Pl) {
#pragma clang fp contract(off)
  __shared__ double th[32][33], tl[32][33];
  const int t = blockIdx.x;
  int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  while (bi * (bi + 1) / 2 > t) --bi;
  const int bj = t - bi * (bi + 1) / 2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Ti = (bi * 32) / TILE, Tj = (bj * 32) / TILE;
  const size_t tbase = (size_t)(Ti * (Ti + 1) / 2 + Tj) * (TILE * TILE), sstride = (size_t)ntiles * (TILE * TILE);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int lr = ty + 8 * k;
    const int gi = bi * 32 + lr, gj = bj * 32 + tx;
    double hi = 0.0, lo = 0.0;
    if (gi >= gj) {
      const size_t off = tbase + (size_t)(gi - Ti * TILE) * TILE + (gj - Tj * TILE);
      for (int sp = 0; sp < nsplit; ++sp) {
        const double a = slab[sp * sstride + off], b = slab_lo[sp * sstride + off];
        const double sum = hi + a;
        const double z = sum - hi;
        lo += ((hi - (sum - z)) + (a - z)) + b;
        hi = sum;
      }
      const double sum = hi + lo;
      lo -= sum - hi;
      hi = sum;
    }
    th[lr][tx] = hi;
    tl[lr][tx] = lo;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int lr = ty + 8 * k;
    const int gi = bi * 32 + lr, gj = bj * 32 + tx;
    if (gi >= gj) { Ph[(size_t)gi * Mp + gj] = th[lr][tx]; Pl[(size_t)gi * Mp + gj] = tl[lr][tx]; }
    const int mi = bj * 32 + lr, mj = bi * 32 + tx;  // mirrored element, row-contiguous store
    if (mj > mi) { Ph[(size_t)mi * Mp + mj] = th[tx][lr]; Pl[(size_t)mi * Mp + mj] = tl[tx][lr]; }
  }
}

// C (double-double) = A (double-double, n x n) B^T (fp64, n x n), all ld n, n a multiple of 64; tr: C is written transposed.  Plain fp64
// VALU: two_prod by fma, two_sum, no contraction -- 64 x 64 tiles of 256 threads, 4 x 4 outputs per thread, 16-deep k-chunks through
// LDS (tools/dd_gemm_proto.hip: both products of W = L^-1 Phi L^-T in 1.14 ms at n = 1024, 3e-18 of max |W| against long doubles).
// NI x NJ outputs per thread (tile 16 NI x 16 NJ).  Round 6: 4 x 4 (64 x 64 tiles: 256 workgroups at n = 1024, ONE wave per SIMD) -> 2 x 2,
// four waves per SIMD (dd_launch below; every output is still accumulated over k in order: the same bits).
constexpr int DDT = 64, DDK = 16;
template <int NI, int NJ>
__global__ __launch_bounds__(256) void dd_gemm_nt_kernel(const double* __restrict__ Ahi, const double* __restrict__ Alo,
                                                         const double* __restrict__ B, int n, double* __restrict__ Chi,
                                                         double* __restrict__ Clo, int tr, int kmode, int lower_only) {
#pragma clang fp contract(off)
  constexpr int TI = 16 * NI, TJ = 16 * NJ;
  __shared__ double sAh[DDK][TI + 1], sAl[DDK][TI + 1], sB[DDK][TJ + 1];
  // (column block rotated by a quarter of the grid per quarter of the rows: the workgroups a CU holds together -- ids 256 apart, i.e. the same
  // blockIdx.x, rows gridDim.y / 4 apart at n = 1024 -- then have k ranges of four different lengths; unrotated they all had the same one and
  // the triangular skipping below shortened the launch by 8 % for half the work)
  const int jrot = (int)((blockIdx.x + (gridDim.x / 4) * (blockIdx.y / (gridDim.y / 4 > 0 ? gridDim.y / 4 : 1))) % gridDim.x);
  const int i0 = blockIdx.y * TI, j0 = jrot * TJ, tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  // Both uses multiply by a TRIANGULAR B (L^-1 or its transpose) and read only the lower triangle of their second product (round 6):
  // kmode 1: B[j][k] = 0 for k < j -- the k loop starts at this tile's first column; kmode 2: B[j][k] = 0 for k > j -- it ends behind its
  // last one; lower_only: tiles strictly above the diagonal are not computed.  The skipped terms are exact zeros: the same bits.
  if (lower_only && i0 + TI <= j0) return;
  const int kbeg = kmode == 1 ? (j0 / DDK) * DDK : 0;
  const int kend = kmode == 2 ? ((j0 + TJ + DDK - 1) / DDK * DDK < n ? (j0 + TJ + DDK - 1) / DDK * DDK : n) : n;
  double hi[NI][NJ], lo[NI][NJ];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b) hi[a][b] = lo[a][b] = 0.0;
  for (int k0 = kbeg; k0 < kend; k0 += DDK) {
    for (int e = tid; e < TI * DDK; e += 256) {
      const int r = e / DDK, kk = e % DDK;
      sAh[kk][r] = Ahi[(size_t)(i0 + r) * n + k0 + kk];
      sAl[kk][r] = Alo[(size_t)(i0 + r) * n + k0 + kk];
    }
    for (int e = tid; e < TJ * DDK; e += 256) {
      const int r = e / DDK, kk = e % DDK;
      sB[kk][r] = B[(size_t)(j0 + r) * n + k0 + kk];
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < DDK; ++kk) {
      double ah[NI], al[NI], bv[NJ];
#pragma unroll
      for (int a = 0; a < NI; ++a) { ah[a] = sAh[kk][ti + 16 * a]; al[a] = sAl[kk][ti + 16 * a]; }
#pragma unroll
      for (int b = 0; b < NJ; ++b) bv[b] = sB[kk][tj + 16 * b];
#pragma unroll
      for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < NJ; ++b) {
          const double pr = ah[a] * bv[b];
          const double e = fma(ah[a], bv[b], -pr) + al[a] * bv[b];
          const double sum = hi[a][b] + pr;
          const double z = sum - hi[a][b];
          lo[a][b] += ((hi[a][b] - (sum - z)) + (pr - z)) + e;
          hi[a][b] = sum;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < NJ; ++b) {
      const double sum = hi[a][b] + lo[a][b];
      const double l = lo[a][b] - (sum - hi[a][b]);
      const int i = i0 + ti + 16 * a, j = j0 + tj + 16 * b;
      const size_t o = tr ? (size_t)j * n + i : (size_t)i * n + j;
      Chi[o] = sum;
      Clo[o] = l;
    }
}
static void dd_launch(const double* Ahi, const double* Alo, const double* B, int n, double* Chi, double* Clo, int tr, hipStream_t st, int kmode,
                      int lower_only) {
  static const int full = getenv("SGP_DD_FULL") ? atoi(getenv("SGP_DD_FULL")) : 0;   // A/B: the whole product, every tile (rounds 4-5)
  if (full) { kmode = 0; lower_only = 0; }
  // SGP_DD_TILE=44: the 64 x 64 tiles of rounds 4-5.  Same box, Phibar's two products at n = 1024: 1.39-1.40 ms (64 x 64), 1.14-1.16 (64 x 32),
  // 1.06-1.07 (32 x 32: four waves per SIMD) -- profiles/r06_dd_gemm_tile_ab.txt
  static const int shape = getenv("SGP_DD_TILE") ? atoi(getenv("SGP_DD_TILE")) : 22;
  if (shape == 44) dd_gemm_nt_kernel<4, 4><<<dim3(n / 64, n / 64), 256, 0, st>>>(Ahi, Alo, B, n, Chi, Clo, tr, kmode, lower_only);
  else dd_gemm_nt_kernel<2, 2><<<dim3(n / 32, n / 32), 256, 0, st>>>(Ahi, Alo, B, n, Chi, Clo, tr, kmode, lower_only);
}
// ---- Phibar in double-double (round 6; tests/studies/explicit_phibar_pass2.py) ---------------------------------
// The extended order's pass 2 takes the EXPLICIT Phibar = L^-T (C / 2 s2) L^-1, whose cond(K_uu)-sized entries cancel in Kbar = 2 K Phibar.
// The study separates three error sources against an 80-bit yardstick: the fp64 FORMATION of Phibar (two MFMA products whose own
// rounding is eps |L^-T| |C| |L^-1| >> eps |Phibar|) is what dominates today (a long-double product with the fp64-formed matrix is no
// better than the fp64 product); formed in double-double and rounded to one word the gradients are 5-15 x closer; with the low word
// applied as well (a 3-digit product suffices) 35-700 x.  This routine is the formation: Cs = C / (2 s2) padded, G = (L^-1)^T,
// Y^T = (Cs G^T)^T = (Cs L^-1)^T and Phibar = Y^T G^T = L^-T Cs L^-1 by the double-double VALU GEMM above (0.6 ms each at M = 1024).
__global__ __launch_bounds__(256) void phibar_dd_prep_kernel(const double* __restrict__ Cw, int M, int Mp, double scale,
                                                             const double* __restrict__ Linv, double* __restrict__ Cs,
                                                             double* __restrict__ zeros, double* __restrict__ G) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.y * 32, bj = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = 0; k < 4; ++k) {
    const int i = bi + ty + 8 * k, j = bj + tx;
    Cs[(size_t)i * Mp + j] = (i < M && j < M) ? scale * Cw[(size_t)i * M + j] : 0.0;
    zeros[(size_t)i * Mp + j] = 0.0;
    tile[ty + 8 * k][tx] = Linv[(size_t)i * Mp + j];
  }
  __syncthreads();
  for (int k = 0; k < 4; ++k) G[(size_t)(bj + ty + 8 * k) * Mp + bi + tx] = tile[tx][ty + 8 * k];
}
// (hi, lo) of the lower triangle on both sides, cropped to M x M (ld M)
__global__ __launch_bounds__(256) void phibar_dd_out_kernel(const double* __restrict__ Ph, const double* __restrict__ Pl, int M, int Mp,
                                                            double* __restrict__ out_hi, double* __restrict__ out_lo) {
  const int64_t total = (int64_t)M * M;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / M), c = (int)(e - (int64_t)r * M);
    const int64_t p = r >= c ? (int64_t)r * Mp + c : (int64_t)c * Mp + r;
    out_hi[e] = Ph[p];
    if (out_lo) out_lo[e] = Pl[p];
  }
}
extern "C" size_t sgp_phibar_dd_workspace_bytes(int M) {
  if (M <= 0 || M > SGP_MAX_INDUCING) return 0;
  const size_t Mp = (size_t)(M + 127) / 128 * 128;
  return 7 * Mp * Mp * sizeof(double) + 256;
}
extern "C" int sgp_phibar_dd(const double* Cw, const double* kuu_linv, int M, double s2, double* Phibar_hi, double* Phibar_lo, void* ws,
                             size_t ws_bytes, sgp_stream_t stream) {
  if (!Cw || !kuu_linv || !Phibar_hi || !(s2 > 0.0)) return SGP_ERR_ARG;
  if (M <= 0 || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (!ws || ws_bytes < sgp_phibar_dd_workspace_bytes(M)) return SGP_ERR_WORKSPACE;
  const int Mp = (M + 127) / 128 * 128;
  const size_t mm = (size_t)Mp * Mp;
  double* base = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
  double *Cs = base, *zeros = Cs + mm, *G = zeros + mm, *Yh = G + mm, *Yl = Yh + mm, *Ph = Yl + mm, *Pl = Ph + mm;
  hipStream_t st = (hipStream_t)stream;
  phibar_dd_prep_kernel<<<dim3(Mp / 32, Mp / 32), 256, 0, st>>>(Cw, M, Mp, 0.5 / s2, kuu_linv, Cs, zeros, G);
  dd_launch(Cs, zeros, G, Mp, Yh, Yl, 1, st, 1, 0);   // Y^T = (Cs L^-1)^T          (G[j][k] = L^-1[k][j]: zero for k < j)
  dd_launch(Yh, Yl, G, Mp, Ph, Pl, 0, st, 1, 1);      // Phibar = Y^T (L^-1) = L^-T Cs L^-1   (phibar_dd_out_kernel reads its lower triangle)
  phibar_dd_out_kernel<<<1024, 256, 0, st>>>(Ph, Pl, M, Mp, Phibar_hi, Phibar_lo);
  return check_launch();
}

// W (M x M, ld M) = scale * (Wh + Wl), the lower triangle's value on both sides (the two products round the two triangles differently)
__global__ __launch_bounds__(256) void ext_w_out_kernel(const double* __restrict__ Wh, const double* __restrict__ Wl, int M, int Mp,
                                                        double scale, double* __restrict__ W) {
  const int64_t total = (int64_t)M * M;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / M), j = (int)(e - (int64_t)i * M);
    const int r = i > j ? i : j, c = i > j ? j : i;
    W[e] = scale * (Wh[(size_t)r * Mp + c] + Wl[(size_t)r * Mp + c]);
  }
}

// phi_diag[i] = scale * Ph[i][i]: what the streaming-order estimate needs from Phi (the extended order returns W, not Phi)
__global__ void ext_phi_diag_kernel(const double* __restrict__ Ph, int M, int Mp, double scale, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) out[i] = scale * Ph[(size_t)i * Mp + i];
}

struct ExtWs {
  double *Xs, *ys, *Zs, *slab, *slab_lo, *bpart, *btmp, *yypart, *Ph, *Pl, *Yh, *Yl, *Wh, *Wl, *bpad, *upad;
  uint8_t* Q;
  size_t bytes;
};
static ExtWs carve_ext(void* ws, const StreamPlan& p, int64_t qrows) {
  Carver c(ws);
  ExtWs w;
  const size_t mm = (size_t)p.Mp * p.Mp, ns = (size_t)i8_nsplit(qrows > 0 ? qrows : 1, p.Mp) * p.ntiles * TILE * TILE;
  w.Xs = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1) * p.DP);
  w.ys = c.take<double>((size_t)(p.Npad > 0 ? p.Npad : 1));
  w.Zs = c.take<double>((size_t)p.Mp * p.DP);
  w.slab = c.take<double>(ns);
  w.slab_lo = c.take<double>(ns);
  w.bpart = c.take<double>((size_t)bpart_rows(p) * p.Mp);
  w.btmp = c.take<double>((size_t)BRED_G * p.Mp);
  w.yypart = c.take<double>(256);
  w.Ph = c.take<double>(mm); w.Pl = c.take<double>(mm);
  w.Yh = c.take<double>(mm); w.Yl = c.take<double>(mm);
  w.Wh = c.take<double>(mm); w.Wl = c.take<double>(mm);
  w.bpad = c.take<double>(p.Mp);
  w.upad = c.take<double>(p.Mp);
  w.Q = c.take<uint8_t>((size_t)(qrows > 0 ? qrows : 1) * p.Mp * 7);
  w.bytes = c.used();
  return w;
}

struct WhRowsWs {
  FwdWs f;
  double *R, *T;
  size_t bytes;
};
static WhRowsWs carve_wh_rows(void* ws, const StreamPlan& p, bool caller_t) {
  WhRowsWs w;
  w.f = carve_fwd(ws, p, true, 0);
  Carver c(ws ? static_cast<char*>(ws) + round_up64((int64_t)w.f.bytes, 256) : nullptr);
  w.R = c.take<double>((size_t)p.Mp * p.Mp);
  w.T = caller_t ? nullptr : c.take<double>((size_t)(p.sc_rows > 0 ? p.sc_rows : 1) * p.Mp);
  w.bytes = (size_t)round_up64((int64_t)w.f.bytes, 256) + c.used();
  return w;
}

}  // namespace sgp

using namespace sgp;

static int ctx_timing_last_ms(Ctx& c, int slot, float* ms) {
  if (slot < 0 || slot >= TIMING_SLOTS || !ms) return SGP_ERR_ARG;
  if (!c.ev_ready || !c.ev_used[slot]) return SGP_ERR_ARG;
  if (hipEventSynchronize(c.ev[slot][1]) != hipSuccess) return SGP_ERR_LAUNCH;
  return hipEventElapsedTime(ms, c.ev[slot][0], c.ev[slot][1]) == hipSuccess ? SGP_OK : SGP_ERR_LAUNCH;
}
extern "C" int sgp_ctx_timing_last_ms(sgp_ctx* ctx, int slot, float* ms) {
  return ctx_timing_last_ms(ctx ? *reinterpret_cast<Ctx*>(ctx) : default_ctx(), slot, ms);
}
extern "C" int64_t sgp_ctx_timing_last_rows(const sgp_ctx* ctx, int slot) {
  return slot == TIMING_SYRK ? (ctx ? *reinterpret_cast<const Ctx*>(ctx) : default_ctx()).syrk_timed_rows : -1;
}
extern "C" int sgp_ctx_contraction_would_use_i8(const sgp_ctx* ctx, int64_t N, int M) {
  if (N <= 0 || M <= 0 || M > SGP_MAX_INDUCING) return 0;
  return i8_rule((ctx ? *reinterpret_cast<const Ctx*>(ctx) : default_ctx()).contraction, round_up64(N, ASM_ROWS), padded_m(M)) ? 1 : 0;
}

// ---- deprecated process-wide switches: shims over the DEFAULT context (include/sgp.h) ----
extern "C" void sgp_timing_enable(int on) { default_ctx().timing = on ? 1 : 0; }
extern "C" int sgp_timing_last_ms(int slot, float* ms) { return ctx_timing_last_ms(default_ctx(), slot, ms); }
extern "C" int64_t sgp_timing_last_rows(int slot) { return sgp_ctx_timing_last_rows(nullptr, slot); }
extern "C" void sgp_set_asm_overlap(int mode) { default_ctx().asm_overlap = (mode < 0 || mode > 2) ? 0 : mode; }
extern "C" int sgp_set_contraction(int mode) {
  const int prev = default_ctx().contraction;
  default_ctx().contraction = (mode < 0 || mode > 2) ? 1 : mode;
  return prev;
}
extern "C" int sgp_contraction_last(void) { return default_ctx().contraction_used; }
extern "C" void sgp_set_pass1_gate(void* hip_event) { default_ctx().pass1_gate = (hipEvent_t)hip_event; }
extern "C" void sgp_set_kfu_budget_bytes(size_t bytes) { default_ctx().kfu_budget = bytes ? bytes : KFU_BUDGET_DEFAULT; }

extern "C" size_t sgp_kfu_len(int64_t N, int M) {
  if (N < 0 || M <= 0 || M > SGP_MAX_INDUCING) return 0;
  return (size_t)round_up64(N > 0 ? N : 1, ASM_ROWS) * padded_m(M);
}

static size_t fwd_workspace_bytes(int64_t N, int M, int d, bool library_kfu) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  StreamPlan p = make_stream_plan(N, M, d);
  const size_t fast = carve_fwd(nullptr, p, library_kfu, p.sc_rows).bytes, comp = comp_fwd_workspace_bytes(N, M);  // one size for every kernel_id
  return fast > comp ? fast : comp;
}
extern "C" size_t sgp_suffstats_workspace_bytes(int64_t N, int M, int d) { return fwd_workspace_bytes(N, M, d, true); }
extern "C" size_t sgp_suffstats_workspace_bytes_ex(int64_t N, int M, int d, int caller_owns_kfu) {
  return fwd_workspace_bytes(N, M, d, caller_owns_kfu == 0);
}

extern "C" int sgp_suffstats_fwd(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                 const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                 double* Phi, double* b, double* yy, double* kappa, double* Kfu_out, void* ws,
                                 size_t ws_bytes, sgp_stream_t stream) {
  // the one-shot gate belongs to THIS call whatever path it takes (an early return must not leave a handle behind for a later
  // call to wait on: the caller's event may be destroyed by then)
  Ctx& cx = cur_ctx();
  hipEvent_t gate = cx.pass1_gate;
  cx.pass1_gate = nullptr;
  if (!Z || !inv_ls || !Phi || !b || !yy || !kappa || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id > SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  if (kernel_id == SGP_KERNEL_COMPOSITE) {  // inv_ls carries the parameter block; materialised path, Kfu_out unused
    CompSpec cs;
    if (comp_parse(inv_ls, d, &cs) != SGP_OK) return SGP_ERR_ARG;
    return comp_suffstats_fwd(X, ldx, y, Z, ldz, cs, N, M, d, Phi, b, yy, kappa, ws, ws_bytes, (hipStream_t)stream);
  }
  StreamPlan p = make_stream_plan(N, M, d);
  const int64_t qrows = p.sc_rows;  // super-chunk of the digit planes (the library's K'_fu budget also when the caller owns K'_fu)
  if (Kfu_out) p.sc_rows = p.Npad;  // caller keeps the whole K'_fu: one super-chunk
  FwdWs w = carve_fwd(ws, p, Kfu_out == nullptr, qrows);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;

  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;

  stream_prologue(p, ka, X, ldx, y, Z, ldz, N, M, w.Xs, w.ys, w.Zs, w.yypart, st);
  double* Kfu = Kfu_out ? Kfu_out : w.Kfu;
  const int grid = p.ntiles * p.nsplit;
  if (p.Npad == 0) {
    // empty shard: run the contraction over zero chunks so every slab tile is written (zeros)
    syrk_tile_kernel<4, false, false><<<grid, 256, 0, st>>>(Kfu, p.Mp, 0, SplitMap{{0, 0, 0, 0}, 1}, p.ntiles, 0, w.slab, p.nsplit);
  }
  auto contract = [&](const double* K, int64_t nchunks, const SplitMap& smap, int nsplit, int accum, double* slab) {
    launch_syrk(cx, K, p.Mp, nchunks, smap, p.ntiles, nsplit, accum, slab, st);
  };
  int nslabs = p.nsplit;
  int head_ns = 0;
  int64_t head_chunks = 0;
  // With a kept K'_fu (value + gradient: pass 2 reads the fp64 block) the assembly writes the fp64 block AND the planes.  Pass 1 alone
  // gains 3.0 ms at C5 (2.6 + 12.9 against 1.7 + 16.0 ms), the leapfrog it belongs to 1.1 ms (50.4 against 51.5, same box, alternating:
  // profiles/r03_i8_leapfrog_ab.jsonl) -- pass 2 runs 1.4 ms longer behind the integer contraction, whose power draw it inherits.
  const bool use_i8 = i8_rule(cx.contraction, p.Npad, p.Mp);
  cx.contraction_used = use_i8 ? 1 : 0;
  if (!use_i8) gate = nullptr;  // the fp64 contraction shares the chip with a side stream: no gate
  if (p.Npad > 0 && !use_i8) head_block(p, p.Npad / NB, &head_ns, &head_chunks);
  if (use_i8) {
    // digit planes (7 bytes per element) live where the library's fp64 K'_fu (8 bytes) would; with a caller-owned K'_fu (value +
    // gradient: pass 2 reads the fp64 block) the assembly writes both and the planes take their own super-chunk of workspace.
    // The split count is that of a full super-chunk.
    const int ns = i8_nsplit(qrows, p.Mp);
    for (int64_t r0 = 0; r0 < p.Npad; r0 += qrows) {
      const int64_t rows = (p.Npad - r0) < qrows ? (p.Npad - r0) : qrows;
      timing_begin(TIMING_ASSEMBLE, st);
      i8_assemble(p, kernel_id, w.Xs, w.ys, w.Zs, r0, rows, N, M, w.Q, Kfu_out ? Kfu_out + (size_t)r0 * p.Mp : nullptr, w.bpart, st);
      timing_end(TIMING_ASSEMBLE, st);
      // the integer contraction leaves next to nothing for anybody else (129 KB of LDS, two 188-register waves per SIMD): a side-stream chain
      // the caller wants done by the end of pass 1 (chol(K_uu)) has to finish beside the ASSEMBLY, so the contraction waits for it
      if (gate && hipStreamWaitEvent(st, gate, 0) != hipSuccess) return SGP_ERR_LAUNCH;
      gate = nullptr;
      timing_begin(TIMING_SYRK, st);
      if (i8_contract(w.Q, p.Mp, rows, ns, r0 > 0 ? 1 : 0, w.slab, st) != SGP_OK) return SGP_ERR_LAUNCH;
      timing_end(TIMING_SYRK, st);
      cx.syrk_timed_rows = rows;
    }
    nslabs = ns;
  } else if (head_ns > 0) {
    // head block [0, hrows) | tail [hrows, Npad): the tail's assembly runs on the side stream beside the head's contraction
    // (serially on the main stream when the side stream cannot be had or SGP_ASM_OVERLAP=2: same blocks, same numbers)
    const int64_t hrows = head_chunks * NB, trows = p.Npad - hrows, tchunks = trows / NB;
    const bool side = cx.asm_overlap == 1 && side_stream_ready(cx);
    double* head_slab = w.slab + (size_t)p.nsplit * p.ntiles * TILE * TILE;
    const int hcps = (int)((head_chunks + head_ns - 1) / head_ns);
    const int tcps = (int)((tchunks + p.nsplit - 1) / p.nsplit);
    const SplitMap hmap{{0, 0, 0, 0}, hcps < 1 ? 1 : hcps};
    const SplitMap tmap{{p.taper[0], p.taper[1], p.taper[2], p.taper[3]}, tcps < 1 ? 1 : tcps};
    timing_begin(TIMING_ASSEMBLE, st);
    stream_assemble(p, kernel_id, w.Xs, w.ys, w.Zs, 0, hrows, N, M, Kfu, w.bpart, st);
    bool forked = false;
    if (side && hipEventRecord(cx.ev_fork, st) == hipSuccess && hipStreamWaitEvent(cx.side, cx.ev_fork, 0) == hipSuccess) {
      stream_assemble(p, kernel_id, w.Xs, w.ys, w.Zs, hrows, trows, N, M, Kfu + (size_t)hrows * p.Mp, w.bpart, cx.side);
      forked = hipEventRecord(cx.ev_join, cx.side) == hipSuccess;
      if (!forked) (void)hipStreamSynchronize(cx.side);  // never expected: fall back to a blocking join
    } else {
      stream_assemble(p, kernel_id, w.Xs, w.ys, w.Zs, hrows, trows, N, M, Kfu + (size_t)hrows * p.Mp, w.bpart, st);
    }
    timing_end(TIMING_ASSEMBLE, st);  // overlapped: the head's assembly only (what stays on the critical path)
    contract(Kfu, head_chunks, hmap, head_ns, 0, head_slab);
    if (forked && hipStreamWaitEvent(st, cx.ev_join, 0) != hipSuccess) (void)hipStreamSynchronize(cx.side);
    timing_begin(TIMING_SYRK, st);  // the dominant launch: the tail's contraction, alone on the device
    contract(Kfu + (size_t)hrows * p.Mp, tchunks, tmap, p.nsplit, 0, w.slab);
    timing_end(TIMING_SYRK, st);
    cx.syrk_timed_rows = trows;
    nslabs = p.nsplit + head_ns;
  } else {
    for (int64_t r0 = 0; r0 < p.Npad; r0 += p.sc_rows) {
      const int64_t rows = (p.Npad - r0) < p.sc_rows ? (p.Npad - r0) : p.sc_rows;
      timing_begin(TIMING_ASSEMBLE, st);
      stream_assemble(p, kernel_id, w.Xs, w.ys, w.Zs, r0, rows, N, M, Kfu, w.bpart, st);
      timing_end(TIMING_ASSEMBLE, st);
      const int64_t nchunks = rows / NB;
      const int cps = (int)((nchunks + p.nsplit - 1) / p.nsplit);
      const SplitMap smap{{p.taper[0], p.taper[1], p.taper[2], p.taper[3]}, cps < 1 ? 1 : cps};
      timing_begin(TIMING_SYRK, st);
      contract(Kfu, nchunks, smap, p.nsplit, r0 > 0 ? 1 : 0, w.slab);
      timing_end(TIMING_SYRK, st);
      cx.syrk_timed_rows = rows;
    }
  }
  const int nb32 = p.Mp / 32;
  reduce_phi_kernel<<<nb32 * (nb32 + 1) / 2, 256, 0, st>>>(w.slab, nslabs, p.ntiles, M, sf2 * sf2, Phi);
  // (the fp64 assembly leaves asm_sub partials per row block, the integer path's assembly one)
  bpart_stage1_kernel<<<dim3(p.Mp / 64, BRED_G), 256, 0, st>>>(w.bpart, p.Npad / ASM_ROWS * (use_i8 ? 1 : p.asm_sub), p.Mp, BRED_G, w.btmp);
  finalize_stats_kernel<<<(M + 255) / 256, 256, 0, st>>>(w.btmp, BRED_G, p.Mp, M, w.yypart, 256, sf2,
                                                         sf2 * (double)N, b, yy, kappa);
  return check_launch();
}

// ---- whitened pass 1 in the streaming layout ---------------------------------------------------------------------------------
// W = A A^T, u = A y with A = L^-1 K_uf (PyMC3's op order) for a LARGE shard: sgp_suffstats_fwd's fp64 pipeline with T = K'_fu L^-T in
// place of K'_fu -- assembly (HBM write bound), ONE N M^2 / 2 product T = K' R with R = L^-T upper triangular (gemm, k clipped to the
// triangle), u's partials (one read of T), the tuned fp64 contraction T^T T, the fixed-order reductions with sf2^2 / sf2 applied there.
// sgp_suffstats_fwd_whitened (sgp_tail.hip) walks the same products in 32768-column chunks of an M x T layout: 31 x 4 launches at
// N = 10^6 whose 136-tile W += A A^T leaves half the chip idle (64.7 ms of kernels at C5 against ~45 here).
// T_out (optional, sgp_kfu_len(N, M) doubles): T (unit amplitude, no sf2) is left there for sgp_suffstats_bwd_factored_ex, which then
// needs neither the assembly nor the product again.
extern "C" size_t sgp_suffstats_whitened_rows_workspace_bytes(int64_t N, int M, int d, int caller_owns_t) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  StreamPlan p = make_stream_plan(N, M, d);
  if (caller_owns_t) p.sc_rows = p.Npad;
  return carve_wh_rows(nullptr, p, caller_owns_t != 0).bytes;
}
extern "C" int sgp_suffstats_fwd_whitened_rows(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                               const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                               const double* kuu_linv, double* W, double* u, double* yy, double* kappa, double* T_out,
                                               void* ws, size_t ws_bytes, sgp_stream_t stream) {
  Ctx& cx = cur_ctx();
  if (!Z || !inv_ls || !kuu_linv || !W || !u || !yy || !kappa || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id >= SGP_KERNEL_COMPOSITE) return SGP_ERR_ARG;  // the composite kernel keeps the chunked routine
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  StreamPlan p = make_stream_plan(N, M, d);
  if (T_out) p.sc_rows = p.Npad;  // the caller keeps all of T: one super-chunk
  WhRowsWs w = carve_wh_rows(ws, p, T_out != nullptr);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;
  stream_prologue(p, ka, X, ldx, y, Z, ldz, N, M, w.f.Xs, w.f.ys, w.f.Zs, w.f.yypart, st);
  transpose_square(kuu_linv, p.Mp, w.R, st);  // R = L^-T as a plain row-major operand
  if (p.Npad == 0)
    syrk_tile_kernel<4, false, false><<<p.ntiles * p.nsplit, 256, 0, st>>>(w.f.Kfu, p.Mp, 0, SplitMap{{0, 0, 0, 0}, 1}, p.ntiles, 0, w.f.slab,
                                                                           p.nsplit);
  for (int64_t r0 = 0; r0 < p.Npad; r0 += p.sc_rows) {
    const int64_t rows = (p.Npad - r0) < p.sc_rows ? (p.Npad - r0) : p.sc_rows;
    double* T = T_out ? T_out + (size_t)r0 * p.Mp : w.T;
    timing_begin(TIMING_ASSEMBLE, st);
    stream_assemble(p, kernel_id, w.f.Xs, w.f.ys, w.f.Zs, r0, rows, N, M, w.f.Kfu, w.f.bpart, st);  // (its K'^T y partials are overwritten below)
    timing_end(TIMING_ASSEMBLE, st);
    GemmDesc g;  // T = K' R: column block c of the upper-triangular R needs k < its end only
    g.A = w.f.Kfu; g.lda = p.Mp; g.B = w.R; g.ldb = p.Mp; g.C = T; g.ldc = p.Mp;
    g.m = (int)rows; g.n = p.Mp; g.k = p.Mp; g.khi_mask = 2;
    gemm(g, st);
    tpart_kernel<<<dim3((unsigned)(rows / ASM_ROWS), (p.Mp + 255) / 256), 256, 0, st>>>(T, w.f.ys, r0, p.Mp, w.f.bpart);
    const int64_t nchunks = rows / NB;
    const int cps = (int)((nchunks + p.nsplit - 1) / p.nsplit);
    const SplitMap smap{{p.taper[0], p.taper[1], p.taper[2], p.taper[3]}, cps < 1 ? 1 : cps};
    timing_begin(TIMING_SYRK, st);
    launch_syrk(cx, T, p.Mp, nchunks, smap, p.ntiles, p.nsplit, r0 > 0 ? 1 : 0, w.f.slab, st);
    timing_end(TIMING_SYRK, st);
    cx.syrk_timed_rows = rows;
  }
  const int nb32 = p.Mp / 32;
  reduce_phi_kernel<<<nb32 * (nb32 + 1) / 2, 256, 0, st>>>(w.f.slab, p.nsplit, p.ntiles, M, sf2 * sf2, W);
  bpart_stage1_kernel<<<dim3(p.Mp / 64, BRED_G), 256, 0, st>>>(w.f.bpart, p.Npad / ASM_ROWS, p.Mp, BRED_G, w.f.btmp);
  finalize_stats_kernel<<<(M + 255) / 256, 256, 0, st>>>(w.f.btmp, BRED_G, p.Mp, M, w.f.yypart, 256, sf2, sf2 * (double)N, u, yy, kappa);
  return check_launch();
}


// ---- the extended streaming order ---------------------------------------------------------------------------------------------
// The same whitened statistics [W = A A^T | u = A y | yy | kappa], A = L^-1 K_uf, from the STREAMING design: Phi = K'^T K' on the integer
// matrix cores with 34 (level 1) or 39 (level 2: the product's default) digit pairs and a double-double fold / slab reduction (exact sums of the fixed-point kernel values to 2^-61 / 2^-69 of the
// largest entry instead of fp64's 2^-53), then W = L^-1 Phi L^-T by two double-double products, u = L^-1 b in fp64 (a CPU study finds it
// harmless, tests/studies/extended_streaming_order.py).  What the explicit-inverse sandwich amplifies is 2^8 times smaller than in
// sgp_suffstats_fwd + sgp_bound_from_stats, so the streaming order's guard (sgp_streaming_error_estimate) passes 256 times later; the cost
// is 14.0 instead of 11.7 ms of contraction and ~1.5 ms of tail at C5 against the whitened order's two extra N M^2 products.
// Stationary kernels; rows x Mp^2 of any size (the integer contraction is used whatever the context's contraction mode says).
// Kfu_out (optional, sgp_kfu_len doubles): the fp64 K'_fu for sgp_suffstats_bwd (explicit Phibar -- good to ~3 x the tolerance only,
// include/sgp.h).  level 1: 34 digit pairs, 2: 39.
extern "C" size_t sgp_suffstats_extended_workspace_bytes(int64_t N, int M, int d) {
  if (N < 0 || M <= 0 || d <= 0 || d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return 0;
  StreamPlan p = make_stream_plan(N, M, d);
  return carve_ext(nullptr, p, p.sc_rows).bytes;
}
extern "C" int sgp_suffstats_fwd_extended(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                          const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                          const double* kuu_linv, int level, double* W, double* u, double* yy, double* kappa,
                                          double* Kfu_out, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  return sgp_suffstats_fwd_extended_ex(X, ldx, y, Z, ldz, inv_ls, sf2, N, M, d, kernel_id, kuu_linv, level, W, u, yy, kappa, Kfu_out,
                                       nullptr, ws, ws_bytes, stream);
}
// ... with one more output (ABI 3): phi_diag (DEVICE, M doubles, or NULL) = diag(K_uf K_fu) of THIS shard, with its amplitude -- what
// sgp_streaming_error_report needs to state the streaming-order estimate exactly at this theta (ranks add their phi_diag up).
extern "C" int sgp_suffstats_fwd_extended_ex(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                             const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                             const double* kuu_linv, int level, double* W, double* u, double* yy, double* kappa,
                                             double* Kfu_out, double* phi_diag, void* ws, size_t ws_bytes, sgp_stream_t stream) {
  return sgp_suffstats_fwd_extended_f16(X, ldx, y, Z, ldz, inv_ls, sf2, N, M, d, kernel_id, kuu_linv, level, W, u, yy, kappa, Kfu_out, nullptr,
                                        phi_diag, ws, ws_bytes, stream);
}
// ... and one more (round 6): Kfu_f16_out (DEVICE, sgp_kfu_len(N, M) 16-bit words, or NULL; needs Kfu_out): the fp16 image of K'_fu that
// sgp_suffstats_bwd_lo_f16 multiplies, written by the assembly kernel that has every value in registers anyway.
extern "C" int sgp_suffstats_fwd_extended_f16(const double* X, int64_t ldx, const double* y, const double* Z, int64_t ldz,
                                              const double* inv_ls, double sf2, int64_t N, int M, int d, int kernel_id,
                                              const double* kuu_linv, int level, double* W, double* u, double* yy, double* kappa,
                                              double* Kfu_out, uint16_t* Kfu_f16_out, double* phi_diag, void* ws, size_t ws_bytes,
                                              sgp_stream_t stream) {
  if (Kfu_f16_out && !Kfu_out) return SGP_ERR_ARG;
  if (!Z || !inv_ls || !kuu_linv || !W || !u || !yy || !kappa || N < 0 || M <= 0 || d <= 0 || ldz < d) return SGP_ERR_ARG;
  if (N > 0 && (!X || !y || ldx < d)) return SGP_ERR_ARG;
  if (kernel_id < 0 || kernel_id >= SGP_KERNEL_COMPOSITE || level < 1 || level > 2) return SGP_ERR_ARG;
  if (d > SGP_MAX_DIM || M > SGP_MAX_INDUCING) return SGP_ERR_DIM;
  StreamPlan p = make_stream_plan(N, M, d);
  const int64_t qrows = p.sc_rows;
  ExtWs w = carve_ext(ws, p, qrows);
  if (!ws || ws_bytes < w.bytes) return SGP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  KernArgs ka;
  for (int j = 0; j < SGP_MAX_DIM; ++j) ka.inv_ls[j] = j < d ? inv_ls[j] : 0.0;
  ka.sf2 = sf2;
  ka.d = d;
  stream_prologue(p, ka, X, ldx, y, Z, ldz, N, M, w.Xs, w.ys, w.Zs, w.yypart, st);
  const int ns = i8_nsplit(qrows > 0 ? qrows : 1, p.Mp);
  const size_t nslab = (size_t)ns * p.ntiles * TILE * TILE;
  if (p.Npad == 0) {
    fill_zero(w.slab, nslab, st);
    fill_zero(w.slab_lo, nslab, st);
  }
  for (int64_t r0 = 0; r0 < p.Npad; r0 += qrows) {
    const int64_t rows = (p.Npad - r0) < qrows ? (p.Npad - r0) : qrows;
    timing_begin(TIMING_ASSEMBLE, st);
    i8_assemble(p, kernel_id, w.Xs, w.ys, w.Zs, r0, rows, N, M, w.Q, Kfu_out ? Kfu_out + (size_t)r0 * p.Mp : nullptr, w.bpart, st,
                Kfu_f16_out ? Kfu_f16_out + (size_t)r0 * p.Mp : nullptr);
    timing_end(TIMING_ASSEMBLE, st);
    timing_begin(TIMING_SYRK, st);
    if (i8_contract(w.Q, p.Mp, rows, ns, r0 > 0 ? 1 : 0, w.slab, st, w.slab_lo, level) != SGP_OK) return SGP_ERR_LAUNCH;
    timing_end(TIMING_SYRK, st);
  }
  const int nb32 = p.Mp / 32;
  reduce_phi_dd_kernel<<<nb32 * (nb32 + 1) / 2, 256, 0, st>>>(w.slab, w.slab_lo, ns, p.ntiles, p.Mp, w.Ph, w.Pl);
  dd_launch(w.Ph, w.Pl, kuu_linv, p.Mp, w.Yh, w.Yl, 1, st, 2, 0);  // Y^T = (Phi L^-T)^T     (L^-1[j][k]: zero for k > j)
  dd_launch(w.Yh, w.Yl, kuu_linv, p.Mp, w.Wh, w.Wl, 0, st, 2, 1);  // W = Y^T L^-T = L^-1 Phi L^-T (symmetric: ext_w_out_kernel reads the lower triangle)
  ext_w_out_kernel<<<1024, 256, 0, st>>>(w.Wh, w.Wl, M, p.Mp, sf2 * sf2, W);
  if (phi_diag) ext_phi_diag_kernel<<<(M + 255) / 256, 256, 0, st>>>(w.Ph, M, p.Mp, sf2 * sf2, phi_diag);
  // b = K_uf y (fp64, with its amplitude), u = L^-1 b
  fill_zero(w.bpad, p.Mp, st);
  bpart_stage1_kernel<<<dim3(p.Mp / 64, BRED_G), 256, 0, st>>>(w.bpart, p.Npad / ASM_ROWS, p.Mp, BRED_G, w.btmp);
  finalize_stats_kernel<<<(M + 255) / 256, 256, 0, st>>>(w.btmp, BRED_G, p.Mp, M, w.yypart, 256, sf2, sf2 * (double)N, w.bpad, yy, kappa);
  gemv(kuu_linv, p.Mp, p.Mp, false, w.bpad, w.upad, st);
  crop_copy(w.upad, 1, u, 1, M, 1, st);
  return check_launch();
}
