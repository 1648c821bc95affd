// Pass-1 contraction on the INTEGER matrix cores: Phi = K'^T K' by an error-free splitting of K'.
//
// Every entry of K'_fu lies in [0, 1] (a stationary profile without its amplitude), so it IS a fixed-point number:
//
//   q = rint(K' 2^54) = sum_p a_p 256^p,   a_p in [-128, 127],  p = 0..6      (balanced digits: the bytes of (q + C) ^ C,
//                                                                              C = 0x80 in each of the seven bytes)
//   Phi_IJ = 2^-108 sum_n q_nI q_nJ = 2^-108 sum_{p + r >= 6} 256^(p + r) sum_n a_p,nI a_r,nJ  +  truncation
//
// * |K' - q 2^-54| <= 2^-55: exact for K' >= 1/4, an ABSOLUTE error below a quarter ulp of 1.0 otherwise.  (Round 3 scaled by 2^53;
//   seven balanced digits hold 2^55, and K' <= 1 (+ an ulp) needs q <= 2^54 + 2, top digit <= 65: the scale 2^54 is free and makes
//   the truncation below four times smaller.)
// * digit products are exact in int32: a group g = p + r - 6 holds 7 - g pairs of |a_p a_r| <= 2^14 -- 7 x 2^14 x 16384 rows < 2^31,
//   so a split never spans more than I8_SPLIT_ROWS = 16384 rows; the fold to fp64 (7 terms) happens once per split and tile.
// * 28 of the 49 digit pairs are kept (p + r >= 6); the dropped ones are < 6 x 2^-54 per product in the worst case, zero-mean
//   (balanced digits) and ~2^-54 typically: below the rounding of ONE fp64 product, where the fp64 SYRK rounds every one of its
//   N accumulation steps.  Measured against long-double arithmetic (tools/i8_syrk_proto.hip, at the 2^53 scale): 2.4-2.8e-16 of
//   max |Phi|.  It is an ABSOLUTE error (against K' <= 1): component-wise statement in DESIGN.md 4d / tests/test_int8_theta_sweep.py.
//
// Why: v_mfma_i32_32x32x32_i8 runs 32 768 MACs in 32 cycles against 1 024 in 64 for v_mfma_f64_16x16x4_f64 -- 64 x the rate for
// 28 x the MACs.  What is left of that on real operands is decided by POWER, not issue slots: on full-entropy bytes the chip
// clocks the int8 pipe down to ~1.78 GHz (profiles/r03_i8_rates.txt: 1 764 TMAC/s sustained), and every byte moved costs
// clock as well.  Measured at C5 (N = 10^6, M = 1024): 12.0-12.4 ms against 16.0-16.9 ms for the fp64 contraction.
//
// Layout.  Digit planes Q[rb][p][m][16 bytes]: digit p of data rows 16 rb .. 16 rb + 15 of inducing column m -- 7 bytes per
// element (the fp64 K'_fu has 8) and ONE ds_read_b128 is a lane's whole MFMA operand (lane l <-> column l % 32, rows
// 16 (l / 32) + 0..15 of a 32-row step; both operands are the same K' rows, so the k-order inside an operand cancels).
// Workgroup = 128 x 64 tile of the lower triangle x one split of the rows; 8 waves (two per SIMD), each one 32 x 32 MFMA tile x 7
// group accumulators = 112 accumulator registers.  32-row stages (42 KB) travel global -> LDS by LDS-DMA through a ring of three;
// the waves run as two groups half a step apart so that one of them always has its operands in registers (see i8_tile_loop).
// Measured alternatives (register staging, lockstep waves, held-back MFMAs, split-major order): tools/i8_syrk_proto.hip, DESIGN.md 4d.
#include <atomic>
#include "sgp_common.hpp"
#include "sgp_stream.hpp"
#include "sgp_ctx.hpp"
#include <cstdlib>
#include <type_traits>

namespace sgp {

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

constexpr int I8_NP = 7;                                  // digit planes
constexpr int I8_TR = 128, I8_TC = 64;                    // tile of Phi per workgroup
constexpr int I8_SCOLS = I8_TR + I8_TC;                   // columns staged per workgroup
constexpr int I8_STAGE_BYTES = 2 * I8_NP * I8_SCOLS * 16; // 32 rows = 2 row blocks of 16
constexpr int I8_NSTAGE = 3;
constexpr int I8_PIECES = 2 * I8_NP * 3;                  // 1 KB LDS-DMA pieces per stage (3 groups of 64 columns)
constexpr int I8_PPW = (I8_PIECES + 7) / 8;               // pieces per wave and stage (8 waves; the last slots repeat a piece)
constexpr int I8_LDS_BYTES = I8_NSTAGE * I8_STAGE_BYTES;  // 129 024

// ---------------------------------------------------------------------------------------------
// 1. kernel assembly into digit planes:  thread <-> inducing column m, 256 rows per workgroup (as kfu_assemble_kernel);
//    per 16 rows a thread stores seven 16-byte vectors, consecutive threads consecutive addresses (4 KB per plane and workgroup).
//    bpart[rowblock][m] = sum_n k'(x_n, z_m) y_n falls out of the same loop in fp64 (the digits are not involved).
// ---------------------------------------------------------------------------------------------
//    WK: the fp64 block K'_fu is written as well (a value + gradient evaluation: pass 2 reads it) -- 15 instead of 7 bytes per
//    element, which puts the kernel back at the HBM-write ceiling.  Kh (optional, with WK): its fp16 image too (17 bytes).
template <int DP, int KID, bool WK>
__global__ __launch_bounds__(256) void kfu_digits_kernel(const double* __restrict__ Xs, const double* __restrict__ ys,
                                                         const double* __restrict__ Zs, int64_t row0, int64_t N, int M, int Mp,
                                                         uint8_t* __restrict__ Q, double* __restrict__ Kfu, uint16_t* __restrict__ Kh,
                                                         double* __restrict__ bpart) {
  __shared__ double xs[ASM_ROWS][DP];
  __shared__ double ysh[ASM_ROWS];
  __shared__ double etab[EXP_TAB_N];
  const int64_t rbase = (int64_t)blockIdx.x * ASM_ROWS;  // row inside this super-chunk's Q
  {
    const double* src = Xs + (row0 + rbase) * DP;
    double* dst = &xs[0][0];
    for (int e = threadIdx.x; e < ASM_ROWS * DP; e += 256) dst[e] = src[e];
    ysh[threadIdx.x] = ys[row0 + rbase + threadIdx.x];
    sgp_exp_tab_load(etab);
  }
  __syncthreads();
  const int m = blockIdx.y * 256 + threadIdx.x;
  if (m >= Mp) return;
  const double zmask = m < M ? 1.0 : 0.0;
  double zr[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) zr[j] = Zs[(size_t)m * DP + j];

  constexpr unsigned long long C = 0x0080808080808080ULL;
  double bacc = 0.0;
  // interior blocks (every row < N, every column < M: all but the last row block and the padding columns) skip the mask -- the
  // kernel is VALU-bound, and the compare + select + multiply per element are 3 of its ~48 lane-operations
#ifdef SGP_AB_DIGITS_NO_INTERIOR  // A/B (tools/ab_build.sh): one masked loop for every block
  const bool interior = false;
#else
  const bool interior = (row0 + rbase + ASM_ROWS <= N) && ((int)(blockIdx.y * 256 + 255) < M);
#endif
  auto run = [&](auto masked_tag) {
  constexpr bool MASKED = decltype(masked_tag)::value;
  for (int g = 0; g < ASM_ROWS / 16; ++g) {
    unsigned lo[16], hi[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = g * 16 + e;
      const int64_t n = row0 + rbase + i;
      double r2 = 0.0;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const double df = xs[i][j] - zr[j];
        r2 = fma(df, df, r2);
      }
      double kv = kprofile_tab<KID>(r2, etab);
      if constexpr (MASKED) kv *= (n < N ? zmask : 0.0);
      bacc = fma(kv, ysh[i], bacc);
      if constexpr (WK) {
        __builtin_nontemporal_store(kv, &Kfu[(rbase + i) * Mp + m]);
        // (round 6) the fp16 image of the same block for sgp_suffstats_bwd_lo_f16, rounded as lo_kfu_f16_kernel rounds it: 2 more bytes per
        // element here instead of a 10 GB pass of its own there
        if (Kh) {
          float kf = (float)kv;
          asm volatile("" : "+v"(kf));   // (two roundings, as lo_kfu_f16_kernel: left alone hipcc folds them into ONE software double -> half conversion)
          Kh[(rbase + i) * Mp + m] = __builtin_bit_cast(uint16_t, (_Float16)kf);
        }
      }
      // q = rint(kv 2^54) without a 64-bit convert: hi = rint(kv 2^22) and the SIGNED remainder r = rint(kv 2^54 - hi 2^32) in
      // [-2^31, 2^31], each read off the mantissa of a magic-constant sum (all four operations exact).  r sits in the low 33
      // mantissa bits of tl as a two's-complement number: q = (hi - bit32) 2^32 + low32 -- also at the ties r = +-2^31, which a
      // 32-bit reading would get wrong about once in 2^33 elements.  A NaN distance (NaN in X / Z / a lengthscale) gives q = 0 --
      // it reaches the bound through b (fp64) instead; a Matern value an ulp above 1 gives q = 2^54 + 2, which seven digits hold.
      const double th = fma(kv, 0x1p22, 0x1p52);
      const double hf = th - 0x1p52;
      const double tl = fma(-hf, 0x1p32, kv * 0x1p54) + 0x1.8p52;
      const unsigned long long tb = (unsigned long long)__double_as_longlong(tl);
      const unsigned q_hi = (unsigned)__double_as_longlong(th) - ((unsigned)(tb >> 32) & 1u);
      const unsigned long long q = ((unsigned long long)q_hi << 32) | (unsigned)tb;
      const unsigned long long qq = (q + C) ^ C;
      lo[e] = (unsigned)qq;
      hi[e] = (unsigned)(qq >> 32);
    }
    const size_t rb = (size_t)(rbase / 16 + g);
    // byte transposition, 4 elements x 4 bytes at a time (8 v_perm_b32 per 4 x 4 block): plane[p][j] = byte p of elements 4 j .. 4 j + 3
    i4v plane[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned* s4 = (h ? hi : lo) + 4 * j;
        const unsigned u0 = __builtin_amdgcn_perm(s4[1], s4[0], 0x05010400u), u1 = __builtin_amdgcn_perm(s4[1], s4[0], 0x07030602u);
        const unsigned v0 = __builtin_amdgcn_perm(s4[3], s4[2], 0x05010400u), v1 = __builtin_amdgcn_perm(s4[3], s4[2], 0x07030602u);
        plane[4 * h + 0][j] = (int)__builtin_amdgcn_perm(v0, u0, 0x05040100u);
        plane[4 * h + 1][j] = (int)__builtin_amdgcn_perm(v0, u0, 0x07060302u);
        plane[4 * h + 2][j] = (int)__builtin_amdgcn_perm(v1, u1, 0x05040100u);
        if (h == 0) plane[3][j] = (int)__builtin_amdgcn_perm(v1, u1, 0x07060302u);  // (byte 7 of q is always zero: no eighth plane)
      }
    }
#pragma unroll
    for (int p = 0; p < I8_NP; ++p)
      __builtin_nontemporal_store(plane[p], reinterpret_cast<i4v*>(Q + ((rb * I8_NP + p) * Mp + m) * 16));
  }
  };
  if (interior) run(std::false_type{}); else run(std::true_type{});
  bpart[((row0 + rbase) / ASM_ROWS) * Mp + m] = bacc;
}

// ---------------------------------------------------------------------------------------------
// 2. contraction
// ---------------------------------------------------------------------------------------------
constexpr int I8_PSPLIT = 5;  // MFMA batches 0 .. 4 (15 MFMAs) in the first half of a step, batches 5, 6 (13) in the second
// MINSUM: the digit pairs p + r >= MINSUM are kept, in 13 - MINSUM significance groups g = p + r - MINSUM.  6 (the statistics of the streaming
// order): 28 products, truncation < 6 x 2^-54 per product.  5 (the EXTENDED streaming order, sgp_suffstats_fwd_extended): 34 products in 8
// groups, truncation 2^-8 of that (14.0 against 11.7 ms at C5, profiles/r04_ab_i8_pairs34.txt), and with DD the fold keeps what fp64
// would round away: the tile leaves as an unevaluated sum hi + lo (lo in a second slab, `lo_off` doubles behind the first).
// A/B (tools/ab_build.sh -DSGP_AB_I8_PAIRS34): the default contraction with 34 pairs.
#ifdef SGP_AB_I8_PAIRS34
constexpr int I8_MINSUM_DEFAULT = 5;
#else
constexpr int I8_MINSUM_DEFAULT = 6;
#endif
static_assert(2 * I8_PPW - 1 <= I8_PSPLIT * (I8_PSPLIT + 1) / 2, "every DMA piece must find its MFMA in the first half");

// Eight waves in two groups (waves 0-3 / 4-7: one of each per SIMD) that run HALF A STEP apart.  A step = 32 data rows = 28 MFMAs per
// wave (batch p = A-plane p against B-planes 6 - p .. 6: the first batches need the fewest operands), two workgroup barriers:
//
//     E(0) O(0) E(1) O(1) .. E(n-1) O(n-1) E(n)      E(s) = group A's top of step s = group B's middle of step s - 1
//                                                    O(s) = group A's middle of step s = group B's top of step s
//
// At its step top a wave issues its 14 operand reads, then its 6 LDS-DMA pieces of stage s + 2 in the shadow of the first 15 MFMAs;
// in its second half it has every operand in registers -- so whenever one group waits for operands (or pays the ~40-60 cycles an
// LDS-DMA piece costs its issuer) the other group's wave on the same SIMD feeds the matrix pipe: 2 180 instead of 2 710 cycles per
// step, the pipe 0.82 instead of 0.66 busy (most of which the chip takes back as clock: 1.73 instead of 1.89 GHz, 12.7 vs 13.5 ms
// in tools/i8_syrk_proto.hip).  Ordering: before every E(s) each wave waits for its own pieces of stage s (counted vmcnt: at most
// one younger stage outstanding); group A reads the stage right after E(s), group B after O(s); the slot of stage s + 2 held
// stage s - 1, whose last reads group B issued after O(s - 1) and retires (lgkmcnt(0)) before E(s); all reads of a step precede its DMAs (a ds_read behind a global_load_lds of the same wave
// waits for that DMA to land).  Lockstep variants measured slower: four waves with 64 x 32 tiles 13.5 ms, eight unstaggered 13.5.
template <bool ACT, int I8_MINSUM, bool DD>
__device__ __forceinline__ void i8_tile_loop(uint8_t* lds, const uint8_t* __restrict__ Q, int Mp, int64_t c0, int64_t c1, int I0,
                                             int J0, int accumulate, double* __restrict__ out, size_t lo_off, int wave, int lane, int prio) {
  constexpr int I8_NG = 13 - I8_MINSUM;
  const int grp = wave >> 2, w4 = wave & 3;
  const int wi = (w4 >> 1) * 2 + grp, wj = w4 & 1;  // 32 x 32 tile (wi, wj) of the 128 x 64 tile: the groups interleave the row blocks
  const int l32 = lane & 31, lh = lane >> 5;
  i16v acc[I8_NG];
#pragma unroll
  for (int g = 0; g < I8_NG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0;

  // DMA piece k of this wave: e = wave + 8 k -> (row block in the stage, plane, column group); the last slots repeat the first
  // pieces (the same bytes to the same place) so that every wave counts 6 per stage
  unsigned goff[I8_PPW];
  int soff[I8_PPW];
#pragma unroll
  for (int k = 0; k < I8_PPW; ++k) {
    int e = wave + 8 * k;
    if (e >= I8_PIECES) e -= I8_PIECES;
    const int rbl = e / (I8_NP * 3), rem = e % (I8_NP * 3), p = rem / 3, cg = rem % 3;
    const int col = cg < 2 ? I0 + cg * 64 : J0;
    goff[k] = (unsigned)(((rbl * I8_NP + p) * Mp + col + lane) * 16);
    soff[k] = __builtin_amdgcn_readfirstlane(((rbl * I8_NP + p) * I8_SCOLS + cg * 64) * 16);
  }
  const size_t gstride = (size_t)2 * I8_NP * Mp * 16;  // bytes of Q per 32-row step
  auto dma_piece = [&](const uint8_t* gbase, int sbase, int k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff[k]),
                                     (__attribute__((address_space(3))) void*)(lds + sbase + soff[k]), 16, 0, 0);
  };
  const int64_t nst = c1 - c0;
  auto wait_stage = [&](int64_t sE) {  // my pieces of stage sE have landed once at most one younger stage's pieces are outstanding
    if (sE + 1 >= nst)
      __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));                                           // vmcnt(0)
    else
      __builtin_amdgcn_s_waitcnt((I8_PPW & 15) | ((I8_PPW >> 4) << 14) | (7 << 4) | (15 << 8));      // vmcnt(6)
  };
  auto bar = [&]() {  // raw barrier, no fence: a __syncthreads() would drain the DMA in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  i4v b[I8_NP], a[I8_NP];
  auto reads = [&](int64_t sidx) {
    if (ACT) {
      const uint8_t* sb = lds + (int)(sidx % I8_NSTAGE) * I8_STAGE_BYTES + lh * (I8_NP * I8_SCOLS * 16);
#pragma unroll
      for (int p = 0; p < I8_NP; ++p) {
        b[I8_NP - 1 - p] = *reinterpret_cast<const i4v*>(sb + ((I8_NP - 1 - p) * I8_SCOLS + I8_TR + wj * 32 + l32) * 16);
        a[p] = *reinterpret_cast<const i4v*>(sb + (p * I8_SCOLS + wi * 32 + l32) * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // MFMA batches [p0, p1); with dma_on the pieces of stage sE + 2 go out in their shadow, one every two MFMAs from the first
  auto half = [&](int p0, int p1, bool dma_on, int64_t sE) {
    // A/B knob SGP_I8_PRIO=1 (off by default): the wave at its step top outranks its SIMD partner.  1.5 % faster in the stand-alone
    // prototype, 5 % SLOWER here (12.03-12.11 vs 11.38-11.53 ms alternating in one process, tools/i8_prio_ab.py).
    if (prio) { if (dma_on) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    const bool pre = dma_on && sE + 2 < nst;
    const uint8_t* gnext = Q + (size_t)(c0 + sE + 2) * gstride;
    const int snext = (int)((sE + 2) % I8_NSTAGE) * I8_STAGE_BYTES;
    int issued = 0, kpiece = 0;
#pragma unroll
    for (int p = p0; p < p1; ++p)
#pragma unroll
      for (int r = (I8_MINSUM - p > 0 ? I8_MINSUM - p : 0); r < I8_NP; ++r) {
        if (ACT) acc[p + r - I8_MINSUM] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b[r], acc[p + r - I8_MINSUM], 0, 0, 0);
        ++issued;
        // after MFMAs 1, 3, .. 11.  Alternating in one process at C5 (profiles/r03_i8_dma_placement_ab.jsonl): 11.27-11.38 ms, against
        // 11.62-11.68 after 2, 4, .. 12, 11.52-11.64 after 1, 2, .. 6 and 11.75-11.83 after 4, 6, .. 14
        if (dma_on && (issued & 1) && kpiece < I8_PPW) {
          __builtin_amdgcn_sched_barrier(0);
          if (pre) dma_piece(gnext, snext, kpiece);
          __builtin_amdgcn_sched_barrier(0);
          ++kpiece;
        }
      }
  };
  if (nst > 0) {
#pragma unroll
    for (int k = 0; k < I8_PPW; ++k) dma_piece(Q + (size_t)c0 * gstride, 0, k);
    if (nst > 1) {
#pragma unroll
      for (int k = 0; k < I8_PPW; ++k) dma_piece(Q + (size_t)(c0 + 1) * gstride, I8_STAGE_BYTES, k);
    }
    if (grp == 0) {
      for (int64_t sidx = 0; sidx < nst; ++sidx) {
        wait_stage(sidx);
        bar();  // E(s): my step top
        reads(sidx);
        half(0, I8_PSPLIT, true, sidx);
        bar();  // O(s): my middle
        half(I8_PSPLIT, I8_NP, false, sidx);
      }
      wait_stage(nst);
      bar();    // E(n): group B's last middle
    } else {
      wait_stage(0);
      bar();    // E(0)
      for (int64_t sidx = 0; sidx < nst; ++sidx) {
        bar();  // O(s): my step top
        reads(sidx);
        half(0, I8_PSPLIT, true, sidx);
        // my reads of stage s (issued a phase ago, the last of them consumed only AFTER this barrier) must have left the LDS queue
        // before group A restages their slot (stage s + 3) right after E(s + 1): lgkmcnt(0), long satisfied by now
        __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8));
        wait_stage(sidx + 1);
        bar();  // E(s + 1): my middle
        half(I8_PSPLIT, I8_NP, false, sidx + 1);
      }
    }
  }
  // fold the significance groups: value = sum_g acc_g 2^(8 g - 60)   (= 2^-108 256^(g + 6)); 128 x 128 slab tile, this half
  if (ACT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wi * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
      double* dst = out + row * TILE + wj * 32 + l32;
      if constexpr (DD) {
#pragma clang fp contract(off)
        // every term acc_g 2^e is exact; the sum of the eight is carried as hi + lo (two_sum per term, least significant first)
        double hi = 0.0, lo = 0.0;
        if (accumulate) { hi = dst[0]; lo = dst[lo_off]; }
#pragma unroll
        for (int g = 0; g < I8_NG; ++g) {
          const double t = (double)acc[g][r] * __builtin_ldexp(1.0, 8 * (g + I8_MINSUM) - 108);
          const double sum = hi + t;
          const double z = sum - hi;
          lo += (hi - (sum - z)) + (t - z);
          hi = sum;
        }
        const double sum = hi + lo;
        dst[lo_off] = lo - (sum - hi);
        dst[0] = sum;
      } else {
        double v = 0.0;
#pragma unroll
        for (int g = 0; g < I8_NG; ++g) v = fma((double)acc[g][r], __builtin_ldexp(1.0, 8 * (g + I8_MINSUM) - 108), v);
        *dst = accumulate ? *dst + v : v;
      }
    }
  }
}

template <int MINSUM, bool DD>
__global__ __launch_bounds__(512, 1) void i8_syrk_tile_kernel(const uint8_t* __restrict__ Q, int Mp, int64_t nsteps, int nsplit,
                                                              int ntiles, int ntiles128, int accumulate, double* __restrict__ slab,
                                                              size_t lo_off, int prio) {
  extern __shared__ __attribute__((aligned(16))) uint8_t i8_lds[];
  // id -> (xcd, tile, split group): the tiles of a split share id % 8, i.e. one XCD under round-robin dispatch (as syrk_tile_kernel)
  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int t = jj % ntiles, split = (jj / ntiles) * 8 + xcd;
  if (split >= nsplit) return;
  // tile t -> (ti, tj): row block ti (128 rows) has 2 (ti + 1) column blocks of 64
  int ti = (int)((sqrtf(4.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) <= t) ++ti;
  while (ti * (ti + 1) > t) --ti;
  const int tj = t - ti * (ti + 1);
  const int I0 = ti * I8_TR, J0 = tj * I8_TC;
  int64_t c0, c1;
  i8_split_steps(nsteps, nsplit, split, c0, c1);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* out = slab + ((size_t)split * ntiles128 + (ti * (ti + 1) / 2 + (tj >> 1))) * (TILE * TILE) + (tj & 1) * I8_TC;
  // the wave's 32 x 32 tile: rows I0 + 32 wi .., columns J0 + 32 wj ..; strictly above the diagonal iff its first column lies beyond
  // its last row -- diagonal workgroups skip those MFMAs (an idle matrix pipe is clock headroom for the other SIMDs here)
  const int grp = wave >> 2, w4 = wave & 3;
  const int wi = (w4 >> 1) * 2 + grp, wj = w4 & 1;
  if (J0 + 32 * wj <= I0 + 32 * wi + 31)
    i8_tile_loop<true, MINSUM, DD>(i8_lds, Q, Mp, c0, c1, I0, J0, accumulate, out, lo_off, wave, lane, prio);
  else
    i8_tile_loop<false, MINSUM, DD>(i8_lds, Q, Mp, c0, c1, I0, J0, accumulate, out, lo_off, wave, lane, prio);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int DP, bool WK>
static void launch_digits(int kid, dim3 grid, hipStream_t st, const double* Xs, const double* ys, const double* Zs, int64_t row0,
                          int64_t N, int M, int Mp, uint8_t* Q, double* Kfu, uint16_t* Kh, double* bpart) {
  switch (kid) {
    case SGP_KERNEL_RBF: kfu_digits_kernel<DP, SGP_KERNEL_RBF, WK><<<grid, 256, 0, st>>>(Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
    case SGP_KERNEL_MATERN32: kfu_digits_kernel<DP, SGP_KERNEL_MATERN32, WK><<<grid, 256, 0, st>>>(Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
    default: kfu_digits_kernel<DP, SGP_KERNEL_MATERN52, WK><<<grid, 256, 0, st>>>(Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
  }
}
template <bool WK>
static void launch_digits_dp(int DP, int kid, dim3 grid, hipStream_t st, const double* Xs, const double* ys, const double* Zs,
                             int64_t row0, int64_t N, int M, int Mp, uint8_t* Q, double* Kfu, uint16_t* Kh, double* bpart) {
  switch (DP) {
    case 2: launch_digits<2, WK>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
    case 4: launch_digits<4, WK>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
    case 8: launch_digits<8, WK>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
    case 16: launch_digits<16, WK>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
    case 24: launch_digits<24, WK>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
    default: launch_digits<32, WK>(kid, grid, st, Xs, ys, Zs, row0, N, M, Mp, Q, Kfu, Kh, bpart); break;
  }
}

// Digit planes of rows [row0, row0 + rows) (rows a multiple of ASM_ROWS) into Q (which starts at row0); Kfu (optional, starts at
// row0 as well): the fp64 block of the same rows; Kh (optional, with Kfu, starts at row0): its fp16 image.
void i8_assemble(const StreamPlan& p, int kid, const double* Xs, const double* ys, const double* Zs, int64_t row0, int64_t rows,
                 int64_t N, int M, uint8_t* Q, double* Kfu, double* bpart, hipStream_t st, uint16_t* Kh) {
  dim3 grid((unsigned)(rows / ASM_ROWS), (p.Mp + 255) / 256);
  if (Kfu)
    launch_digits_dp<true>(p.DP, kid, grid, st, Xs, ys, Zs, row0, N, M, p.Mp, Q, Kfu, Kh, bpart);
  else
    launch_digits_dp<false>(p.DP, kid, grid, st, Xs, ys, Zs, row0, N, M, p.Mp, Q, nullptr, nullptr, bpart);
}

// slab[split][128 x 128 tile of the lower triangle] (+)= this split's part of K'^T K' (without sf2^2), as syrk_tile_kernel
// leaves it for reduce_phi_kernel.  rows: a multiple of 32; nsplit from i8_nsplit() (the same for every super-chunk).
int i8_contract(const uint8_t* Q, int Mp, int64_t rows, int nsplit, int accumulate, double* slab, hipStream_t st, double* slab_lo, int level) {
  Ctx& cx = cur_ctx();
  // once per DEVICE (the attribute belongs to the device's copy of the kernels), whatever context asks: a context created for another
  // device, or an entry point running in the default context on a second device, must not inherit the first device's "done"
  // (atomics -- any host thread may enter here; devices beyond the table set the attribute on every call instead of failing)
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return SGP_ERR_LAUNCH;
  if (dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute((const void*)i8_syrk_tile_kernel<I8_MINSUM_DEFAULT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            I8_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)i8_syrk_tile_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, I8_LDS_BYTES) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)i8_syrk_tile_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, I8_LDS_BYTES) !=
            hipSuccess)
      return SGP_ERR_LAUNCH;
    if (dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  const int nrt = Mp / I8_TR, ntiles = nrt * (nrt + 1), ntiles128 = nrt * (nrt + 1) / 2;
  const int prio = cx.i8_prio;  // A/B knob SGP_I8_PRIO (read when the context is created; measured a loss)
  if (slab_lo && level >= 2)  // the extended order, second level: 39 pairs (p + r >= 4), tiles as hi + lo
    i8_syrk_tile_kernel<4, true><<<nsplit * ntiles, 512, I8_LDS_BYTES, st>>>(Q, Mp, rows / 32, nsplit, ntiles, ntiles128, accumulate, slab,
                                                                             (size_t)(slab_lo - slab), prio);
  else if (slab_lo)  // the extended order: 34 pairs, tiles as hi + lo
    i8_syrk_tile_kernel<5, true><<<nsplit * ntiles, 512, I8_LDS_BYTES, st>>>(Q, Mp, rows / 32, nsplit, ntiles, ntiles128, accumulate, slab,
                                                                             (size_t)(slab_lo - slab), prio);
  else
    i8_syrk_tile_kernel<I8_MINSUM_DEFAULT, false><<<nsplit * ntiles, 512, I8_LDS_BYTES, st>>>(Q, Mp, rows / 32, nsplit, ntiles, ntiles128,
                                                                                               accumulate, slab, 0, prio);
  return SGP_OK;
}

}  // namespace sgp
