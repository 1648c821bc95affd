// Prototype (diagnostic, not part of the library): Phi = K'^T K' for K' in [0, 1] on the INTEGER matrix cores by an error-free
// splitting -- the feasibility measurement behind DESIGN.md section 8's "integer-matrix-core lever".
//
//   q = rint(K' 2^53) = sum_p a_p 256^p,  a_p in [-128, 127]  (p = 0..6; balanced digits = bytes of (q + C) ^ C, C = 0x80 x 7)
//   Phi_IJ = 2^-106 sum_n q_nI q_nJ = 2^-106 sum_{p + r >= 6} 256^(p + r) sum_n a_p,nI a_r,nJ     (+ a truncation < 2^-52 per product,
//            zero-mean: 28 of the 49 digit pairs; the 7 significance groups g = p + r - 6 are exact int32 sums over <= 16384 rows)
//
// Digit planes in HBM: Q[rb][p][m][16 bytes] = digit p of rows 16 rb .. 16 rb + 15 of column m -- 7 bytes per element, and
// one ds_read_b128 is a lane's whole MFMA operand (v_mfma_i32_32x32x32_i8: lane l <-> column l % 32, rows 16 (l / 32) + 0..15).
// Workgroup = 128 x 64 tile of the lower triangle x one split of the rows; 4 waves (one per SIMD), each 64 x 32 = two 32 x 32
// MFMA tiles x 7 group accumulators = 224 accumulator registers.  32-row stages travel global -> LDS by LDS-DMA through a ring.
//
//   build: hipcc --offload-arch=gfx950 -O3 tools/i8_syrk_proto.hip -o tools/i8_syrk_proto
//   run:   tools/i8_syrk_proto [N] [M]     (checks a small case against long-double host arithmetic first)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));

constexpr int NP = 7;            // digit planes
constexpr int TR = 128, TC = 64; // tile
constexpr int SCOLS = TR + TC;   // columns staged per workgroup
constexpr int STAGE_BYTES = 2 * NP * SCOLS * 16;  // 32 rows = 2 row blocks
#ifndef NSTAGE
#define NSTAGE 3
#endif
constexpr int PIECES = 2 * NP * 3;   // 1 KB LDS-DMA pieces per stage (3 groups of 64 columns)
constexpr int PPW = (PIECES + 3) / 4;  // pieces per wave (the last slots of waves 2, 3 repeat a piece)

__host__ __device__ inline uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
// synthetic K'[n][m] 2^53: smooth-ish magnitude spread with full-entropy low digits
__host__ __device__ inline uint64_t kq(int64_t n, int m) {
  const uint64_t h = mix((uint64_t)n * 1315423911ULL + (uint64_t)m * 2654435761ULL + 12345);
  const int sh = (int)(mix(h) % 12);  // magnitudes from 1 down to 2^-11
  return (h >> 11) >> sh;             // < 2^53
}

// digit planes of 16 rows x 1 column per thread
__global__ __launch_bounds__(256) void digits_kernel(int64_t nrb, int Mp, uint8_t* __restrict__ Q) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int64_t rb = blockIdx.y;
  if (m >= Mp) return;
  constexpr uint64_t C = 0x0080808080808080ULL;
  unsigned lo[16], hi[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const uint64_t qq = (kq(rb * 16 + e, m) + C) ^ C;
    lo[e] = (unsigned)qq;
    hi[e] = (unsigned)(qq >> 32);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const unsigned* src = p < 4 ? lo : hi;
    const unsigned b = p & 3;
    const unsigned sel = b | ((4 + b) << 8);  // byte b of the second operand, byte b of the first
    i4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned t01 = __builtin_amdgcn_perm(src[4 * j + 1], src[4 * j], sel);
      const unsigned t23 = __builtin_amdgcn_perm(src[4 * j + 3], src[4 * j + 2], sel);
      v[j] = (int)__builtin_amdgcn_perm(t23, t01, 0x05040100u);
    }
    *reinterpret_cast<i4*>(Q + (((size_t)rb * NP + p) * Mp + m) * 16) = v;
  }
}

// MODE 0: the kernel; 1: no LDS-DMA inside the loop (matrix pipe + LDS reads alone); 2: no MFMA work (data movement alone)
template <int MODE>
__global__ __launch_bounds__(256, 1) void i8_syrk_kernel(const uint8_t* __restrict__ Q, int Mp, int64_t nsteps, int nsplit, int ntiles,
                                                         double* __restrict__ slab, unsigned long long* __restrict__ stamp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int t = jj % ntiles, split = (jj / ntiles) * 8 + xcd;
  if (split >= nsplit) return;
  // tile t -> (ti, tj): row block ti (128 rows) has 2 (ti + 1) column blocks of 64
  int ti = (int)((sqrtf(4.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) <= t) ++ti;
  while (ti * (ti + 1) > t) --ti;
  const int tj = t - ti * (ti + 1);
  const int I0 = ti * TR, J0 = tj * TC;
  const int64_t per = (nsteps + nsplit - 1) / nsplit;
  const int64_t c0 = split * per, c1 = (c0 + per < nsteps) ? c0 + per : nsteps;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int l32 = lane & 31, lh = lane >> 5;

  i16 acc[2][NP];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int g = 0; g < NP; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][g][r] = 0;

  // DMA pieces of this wave: e = wave + 4 k -> (row block in stage, plane, column group)
  auto dma_piece = [&](int64_t c, int slot, int k) {
    const uint8_t* gbase = Q + (size_t)(2 * (MODE == 5 ? (c & 63) : c)) * NP * Mp * 16;  // MODE 5: every workgroup re-reads 2048 rows (L2-resident)
    uint8_t* sbase = lds + slot * STAGE_BYTES;
    int e = wave + 4 * k;
    if (e >= PIECES) e -= PIECES;  // repeat: same bytes to the same place
    const int rbl = e / (NP * 3), rem = e % (NP * 3), p = rem / 3, cg = rem % 3;
    const int col = cg < 2 ? I0 + cg * 64 : J0;
    const uint8_t* g = gbase + (((size_t)rbl * NP + p) * Mp + col + lane) * 16;
    uint8_t* s = sbase + ((rbl * NP + p) * SCOLS + cg * 64) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)s, 16, 0, 0);
  };
  constexpr int NPC = MODE == 3 ? 6 : MODE == 4 ? 3 : PPW;
  auto dma = [&](int64_t c, int slot) {
#pragma unroll
    for (int k = 0; k < NPC; ++k) dma_piece(c, slot, k);
  };
  // One wave per SIMD: nobody else hides LDS latency, and a ds_read issued after a global_load_lds of the same wave waits for
  // that DMA to land (measured: + 430 cycles per step with ANY piece issued ahead of the reads).  So: all 21 operand reads
  // first, in the order the MFMA batches need them (batch p = A-plane p against B-planes 6 - p .. 6: 2, 4, .. 14 MFMAs), then
  // the DMA pieces of the next-but-one stage, then the MFMAs.
  auto compute = [&](int slot, int64_t cn, int slotn, bool pre) {
    const uint8_t* sbase = lds + slot * STAGE_BYTES + lh * (NP * SCOLS * 16);
    i4 b[NP], a[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      b[NP - 1 - p] = *reinterpret_cast<const i4*>(sbase + ((NP - 1 - p) * SCOLS + TR + wj * 32 + l32) * 16);
#pragma unroll
      for (int u = 0; u < 2; ++u) a[p][u] = *reinterpret_cast<const i4*>(sbase + (p * SCOLS + wi * 64 + u * 32 + l32) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 1 && pre) {
#pragma unroll
      for (int k = 0; k < NPC; ++k) dma_piece(cn, slotn, k);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int r = NP - 1 - p; r < NP; ++r)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          acc[u][p + r - (NP - 1)] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p][u], b[r], acc[u][p + r - (NP - 1)], 0, 0, 0);
    }
  };

  const unsigned long long cy0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  if (c0 < c1) {
    dma(c0, 0);
    if (c0 + 1 < c1) dma(c0 + 1, 1);
    for (int64_t c = c0; c < c1; ++c) {
      const int slot = (int)((c - c0) % NSTAGE);
      // my pieces of stage c have landed once at most one younger stage's pieces are outstanding
      constexpr int NW_ = NPC;
      if (MODE != 1 && c + 1 < c1)
        __builtin_amdgcn_s_waitcnt((NW_ & 15) | ((NW_ >> 4) << 14) | (7 << 4) | (15 << 8));
      else
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (MODE == 2 && c + 2 < c1) dma(c + 2, (int)((c + 2 - c0) % NSTAGE));
      if (MODE != 2) compute(slot, c + 2, (int)((c + 2 - c0) % NSTAGE), c + 2 < c1);
    }
  }
  if (tid == 0 && id == 0) {
    stamp[0] = __builtin_amdgcn_s_memtime() - cy0;
    stamp[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    stamp[2] = (unsigned long long)(c1 - c0);
  }
  // fold the significance groups: value = sum_g acc_g 2^(8 g - 58)   (2^-106 256^(g + 6))
  double* out = slab + ((size_t)split * ntiles + t) * (TR * TC);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < NP; ++g) v = fma((double)acc[u][g][r], __builtin_ldexp(1.0, 8 * g - 58), v);
      const int row = wi * 64 + u * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
      out[row * TC + wj * 32 + l32] = v;
    }
}


// ---- v2: register staging (global -> VGPR -> ds_write_b128), operand prefetch across the barrier ---------------------------------
// LDS-DMA cost this loop twice (measured above): a ds_read behind a global_load_lds of the same wave waits for it to land, and a
// dwordx4 DMA piece takes ~14-16 LDS cycles per KB (4 dword passes with 4-way conflicts) against 8 for ds_write_b128.  With
// the stage written by ds_write one step ahead, stage c + 1 is complete at barrier c, so the first operands of step c + 1 are
// read during the last MFMA batch of step c and the matrix pipe never waits at the top of a step.
#ifndef WB
#define WB 4
#endif
template <int MODE>
__global__ __launch_bounds__(256, 1) void i8_syrk_rs_kernel(const uint8_t* __restrict__ Q, int Mp, int64_t nsteps, int nsplit, int ntiles,
                                                            double* __restrict__ slab, unsigned long long* __restrict__ stamp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int t = jj % ntiles, split = (jj / ntiles) * 8 + xcd;
  if (split >= nsplit) return;
  int ti = (int)((sqrtf(4.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) <= t) ++ti;
  while (ti * (ti + 1) > t) --ti;
  const int tj = t - ti * (ti + 1);
  const int I0 = ti * TR, J0 = tj * TC;
  const int64_t per = (nsteps + nsplit - 1) / nsplit;
  const int64_t c0 = split * per, c1 = (c0 + per < nsteps) ? c0 + per : nsteps;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int l32 = lane & 31, lh = lane >> 5;

  i16 acc[2][NP];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int g = 0; g < NP; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][g][r] = 0;

  // staging: piece e = wave + 4 k (k < 11; 42 pieces, the last two slots of waves 2, 3 repeat pieces 0, 1 of... their own list)
  i4 R[PPW];
  unsigned goff[PPW], soff[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    int e = wave + 4 * k;
    if (e >= PIECES) e -= PIECES;
    const int rbl = e / (NP * 3), rem = e % (NP * 3), p = rem / 3, cg = rem % 3;
    const int col = cg < 2 ? I0 + cg * 64 : J0;
    goff[k] = (unsigned)(((rbl * NP + p) * Mp + col + lane) * 16);
    soff[k] = (unsigned)(((rbl * NP + p) * SCOLS + cg * 64 + lane) * 16);
  }
  auto gload = [&](int64_t c) {
    const uint8_t* gbase = Q + (size_t)(2 * c) * NP * Mp * 16;
#pragma unroll
    for (int k = 0; k < PPW; ++k) R[k] = *reinterpret_cast<const i4*>(gbase + goff[k]);
  };
  auto swrite = [&](int slot) {
    uint8_t* sbase = lds + slot * STAGE_BYTES;
#pragma unroll
    for (int k = 0; k < PPW; ++k) *reinterpret_cast<i4*>(sbase + soff[k]) = R[k];
  };
  auto rdA = [&](int slot, int p, int u) {
    return *reinterpret_cast<const i4*>(lds + slot * STAGE_BYTES + lh * (NP * SCOLS * 16) + (p * SCOLS + wi * 64 + u * 32 + l32) * 16);
  };
  auto rdB = [&](int slot, int r) {
    return *reinterpret_cast<const i4*>(lds + slot * STAGE_BYTES + lh * (NP * SCOLS * 16) + (r * SCOLS + TR + wj * 32 + l32) * 16);
  };
  auto batch = [&](int p, const i4 (&a)[NP][2], const i4 (&b)[NP]) {
#pragma unroll
    for (int r = NP - 1 - p; r < NP; ++r)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        acc[u][p + r - (NP - 1)] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p][u], b[r], acc[u][p + r - (NP - 1)], 0, 0, 0);
  };

  const unsigned long long cy0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  i4 a[NP][2], b[NP];
  constexpr int NPRE = 3;  // A-planes (and B-planes 6, 5, 4) whose operands are read one step ahead
  if (c0 < c1) {
    gload(c0);
    swrite(0);
    if (c0 + 1 < c1) {
      gload(c0 + 1);
      swrite(1);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPRE; ++p) {
      b[NP - 1 - p] = rdB(0, NP - 1 - p);
      a[p][0] = rdA(0, p, 0);
      a[p][1] = rdA(0, p, 1);
    }
    for (int64_t c = c0; c < c1; ++c) {
      const int slot = (int)((c - c0) % NSTAGE), slot1 = (int)((c + 1 - c0) % NSTAGE), slot2 = (int)((c + 2 - c0) % NSTAGE);
      const bool more2 = c + 2 < c1;
#pragma unroll
      for (int p = NPRE; p < NP; ++p) {
        b[NP - 1 - p] = rdB(slot, NP - 1 - p);
        a[p][0] = rdA(slot, p, 0);
        a[p][1] = rdA(slot, p, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 1) gload(more2 ? c + 2 : c1 - 1);  // (branch-free: a merge point makes the waitcnt pass drain the prefetch)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < NP - 1; ++p) {
        batch(p, a, b);
        __builtin_amdgcn_sched_barrier(0);
        if (p == WB && MODE != 1) {
          swrite(slot2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // first operands of the next step (its stage was complete at this step's barrier) under the last, longest batch
      i4 nb[NPRE], na[NPRE][2];
      {
#pragma unroll
        for (int p = 0; p < NPRE; ++p) {
          nb[p] = rdB(slot1, NP - 1 - p);
          na[p][0] = rdA(slot1, p, 0);
          na[p][1] = rdA(slot1, p, 1);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      batch(NP - 1, a, b);
      __builtin_amdgcn_sched_barrier(0);
      {
#pragma unroll
        for (int p = 0; p < NPRE; ++p) {
          b[NP - 1 - p] = nb[p];
          a[p][0] = na[p][0];
          a[p][1] = na[p][1];
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0 && id == 0) {
    stamp[0] = __builtin_amdgcn_s_memtime() - cy0;
    stamp[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    stamp[2] = (unsigned long long)(c1 - c0);
  }
  double* out = slab + ((size_t)split * ntiles + t) * (TR * TC);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < NP; ++g) v = fma((double)acc[u][g][r], __builtin_ldexp(1.0, 8 * g - 58), v);
      const int row = wi * 64 + u * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
      out[row * TC + wj * 32 + l32] = v;
    }
}


// ---- v4: DMA pieces issued in the shadow of the MFMAs (precomputed offsets: saddr + 32-bit lane offset, M0 = base + constant), and
//      the last HOLD_N MFMAs of a step held back to cover the next step's operand reads ----------------------------------------
#ifndef HOLD_R
#define HOLD_R 3   // batch 6 (A-plane 6) against B-planes HOLD_R .. 6 is issued after the NEXT step's reads
#endif
#ifndef DMA_EVERY
#define DMA_EVERY 4
#endif
#ifndef DMA_FIRST
#define DMA_FIRST 6
#endif
__constant__ int g_map;  // 0: the tiles of a split on one XCD (id % 8), 1: split-major ids (a split's tiles dispatched together, over all XCDs)
template <int MODE>
__global__ __launch_bounds__(256, 1) void i8_syrk_v4_kernel(const uint8_t* __restrict__ Q, int Mp, int64_t nsteps, int nsplit, int ntiles,
                                                            double* __restrict__ slab, unsigned long long* __restrict__ stamp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int id = blockIdx.x;
  int t, split;
  if (g_map == 0) {
    const int xcd = id & 7, jj = id >> 3;
    t = jj % ntiles;
    split = (jj / ntiles) * 8 + xcd;
  } else {
    t = id % ntiles;
    split = id / ntiles;
  }
  if (split >= nsplit) return;
  int ti = (int)((sqrtf(4.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) <= t) ++ti;
  while (ti * (ti + 1) > t) --ti;
  const int tj = t - ti * (ti + 1);
  const int I0 = ti * TR, J0 = tj * TC;
  const int64_t per = (nsteps + nsplit - 1) / nsplit;
  const int64_t c0 = split * per, c1 = (c0 + per < nsteps) ? c0 + per : nsteps;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int l32 = lane & 31, lh = lane >> 5;

  i16 acc[2][NP];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int g = 0; g < NP; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][g][r] = 0;

  unsigned goff[PPW];
  int soff[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    int e = wave + 4 * k;
    if (e >= PIECES) e -= PIECES;
    const int rbl = e / (NP * 3), rem = e % (NP * 3), p = rem / 3, cg = rem % 3;
    const int col = cg < 2 ? I0 + cg * 64 : J0;
    goff[k] = (unsigned)(((rbl * NP + p) * Mp + col + lane) * 16);
    soff[k] = __builtin_amdgcn_readfirstlane(((rbl * NP + p) * SCOLS + cg * 64) * 16);
  }
  const size_t gstride = (size_t)2 * NP * Mp * 16;  // bytes per 32-row step
  auto dma_piece = [&](const uint8_t* gbase, int sbase, int k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff[k]),
                                     (__attribute__((address_space(3))) void*)(lds + sbase + soff[k]), 16, 0, 0);
  };

  const unsigned long long cy0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  i4 ao[2], bo[NP];  // operands of the held-back MFMAs
  if (c0 < c1) {
#pragma unroll
    for (int k = 0; k < PPW; ++k) dma_piece(Q + (size_t)c0 * gstride, 0, k);
    if (c0 + 1 < c1) {
#pragma unroll
      for (int k = 0; k < PPW; ++k) dma_piece(Q + (size_t)(c0 + 1) * gstride, STAGE_BYTES, k);
    }
    for (int64_t c = c0; c < c1; ++c) {
      const int slot = (int)((c - c0) % NSTAGE), slot2 = (int)((c + 2 - c0) % NSTAGE);
      if (c + 1 < c1)
        __builtin_amdgcn_s_waitcnt((PPW & 15) | ((PPW >> 4) << 14) | (7 << 4) | (15 << 8));
      else
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const uint8_t* sb = lds + slot * STAGE_BYTES + lh * (NP * SCOLS * 16);
      i4 b[NP], a[NP][2];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        b[NP - 1 - p] = *reinterpret_cast<const i4*>(sb + ((NP - 1 - p) * SCOLS + TR + wj * 32 + l32) * 16);
#pragma unroll
        for (int u = 0; u < 2; ++u) a[p][u] = *reinterpret_cast<const i4*>(sb + (p * SCOLS + wi * 64 + u * 32 + l32) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c > c0) {  // the previous step's held-back MFMAs: the matrix pipe works while this step's operands arrive
#pragma unroll
        for (int r = HOLD_R; r < NP; ++r)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[u][r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ao[u], bo[r], acc[u][r], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      const bool pre = MODE != 1 && c + 2 < c1;
      const uint8_t* gnext = Q + (size_t)(MODE == 5 ? ((c + 2) & 63) : (c + 2)) * gstride;
      const int snext = slot2 * STAGE_BYTES;
      int issued = 0, kpiece = 0;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int r = NP - 1 - p; r < (p == NP - 1 ? HOLD_R : NP); ++r) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            acc[u][p + r - (NP - 1)] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p][u], b[r], acc[u][p + r - (NP - 1)], 0, 0, 0);
            ++issued;
            if (issued >= DMA_FIRST && (issued - DMA_FIRST) % DMA_EVERY == 0 && kpiece < PPW) {
              __builtin_amdgcn_sched_barrier(0);
              if (pre) dma_piece(gnext, snext, kpiece);
              __builtin_amdgcn_sched_barrier(0);
              ++kpiece;
            }
          }
        }
      }
      static_assert(DMA_FIRST + DMA_EVERY * (PPW - 1) <= 56 - 2 * (NP - HOLD_R), "every DMA piece must find its MFMA");
      ao[0] = a[NP - 1][0];
      ao[1] = a[NP - 1][1];
#pragma unroll
      for (int r = HOLD_R; r < NP; ++r) bo[r] = b[r];
    }
#pragma unroll
    for (int r = HOLD_R; r < NP; ++r)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u][r] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ao[u], bo[r], acc[u][r], 0, 0, 0);
  }
  if (tid == 0 && id == 0) {
    stamp[0] = __builtin_amdgcn_s_memtime() - cy0;
    stamp[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    stamp[2] = (unsigned long long)(c1 - c0);
  }
  double* out = slab + ((size_t)split * ntiles + t) * (TR * TC);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < NP; ++g) v = fma((double)acc[u][g][r], __builtin_ldexp(1.0, 8 * g - 58), v);
      const int row = wi * 64 + u * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
      out[row * TC + wj * 32 + l32] = v;
    }
}


// ---- v5: eight waves (two per SIMD), each one 32 x 32 MFMA tile x 7 group accumulators (112 registers): when one wave of a SIMD waits
//      (barrier, operand reads, the ~60 cycles an LDS-DMA piece costs its issuer) the other one feeds the matrix pipe.  LDS reads
//      go up by a third (14 per 28 MFMAs instead of 21 per 56); the LDS array has room (12 % active in v4).
constexpr int PPW8 = (PIECES + 7) / 8;  // 6 pieces per wave and stage (the last slots repeat pieces)
#ifndef DMA8_FIRST
#define DMA8_FIRST 4
#endif
#ifndef DMA8_EVERY
#define DMA8_EVERY 4
#endif
template <int MODE>
__global__ __launch_bounds__(512, 1) void i8_syrk_v5_kernel(const uint8_t* __restrict__ Q, int Mp, int64_t nsteps, int nsplit, int ntiles,
                                                            double* __restrict__ slab, unsigned long long* __restrict__ stamp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int t = jj % ntiles, split = (jj / ntiles) * 8 + xcd;
  if (split >= nsplit) return;
  int ti = (int)((sqrtf(4.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) <= t) ++ti;
  while (ti * (ti + 1) > t) --ti;
  const int tj = t - ti * (ti + 1);
  const int I0 = ti * TR, J0 = tj * TC;
  const int64_t per = (nsteps + nsplit - 1) / nsplit;
  const int64_t c0 = split * per, c1 = (c0 + per < nsteps) ? c0 + per : nsteps;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int l32 = lane & 31, lh = lane >> 5;

  i16 acc[NP];
#pragma unroll
  for (int g = 0; g < NP; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0;

  unsigned goff[PPW8];
  int soff[PPW8];
#pragma unroll
  for (int k = 0; k < PPW8; ++k) {
    int e = wave + 8 * k;
    if (e >= PIECES) e -= PIECES;
    const int rbl = e / (NP * 3), rem = e % (NP * 3), p = rem / 3, cg = rem % 3;
    const int col = cg < 2 ? I0 + cg * 64 : J0;
    goff[k] = (unsigned)(((rbl * NP + p) * Mp + col + lane) * 16);
    soff[k] = __builtin_amdgcn_readfirstlane(((rbl * NP + p) * SCOLS + cg * 64) * 16);
  }
  const size_t gstride = (size_t)2 * NP * Mp * 16;
  auto dma_piece = [&](const uint8_t* gbase, int sbase, int k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff[k]),
                                     (__attribute__((address_space(3))) void*)(lds + sbase + soff[k]), 16, 0, 0);
  };

  const unsigned long long cy0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  if (c0 < c1) {
#pragma unroll
    for (int k = 0; k < PPW8; ++k) dma_piece(Q + (size_t)c0 * gstride, 0, k);
    if (c0 + 1 < c1) {
#pragma unroll
      for (int k = 0; k < PPW8; ++k) dma_piece(Q + (size_t)(c0 + 1) * gstride, STAGE_BYTES, k);
    }
    for (int64_t c = c0; c < c1; ++c) {
      const int slot = (int)((c - c0) % NSTAGE), slot2 = (int)((c + 2 - c0) % NSTAGE);
      if (MODE != 1 && c + 1 < c1)
        __builtin_amdgcn_s_waitcnt((PPW8 & 15) | ((PPW8 >> 4) << 14) | (7 << 4) | (15 << 8));
      else
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const uint8_t* sb = lds + slot * STAGE_BYTES + lh * (NP * SCOLS * 16);
      i4 b[NP], a[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        b[NP - 1 - p] = *reinterpret_cast<const i4*>(sb + ((NP - 1 - p) * SCOLS + TR + wj * 32 + l32) * 16);
        a[p] = *reinterpret_cast<const i4*>(sb + (p * SCOLS + wi * 32 + l32) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      const bool pre = MODE != 1 && c + 2 < c1;
      const uint8_t* gnext = Q + (size_t)(c + 2) * gstride;
      const int snext = slot2 * STAGE_BYTES;
      int issued = 0, kpiece = 0;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int r = NP - 1 - p; r < NP; ++r) {
          acc[p + r - (NP - 1)] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b[r], acc[p + r - (NP - 1)], 0, 0, 0);
          ++issued;
          if (issued >= DMA8_FIRST && (issued - DMA8_FIRST) % DMA8_EVERY == 0 && kpiece < PPW8) {
            __builtin_amdgcn_sched_barrier(0);
            if (pre) dma_piece(gnext, snext, kpiece);
            __builtin_amdgcn_sched_barrier(0);
            ++kpiece;
          }
        }
      }
      static_assert(DMA8_FIRST + DMA8_EVERY * (PPW8 - 1) <= 28, "every DMA piece must find its MFMA");
    }
  }
  if (tid == 0 && id == 0) {
    stamp[0] = __builtin_amdgcn_s_memtime() - cy0;
    stamp[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    stamp[2] = (unsigned long long)(c1 - c0);
  }
  double* out = slab + ((size_t)split * ntiles + t) * (TR * TC);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    double v = 0.0;
#pragma unroll
    for (int g = 0; g < NP; ++g) v = fma((double)acc[g][r], __builtin_ldexp(1.0, 8 * g - 58), v);
    const int row = wi * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
    out[row * TC + wj * 32 + l32] = v;
  }
}


// ---- v6: v5's eight waves in two groups (waves 0-3 / 4-7: one of each per SIMD) that run HALF A STEP apart: two barriers per step;
//      at every barrier one group is at the top of a step (issues its 14 operand reads, waits for them) while the other is in
//      the middle of one (all operands in registers, 13 MFMAs to go) -- so the matrix pipe has work while operands are in flight.
//      DMA: every wave issues its pieces of stage c + 2 after the EVEN barrier of step c (group A's step top, group B's middle)
//      and waits for its pieces of stage c before the even barrier of step c: A reads the stage right after it, B half a step later.
template <int MODE>
__global__ __launch_bounds__(512, 1) void i8_syrk_v6_kernel(const uint8_t* __restrict__ Q, int Mp, int64_t nsteps, int nsplit, int ntiles,
                                                            double* __restrict__ slab, unsigned long long* __restrict__ stamp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int t = jj % ntiles, split = (jj / ntiles) * 8 + xcd;
  if (split >= nsplit) return;
  int ti = (int)((sqrtf(4.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) <= t) ++ti;
  while (ti * (ti + 1) > t) --ti;
  const int tj = t - ti * (ti + 1);
  const int I0 = ti * TR, J0 = tj * TC;
  const int64_t per = (nsteps + nsplit - 1) / nsplit;
  const int64_t c0 = split * per, c1 = (c0 + per < nsteps) ? c0 + per : nsteps;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                  // 0: group A, 1: group B (half a step behind)
  const int w4 = wave & 3;
  const int wi = (w4 >> 1) * 2 + grp, wj = w4 & 1;  // the two groups interleave the four 32-row blocks
  const int l32 = lane & 31, lh = lane >> 5;

  i16 acc[NP];
#pragma unroll
  for (int g = 0; g < NP; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0;

  unsigned goff[PPW8];
  int soff[PPW8];
#pragma unroll
  for (int k = 0; k < PPW8; ++k) {
    int e = wave + 8 * k;
    if (e >= PIECES) e -= PIECES;
    const int rbl = e / (NP * 3), rem = e % (NP * 3), p = rem / 3, cg = rem % 3;
    const int col = cg < 2 ? I0 + cg * 64 : J0;
    goff[k] = (unsigned)(((rbl * NP + p) * Mp + col + lane) * 16);
    soff[k] = __builtin_amdgcn_readfirstlane(((rbl * NP + p) * SCOLS + cg * 64) * 16);
  }
  const size_t gstride = (size_t)2 * NP * Mp * 16;
  auto dma_piece = [&](const uint8_t* gbase, int sbase, int k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff[k]),
                                     (__attribute__((address_space(3))) void*)(lds + sbase + soff[k]), 16, 0, 0);
  };
  auto mfma = [&](const i4& a, const i4& b, i16& c) { c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); };
#ifndef PSPLIT_V
#define PSPLIT_V 5
#endif
#ifndef DMA6_FIRST
#define DMA6_FIRST 1
#endif
#ifndef DMA6_EVERY
#define DMA6_EVERY 2
#endif
  static_assert(DMA6_FIRST + DMA6_EVERY * (PPW8 - 1) <= PSPLIT_V * (PSPLIT_V + 1) / 2, "every DMA piece must find its MFMA in the first half");
  constexpr int PSPLIT = PSPLIT_V;  // batches 0 .. PSPLIT - 1 (15 MFMAs at 5) in the first half of a step, the rest (13) in the second

  const unsigned long long cy0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  const int64_t nst = c1 - c0;
  // barrier sequence of the workgroup: E(0) O(0) E(1) O(1) .. E(nst-1) O(nst-1) E(nst).  E(s) = group A's top of step s and group
  // B's middle of step s - 1; O(s) = A's middle of step s and B's top of step s.  Before every E(s) each wave waits for its own
  // pieces of stage s; after every E(s) each wave issues its pieces of stage s + 2 (that slot held stage s - 1, last read after
  // O(s - 1) by group B).
  auto wait_stage = [&](int64_t sE) {
    if (MODE == 1 || sE + 1 >= nst)
      __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
    else
      __builtin_amdgcn_s_waitcnt((PPW8 & 15) | ((PPW8 >> 4) << 14) | (7 << 4) | (15 << 8));
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  i4 b[NP], a[NP];
  auto reads = [&](int64_t sidx) {
    const uint8_t* sb = lds + (int)(sidx % NSTAGE) * STAGE_BYTES + lh * (NP * SCOLS * 16);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      b[NP - 1 - p] = *reinterpret_cast<const i4*>(sb + ((NP - 1 - p) * SCOLS + TR + wj * 32 + l32) * 16);
      a[p] = *reinterpret_cast<const i4*>(sb + (p * SCOLS + wi * 32 + l32) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // MFMA batches [p0, p1) with (optionally) the DMA pieces of stage sE + 2 in their shadow, one every two MFMAs
#ifndef PRIO_FIRST
#define PRIO_FIRST 0
#endif
#ifndef PRIO_SECOND
#define PRIO_SECOND 0
#endif
  auto half = [&](int p0, int p1, bool dma_on, int64_t sE) {
    if (dma_on) __builtin_amdgcn_s_setprio(PRIO_FIRST); else __builtin_amdgcn_s_setprio(PRIO_SECOND);
    const bool pre = dma_on && MODE != 1 && sE + 2 < nst;
    const uint8_t* gnext = Q + (size_t)(c0 + sE + 2) * gstride;
    const int snext = (int)((sE + 2) % NSTAGE) * STAGE_BYTES;
    int issued = 0, kpiece = 0;
#pragma unroll
    for (int p = p0; p < p1; ++p)
#pragma unroll
      for (int r = NP - 1 - p; r < NP; ++r) {
        mfma(a[p], b[r], acc[p + r - (NP - 1)]);
        ++issued;
        if (dma_on && issued >= DMA6_FIRST && (issued - DMA6_FIRST) % DMA6_EVERY == 0 && kpiece < PPW8) {
          __builtin_amdgcn_sched_barrier(0);
          if (pre) dma_piece(gnext, snext, kpiece);
          __builtin_amdgcn_sched_barrier(0);
          ++kpiece;
        }
      }
  };
  if (nst > 0) {
#pragma unroll
    for (int k = 0; k < PPW8; ++k) dma_piece(Q + (size_t)c0 * gstride, 0, k);
    if (nst > 1) {
#pragma unroll
      for (int k = 0; k < PPW8; ++k) dma_piece(Q + (size_t)(c0 + 1) * gstride, STAGE_BYTES, k);
    }
    if (grp == 0) {
      for (int64_t sidx = 0; sidx < nst; ++sidx) {
        wait_stage(sidx);
        bar();                                  // E(s): my step top
        reads(sidx);
        half(0, PSPLIT, true, sidx);
        bar();                                  // O(s): my middle
        half(PSPLIT, NP, false, sidx);
      }
      wait_stage(nst);
      bar();                                    // E(nst): group B's last middle
    } else {
      wait_stage(0);
      bar();                                    // E(0): nothing of mine to do yet
      for (int64_t sidx = 0; sidx < nst; ++sidx) {
        bar();                                  // O(s): my step top
        reads(sidx);
        half(0, PSPLIT, true, sidx);            // (DMA after my reads, as in group A: a ds_read behind an LDS-DMA of the same wave waits for it)
        __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8));  // lgkmcnt(0): my reads of stage s are out of the LDS queue before E(s + 1)
        wait_stage(sidx + 1);
        bar();                                  // E(s + 1): my middle
        half(PSPLIT, NP, false, sidx + 1);
      }
    }
  }
  if (tid == 0 && id == 0) {
    stamp[0] = __builtin_amdgcn_s_memtime() - cy0;
    stamp[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    stamp[2] = (unsigned long long)(c1 - c0);
  }
  double* out = slab + ((size_t)split * ntiles + t) * (TR * TC);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    double v = 0.0;
#pragma unroll
    for (int g = 0; g < NP; ++g) v = fma((double)acc[g][r], __builtin_ldexp(1.0, 8 * g - 58), v);
    const int row = wi * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
    out[row * TC + wj * 32 + l32] = v;
  }
}

// ---- v7 = v6 with the B operand (the 64 columns of the J block) loaded global -> VGPR one step ahead instead of through LDS: a third
//      fewer LDS-DMA pieces and half the LDS operand reads, for 7 global_load_dwordx4 per wave and step.
// ---- (v6:) eight waves in two groups (waves 0-3 / 4-7: one of each per SIMD) that run HALF A STEP apart: two barriers per step;
//      at every barrier one group is at the top of a step (issues its 14 operand reads, waits for them) while the other is in
//      the middle of one (all operands in registers, 13 MFMAs to go) -- so the matrix pipe has work while operands are in flight.
//      DMA: every wave issues its pieces of stage c + 2 after the EVEN barrier of step c (group A's step top, group B's middle)
//      and waits for its pieces of stage c before the even barrier of step c: A reads the stage right after it, B half a step later.
template <int MODE>
__global__ __launch_bounds__(512, 1) void i8_syrk_v7_kernel(const uint8_t* __restrict__ Q, int Mp, int64_t nsteps, int nsplit, int ntiles,
                                                            double* __restrict__ slab, unsigned long long* __restrict__ stamp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int id = blockIdx.x;
  const int xcd = id & 7, jj = id >> 3;
  const int t = jj % ntiles, split = (jj / ntiles) * 8 + xcd;
  if (split >= nsplit) return;
  int ti = (int)((sqrtf(4.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) <= t) ++ti;
  while (ti * (ti + 1) > t) --ti;
  const int tj = t - ti * (ti + 1);
  const int I0 = ti * TR, J0 = tj * TC;
  const int64_t per = (nsteps + nsplit - 1) / nsplit;
  const int64_t c0 = split * per, c1 = (c0 + per < nsteps) ? c0 + per : nsteps;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                  // 0: group A, 1: group B (half a step behind)
  const int w4 = wave & 3;
  const int wi = (w4 >> 1) * 2 + grp, wj = w4 & 1;  // the two groups interleave the four 32-row blocks
  const int l32 = lane & 31, lh = lane >> 5;

  i16 acc[NP];
#pragma unroll
  for (int g = 0; g < NP; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0;

  constexpr int PIECES7 = 2 * NP * 2, PPW8 = (PIECES7 + 7) / 8;  // 28 pieces (A columns only), 4 per wave
  unsigned goff[PPW8];
  int soff[PPW8];
#pragma unroll
  for (int k = 0; k < PPW8; ++k) {
    int e = wave + 8 * k;
    if (e >= PIECES7) e -= PIECES7;
    const int rbl = e / (NP * 2), rem = e % (NP * 2), p = rem / 2, cg = rem % 2;
    const int col = I0 + cg * 64;
    goff[k] = (unsigned)(((rbl * NP + p) * Mp + col + lane) * 16);
    soff[k] = __builtin_amdgcn_readfirstlane(((rbl * NP + p) * SCOLS + cg * 64) * 16);
  }
  const size_t gstride = (size_t)2 * NP * Mp * 16;
  auto dma_piece = [&](const uint8_t* gbase, int sbase, int k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff[k]),
                                     (__attribute__((address_space(3))) void*)(lds + sbase + soff[k]), 16, 0, 0);
  };
  auto mfma = [&](const i4& a, const i4& b, i16& c) { c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); };
  constexpr int PSPLIT = 5;  // batches 0 .. PSPLIT - 1 (15 MFMAs at 5) in the first half of a step, the rest (13) in the second

  const unsigned long long cy0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  const int64_t nst = c1 - c0;
  // barrier sequence of the workgroup: E(0) O(0) E(1) O(1) .. E(nst-1) O(nst-1) E(nst).  E(s) = group A's top of step s and group
  // B's middle of step s - 1; O(s) = A's middle of step s and B's top of step s.  Before every E(s) each wave waits for its own
  // pieces of stage s; after every E(s) each wave issues its pieces of stage s + 2 (that slot held stage s - 1, last read after
  // O(s - 1) by group B).
  // vmcnt in program order per step: [7 B loads of step s + 1][4 DMA pieces of stage s + 2].  NTOP: before my step top everything but
  // the youngest DMA group must be back (my B operands, my pieces of the stage about to be read); NMID (group B's middle): my pieces
  // of the next stage, older than the B loads and DMA pieces of this step
  auto wait_n = [&](int64_t sE, bool mid) {
    if (MODE == 1 || sE + 2 >= nst)
      __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
    else if (mid)
      __builtin_amdgcn_s_waitcnt(((PPW8 + NP) & 15) | (((PPW8 + NP) >> 4) << 14) | (7 << 4) | (15 << 8));
    else
      __builtin_amdgcn_s_waitcnt((PPW8 & 15) | ((PPW8 >> 4) << 14) | (7 << 4) | (15 << 8));
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  i4 b[NP], bn[NP], a[NP];
  const size_t boff = (size_t)((lh * NP) * Mp + J0 + wj * 32 + l32) * 16;  // + p Mp 16 per plane
  auto bload = [&](int64_t sidx, i4 (&dst)[NP]) {
    const uint8_t* gb = Q + (size_t)(c0 + sidx) * gstride + boff;
#pragma unroll
    for (int p = 0; p < NP; ++p) dst[p] = *reinterpret_cast<const i4*>(gb + (size_t)p * Mp * 16);
  };
  auto reads = [&](int64_t sidx) {
    const uint8_t* sb = lds + (int)(sidx % NSTAGE) * STAGE_BYTES + lh * (NP * SCOLS * 16);
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = *reinterpret_cast<const i4*>(sb + (p * SCOLS + wi * 32 + l32) * 16);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 1 && sidx + 1 < nst) bload(sidx + 1, bn);   // next step's B operands: one step of latency cover
    __builtin_amdgcn_sched_barrier(0);
  };
  // MFMA batches [p0, p1) with (optionally) the DMA pieces of stage sE + 2 in their shadow, one every two MFMAs
  auto half = [&](int p0, int p1, bool dma_on, int64_t sE) {
    const bool pre = dma_on && MODE != 1 && sE + 2 < nst;
    const uint8_t* gnext = Q + (size_t)(c0 + sE + 2) * gstride;
    const int snext = (int)((sE + 2) % NSTAGE) * STAGE_BYTES;
    int issued = 0, kpiece = 0;
#pragma unroll
    for (int p = p0; p < p1; ++p)
#pragma unroll
      for (int r = NP - 1 - p; r < NP; ++r) {
        mfma(a[p], b[r], acc[p + r - (NP - 1)]);
        ++issued;
        if (dma_on && (issued & 1) && issued >= 3 && kpiece < PPW8) {
          __builtin_amdgcn_sched_barrier(0);
          if (pre) dma_piece(gnext, snext, kpiece);
          __builtin_amdgcn_sched_barrier(0);
          ++kpiece;
        }
      }
  };
  auto rotate = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p) b[p] = bn[p];
  };
  if (nst > 0) {
#pragma unroll
    for (int k = 0; k < PPW8; ++k) dma_piece(Q + (size_t)c0 * gstride, 0, k);
    if (nst > 1) {
#pragma unroll
      for (int k = 0; k < PPW8; ++k) dma_piece(Q + (size_t)(c0 + 1) * gstride, STAGE_BYTES, k);
    }
    bload(0, b);
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));  // prologue: everything back (stage 0, stage 1, my first B operands)
    if (grp == 0) {
      for (int64_t sidx = 0; sidx < nst; ++sidx) {
        if (sidx > 0) wait_n(sidx - 1, false);  // B operands of this step + my pieces of stage s (issued in step s - 2)
        bar();                                  // E(s): my step top
        reads(sidx);
        half(0, PSPLIT, true, sidx);
        bar();                                  // O(s): my middle
        half(PSPLIT, NP, false, sidx);
        rotate();
      }
      __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
      bar();                                    // E(nst)
    } else {
      bar();                                    // E(0)
      for (int64_t sidx = 0; sidx < nst; ++sidx) {
        if (sidx > 0) wait_n(sidx - 1, false);  // my B operands of this step
        bar();                                  // O(s): my step top
        reads(sidx);
        half(0, PSPLIT, true, sidx);
        __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8));  // lgkmcnt(0)
        wait_n(sidx, true);                     // my pieces of stage s + 1 (older than this step's B loads and DMA pieces)
        bar();                                  // E(s + 1): my middle
        half(PSPLIT, NP, false, sidx + 1);
        rotate();
      }
    }
  }
  if (tid == 0 && id == 0) {
    stamp[0] = __builtin_amdgcn_s_memtime() - cy0;
    stamp[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    stamp[2] = (unsigned long long)(c1 - c0);
  }
  double* out = slab + ((size_t)split * ntiles + t) * (TR * TC);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    double v = 0.0;
#pragma unroll
    for (int g = 0; g < NP; ++g) v = fma((double)acc[g][r], __builtin_ldexp(1.0, 8 * g - 58), v);
    const int row = wi * 32 + (r >> 2) * 8 + lh * 4 + (r & 3);
    out[row * TC + wj * 32 + l32] = v;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int RS = 0>
static double run(int64_t N, int M, bool check, int reps, int split_rows = 16384) {
  const int Mp = (M + 127) / 128 * 128;
  const int64_t nsteps = N / 32;
  int nsplit = (int)((N + split_rows - 1) / split_rows);
  nsplit = (nsplit + 7) / 8 * 8;
  const int nrt = Mp / TR;
  const int ntiles = nrt * (nrt + 1);
  uint8_t* Q;
  double* slab;
  unsigned long long* stamp;
  CK(hipMalloc(&stamp, 64));
  CK(hipMalloc(&Q, (size_t)(N / 16) * NP * Mp * 16));
  CK(hipMalloc(&slab, (size_t)nsplit * ntiles * TR * TC * 8));
  digits_kernel<<<dim3((Mp + 255) / 256, (unsigned)(N / 16)), 256>>>(N / 16, Mp, Q);
  CK(hipDeviceSynchronize());
  const size_t shm = (size_t)NSTAGE * STAGE_BYTES;
  auto kern = RS == 5 ? i8_syrk_v7_kernel<MODE> : RS == 4 ? i8_syrk_v6_kernel<MODE> : RS == 3 ? i8_syrk_v5_kernel<MODE> : RS == 2 ? i8_syrk_v4_kernel<MODE> : RS == 1 ? i8_syrk_rs_kernel<MODE> : i8_syrk_kernel<MODE>;
  const int nthreads = RS >= 3 ? 512 : 256;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const int grid = nsplit * ntiles;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  kern<<<grid, nthreads, shm>>>(Q, Mp, nsteps, nsplit, ntiles, slab, stamp);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) kern<<<grid, nthreads, shm>>>(Q, Mp, nsteps, nsplit, ntiles, slab, stamp);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double f64_equiv = (double)N * M * M * 1e-9 / ms;  // the accounting of bench.py: N M^2 flops for the lower triangle, GFLOP/ms = TFLOP/s
  const double macs = (double)N * ntiles * TR * TC * 28.0;
  printf("%s mode %d  N %lld M %d: %d splits x %d tiles = %d workgroups, %.3f ms, int8 %.1f TMAC/s (%.2f of 2447), fp64-equivalent %.1f TFLOP/s\n",
         RS == 5 ? "v7 (v6, B operand direct)" : RS == 4 ? "v6 (2 staggered groups)" : RS == 3 ? "v5 (8 waves)" : RS == 2 ? "v4" : RS == 1 ? "reg-staged" : "lds-dma", MODE, (long long)N, Mp, nsplit, ntiles, grid, ms, macs / (ms * 1e-3) / 1e12, macs / (ms * 1e-3) / 1e12 / 2447.0, f64_equiv);
  unsigned long long hs[3];
  CK(hipMemcpy(hs, stamp, 24, hipMemcpyDeviceToHost));
  printf("  workgroup 0: %.0f cycles per 32-row step (1792 = matrix pipe alone), clock %.0f MHz\n", (double)hs[0] / (double)hs[2], (double)hs[0] / ((double)hs[1] * 10.0) * 1e3);
  if (check) {
    std::vector<double> h((size_t)nsplit * ntiles * TR * TC);
    CK(hipMemcpy(h.data(), slab, h.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0, big = 0.0;
    for (int t = 0; t < ntiles; ++t) {
      int ti = 0;
      while ((ti + 1) * (ti + 2) <= t) ++ti;
      const int tj = t - ti * (ti + 1);
      for (int r = 0; r < TR; r += 5)
        for (int c = 0; c < TC; c += 3) {
          const int I = ti * TR + r, J = tj * TC + c;
          long double ref = 0.0L;
          for (int64_t n = 0; n < N; ++n) ref += (long double)kq(n, I) * (long double)kq(n, J);
          ref = ldexpl(ref, -106);
          double got = 0.0;
          for (int s = 0; s < nsplit; ++s) got += h[((size_t)s * ntiles + t) * TR * TC + r * TC + c];
          const double err = fabs((double)(got - ref));
          if (err > worst) worst = err;
          if (fabsl(ref) > big) big = (double)fabsl(ref);
        }
    }
    printf("  check vs long double: max |err| %.3e, max |Phi| %.3e, ratio %.2e (fp64 eps 1.1e-16)\n", worst, big, worst / big);
  }
  CK(hipFree(Q));
  CK(hipFree(slab));
  return ms;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 1048576;
  const int M = argc > 2 ? atoi(argv[2]) : 1024;
  int one = 1, zero = 0;
  (void)one;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_map), &zero, 4));
  if (argc > 3) {  // parameter sweep builds (-DHOLD_R= -DDMA_FIRST= -DDMA_EVERY= for v4, -DPSPLIT_V= for v6)
    run<0, 4>(8192, 256, true, 1);
    run<0, 5>(8192, 256, true, 1);
    run<0, 5>(40960, 384, true, 1, 4096);
    for (int rep = 0; rep < 3; ++rep) {
      run<0, 4>(N, M, false, 5);
      run<0, 5>(N, M, false, 5);
    }
    return 0;
  }
  run<0, 2>(8192, 256, true, 1);
  run<0, 4>(8192, 256, true, 1);
  run<0, 4>(40960, 384, true, 1, 4096);
  run<0, 4>(N, M, false, 3);
  run<1, 4>(N, M, false, 3);
  run<0, 3>(8192, 256, true, 1);
  run<0, 3>(N, M, false, 3);
  run<1, 3>(N, M, false, 3);
  run<0>(N, M, false, 3);
  run<0, 1>(N, M, false, 3);
  run<0, 2>(N, M, false, 3);
  run<5, 2>(N, M, false, 3);
  run<1, 2>(N, M, false, 3);
  return 0;
}
