// Single-launch tile-dataflow Cholesky: device code shared by the stand-alone kernel (sgp_dense.hip) and the fused
// small-problem kernel (sgp_small.hip).  See sgp_dense.hip for the design notes of the M x M back end.
#pragma once
#include "sgp_dense.hpp"

namespace sgp {

constexpr int GK = 16;        // k-chunk of the MFMA GEMMs
constexpr int GLD = 64 + 16;  // LDS row stride of an operand chunk (80 doubles: +128 B bank shift per k)

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

#ifdef SGP_POTRF_STAMPS  // measurement build only (tools/potrf_phases.py): s_memrealtime stamps of the critical path
__device__ unsigned long long g_potrf_stamps[64 * 16];
#define SGP_PSTAMP(step, k) do { if (threadIdx.x == 0 && (step) < 64) g_potrf_stamps[(step) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SGP_PSTAMP(step, k) do { } while (0)
#endif

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
  int ticket;              // (ticketed claim: the item number thread 0 drew, for the other waves)
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
// `acquire` (the chain-workgroup kernel's SGP_POTRF_ACQUIRE mode only): an agent-scope acquire behind the poll -- the CU's vector cache and
// the L2's lines of other agents' data are invalidated, so a line fetched before the flag can no longer be read stale behind it.
__device__ __forceinline__ bool df_wait(const int* flag, int* abort_flag, int* dead, bool acquire = false) {
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
    if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return *dead == 0;
}

// acc += P(64 x 64) Q(64 x 64)^T, both row-major with the contraction index contiguous (tiles of L)
// PL / QL: that operand is the tile held in sh.Ts (row-major, stride TLD) instead of global memory
template <bool PL = false, bool QL = false>
__device__ __forceinline__ void df_mac(const double* P, const double* Q, int64_t ld, int64_t ldq, DfShared& sh, d4 (&acc)[2][2]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  const int row = t >> 2, kq = (t & 3) * 4;
  double va[4][4], vb[4][4];  // the whole of both tiles: one round trip to L2, then four LDS chunks
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if constexpr (PL) {
#pragma unroll
      for (int e = 0; e < 4; ++e) va[c][e] = sh.Ts[row][c * GK + kq + e];
    } else {
      const double* sa = P + (int64_t)row * ld + c * GK + kq;
      const d2 a0 = *reinterpret_cast<const d2*>(sa), a1 = *reinterpret_cast<const d2*>(sa + 2);
      va[c][0] = a0[0]; va[c][1] = a0[1]; va[c][2] = a1[0]; va[c][3] = a1[1];
    }
    if constexpr (QL) {
#pragma unroll
      for (int e = 0; e < 4; ++e) vb[c][e] = sh.Ts[row][c * GK + kq + e];
    } else {
      const double* sb = Q + (int64_t)row * ldq + c * GK + kq;
      const d2 b0 = *reinterpret_cast<const d2*>(sb), b1 = *reinterpret_cast<const d2*>(sb + 2);
      vb[c][0] = b0[0]; vb[c][1] = b0[1]; vb[c][2] = b1[0]; vb[c][3] = b1[1];
    }
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
                                             const double* rhs, double* sol, DfShared& sh, bool acquire = false) {
  const int tid = threadIdx.x, row = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* qs = &sh.Ts[0][0];          // the solution so far (<= 4096 doubles fit the T tile)
  double* red = &sh.Dinv[0][0][0];    // 4 x 64 partial sums
  auto tile_no = [&](int ti, int tj) { return tj * nb - (tj * (tj - 1)) / 2 + (ti - tj); };
  for (int jb = 0; jb < nb; ++jb) {
    double part = 0.0;
    for (int p = 0; p < jb; ++p) {
      if (!df_wait(ready + tile_no(jb, p) * DF_FLAG_STRIDE, abort_flag, &sh.dead, acquire)) return;
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
    if (!df_wait(ready + tile_no(jb, jb) * DF_FLAG_STRIDE, abort_flag, &sh.dead, acquire)) return;
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
// wg / nwg: this workgroup's index among the nwg workgroups that share the factorization (a __global__ wrapper passes
// blockIdx.x / gridDim.x; a fused kernel may run the whole factorization in one workgroup with wg = 0, nwg = 1: every
// flag an item waits for has then been raised by the same workgroup earlier).
__device__ __forceinline__ void potrf_dataflow_body(double* A, int64_t ld, int nb, int* ready, double* dinv_g, int* info,
                                                    int info_base, const double* rhs, double* sol, double* Linv,
                                                    DfShared& sh, int wg, int nwg, int* ticket = nullptr) {
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
    SGP_PSTAMP(jd, 6);
    diag_factor64_fast(x, sh.Sp, sh.Lt, sh.prog, sh.rdiag3, sh.Dinv, &sh.bad, r, g);
    SGP_PSTAMP(jd, 7);
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
    SGP_PSTAMP(jd, 8);
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

  // Items in their own (dependency) order: dealt statically -- workgroup wg takes wg, wg + nwg, ... -- or, with `ticket` (a zeroed counter:
  // SGP_OPT_SHARED_DEVICE), drawn from it by whichever workgroup is free, so that an item is only ever held by a workgroup that is running
  // and waits only for items drawn before it (two processes on one GPU can otherwise starve each other's spinning launches).
  int j = 0, start = 0;  // column of the current tile and number of the first tile of that column
  for (int t = wg;; t += nwg) {
    if (ticket) {
      __syncthreads();
      if (tid == 0) sh.ticket = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      t = sh.ticket;
    }
    if (t >= nitem) break;
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
      const double* Lip = A + (int64_t)i * DB * ld + (int64_t)p * DB;
      // the diagonal tile's update first: it needs row i only, while for p = j - 1 the other operand of `acc`, tile
      // (j, j - 1), is the X its owner publishes just before it starts to factor (j, j) -- what is still to do after
      // that flag decides whether this item is ready when that factorization ends (stamps, tools/potrf_phases.py)
      if (head) df_mac(Lip, Lip, ld, ld, sh, accd);
      if (!df_wait(ready + tile_no(j, p) * DF_FLAG_STRIDE, abort_flag, &sh.dead)) return;
      if (head && p == j - 1) SGP_PSTAMP(i, 0);
      df_mac(Lip, A + (int64_t)j * DB * ld + (int64_t)p * DB, ld, ld, sh, acc);
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
    if (head) SGP_PSTAMP(i, 1);
    if (!df_wait(ready + tile_no(j, j) * DF_FLAG_STRIDE, abort_flag, &sh.dead)) return;
    if (head) SGP_PSTAMP(i, 2);
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
      SGP_PSTAMP(i, 3);
      __syncthreads();
      df_mac_lds(sh.Ts, accd, wi, wj, l15, l4);
      SGP_PSTAMP(i, 4);
      publish(t);  // (barrier: everybody is done reading X)
      SGP_PSTAMP(i, 5);
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

}  // namespace sgp
