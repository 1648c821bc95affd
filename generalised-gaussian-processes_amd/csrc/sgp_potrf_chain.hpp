// Single-launch Cholesky, second generation (round 5): ONE workgroup owns the whole critical path.
//
// sgp_potrf.hpp's dataflow factorization hands the critical path from workgroup to workgroup: per 64-column block
//     factor L(j,j) 10 us -> publish 2 -> seen 0.5 -> solve X = T L(j,j)^-T 4.2 -> X X^T 2 -> convert 1.5 -> factor L(j+1,j+1) ...
// i.e. 20 us per block column of which 10 are the pivot chain (profiles/r02_v7_potrf_phases_m1024.txt).  Here the CHAIN
// WORKGROUP (blockIdx 0, eight waves) keeps the diagonal tile D(j) AND the tile below it S(j) = (j+1, j) on chip:
//   * waves 0-3 ("D-waves") factor D(j) exactly as diag_factor64_fast does: wave pb runs the pivot chain of 16-column panel pb in
//     registers, the waves to its right follow column by column;
//   * waves 4-7 ("S-waves", wave 4 + g <-> rows 16 g .. of S) solve X = S L(j,j)^-T on the matrix cores PANEL BY PANEL behind the
//     chain (X_k^T = Dinv_k Y_k; Y_q -= L[q][k] X_k^T, q > k: 4 + 4 (3 - k) MFMAs per wave), and accumulate the rank-16 pieces of
//     X X^T for D(j+1) as the panels of X appear;
//   * the 16 x 16 block inverses Dinv_k are formed row by row BEHIND the chain by a wave that is idle then;
// so that when the chain of D(j) ends, what is left before the chain of D(j+1) starts is one panel step of the solve (4 MFMAs),
// one rank-16 update (<= 12 MFMAs) and the change of register layout through LDS: ~1.2 us instead of 10.
// Everything else is done by the other workgroups, as before one work item per tile, and published through flags:
//   * item (i, j), i >= j + 2: T = A(i,j) - sum_{p<j} L(i,p) L(j,p)^T (MFMA), X = T L(j,j)^-T -- now also panel by panel: the chain
//     workgroup publishes each 16-column panel of L(j,j) (+ its Dinv) as soon as it is final, so only the last panel step is
//     exposed after the chain ends;
//   * item (j+1, j) ("prep item"): the partial sums the chain workgroup needs but has no time for,
//         US(j)   = sum_{p<j}  L(j+1,p) L(j,p)^T     -> scratch `upre`, in the S-waves' register order, flag preS[j]
//         UD(j+1) = sum_{p<j}  L(j+1,p) L(j+1,p)^T   -> scratch `dpre`, in the accumulator order, flag preD[j+1]
//     (never in place: a tile must have ONE writer and no reader before it is published -- the L2s of the eight XCDs are not
//     coherent with each other for ordinary lines, see df_wait in sgp_potrf.hpp);
//   * the blocks of L^-1 (when the caller wants the inverse) are items too, row i of blocks behind column i of tiles: the transposed block
//         X(i,j)^T = [ -sum_{j<=p<i} X(p,j)^T L(i,p)^T   or   I for i = j ] L(i,i)^-T
//     is the tiles' own panel-by-panel solve against the panels the chain workgroup publishes, so the whole inverse is complete one
//     panel step behind the last chain -- tri_inverse()'s log2(nb) levels of two dependent launches each (40 us at M = 512, on the
//     critical path of C3: profiles/r05_v2_c3_timeline.txt) are gone; the optional right-hand side is the last item.
// Round-trip costs between CUs decide the rest (profiles/r05_potrf_chain_phases_*.txt: a publication with an agent-scope release fence,
// i.e. an L2 write-back, 1 - 2.5 us; the consumer's first load of the published lines 1.5 - 2 us when they come from memory): the
// chain of hops  last panel of L(j-1,j-1) -> tile (j+1,j-1) -> its product with X(j-1) -> US(j) -> the S-waves  is ~10 us long after
// the chain of D(j-1) ends.  The eight XCDs have one L2 each, so a producer and a consumer ON THE SAME XCD need no write-back: data
// that has reached the L2 (s_waitcnt vmcnt(0)) is visible to every CU of that XCD.  The chain workgroup therefore publishes twice --
// a "light" flag as soon as its stores have completed, the ordinary flag after the write-back -- and the two items per column that sit
// on the critical path (tile (j+2, j) + the last term of US / UD) are dealt to workgroups that are expected on the chain workgroup's
// XCD (workgroup w -> XCD w mod 8).  Expected, not assumed: every workgroup reads its XCC_ID, and a light flag / a light publication
// is used only where the two ids really agree; anything else takes the ordinary path.
// Deadlock freedom is the old argument: every dependency of an item is an item with a smaller number or a step of the chain
// workgroup with a smaller-or-equal column, the chain workgroup's step j depends on items of columns < j only (+ the first half of
// the prep item of column j, which depends on columns <= j - 2), and all workgroups of the launch are resident.
#pragma once
#include "sgp_potrf.hpp"
#include "sgp_potrf_items.hpp"

namespace sgp {

constexpr int CH_THREADS = 512;

// accumulator blocks of X X^T (lower triangle of the 4 x 4 grid of 16 x 16 blocks) dealt to the four S-waves: (row, column) block
__device__ __forceinline__ constexpr int ch_nblk(int g) { return g < 2 ? 3 : 2; }
__device__ __forceinline__ constexpr int ch_bg(int g, int b) { return b == 0 ? g : (b == 1 ? (g == 0 ? 2 : g) : 3); }
__device__ __forceinline__ constexpr int ch_bh(int g, int b) { return b == 0 ? g : (b == 1 ? (g == 0 ? 0 : g - 1) : g); }
//   g = 0: (0,0) (2,0) (3,0)    g = 1: (1,1) (1,0) (3,1)    g = 2: (2,2) (2,1)    g = 3: (3,3) (3,2)
// the inverse: lower block (R, C) -> (owner wave) * 3 + (index among its blocks)
__device__ __forceinline__ int ch_owner_slot(int R, int C) {
  if (R == C) return R * 3;
  if (C == 0) return R == 1 ? 4 : (R == 2 ? 1 : 2);   // (1,0) -> g 1 b 1; (2,0) -> g 0 b 1; (3,0) -> g 0 b 2
  if (C == 1) return R == 2 ? 7 : 5;                  // (2,1) -> g 2 b 1; (3,1) -> g 1 b 2
  return 10;                                          // (3,2) -> g 3 b 1
}

struct ChShared {
  union {
    struct { double Sp[4][DB][PLD]; double Dinv[4][16][17]; } f;  // panels of L(j,j) (row-major) and the block inverses
    double Ts[DB][TLD];                                           // step boundary: the accumulators of D(j+1) on their way to the D-waves
  };
  double Lt[4][16][DB];    // finished columns of D(j), transposed (followers of the pivot chain)
  double Xp[4][4][4][64];  // X = L(j+1,j): [panel k][row block g][k-step sq][lane] = X[16 g + (lane & 15)][16 k + 4 sq + (lane >> 4)]
  double rdiag[4][16];     // reciprocal pivots (the block inverses are formed behind the chain)
  int prog[4];             // columns of panel pb finished
  int dinv_done[4];
  int xa_done[4];          // S-waves that have stored panel k of X
  int bad;
  int dead;
  int xpub;                // S-waves that have written back their rows of X (cumulative over the steps)
};

// scratch layout (ints): [ready: ntile | abort | preS: nb | preD: nb | pready: 4 nb | preSE: nb | preDE: nb | pready_l: 4 nb | xready_l: nb | cwx |
// iready: ntile | lo2r: nb | lo2r_l: nb] x DF_FLAG_STRIDE, then doubles:
// dinv_g nb x 1024 | upre nb x 4096 | dpre nb x 4096 | lo2 nb x 4096 | upe nb x 4096 | dpe nb x 4096 | xt ntile x 4096
struct ChScratch {
  int* ready;
  int* abort_flag;
  int* preS;
  int* preD;
  int* pready;
  double* dinv_g;
  double* upre;
  double* dpre;
  double* lo2;   // the ORIGINAL entries of tile (c+2, c), handed from FUSED_D(c) to FUSED_S(c) (flag lo2r[c]): nb x 4096, thread for thread
  double* upe;   // early terms of US / UD (slots 2 / 3), in the accumulator order of the workgroup that adds the last term
  double* dpe;
  int* preSE;
  int* preDE;
  int* pready_l;  // "light" twins of pready / of the flags of the tiles (j+1, j): the data is in the chain workgroup's L2, not written back
  int* xready_l;
  int* cwx;       // the chain workgroup's XCC id + 1
  int* iready;    // blocks of L^-1 (same numbering as the tiles)
  int* lo2r;      // lo2[c] is there (written back) ...
  int* lo2r_l;    // ... or in the L2 of the chain workgroup's XCD (light twin, as pready_l)
  double* xt;     // ... and their transposes X(i,j)^T, row-major 64 x 64: the left operand of the rows below
  int* ticket;    // claim counters of the ticketed deal: the rest list (or the one list of a launch without critical workgroups) ...
  int* ticket_crit;  // ... and the critical list
  int* base;      // the scratch's first word (flag slot numbers of the trace build)
};
// one flag = one 128-byte line; every scratch block (1024 / 4096 doubles) and every tile row segment (64 doubles, ld a multiple of 16 --
// checked per call in potrf_lower) is a whole number of lines: no two writers, and no writer and an early reader, ever share a line
static_assert(DF_FLAG_STRIDE * sizeof(int) == 128 && (DB * sizeof(double)) % 128 == 0 && (1024 * sizeof(double)) % 128 == 0, "line granularity of the hand-overs");
__host__ __device__ inline size_t ch_flag_slots(int nb) { return (size_t)ch_flag_base(CF_NKIND, nb); }  // (sgp_potrf_items.hpp: the access table's numbering)
__host__ __device__ inline size_t ch_scratch_doubles(int nb) { return (size_t)nb * 1024 + (size_t)nb * 4096 * 5 + (size_t)nb * (nb + 1) / 2 * 4096; }
__device__ __forceinline__ ChScratch ch_scratch(int* scratch, int nb) {
  ChScratch s;
  auto at = [&](int kind) { return scratch + (size_t)ch_flag_base(kind, nb) * DF_FLAG_STRIDE; };
  s.ready = at(CF_READY);
  s.abort_flag = at(CF_ABORT);
  s.preS = at(CF_PRES);
  s.preD = at(CF_PRED);
  s.pready = at(CF_PREADY);
  s.preSE = at(CF_PRESE);
  s.preDE = at(CF_PREDE);
  s.pready_l = at(CF_PREADY_L);
  s.xready_l = at(CF_XREADY_L);
  s.cwx = at(CF_CWX);
  s.iready = at(CF_IREADY);
  s.lo2r = at(CF_LO2R);
  s.lo2r_l = at(CF_LO2R_L);
  s.ticket = at(CF_TICKET);
  s.ticket_crit = at(CF_TICKET_CRIT);
  s.base = scratch;
  s.dinv_g = reinterpret_cast<double*>(scratch + ch_flag_slots(nb) * DF_FLAG_STRIDE);
  s.upre = s.dinv_g + (size_t)nb * 1024;
  s.dpre = s.upre + (size_t)nb * 4096;
  s.lo2 = s.dpre + (size_t)nb * 4096;
  s.upe = s.lo2 + (size_t)nb * 4096;
  s.dpe = s.upe + (size_t)nb * 4096;
  s.xt = s.dpe + (size_t)nb * 4096;
  return s;
}

// Launch modes (potrf_lower: environment knobs read once, A/B and diagnosis; the ticketed claim is SGP_OPT_SHARED_DEVICE of the call's context)
constexpr int CH_MODE_ACQUIRE = 1;   // SGP_POTRF_ACQUIRE=1: an agent-scope acquire (buffer_inv sc1) behind every successful flag poll -- drops the
                                     // "never touched before its publication" invariant the consumer side otherwise relies on (checker: H4)
constexpr int CH_MODE_NOLIGHT = 2;   // SGP_POTRF_LIGHT=0: no same-XCD light flags / publications, every hand-over takes the ordinary release path
constexpr int CH_MODE_TICKET = 4;    // SGP_OPT_SHARED_DEVICE (or SGP_POTRF_TICKET=1): the ticketed claim of sgp_potrf_items.hpp instead of the static deal

// Trace build (-DSGP_CH_TRACE, tools/potrf_trace_check.py): every wait that returned and every flag raised, per workgroup, in the flag
// numbering of the access table -- the kernel's real synchronisation held against ch_item_program / ch_chain_*_program.
#ifdef SGP_CH_TRACE
constexpr int CH_TRACE_WG = 258, CH_TRACE_LEN = 8192;
__device__ int g_ch_trace[CH_TRACE_WG][CH_TRACE_LEN];
__device__ int g_ch_trace_n[CH_TRACE_WG];
__device__ __forceinline__ void ch_trace_put(int slot_wg, int code) {
  const int n = atomicAdd(&g_ch_trace_n[slot_wg], 1);
  if (n < CH_TRACE_LEN) g_ch_trace[slot_wg][n] = code;
}
// codes: 1 << 28 | slot: wait returned; 2 << 28 | light << 27 | slot: raise; 4 << 28 | kind << 20 | c << 10 | i: an item begins; 5 << 28 | lite: its protocol;
// 6 << 28 | j: a step of the chain workgroup begins.  Trace slots: 0 = D-waves, 257 = S-waves, 1 + ow = the other workgroups.
#define CH_TR_WAIT(sc, wg, flagptr) ch_trace_put((wg), (1 << 28) | (int)(((flagptr) - (sc).base) / DF_FLAG_STRIDE))
#define CH_TR_RAISE(sc, wg, flagptr, light) ch_trace_put((wg), (2 << 28) | ((light) ? (1 << 27) : 0) | (int)(((flagptr) - (sc).base) / DF_FLAG_STRIDE))
#define CH_TR_MARK(wg, code) ch_trace_put((wg), (code))
#else
#define CH_TR_WAIT(sc, wg, flagptr) do { } while (0)
#define CH_TR_RAISE(sc, wg, flagptr, light) do { } while (0)
#define CH_TR_MARK(wg, code) do { } while (0)
#endif

#ifdef SGP_POTRF_STAMPS
#define CH_STAMP(step, k) do { if ((threadIdx.x & 63) == 0 && (step) < 64) g_potrf_stamps[(step) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CH_STAMP(step, k) do { } while (0)
#endif

// "my stores have reached the L2" (an explicit wait: in the default execution mode a workgroup-scope release fence does not wait for
// vector-memory stores).  Enough for a consumer on the same XCD; a consumer elsewhere needs the agent-scope release (L2 write-back).
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// XCC (= XCD) this wave runs on: HW_REG_XCC_ID (id 20), bits 3:0
__device__ __forceinline__ int my_xcc() { return (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }

__device__ __forceinline__ int lds_flag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_flag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// A wave of the chain workgroup waits for a global flag to reach `want` (every lane polls the same word: one request).  Bounded:
// after DF_SPIN_LIMIT polls, or when another workgroup has given up, the launch is marked aborted and the wave carries on with
// whatever it has -- every loop of the chain workgroup stays finite, the caller turns the abort flag into SGP_INFO_TIMEOUT.
__device__ __forceinline__ void ch_wait_global(const int* flag, int want, int* abort_flag, int* dead, bool acquire = false) {
  if (lds_flag_load(dead)) return;
  int spins = 0;
  while (df_flag_load(flag) < want) {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 255) == 0 && (df_flag_load(abort_flag) != 0 || lds_flag_load(dead))) { lds_flag_store(dead, 1); break; }
    if (spins > DF_SPIN_LIMIT) {
      __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lds_flag_store(dead, 1);
      break;
    }
  }
  if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("" ::: "memory");
}

// 16 x 16 block inverse of panel pb, one row behind the pivot chain (lanes < 16 <-> column c); then to global + the panel's flag
__device__ __forceinline__ void ch_dinv_follow(ChShared& sh, int pb, int lane, double* dg, int* pflag, int* pflag_l, const ChScratch& sc, int trwg) {
  if (lane < 16) {
    double y[16], lrow[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      double rd;
      for (;;) {  // counter and row prefix back to back (LDS keeps a wave's order: a row read after a counter value > rr is complete)
        const int done = lds_flag_load(&sh.prog[pb]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; 2 * e < rr; ++e) {
          const d2 t2 = *reinterpret_cast<const d2*>(&sh.f.Sp[pb][16 * pb + rr][2 * e]);
          lrow[2 * e] = t2[0];
          lrow[2 * e + 1] = t2[1];
        }
        rd = sh.rdiag[pb][rr];
        asm volatile("" ::: "memory");
        if (done > rr) break;
        __builtin_amdgcn_s_sleep(1);
      }
      dinv_row(y, lrow, rr, lane, rd);
      sh.f.Dinv[pb][rr][lane] = y[rr];
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) dg[pb * 256 + rr * 16 + lane] = y[rr];
  }
  asm volatile("" ::: "memory");
  lds_flag_store(&sh.dinv_done[pb], 1);
  stores_done();
  if (lane == 0) { __hip_atomic_fetch_add(pflag_l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); CH_TR_RAISE(sc, trwg, pflag_l, 1); }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (lane == 0) { __hip_atomic_fetch_add(pflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); CH_TR_RAISE(sc, trwg, pflag, 0); }
  (void)sc; (void)trwg;
}

// One D-wave's share of the factorization of D(j): a[] = row `lane`, columns 16 g .. of the fully updated tile.
__device__ __forceinline__ void ch_dwave(double (&a)[16], ChShared& sh, int g, int lane, double* Ajj, int64_t ld, double* dg,
                                         int* pready, int* pready_l, int step, const ChScratch& sc) {
  const int i = lane;
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    if (g == pb) {
#ifndef SGP_CH_PRIO
#define SGP_CH_PRIO 3
#endif
      __builtin_amdgcn_s_setprio(SGP_CH_PRIO);
      const int base = 16 * pb;
      double p = a[0], lprev = 0.0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = base + jj;
        double pnext = 0.0;
        if (jj < 15) {
          double s0 = a[jj + 1], s1 = 0.0;
          const double* row = &sh.f.Sp[pb][j + 1][0];
          const int nl = jj > 0 ? jj - 1 : 0;
#pragma unroll
          for (int k = 0; k + 1 < nl; k += 2) {
            const d2 r2 = *reinterpret_cast<const d2*>(row + k);
            s0 = fma(-a[k], r2[0], s0);
            s1 = fma(-a[k + 1], r2[1], s1);
          }
          if (nl & 1) s0 = fma(-a[nl - 1], row[nl - 1], s0);
          if (jj > 0) s1 = fma(-a[jj - 1], readlane_f64(lprev, j + 1), s1);
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
        if (!(d > 0.0)) {
          if (i == 0 && sh.bad == 0) sh.bad = j + 1;
          d = 1.0;
        }
        const double rs = rsqrt_newton(d);
        const double l = afull * rs;
        a[jj] = l;
        sh.f.Sp[pb][i][jj] = (i >= j) ? l : 0.0;
        sh.Lt[pb][jj][i] = l;
        if (i == 0) sh.rdiag[pb][jj] = rs;
        asm volatile("" ::: "memory");
        lds_flag_store(&sh.prog[pb], jj + 1);
        lprev = l;
        p = pnext;
      }
      __builtin_amdgcn_s_setprio(0);
      CH_STAMP(step, 1 + pb);
      // the panel is final: to global at once (the other workgroups solve panel by panel), zeros above the diagonal
      double* dst = Ajj + (int64_t)i * ld + base;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        *reinterpret_cast<d2*>(dst + 2 * k) = d2{(base + 2 * k <= i) ? a[2 * k] : 0.0, (base + 2 * k + 1 <= i) ? a[2 * k + 1] : 0.0};
      stores_done();
      if (i == 0) { __hip_atomic_fetch_add(pready_l + pb * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); CH_TR_RAISE(sc, 0, pready_l + pb * DF_FLAG_STRIDE, 1); }
      if (pb == 3) CH_STAMP(step, 9);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (i == 0) { __hip_atomic_fetch_add(pready + pb * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); CH_TR_RAISE(sc, 0, pready + pb * DF_FLAG_STRIDE, 0); }
    } else if (g > pb) {
#pragma unroll 1
      for (int jj = 0; jj < 16; ++jj) {
        while (lds_flag_load(&sh.prog[pb]) <= jj) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const double own = sh.Lt[pb][jj][i];
        double lc[16];
        lds_row16(lc, &sh.Lt[pb][jj][16 * g]);
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = fma(-own, lc[k], a[k]);
      }
    }
    if (g + 1 == pb) ch_dinv_follow(sh, pb, lane, dg, pready + pb * DF_FLAG_STRIDE, pready_l + pb * DF_FLAG_STRIDE, sc, 0);  // idle by now: the next panel's block inverse
  }
}

// The chain workgroup.  A: the matrix (in place), Linv is NOT written here (block inverses are items of the other workgroups).
// The two roles run separate loops (their register sets never coexist in one wave) with the same three barriers per step:
//   B1  accumulators of D(j) are in sh.Ts, the step's LDS flags are cleared     B2  the D-waves have read their rows: Ts -> panels
//   B3  step done: every wave has written back (release) what it stored to global; thread 0 raises the tiles' flags
__device__ __forceinline__ void chain_d_role(double* A, int64_t ld, int nb, ChScratch sc, int* info, int info_base, ChShared& sh) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto tile_no = [&](int ti, int tj) { return tj * nb - (tj * (tj - 1)) / 2 + (ti - tj); };
  if (tid == 0) {
    sh.dead = 0;
    sh.xpub = 0;
    __hip_atomic_store(sc.cwx, my_xcc() + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CH_TR_RAISE(sc, 0, sc.cwx, 0);
  }
  for (int j = 0; j < nb; ++j) {
    double* Ajj = A + (int64_t)j * DB * (ld + 1);
    if (tid < 4) { sh.prog[tid] = 0; sh.dinv_done[tid] = 0; sh.xa_done[tid] = 0; }
    if (tid == 0) sh.bad = 0;
    if (tid == 0) CH_TR_MARK(0, (6 << 28) | j);
    __syncthreads();  // B1
    double xd[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const d2 t2 = *reinterpret_cast<const d2*>(&sh.Ts[lane][16 * g + 2 * k]);
      xd[2 * k] = -t2[0];
      xd[2 * k + 1] = -t2[1];
    }
    __syncthreads();  // B2
    CH_STAMP(j, 0);
    ch_dwave(xd, sh, g, lane, Ajj, ld, sc.dinv_g + (size_t)j * 1024, sc.pready + (size_t)4 * j * DF_FLAG_STRIDE,
             sc.pready_l + (size_t)4 * j * DF_FLAG_STRIDE, j, sc);
    __syncthreads();  // B3
    CH_STAMP(j, 8);
    if (tid == 0) {
      if (sh.bad != 0 && *info == 0) *info = info_base + j * DB + sh.bad;
      __hip_atomic_store(sc.ready + (size_t)tile_no(j, j) * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      CH_TR_RAISE(sc, 0, sc.ready + (size_t)tile_no(j, j) * DF_FLAG_STRIDE, 0);
      if (j + 1 < nb) {  // X = L(j+1,j): in this XCD's L2 (the S-waves waited for their stores); its ordinary flag follows the write-back
        __hip_atomic_store(sc.xready_l + (size_t)j * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CH_TR_RAISE(sc, 0, sc.xready_l + (size_t)j * DF_FLAG_STRIDE, 1);
      }
    }
  }
}

__device__ __forceinline__ void chain_s_role(double* A, int64_t ld, int nb, ChScratch sc, ChShared& sh, int mode) {
  const bool acq = (mode & CH_MODE_ACQUIRE) != 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nblk = ch_nblk(g);
  auto tile_no = [&](int ti, int tj) { return tj * nb - (tj * (tj - 1)) / 2 + (ti - tj); };
  int brow[3], bcol[3];  // this wave's accumulator blocks
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    brow[b] = ch_bg(g, b);
    bcol[b] = ch_bh(g, b);
  }
  d4 accd[3];  // blocks of [X X^T + UD - A] for the NEXT diagonal tile (the D-waves take -accd)
  auto load_diag_blocks = [&](int jd) __attribute__((always_inline)) {
    const double* Ad = A + (int64_t)jd * DB * (ld + 1);
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if (b < nblk) {
        const double* src = Ad + (int64_t)(16 * brow[b] + l4) * ld + 16 * bcol[b] + l15;
#pragma unroll
        for (int q = 0; q < 4; ++q) accd[b][q] = -src[(int64_t)4 * q * ld];
      }
  };
  load_diag_blocks(0);

  for (int j = 0; j < nb; ++j) {
    double* dg = sc.dinv_g + (size_t)j * 1024;
    int* pready = sc.pready + (size_t)4 * j * DF_FLAG_STRIDE;
    // ---- step boundary: accumulators (MFMA layout) -> LDS -> rows (D-waves) ----
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if (b < nblk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sh.Ts[16 * brow[b] + l4 + 4 * q][16 * bcol[b] + l15] = accd[b][q];
      }
    if (tid == 256) CH_TR_MARK(257, (6 << 28) | j);
    __syncthreads();  // B1
    __syncthreads();  // B2: Ts becomes the panels / block inverses of this step
    if (j > 0) {
      // X of the previous step: written back now, beside this step's chain (the write-back took 1 - 2.5 us between two chains);
      // every S-wave releases its own rows, the last one to do so raises the tile's ordinary flag
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      int last = 0;
      if (lane == 0) last = __hip_atomic_fetch_add(&sh.xpub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 4 * j - 1;
      if (__builtin_amdgcn_readfirstlane(last)) {
        __hip_atomic_store(sc.ready + (size_t)tile_no(j, j - 1) * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) CH_TR_RAISE(sc, 257, sc.ready + (size_t)tile_no(j, j - 1) * DF_FLAG_STRIDE, 0);
      }
    }
    if (g == 0) ch_dinv_follow(sh, 0, lane, dg, pready, sc.pready_l + (size_t)4 * j * DF_FLAG_STRIDE, sc, 257);  // beside the chain of panel 0
    if (j + 1 < nb) {
      double* Asj = A + (int64_t)(j + 1) * DB * ld + (int64_t)j * DB;
      // S = A(j+1,j) - US(j): this wave's 16 rows as four 16 x 16 blocks in the accumulator layout
      // (component sq of block pb, lane l: S[16 g + (l & 15)][16 pb + 4 sq + (l >> 4)])
      d4 yb[4];
      {
        const double* src = Asj + (int64_t)(16 * g + l15) * ld + l4;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int sq = 0; sq < 4; ++sq) yb[pb][sq] = src[16 * pb + 4 * sq];
      }
      load_diag_blocks(j + 1);
      {  // the mirror of this tile is the strictly-upper part of the result: zero (this wave: 16 of its rows)
        double* U = A + (int64_t)j * DB * ld + (int64_t)(j + 1) * DB + (int64_t)(16 * g + (lane >> 2)) * ld + (lane & 3) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<d2*>(U + 2 * e) = d2{0.0, 0.0};
      }
      if (j >= 1) {  // US(j) = sum_{p<j} L(j+1,p) L(j,p)^T from the prep item of this column
        ch_wait_global(sc.preS + (size_t)j * DF_FLAG_STRIDE, 1, sc.abort_flag, &sh.dead, acq);
        if (g == 0 && lane == 0) CH_TR_WAIT(sc, 257, sc.preS + (size_t)j * DF_FLAG_STRIDE);
        const double* up = sc.upre + (size_t)j * 4096 + (size_t)g * 1024 + lane;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int sq = 0; sq < 4; ++sq) yb[pb][sq] -= up[(pb * 4 + sq) * 64];
      }
      if (g == 0) CH_STAMP(j, 5);
      d4 dp[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) dp[b] = d4{0.0, 0.0, 0.0, 0.0};
      // UD(j+1) = sum_{p<j} L(j+1,p) L(j+1,p)^T (FUSED_D item of column j - 1): wanted at the very end only -- asked for before the last
      // panel step when it is there by then, otherwise waited for behind the last update (never in front of work that could proceed)
      bool have_ud = false;
      auto fetch_ud = [&]() __attribute__((always_inline)) {
        const double* dpp = sc.dpre + (size_t)(j + 1) * 4096 + (size_t)g * 768 + lane;
#pragma unroll
        for (int b = 0; b < 3; ++b)
          if (b < nblk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dp[b][q] = dpp[b * 256 + q * 64];
          }
      };
      // rank-16 update of the next diagonal tile by panel kk of X -- whenever all four S-waves have stored that panel; never waited
      // for while a panel step of the solve is possible (a late S = A - US then catches up without a cross-wave wait per panel)
      int pending = 0;
      auto update_ready = [&]() __attribute__((always_inline)) { return lds_flag_load(&sh.xa_done[pending]) >= 4; };
      auto update = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        const double (*xp)[4][64] = sh.Xp[pending];
#pragma unroll
        for (int b = 0; b < 3; ++b)
          if (b < nblk) {
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) accd[b] = mfma16(xp[brow[b]][sq][lane], xp[bcol[b]][sq][lane], accd[b]);
          }
        ++pending;
      };
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k == 3 && j >= 1 && df_flag_load(sc.preD + (size_t)(j + 1) * DF_FLAG_STRIDE) != 0) {
          if (acq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          fetch_ud();  // already there: its loads fly while the chain is in its last panel
          have_ud = true;
        }
        while (lds_flag_load(&sh.prog[k]) < 16 || lds_flag_load(&sh.dinv_done[k]) == 0) {
          if (pending < k && update_ready()) update();
          else __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        d4 xa = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) xa = mfma16(sh.f.Dinv[k][l15][4 * sq + l4], yb[k][sq], xa);
#pragma unroll
        for (int q = k + 1; q < 4; ++q)
#pragma unroll
          for (int sq = 0; sq < 4; ++sq) yb[q] = mfma16(-sh.f.Sp[k][16 * q + l15][4 * sq + l4], xa[sq], yb[q]);
        {
          double* dst = Asj + (int64_t)(16 * g + l15) * ld + 16 * k + l4;
#pragma unroll
          for (int sq = 0; sq < 4; ++sq) {
            sh.Xp[k][g][sq][lane] = xa[sq];
            dst[4 * sq] = xa[sq];
          }
        }
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&sh.xa_done[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == 3 && g == 0) CH_STAMP(j, 6);
      }
      while (pending < 4) {
        while (!update_ready()) __builtin_amdgcn_s_sleep(1);
        update();
      }
      if (j >= 1 && !have_ud) {
        ch_wait_global(sc.preD + (size_t)(j + 1) * DF_FLAG_STRIDE, 1, sc.abort_flag, &sh.dead, acq);
        fetch_ud();
      }
      if (j >= 1 && g == 0 && lane == 0) CH_TR_WAIT(sc, 257, sc.preD + (size_t)(j + 1) * DF_FLAG_STRIDE);
#pragma unroll
      for (int b = 0; b < 3; ++b) accd[b] += dp[b];
      if (g == 0) CH_STAMP(j, 7);
      stores_done();  // this wave's rows of X are in the L2: thread 0 raises the tile's light flag behind B3
    }
    __syncthreads();  // B3
  }
}

// df_wait for a counter: one lane polls until the flag reaches `want`
__device__ __forceinline__ bool df_wait_count(const int* flag, int want, int* abort_flag, int* dead) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (df_flag_load(flag) < want) {
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

// Waits for panel pb of a diagonal tile (two contributions: the columns and their block inverse) and reports how many panels are
// public by then (pb + 1 .. 4; the same number in every thread), or -1 when the launch has been aborted.
__device__ __forceinline__ int df_wait_panels(const int* pready, int pb, int* abort_flag, DfShared& sh, bool acquire = false) {
  if (threadIdx.x == 0) {
    // all four counters in one round trip (an agent-scope load goes past the L2: ~1 us each when asked one after the other)
    int f[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = df_flag_load(pready + q * DF_FLAG_STRIDE);
    int spins = 0;
    while (f[pb] < 2) {
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int q = 0; q < 4; ++q) f[q] = df_flag_load(pready + q * DF_FLAG_STRIDE);
      ++spins;
      if ((spins & 255) == 0 && df_flag_load(abort_flag) != 0) { sh.dead = 1; break; }
      if (spins > DF_SPIN_LIMIT) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh.dead = 1;
        break;
      }
    }
    int upto = pb + 1;
    while (upto < 4 && f[upto] >= 2) ++upto;
    sh.bad = upto;
    if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return sh.dead ? -1 : sh.bad;
}

// The other workgroups (four waves each).  Work items:
//   EARLY_S(c), EARLY_D(c)   the terms of US(c+1) / UD(c+2) from the columns p < c (needed late, ready early)
//   FUSED_D(c)               tile (c+2, c) in place (its original entries copied to scratch first; published at once), then UD(c+2) = early terms + X X^T
//   FUSED_S(c)               tile (c+2, c) once more, in registers, from that copy; then US(c+1) = early terms + L(c+1,c) X^T   -> the chain workgroup's S-waves
//   TILE(i, c), i >= c + 3   rank-64 updates, then the solve panel by panel as the chain workgroup publishes L(c,c)
//   INV(i, j), i >= j          block (i, j) of L^-1 (and its transpose into scratch), behind column i's tiles
//   RHS                      sol = L^-1 rhs
// Two lists, each in dependency order (every dependency of an item is an earlier item of one of the lists or a step of the chain
// workgroup): the CRITICAL one [FUSED_D(0), FUSED_S(0), FUSED_D(1), ...] is dealt round-robin to the workgroups expected on the chain
// workgroup's XCD (blockIdx = 0 mod 8), the other one to the rest; with fewer than eight workgroups there is one list for all.
__device__ __forceinline__ void chain_outside(double* A, int64_t ld, int nb, ChScratch sc, const double* rhs, double* sol, double* Linv,
                                              DfShared& sh, int ow, int nout, int mode) {
  const bool acq = (mode & CH_MODE_ACQUIRE) != 0;
  const int trwg = 1 + ow;  // (trace build: this workgroup's log)
  (void)trwg;
  const int tid = threadIdx.x, r = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, l4 = lane >> 4;
  int* ready = sc.ready;
  int* abort_flag = sc.abort_flag;
  if (tid == 0) sh.dead = 0;
  __syncthreads();
  auto tile_no = [&](int ti, int tj) { return tj * nb - (tj * (tj - 1)) / 2 + (ti - tj); };
  auto zero_acc = [&](d4 (&acc)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) acc[u][v] = d4{0.0, 0.0, 0.0, 0.0};
  };
  auto acc_to_ts = [&](const d4 (&acc)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) sh.Ts[wi * 32 + u * 16 + l4 + 4 * q][wj * 32 + v * 16 + l15] = acc[u][v][q];
  };
  auto raise = [&](int* flag) __attribute__((always_inline)) {  // stores written back (release) -> barrier -> flag
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); CH_TR_RAISE(sc, trwg, flag, 0); }
  };
  auto raise_light = [&](int* flag) __attribute__((always_inline)) {  // the only reader shares this XCD's L2: completed -> barrier -> flag
    stores_done();
    __syncthreads();
    if (tid == 0) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); CH_TR_RAISE(sc, trwg, flag, 1); }
  };
  auto wait_flag = [&](const int* flag) __attribute__((always_inline)) {
    const bool ok = df_wait(flag, abort_flag, &sh.dead, acq);
    if (tid == 0) CH_TR_WAIT(sc, trwg, flag);
    return ok;
  };

  // ---- which list, and: does this workgroup really share the chain workgroup's L2? ----
  const bool want_inv = Linv != nullptr, want_rhs = rhs != nullptr;
  // Which items: the static deal (default: fastest when the launch has the device to itself -- C3 0.41 against 0.45 ms per evaluation),
  // or a TICKETED claim (SGP_OPT_SHARED_DEVICE; sgp_potrf_items.hpp: ch_claim_next) -- an item is only ever held by a RUNNING workgroup and
  // waits only for items claimed before it and for the chain workgroup, so progress no longer needs every workgroup of the launch to be
  // resident at once (two processes on one GPU: VERDICT r5 weak-5).  The arithmetic of an item does not depend on who runs it: same bits.
  const bool ticketed = (mode & CH_MODE_TICKET) != 0;
  const ChDeal deal = ch_deal(ow, nout, nb, want_inv, want_rhs);  // (sgp_potrf_items.hpp)
  const int first = deal.first, stride = deal.stride, count = deal.count;
  struct Atomics {   // ch_claim_next's view of the two counters: thread 0 asks, the answer goes round through sh.bad
    int* ctr[2];
    DfShared& sh;
    int tid;
    int first_wave;   // critical tickets the critical workgroups draw by themselves when they start
    __device__ __forceinline__ int bcast(int v) {
      __syncthreads();  // (sh.bad: its previous readers are behind this barrier)
      if (tid == 0) sh.bad = v;
      __syncthreads();
      const int r = sh.bad;
      __syncthreads();
      return r;
    }
    __device__ __forceinline__ int take(int list) {
      int v = 0;
      if (tid == 0) v = __hip_atomic_fetch_add(ctr[list], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return bcast(v);
    }
    __device__ __forceinline__ int take_below(int list, int bound) {
      int v = -1;
      if (tid == 0) {
        int cur = df_flag_load(ctr[list]);
        // Patience, but only while the launch is still starting: the critical workgroups (blockIdx 8, 16, ...) are dispatched behind the
        // others and draw their first tickets a few microseconds into the launch -- a helper that steps in before that only moves a
        // critical item away from the chain workgroup's XCD (first version, 64 polls of patience: 10 of 28 critical items at M = 1024 ran
        // elsewhere).  Once `first_wave` tickets are out the counter says what it says: no waiting.
        for (int spins = 0; cur < bound && cur < first_wave && spins < 512; ++spins) {
          __builtin_amdgcn_s_sleep(2);
          cur = df_flag_load(ctr[list]);
        }
        while (cur < bound) {
          if (__hip_atomic_compare_exchange_strong(ctr[list], &cur, cur + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            v = cur;
            break;
          }
        }
      }
      return bcast(v);
    }
  } atomics{{sc.ticket, sc.ticket_crit}, sh, tid, nout / 8 < ch_crit_items(nb) ? nout / 8 : ch_crit_items(nb)};
  ChClaim claim;
  bool local = false;  // same XCD as the chain workgroup (decided once it has said where it runs; asked only by the fused items)
  bool local_known = false;
  auto ask_local = [&]() __attribute__((always_inline)) {
    if (local_known) return true;
    if (tid == 0) {
      int spins = 0, v;
      while ((v = df_flag_load(sc.cwx)) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > DF_SPIN_LIMIT || ((spins & 255) == 0 && df_flag_load(abort_flag) != 0)) {
          __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sh.dead = 1;
          break;
        }
      }
      CH_TR_WAIT(sc, trwg, sc.cwx);
      sh.bad = (v - 1 == my_xcc() && !(mode & CH_MODE_NOLIGHT)) ? 1 : 0;
    }
    __syncthreads();
    if (sh.dead) return false;
    local = sh.bad != 0;
    local_known = true;
    __syncthreads();  // (sh.bad is reused by df_wait_panels)
    return true;
  };

  // X = Y L(jd,jd)^-T panel by panel as the chain workgroup publishes L(jd,jd): yb / xb = this wave's 16 rows of Y / X as four 16 x 16
  // blocks in the accumulator layout.  Panels already public are fetched in one go (an item whose updates end late finds all four
  // there: one round trip, not four).  false: the launch has been aborted.
  auto solve_panels = [&](int jd, const int* pready, d4 (&yb)[4], d4 (&xb)[4], bool stamp) __attribute__((always_inline)) {
    const double* Ljj = A + (int64_t)jd * DB * (ld + 1);
    const double* dg = sc.dinv_g + (size_t)jd * 1024;
    auto fetch_panel = [&](int pb) __attribute__((always_inline)) {
      const double* src = Ljj + (int64_t)r * ld + 16 * pb + 4 * g;  // 64 rows x 16 columns: thread <-> (row, 4 columns)
      const d2 v0 = *reinterpret_cast<const d2*>(src), v1 = *reinterpret_cast<const d2*>(src + 2);
      const double dv = dg[pb * 256 + tid];
      sh.Sp[pb][r][4 * g] = v0[0];
      sh.Sp[pb][r][4 * g + 1] = v0[1];
      sh.Sp[pb][r][4 * g + 2] = v1[0];
      sh.Sp[pb][r][4 * g + 3] = v1[1];
      sh.Dinv[pb][tid >> 4][tid & 15] = dv;
    };
    int have = 0;  // panels in LDS
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      if (pb >= have) {
        const int upto = df_wait_panels(pready, pb, abort_flag, sh, acq);
        if (upto < 0) return false;
        for (int q = pb; q < upto; ++q) {
          fetch_panel(q);
          if (tid == 0) CH_TR_WAIT(sc, trwg, pready + q * DF_FLAG_STRIDE);
        }
        have = upto;
        __syncthreads();
      }
      if (pb == 3 && stamp && tid == 0) CH_STAMP(jd, 12);
      d4 xa = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) xa = mfma16(sh.Dinv[pb][l15][4 * sq + l4], yb[pb][sq], xa);
      xb[pb] = xa;
#pragma unroll
      for (int q = pb + 1; q < 4; ++q)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) yb[q] = mfma16(-sh.Sp[pb][16 * q + l15][4 * sq + l4], xa[sq], yb[q]);
    }
    return true;
  };

  for (int k = first;; k += stride) {
    if (!ticketed && k >= count) break;
    const ChItem it = ticketed ? ch_claim_next(claim, deal.split, deal.crit_wg, nb, want_inv, want_rhs, atomics)
                               : ch_dealt_item(deal, k, nb, want_inv, want_rhs);
    const int kind = it.kind, j = it.c, i = it.i;
    if (kind == CH_NONE) break;
    if (tid == 0) CH_TR_MARK(trwg, (4 << 28) | (kind << 20) | (j << 10) | i);
    if (kind == CH_RHS) {
      df_solve_rhs(A, ld, nb, ready, abort_flag, rhs, sol, sh, acq);
      continue;
    }
    if (kind == CH_INV) {
      // block (i, j) of L^-1, formed transposed: Y = X(i,j)^T = [ -(sum_{j<=p<i} X(p,j)^T L(i,p)^T)  or  I ] L(i,i)^-T
      d4 acci[2][2];
      zero_acc(acci);
      for (int p = j; p < i; ++p) {
        if (!wait_flag(sc.iready + (size_t)tile_no(p, j) * DF_FLAG_STRIDE)) return;
        if (!wait_flag(ready + (size_t)tile_no(i, p) * DF_FLAG_STRIDE)) return;
        df_mac(sc.xt + (size_t)tile_no(p, j) * 4096, A + (int64_t)i * DB * ld + (int64_t)p * DB, DB, ld, sh, acci);
      }
      d4 yi[4], xi[4];
      if (i > j) {
        acc_to_ts(acci);
        __syncthreads();
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int sq = 0; sq < 4; ++sq) yi[pb][sq] = -sh.Ts[16 * g + l15][16 * pb + 4 * sq + l4];
      } else {
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
          for (int sq = 0; sq < 4; ++sq) yi[pb][sq] = (16 * g + l15 == 16 * pb + 4 * sq + l4) ? 1.0 : 0.0;
      }
      if (!solve_panels(i, sc.pready + (size_t)4 * i * DF_FLAG_STRIDE, yi, xi, false)) return;
      // Y[m][n], m = 16 g + l15, n = 16 pb + 4 sq + l4  ->  scratch (as it is) and L^-1[64 i + n][64 j + m] (a diagonal block: lower triangle only)
      double* xtd = sc.xt + (size_t)tile_no(i, j) * 4096 + (size_t)(16 * g + l15) * DB + l4;
      double* lid = Linv + ((int64_t)i * DB + l4) * ld + (int64_t)j * DB + 16 * g + l15;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
          const double v = (i > j || 16 * g + l15 <= 16 * pb + 4 * sq + l4) ? xi[pb][sq] : 0.0;
          xtd[16 * pb + 4 * sq] = v;
          lid[(int64_t)(16 * pb + 4 * sq) * ld] = v;
        }
      raise(sc.iready + (size_t)tile_no(i, j) * DF_FLAG_STRIDE);
      continue;
    }
    const int jn = j + 1;  // the column whose prep sums the EARLY / FUSED items of column j deliver
    if (kind == CH_EARLY_S || kind == CH_EARLY_D) {
      if (j == 0) continue;  // nothing to the left of column 0
      d4 acce[2][2];
      zero_acc(acce);
      for (int p = 0; p < j; ++p) {
        if (!wait_flag(ready + (size_t)tile_no(i, p) * DF_FLAG_STRIDE)) return;
        const double* Lip = A + (int64_t)i * DB * ld + (int64_t)p * DB;
        if (kind == CH_EARLY_S) {  // (US)^T: rows = columns of S
          if (!wait_flag(ready + (size_t)tile_no(jn, p) * DF_FLAG_STRIDE)) return;
          df_mac(A + (int64_t)jn * DB * ld + (int64_t)p * DB, Lip, ld, ld, sh, acce);
        } else {
          df_mac(Lip, Lip, ld, ld, sh, acce);
        }
      }
      double* dste = (kind == CH_EARLY_S ? sc.upe + (size_t)jn * 4096 : sc.dpe + (size_t)i * 4096) + (size_t)wave * 1024 + lane;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int q = 0; q < 4; ++q) dste[((u * 2 + v) * 4 + q) * 64] = acce[u][v][q];  // the accumulator as it is: the reader is the FUSED item's same thread
      raise(kind == CH_EARLY_S ? sc.preSE + (size_t)jn * DF_FLAG_STRIDE : sc.preDE + (size_t)i * DF_FLAG_STRIDE);
      continue;
    }

    // ---- a tile (i, j): rank-64 updates, solve panel by panel; FUSED items carry on with the last term of their prep sum ----
    const bool fused_s = kind == CH_FUSED_S, fused_d = kind == CH_FUSED_D;
    if ((fused_s || fused_d) && !ask_local()) return;
    const bool lite = (fused_s || fused_d) && local;  // read the chain workgroup's light flags, publish to it without a write-back
    if ((fused_s || fused_d) && tid == 0) CH_TR_MARK(trwg, (5 << 28) | (lite ? 1 : 0));
    double* Aij = A + (int64_t)i * DB * ld + (int64_t)j * DB;
    if (!fused_s) {  // the mirrored tile is the strictly-upper part of the result: zero
      double* U = A + (int64_t)j * DB * ld + (int64_t)i * DB;
      for (int e = tid; e < DB * DB / 2; e += 256) {
        const int rr = e >> 5, cc = (e & 31) * 2;
        *reinterpret_cast<d2*>(U + (int64_t)rr * ld + cc) = d2{0.0, 0.0};
      }
    }
    d4 acc[2][2];
    zero_acc(acc);
    if (fused_s && tid == 0) CH_STAMP(j, 10);
    d4 yb[4], xb[4];
    if (!fused_s) {  // this wave's 16 rows of A(i,j) in the accumulator layout: asked for now, wanted after the updates
      const double* src = Aij + (int64_t)(16 * g + l15) * ld + l4;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) yb[pb][sq] = src[16 * pb + 4 * sq];
    }
    if (fused_d) {
      // Tile (c+2, c) is computed twice -- here in place, by FUSED_S(c) into registers only -- from the same ORIGINAL entries, which this
      // item is about to overwrite.  FUSED_S must therefore never read them in place (it did until tools/potrf_budget_check.py: with few
      // workgroups it could start after this item had finished and solved the FINAL tile a second time -- factors off by 3e-3 under CU
      // budgets 3 ... 7 and, now and then, 17 ... 47; the default deal starts both items at once): the entries go to scratch now, thread
      // for thread, behind a flag of their own.  (Nor does FUSED_S then cache a line of a tile before its publication.)
      double* cp = sc.lo2 + (size_t)j * 4096 + tid;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) cp[(pb * 4 + sq) * 256] = yb[pb][sq];
      // on the chain workgroup's XCD (where the default deal puts every fused item) the copy is handed over through the shared L2 at
      // once -- no write-back of that L2 in front of this item's work; otherwise both flags follow this item's own release below
      if (lite) raise_light(sc.lo2r_l + (size_t)j * DF_FLAG_STRIDE);
    }
    for (int p = 0; p < j; ++p) {
      if (!wait_flag(ready + (size_t)tile_no(i, p) * DF_FLAG_STRIDE)) return;
      if (!wait_flag(ready + (size_t)tile_no(j, p) * DF_FLAG_STRIDE)) return;
      df_mac(A + (int64_t)i * DB * ld + (int64_t)p * DB, A + (int64_t)j * DB * ld + (int64_t)p * DB, ld, ld, sh, acc);
    }
    acc_to_ts(acc);
    if (fused_s) {  // the original entries of the tile, from FUSED_D(c) (df_wait's barrier also orders the stores to sh.Ts above)
      if (!wait_flag((lite ? sc.lo2r_l : sc.lo2r) + (size_t)j * DF_FLAG_STRIDE)) return;
      const double* cp = sc.lo2 + (size_t)j * 4096 + tid;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) yb[pb][sq] = cp[(pb * 4 + sq) * 256];
    } else {
      __syncthreads();
    }
    if (fused_s && tid == 0) CH_STAMP(j, 11);
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) yb[pb][sq] -= sh.Ts[16 * g + l15][16 * pb + 4 * sq + l4];
    if (!solve_panels(j, (lite ? sc.pready_l : sc.pready) + (size_t)4 * j * DF_FLAG_STRIDE, yb, xb, fused_s)) return;
    if (!fused_s) {
      double* dst = Aij + (int64_t)(16 * g + l15) * ld + l4;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) dst[16 * pb + 4 * sq] = xb[pb][sq];
    }
    if (kind == CH_TILE) {
      raise(ready + (size_t)tile_no(i, j) * DF_FLAG_STRIDE);
      continue;
    }
    // ---- FUSED items: the last term of the prep sum with the X just solved, which stays on chip (sh.Ts, row-major) ----
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
      for (int sq = 0; sq < 4; ++sq) sh.Ts[16 * g + l15][16 * pb + 4 * sq + l4] = xb[pb][sq];
    if (fused_d) {  // the tile itself goes public first: the next column's items wait for it (the same release covers the copy for FUSED_S)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(ready + (size_t)tile_no(i, j) * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CH_TR_RAISE(sc, trwg, ready + (size_t)tile_no(i, j) * DF_FLAG_STRIDE, 0);
        __hip_atomic_store(sc.lo2r + (size_t)j * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CH_TR_RAISE(sc, trwg, sc.lo2r + (size_t)j * DF_FLAG_STRIDE, 0);
        if (!lite) {
          __hip_atomic_store(sc.lo2r_l + (size_t)j * DF_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          CH_TR_RAISE(sc, trwg, sc.lo2r_l + (size_t)j * DF_FLAG_STRIDE, 0);
        }
      }
    }
    else __syncthreads();
    if (fused_s && tid == 0) CH_STAMP(j, 13);
    // the terms from the columns p < j (EARLY items of this column, stored thread for thread): asked for now, added behind the product
    d4 acc2[2][2], early[2][2];
    zero_acc(acc2);
    zero_acc(early);
    if (j > 0) {
      if (!wait_flag((fused_s ? sc.preSE + (size_t)jn * DF_FLAG_STRIDE : sc.preDE + (size_t)i * DF_FLAG_STRIDE))) return;
      const double* srce = (fused_s ? sc.upe + (size_t)jn * 4096 : sc.dpe + (size_t)i * 4096) + (size_t)wave * 1024 + lane;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int q = 0; q < 4; ++q) early[u][v][q] = srce[((u * 2 + v) * 4 + q) * 64];
    }
    if (fused_s) {
      // US(jn)^T += L(jn,j) X^T: L(jn,j) is the chain workgroup's X of step j (its light flag when we share the L2)
      if (!wait_flag(lite ? sc.xready_l + (size_t)j * DF_FLAG_STRIDE : ready + (size_t)tile_no(jn, j) * DF_FLAG_STRIDE)) return;
      if (tid == 0) CH_STAMP(j, 14);
      df_mac<false, true>(A + (int64_t)jn * DB * ld + (int64_t)j * DB, nullptr, ld, 0, sh, acc2);
      if (tid == 0) CH_STAMP(j, 15);
      // accumulator (rows: columns of S, columns: rows of S) == the S-waves' register order: straight to `upre`
      double* up = sc.upre + (size_t)jn * 4096;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int q = 0; q < 4; ++q) up[(((wj * 2 + v) * 4 + (wi * 2 + u)) * 4 + q) * 64 + lane] = acc2[u][v][q] + early[u][v][q];
      if (lite) raise_light(sc.preS + (size_t)jn * DF_FLAG_STRIDE);
      else raise(sc.preS + (size_t)jn * DF_FLAG_STRIDE);
    } else {
      df_mac<true, true>(nullptr, nullptr, 0, 0, sh, acc2);
      // lower blocks of UD(i) in the order of the S-waves' accumulators
      double* dpp = sc.dpre + (size_t)i * 4096;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int R = wi * 2 + u, Cc = wj * 2 + v;
          if (R >= Cc) {
            const int slotb = ch_owner_slot(R, Cc);  // (owner wave) * 3 + (its block index)
#pragma unroll
            for (int q = 0; q < 4; ++q) dpp[slotb * 256 + q * 64 + lane] = acc2[u][v][q] + early[u][v][q];
          }
        }
      if (lite) raise_light(sc.preD + (size_t)i * DF_FLAG_STRIDE);
      else raise(sc.preD + (size_t)i * DF_FLAG_STRIDE);
    }
  }
}

}  // namespace sgp
