// Shared geometry of the two streaming passes (sgp_suffstats_fwd.hip / sgp_suffstats_bwd.hip).
#pragma once
#include <cstdlib>
#include "sgp_common.hpp"

namespace sgp {

constexpr int TILE = 128;            // tile edge (inducing columns) of both passes
constexpr int NB = 16;               // data rows per SYRK chunk
constexpr int KROW = 2 * TILE + 16;  // LDS row stride of a SYRK chunk in doubles (272: +128 B bank shift per row)
constexpr int ASM_ROWS = 256;        // data rows per assembly workgroup; K'_fu is padded to a multiple of this
constexpr int TARGET_WGS = 2048;     // ~4 waves of the 512 resident workgroups (2 per CU): small problems
constexpr int RESIDENT_WGS = 512;    // 256 CUs x 2 workgroups of the SYRK / K-bar kernels
constexpr size_t KFU_BUDGET_DEFAULT = size_t(16) << 30;  // K'_fu bytes kept in flight when the library owns the buffer
size_t stream_kfu_budget();  // current budget (sgp_set_kfu_budget_bytes); defined in sgp_suffstats_fwd.hip

static inline int dp_for(int d) {
  const int opts[] = {2, 4, 8, 16, 24, 32};
  for (int o : opts)
    if (d <= o) return o;
  return -1;
}

struct StreamPlan {
  int Mp, ntr, ntiles, DP;
  int64_t Npad;     // N rounded up to ASM_ROWS
  int asm_sub;      // fp64 assembly: workgroups per ASM_ROWS row block (1, or 4 / 8 on small shards: 64 / 32 rows each, see make_stream_plan)
  int64_t sc_rows;  // rows of K'_fu materialised at a time (multiple of ASM_ROWS)
  int nsplit;       // pass 1: row-range splits per tile (multiple of 8: one XCD per residue)
  int taper[4];     // pass 1: groups of 8 splits at relative sizes 8, 4, 2, 1 (all 0: equal splits), see split_range()
  int nmb;          // pass 2: 128-column blocks of Phibar
  int nsplit_b;     // pass 2: row-range splits per column block
  int taper_b[4];   // pass 2: tapered split sizes (as taper[])
};

// rows of the partial sums of K'^T y the fp64 assembly leaves (the integer path's assembly and the whitened layout's tpart_kernel keep
// one per ASM_ROWS; their consumers say so)
static inline int64_t bpart_rows(const StreamPlan& p) { return p.Npad / ASM_ROWS > 0 ? p.Npad / ASM_ROWS * p.asm_sub : 1; }

// chunk range [c0, c1) of split `split` (8 splits per group; group sizes taper 8 : 4 : 2 : 1 when taper[] is set)
struct SplitMap {
  int g[4];
  int cps;
};
__host__ __device__ inline void split_range(const SplitMap& m, int split, int64_t nchunks, int64_t& c0, int64_t& c1) {
  const int64_t groups = (int64_t)m.g[0] + m.g[1] + m.g[2] + m.g[3];
  if (groups == 0) {
    c0 = (int64_t)split * m.cps;
    c1 = c0 + m.cps;
    if (c1 > nchunks) c1 = nchunks;
    if (c0 > nchunks) c0 = nchunks;
    return;
  }
  const int64_t total = 8 * (8 * (int64_t)m.g[0] + 4 * (int64_t)m.g[1] + 2 * (int64_t)m.g[2] + (int64_t)m.g[3]);
  int64_t gg = split >> 3, before = 0;
  int w = 8;
  for (int l = 0; l < 4; ++l) {
    if (gg >= m.g[l]) {
      before += (int64_t)m.g[l] * 8 * w;
      gg -= m.g[l];
      w >>= 1;
    } else {
      before += gg * 8 * w;
      break;
    }
  }
  if (w == 0) w = 1;
  const int64_t cum0 = before + (int64_t)(split & 7) * w;
  c0 = cum0 * nchunks / total;
  c1 = (cum0 + w) * nchunks / total;
}

static inline StreamPlan make_stream_plan(int64_t N, int M, int d) {
  StreamPlan p;
  p.Mp = padded_m(M);
  p.ntr = p.Mp / TILE;
  p.ntiles = p.ntr * (p.ntr + 1) / 2;
  p.DP = dp_for(d);
  p.Npad = N > 0 ? round_up64(N, ASM_ROWS) : 0;
  // Small shards (C3: 52 row blocks x 2 column groups = 104 workgroups of four waves on 256 CUs, every thread walking 256 rows with one
  // wave per SIMD: 125 us for 6.8 M kernel values, profiles/r05_v2_c3_timeline.txt): fewer rows per workgroup, more workgroups -- 64 rows
  // 47 us, 32 rows 34 us, 16 rows 34 us at C3 (same-box A/B, C3 value 431 -> 414 us from 64 to 32 rows).  The partials of K'^T y are then
  // per 64 / 32 rows (bpart_rows(): the fixed-order reduction behind them takes any count).
  {
    const int64_t base = (p.Npad / ASM_ROWS) * ((p.Mp + 255) / 256);
    p.asm_sub = base < 128 ? 8 : (base < 256 ? 4 : 1);
  }
  static const int asm_sub_override = getenv("SGP_ASM_SUB") ? atoi(getenv("SGP_ASM_SUB")) : 0;  // tuning knob: 1, 4, 8 or 16
  if (asm_sub_override == 1 || asm_sub_override == 4 || asm_sub_override == 8 || asm_sub_override == 16) p.asm_sub = asm_sub_override;
  int64_t cap = (int64_t)(stream_kfu_budget() / ((size_t)p.Mp * 8)) / ASM_ROWS * ASM_ROWS;
  if (cap < ASM_ROWS) cap = ASM_ROWS;
  p.sc_rows = p.Npad < cap ? p.Npad : cap;
  const int64_t nchunks = p.sc_rows / NB;
  static const int target_wgs = getenv("SGP_TARGET_WGS") ? atoi(getenv("SGP_TARGET_WGS")) : TARGET_WGS;  // tuning knob
  int64_t k = (target_wgs + 8 * p.ntiles - 1) / (8 * p.ntiles);
  int64_t ns = 8 * k;
  // Wave quantisation: workgroups take 3.3-4.7 ms each at C5 and 512 run at a time, so a launch whose last round is
  // half empty idles a tenth of the chip (per-workgroup stamps: 14.9 ms of work in a 16.5 ms launch at 3.94 rounds).
  // With enough rows, take the split count (multiple of 8, 6-11 rounds of workgroups that still get >= 64 chunks
  // each) whose last round is fullest: 128 splits = exactly 9 rounds at M = 1024 (17.6 vs 18.5 ms on one box).
  if (nchunks >= 64 * 8 && !getenv("SGP_TARGET_WGS")) {
    double best_waste = 2.0;
    for (int64_t kk = (3072 + 8 * p.ntiles - 1) / (8 * p.ntiles); 8 * kk * p.ntiles <= 5632; ++kk) {
      const int64_t cand = 8 * kk;
      if (kk < 1 || nchunks / cand < 64) break;
      const double r = (double)p.ntiles * (double)cand / (double)RESIDENT_WGS;
      const double rounds = (double)(int64_t)(r + 0.999999);
      const double waste = (rounds - r) / rounds;
      if (waste < best_waste - 1e-9) {
        best_waste = waste;
        ns = cand;
      }
    }
  }
  // Mid-size shards (C3: 13 279 rows): ~4 waves of workgroups would leave each with a handful of 16-row chunks and a
  // 128 KB slab to write and reduce (2080 workgroups, 272 MB of slabs: contraction 107 us, reduction 70 us).  One round of
  // resident workgroups instead -- 56 splits at M = 512: value 735 -> 628 us, value+gradient 1311 -> 1185 us (same box).
  // (round 5: 0.8 of a round, rounded DOWN -- the K_uu factorization's workgroups hold a quarter of the CUs' LDS while this kernel runs, and a
  // launch that does not fit beside them pays a second round: 56 / 48 / 40 / 32 / 24 splits at C3 = 104 / 106 / 89 / 105 / 136 us + a reduction
  // of 15 / 13 / 11 / 9 / 8 us, rocprofv3, same box)
  if (nchunks / ns < 16 && !getenv("SGP_TARGET_WGS")) {
    int64_t one_round = 8 * ((RESIDENT_WGS * 4 / 5) / (8 * p.ntiles));
    if (one_round < 8) one_round = 8;
    if (one_round < ns) ns = one_round;
  }
  static const int ns_override = getenv("SGP_SYRK_NSPLIT") ? atoi(getenv("SGP_SYRK_NSPLIT")) : 0;  // tuning knob
  if (ns_override > 0) ns = 8 * ((ns_override + 7) / 8);
  const int64_t lim = round_up64(nchunks > 0 ? nchunks : 1, 8);
  if (ns > lim) ns = lim;
  p.nsplit = (int)ns;
  p.taper[0] = p.taper[1] = p.taper[2] = p.taper[3] = 0;
  static const int taper_on = getenv("SGP_SYRK_TAPER") ? atoi(getenv("SGP_SYRK_TAPER")) : 1;  // A/B knob (20.55 vs 20.76 ms)
  if (taper_on && !getenv("SGP_TARGET_WGS") && ns % 32 == 0 && nchunks / ns >= 64) {
    // big splits first, then halves, quarters and eighths: the last round of workgroups is short, so the ragged end of
    // the launch (workgroup durations differ by +-15 %) shrinks with it
    const int B = (int)(ns / 8);
    p.taper[0] = 3 * B / 4; p.taper[1] = B / 4; p.taper[2] = B / 4; p.taper[3] = B / 2;
    p.nsplit = 8 * (p.taper[0] + p.taper[1] + p.taper[2] + p.taper[3]);
  }
  p.nmb = p.Mp / TILE;
  const int64_t nblocks = p.sc_rows / TILE;
  int64_t nsb = 8 * ((TARGET_WGS + 8 * p.nmb - 1) / (8 * p.nmb));  // multiple of 8: one XCD per residue
  if (nblocks / nsb < 2) {  // mid-size shards: one round of resident workgroups (two per CU), 1309 -> 1213 us at C3
    const int64_t one_round = 8 * ((RESIDENT_WGS + 8 * p.nmb - 1) / (8 * p.nmb));
    if (one_round < nsb) nsb = one_round;
  }
  static const int nsb_override = getenv("SGP_KBAR_NSPLIT") ? atoi(getenv("SGP_KBAR_NSPLIT")) : 0;  // tuning knob
  if (nsb_override > 0) nsb = 8 * ((nsb_override + 7) / 8);
  const int64_t limb = round_up64(nblocks > 0 ? nblocks : 1, 8);
  if (nsb > limb) nsb = limb;
  p.nsplit_b = (int)nsb;
  p.taper_b[0] = p.taper_b[1] = p.taper_b[2] = p.taper_b[3] = 0;
  static const int taper_b_on = getenv("SGP_KBAR_TAPER") ? atoi(getenv("SGP_KBAR_TAPER")) : 1;  // A/B knob (55.4 vs 56.2 ms)
  if (taper_b_on && nsb % 32 == 0 && nblocks / nsb >= 8) {
    const int B = (int)(nsb / 8);
    p.taper_b[0] = 3 * B / 4; p.taper_b[1] = B / 4; p.taper_b[2] = B / 4; p.taper_b[3] = B / 2;
    p.nsplit_b = 8 * (p.taper_b[0] + p.taper_b[1] + p.taper_b[2] + p.taper_b[3]);
  }
  return p;
}

// Optional per-kernel timing (sgp_timing_enable): HIP events recorded on the launch stream around the
// dominant kernels.  Slots: 0 = kernel assembly, 1 = SYRK contraction, 2 = Kbar contraction (pass 2).
enum { TIMING_ASSEMBLE = 0, TIMING_SYRK = 1, TIMING_KBAR = 2, TIMING_SLOTS = 3 };
void timing_begin(int slot, hipStream_t st);
void timing_end(int slot, hipStream_t st);

// implemented in sgp_suffstats_fwd.hip
// pad (optional): a job pass 2 used to launch separately rides along -- out (Mp x Mp) <- (P + P^T) / 2 zero-padded, vout (Mp) <- vec zero-padded
struct PadSymJob {
  const double* P = nullptr;
  double* out = nullptr;
  const double* vec = nullptr;
  double* vout = nullptr;
};
void stream_prologue(const StreamPlan& p, const KernArgs& ka, const double* X, int64_t ldx, const double* y,
                     const double* Z, int64_t ldz, int64_t N, int M, double* Xs, double* ys, double* Zs, double* yypart,
                     hipStream_t st, const PadSymJob& pad = PadSymJob());
void stream_assemble(const StreamPlan& p, int kid, const double* Xs, const double* ys, const double* Zs, int64_t row0,
                     int64_t rows, int64_t N, int M, double* Kfu, double* bpart, hipStream_t st);

// implemented in sgp_suffstats_i8.hip: the pass-1 contraction on the integer matrix cores (error-free digit planes of K'_fu)
constexpr int I8_SPLIT_ROWS = 16384;  // rows per split at most: 7 digit pairs x 2^14 x 16384 rows < 2^31
// splits of at most I8_SPLIT_ROWS rows (the int32 bound), a multiple of 8 (one XCD per residue), and -- when the rows allow
// splits of >= 2048 rows -- the count below twice the minimum whose last round of 256 resident workgroups is fullest
static inline int i8_nsplit(int64_t rows, int Mp) {
  const int nrt = Mp / 128, ntiles = nrt * (nrt + 1);  // 128 x 64 tiles of the lower triangle
  int64_t lo = (rows + I8_SPLIT_ROWS - 1) / I8_SPLIT_ROWS;
  lo = (lo + 7) / 8 * 8;
  if (lo < 8) lo = 8;
  int64_t best = lo;
  double best_waste = 2.0;
  for (int64_t cand = lo; cand <= 2 * lo + 8; cand += 8) {
    if (cand > lo && rows / cand < 2048) break;
    const double r = (double)ntiles * (double)cand / 256.0;
    const double rounds = (double)(int64_t)(r + 0.999999);
    const double waste = (rounds - r) / rounds;
    if (waste < best_waste - 1e-9) {
      best_waste = waste;
      best = cand;
    }
  }
  return (int)best;
}
// 32-row steps [c0, c1) of split `split`: equal shares, the last ones short or empty
__host__ __device__ inline void i8_split_steps(int64_t nsteps, int nsplit, int split, int64_t& c0, int64_t& c1) {
  const int64_t per = (nsteps + nsplit - 1) / nsplit;
  c0 = (int64_t)split * per;
  c1 = c0 + per;
  if (c0 > nsteps) c0 = nsteps;
  if (c1 > nsteps) c1 = nsteps;
}
void i8_assemble(const StreamPlan& p, int kid, const double* Xs, const double* ys, const double* Zs, int64_t row0, int64_t rows,
                 int64_t N, int M, uint8_t* Q, double* Kfu /* optional: the fp64 block too */, double* bpart, hipStream_t st,
                 uint16_t* Kh = nullptr /* optional, with Kfu: its fp16 image too */);
// slab_lo (optional): the extended contraction -- 34 digit pairs, every tile as an unevaluated sum slab + slab_lo (sgp_suffstats_fwd_extended)
// level (with slab_lo): 1 = the pairs p + r >= 5 (34), 2 = p + r >= 4 (39)
int i8_contract(const uint8_t* Q, int Mp, int64_t rows, int nsplit, int accumulate, double* slab, hipStream_t st, double* slab_lo = nullptr,
                int level = 1);

}  // namespace sgp
