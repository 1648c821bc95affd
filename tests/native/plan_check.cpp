// Host-only check of the streaming geometry (sgp_stream.hpp): for many (N, M, d) the tapered / equal split ranges of both
// passes must tile [0, nchunks) and [0, nblocks) exactly once, in order, with no empty holes.
#include "sgp_stream.hpp"
#include <cstdio>
namespace sgp { size_t stream_kfu_budget() { return KFU_BUDGET_DEFAULT; } }
using namespace sgp;
static int check(const SplitMap& m, int nsplit, int64_t n, const char* what, int64_t N, int M) {
  int64_t expect = 0;
  for (int s = 0; s < nsplit; ++s) {
    int64_t c0, c1;
    split_range(m, s, n, c0, c1);
    if (c0 > c1 || c0 < 0 || c1 > n) { printf("FAIL %s N=%lld M=%d split %d: [%lld,%lld) of %lld\n", what, (long long)N, M, s, (long long)c0, (long long)c1, (long long)n); return 1; }
    if (c0 != c1) {
      if (c0 != expect) { printf("FAIL %s N=%lld M=%d split %d starts at %lld, expected %lld\n", what, (long long)N, M, s, (long long)c0, (long long)expect); return 1; }
      expect = c1;
    }
  }
  if (expect != n) { printf("FAIL %s N=%lld M=%d covers %lld of %lld\n", what, (long long)N, M, (long long)expect, (long long)n); return 1; }
  return 0;
}
int main() {
  int bad = 0, cases = 0;
  const int64_t Ns[] = {0, 1, 255, 256, 257, 634, 4096, 13279, 65536, 100000, 125000, 250000, 500000, 999999, 1000000, 1000001, 3000000};
  const int Ms[] = {1, 25, 50, 100, 128, 129, 512, 1000, 1024, 2048, 4096};
  for (int64_t N : Ns)
    for (int M : Ms)
      for (int d : {1, 8, 18}) {
        const StreamPlan p = make_stream_plan(N, M, d);
        ++cases;
        if (p.nsplit % 8 != 0 || p.nsplit_b % 8 != 0 || p.nsplit <= 0 || p.nsplit_b <= 0) { printf("FAIL nsplit N=%lld M=%d\n", (long long)N, M); ++bad; continue; }
        if (p.Npad == 0) continue;
        const int64_t nchunks = p.sc_rows / NB, nblocks = p.sc_rows / TILE;
        const int cps = (int)((nchunks + p.nsplit - 1) / p.nsplit), bps = (int)((nblocks + p.nsplit_b - 1) / p.nsplit_b);
        bad += check(SplitMap{{p.taper[0], p.taper[1], p.taper[2], p.taper[3]}, cps < 1 ? 1 : cps}, p.nsplit, nchunks, "pass1", N, M);
        bad += check(SplitMap{{p.taper_b[0], p.taper_b[1], p.taper_b[2], p.taper_b[3]}, bps < 1 ? 1 : bps}, p.nsplit_b, nblocks, "pass2", N, M);
        // the integer contraction's splits: a multiple of 8, at most I8_SPLIT_ROWS rows each (the int32 bound of the digit-pair sums),
        // tiling the 32-row steps of a super-chunk exactly once -- also for the short last super-chunk contracted with the same count
        const int ns8 = i8_nsplit(p.sc_rows, p.Mp);
        if (ns8 < 8 || ns8 % 8 != 0) { printf("FAIL i8 nsplit %d N=%lld M=%d\n", ns8, (long long)N, M); ++bad; }
        for (int64_t rows : {p.sc_rows, p.Npad % p.sc_rows}) {
          if (rows == 0) continue;
          const int64_t nsteps = rows / 32;
          int64_t expect = 0;
          for (int sidx = 0; sidx < ns8; ++sidx) {
            int64_t c0, c1;
            i8_split_steps(nsteps, ns8, sidx, c0, c1);
            if (c0 > c1 || (c1 - c0) * 32 > I8_SPLIT_ROWS || (c0 != c1 && c0 != expect)) { printf("FAIL i8 split %d N=%lld M=%d [%lld,%lld)\n", sidx, (long long)N, M, (long long)c0, (long long)c1); ++bad; break; }
            if (c0 != c1) expect = c1;
          }
          if (expect != nsteps) { printf("FAIL i8 splits cover %lld of %lld steps N=%lld M=%d\n", (long long)expect, (long long)nsteps, (long long)N, M); ++bad; }
        }
      }
  printf("stream plan check: %d cases, %d failures\n", cases, bad);
  return bad != 0;
}
