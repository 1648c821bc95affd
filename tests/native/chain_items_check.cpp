// Host walk of the chain-workgroup Cholesky's work items (csrc/sgp_potrf_items.hpp): for every block count nb, every combination of
// "inverse wanted" / "right-hand side wanted" and a range of workgroup counts it checks that
//   1. the deal hands out every item of the launch exactly once (and nothing else), and
//   2. the dataflow cannot deadlock: with every workgroup taking its items strictly in its own order, and every item / chain step
//      completing only after what the kernel makes it wait for (the flags of sgp_potrf_chain.hpp, restated below), everything completes.
// Plain C++ (g++ -fsanitize=address,undefined): the header is index arithmetic only.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include <vector>
#include "sgp_potrf_items.hpp"

using namespace sgp;
using Key = std::tuple<int, int, int>;  // (kind, c, i)

struct Sim {
  int nb;
  bool want_inv, want_rhs;
  std::map<Key, bool> done;   // items
  std::map<Key, bool> started;  // ... and: its workgroup has reached it (FUSED_D copies the tile's original entries for FUSED_S when it STARTS)
  std::vector<bool> step;     // chain workgroup: step j complete (L(j,j), the tile below it, the next diagonal tile's updates)
  bool has(int kind, int c, int i) const { return done.count(Key(kind, c, i)) != 0; }
  bool is_done(int kind, int c, int i) const {
    auto it = done.find(Key(kind, c, i));
    return it != done.end() && it->second;
  }
  // tile (r, c) of L public?
  bool tile(int r, int c) const {
    if (r == c || r == c + 1) return step[c];           // the chain workgroup's own two tiles of column c
    if (r == c + 2) return is_done(CH_FUSED_D, c, r);   // published by the fused item that computes it in place
    return is_done(CH_TILE, c, r);
  }
  bool item_ready(int kind, int c, int i) const {
    switch (kind) {
      case CH_EARLY_S:
        for (int p = 0; p < c; ++p)
          if (!tile(i, p) || !tile(c + 1, p)) return false;
        return true;
      case CH_EARLY_D:
        for (int p = 0; p < c; ++p)
          if (!tile(i, p)) return false;
        return true;
      case CH_FUSED_S:
      case CH_FUSED_D:
      case CH_TILE:
        for (int p = 0; p < c; ++p)
          if (!tile(i, p) || !tile(c, p)) return false;
        if (!step[c]) return false;  // the panels of L(c,c) (and, fused items, the X of the chain workgroup's step c)
        // the original entries of tile (c+2, c) come from FUSED_D(c): at its start when both items share the chain workgroup's L2, with its
        // publication of the tile otherwise -- the stronger requirement is the one checked
        if (kind == CH_FUSED_S && !is_done(CH_FUSED_D, c, i)) return false;
        if (kind == CH_FUSED_S && c > 0 && !is_done(CH_EARLY_S, c, i)) return false;
        if (kind == CH_FUSED_D && c > 0 && !is_done(CH_EARLY_D, c, i)) return false;
        return true;
      case CH_INV:  // block (i, c) of L^-1
        for (int p = c; p < i; ++p)
          if (!is_done(CH_INV, c, p) || !tile(i, p)) return false;
        return step[i];
      case CH_RHS:
        for (int c2 = 0; c2 < nb; ++c2)
          for (int r = c2; r < nb; ++r)
            if (!tile(r, c2)) return false;
        return true;
      default:
        return false;
    }
  }
  bool step_ready(int j) const {
    if (j > 0 && !step[j - 1]) return false;
    // the S-waves of step j (j + 1 < nb) wait for US(j) and UD(j+1): the fused items of column j - 1
    if (j >= 1 && j + 1 < nb) return is_done(CH_FUSED_S, j - 1, j + 1) && is_done(CH_FUSED_D, j - 1, j + 1);
    return true;
  }
};

static int check(int nb, bool want_inv, bool want_rhs, int nout) {
  int failures = 0;
  Sim s;
  s.nb = nb; s.want_inv = want_inv; s.want_rhs = want_rhs;
  s.step.assign(nb, false);
  // what the launch must contain
  std::map<Key, int> expect;
  for (int c = 0; c + 3 <= nb; ++c) {
    expect[Key(CH_EARLY_S, c, c + 2)] = 0;
    expect[Key(CH_EARLY_D, c, c + 2)] = 0;
    expect[Key(CH_FUSED_S, c, c + 2)] = 0;
    expect[Key(CH_FUSED_D, c, c + 2)] = 0;
    for (int i = c + 3; i < nb; ++i) expect[Key(CH_TILE, c, i)] = 0;
  }
  if (want_inv)
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j <= i; ++j) expect[Key(CH_INV, j, i)] = 0;
  if (want_rhs) expect[Key(CH_RHS, 0, 0)] = 0;
  const int total = ch_tile_items(nb) + (want_inv ? ch_inv_items(nb) : 0) + (want_rhs ? 1 : 0);
  if ((int)expect.size() != total) { std::printf("nb %d: %zu items expected by the walk, %d by the counts\n", nb, expect.size(), total); ++failures; }
  // the deal
  std::vector<std::vector<Key>> mine(nout);
  for (int ow = 0; ow < nout; ++ow) {
    const ChDeal d = ch_deal(ow, nout, nb, want_inv, want_rhs);
    for (int k = d.first; k < d.count; k += d.stride) {
      const ChItem it = ch_dealt_item(d, k, nb, want_inv, want_rhs);
      if (it.kind == CH_NONE) break;
      const Key key(it.kind, it.c, it.i);
      auto e = expect.find(key);
      if (e == expect.end()) { std::printf("nb %d nout %d: unexpected item (%d, %d, %d)\n", nb, nout, it.kind, it.c, it.i); ++failures; continue; }
      ++e->second;
      mine[ow].push_back(key);
      s.done[key] = false;
      s.started[key] = false;
    }
  }
  for (auto& e : expect)
    if (e.second != 1) {
      std::printf("nb %d inv %d rhs %d nout %d: item (%d, %d, %d) dealt %d times\n", nb, want_inv, want_rhs, nout, std::get<0>(e.first),
                  std::get<1>(e.first), std::get<2>(e.first), e.second);
      ++failures;
    }
  if (failures) return failures;
  // the dataflow: sweep until nothing moves
  std::vector<size_t> at(nout, 0);
  bool moved = true;
  while (moved) {
    moved = false;
    for (int j = 0; j < nb; ++j)
      if (!s.step[j] && s.step_ready(j)) { s.step[j] = true; moved = true; }
    for (int ow = 0; ow < nout; ++ow)
      while (at[ow] < mine[ow].size()) {
        const Key& k = mine[ow][at[ow]];
        if (!s.started[k]) { s.started[k] = true; moved = true; }
        if (!s.item_ready(std::get<0>(k), std::get<1>(k), std::get<2>(k))) break;
        s.done[k] = true;
        ++at[ow];
        moved = true;
      }
  }
  for (int j = 0; j < nb; ++j)
    if (!s.step[j]) { std::printf("nb %d inv %d rhs %d nout %d: chain step %d never completes\n", nb, want_inv, want_rhs, nout, j); ++failures; break; }
  for (int ow = 0; ow < nout && !failures; ++ow)
    if (at[ow] < mine[ow].size()) {
      const Key& k = mine[ow][at[ow]];
      std::printf("nb %d inv %d rhs %d nout %d: workgroup %d stuck at item (%d, %d, %d)\n", nb, want_inv, want_rhs, nout, ow, std::get<0>(k),
                  std::get<1>(k), std::get<2>(k));
      ++failures;
    }
  return failures;
}

int main() {
  int failures = 0, cases = 0;
  const int nouts[] = {1, 2, 3, 7, 8, 9, 15, 16, 17, 39, 64, 255};
  for (int nb = 2; nb <= 64; nb = nb < 20 ? nb + 1 : nb + 11)
    for (int inv = 0; inv < 2; ++inv)
      for (int rhs = 0; rhs < 2; ++rhs)
        for (int nout : nouts) {
          const int items = ch_tile_items(nb) + (inv ? ch_inv_items(nb) : 0) + rhs;
          if (items == 0) continue;            // (nb = 2 without inverse / rhs: the launch has no other workgroups' items at all)
          if (nout > items && nout != 1) continue;  // potrf_lower never launches more workgroups than items
          failures += check(nb, inv != 0, rhs != 0, nout);
          ++cases;
        }
  std::printf("%d cases, %d failures\n", cases, failures);
  return failures ? 1 : 0;
}
