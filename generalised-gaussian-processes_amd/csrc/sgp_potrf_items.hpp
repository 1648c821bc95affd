// Work items of the chain-workgroup Cholesky (sgp_potrf_chain.hpp) and how they are dealt to the workgroups: pure index arithmetic, kept
// apart from the kernel so that a host build can walk it (tests/native/chain_items_check.cpp simulates the dataflow for every block count
// and many workgroup counts: every item is dealt exactly once and no dealing can deadlock).
#pragma once
#if defined(__HIPCC__)
#define SGP_HD __host__ __device__
#else
#define SGP_HD
#endif

namespace sgp {

enum ChKind { CH_EARLY_S, CH_EARLY_D, CH_FUSED_S, CH_FUSED_D, CH_TILE, CH_INV, CH_RHS, CH_NONE };
struct ChItem { int kind, c, i; };
SGP_HD inline int ch_crit_items(int nb) { return nb >= 3 ? 2 * (nb - 2) : 0; }
SGP_HD inline int ch_rest_tile_items(int nb) { return nb >= 3 ? (nb - 2) * (nb + 1) / 2 : 0; }  // per column c: 2 + (nb - 3 - c)
SGP_HD inline int ch_tile_items(int nb) { return ch_crit_items(nb) + ch_rest_tile_items(nb); }
SGP_HD inline int ch_inv_items(int nb) { return nb * (nb + 1) / 2; }
// item k of a list: per column c the tile items ([EARLY_S, EARLY_D, (single list only: FUSED_S, FUSED_D,) TILE(c+3..)], columns c <= nb - 3),
// then -- want_inv -- row c of L^-1: INV(c, 0 .. c); behind the last column RHS if want_rhs.  `fused`: the single list (few workgroups),
// with FUSED_D, FUSED_S in that order
SGP_HD inline ChItem ch_list_item(int k, int nb, bool want_inv, bool want_rhs, bool fused) {
  for (int c = 0; c < nb; ++c) {
    const int nt = c <= nb - 3 ? nb - 1 - c + (fused ? 2 : 0) : 0;
    if (k < nt) {
      if (k == 0) return ChItem{CH_EARLY_S, c, c + 2};
      if (k == 1) return ChItem{CH_EARLY_D, c, c + 2};
      if (fused) {
        if (k == 2) return ChItem{CH_FUSED_D, c, c + 2};  // (ahead of FUSED_S, which waits for the copy of the tile's original entries
        if (k == 3) return ChItem{CH_FUSED_S, c, c + 2};  //  this item makes when it STARTS: with one workgroup the order must be this one)
        return ChItem{CH_TILE, c, c - 1 + k};
      }
      return ChItem{CH_TILE, c, c + 1 + k};
    }
    k -= nt;
    if (want_inv) {
      if (k <= c) return ChItem{CH_INV, k, c};
      k -= c + 1;
    }
  }
  if (want_rhs && k == 0) return ChItem{CH_RHS, 0, 0};
  return ChItem{CH_NONE, 0, 0};
}
// The deal: workgroup `ow` of the `nout` non-chain workgroups takes items first, first + stride, ... < count of its list.  With eight
// workgroups or more the CRITICAL list [FUSED_D(0), FUSED_S(0), FUSED_D(1), ...] goes round-robin to the workgroups expected on the chain
// workgroup's XCD (ow = 7, 15, ...: blockIdx = 0 mod 8), the other list to the rest; with fewer there is one list for all.
struct ChDeal {
  bool split, crit_wg;
  int first, stride, count;
};
SGP_HD inline ChDeal ch_deal(int ow, int nout, int nb, bool want_inv, bool want_rhs) {
  ChDeal d;
  const int nl = nout / 8;
  const int extra = (want_inv ? ch_inv_items(nb) : 0) + (want_rhs ? 1 : 0);
  d.split = nl > 0 && ch_crit_items(nb) > 0;
  d.crit_wg = d.split && ((ow + 1) & 7) == 0;
  if (!d.split) {
    d.first = ow; d.stride = nout; d.count = ch_tile_items(nb) + extra;
  } else if (d.crit_wg) {
    d.first = (ow + 1) / 8 - 1; d.stride = nl; d.count = ch_crit_items(nb);
  } else {
    d.first = ow - (ow + 1) / 8; d.stride = nout - nl; d.count = ch_rest_tile_items(nb) + extra;
  }
  return d;
}
// item k of the list of a workgroup dealt `d`
SGP_HD inline ChItem ch_dealt_item(const ChDeal& d, int k, int nb, bool want_inv, bool want_rhs) {
  if (!d.split) return ch_list_item(k, nb, want_inv, want_rhs, true);
  if (d.crit_wg) return ChItem{(k & 1) ? CH_FUSED_S : CH_FUSED_D, k >> 1, (k >> 1) + 2};
  return ch_list_item(k, nb, want_inv, want_rhs, false);
}

}  // namespace sgp
