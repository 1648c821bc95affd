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
// (`legacy`: FUSED_S ahead of FUSED_D as until 59da5e9 -- only the host checker's negative test asks for it)
SGP_HD inline ChItem ch_list_item(int k, int nb, bool want_inv, bool want_rhs, bool fused, bool legacy = false) {
  for (int c = 0; c < nb; ++c) {
    const int nt = c <= nb - 3 ? nb - 1 - c + (fused ? 2 : 0) : 0;
    if (k < nt) {
      if (k == 0) return ChItem{CH_EARLY_S, c, c + 2};
      if (k == 1) return ChItem{CH_EARLY_D, c, c + 2};
      if (fused) {
        if (k == 2) return ChItem{legacy ? CH_FUSED_S : CH_FUSED_D, c, c + 2};  // (ahead of FUSED_S, which waits for the copy of the tile's original entries
        if (k == 3) return ChItem{legacy ? CH_FUSED_D : CH_FUSED_S, c, c + 2};  //  this item makes when it STARTS: with one workgroup the order must be this one)
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
SGP_HD inline ChItem ch_dealt_item(const ChDeal& d, int k, int nb, bool want_inv, bool want_rhs, bool legacy = false) {
  if (!d.split) return ch_list_item(k, nb, want_inv, want_rhs, true, legacy);
  if (d.crit_wg) return ChItem{((k & 1) != 0) != legacy ? CH_FUSED_S : CH_FUSED_D, k >> 1, (k >> 1) + 2};
  return ch_list_item(k, nb, want_inv, want_rhs, false);
}

// -------------------------------------------------------------------------------------------------------------------------------------
// The TICKETED claim (round 6; the default -- the static deal above stays for A/B: SGP_POTRF_TICKET=0).  Items are taken in list order
// from a counter by whichever workgroup is free, so that an item is only ever held by a workgroup that is RUNNING: progress no longer
// needs every workgroup of the launch to be resident at once (two processes on one GPU, CU-masked streams).  With critical workgroups
// (`split`, as in the static deal: the workgroups expected on the chain workgroup's XCD) there are two counters -- the critical list
// [FUSED_D(0), FUSED_S(0), FUSED_D(1), ...] for them, the rest for the others, each kind carrying on with the other list when its own is
// exhausted -- and ONE rule that keeps the two lists from starving each other: a rest item of column c is not STARTED before the critical
// tickets of all columns < c have been taken; a workgroup that finds them missing takes them itself, one after the other, and runs them
// first (the rest item stays claimed by a workgroup that is working on its dependencies).  The critical workgroups claim unconditionally:
// should only they be running, their items would wait for rest items nobody can take -- but workgroups are dispatched in index order, so
// a running critical workgroup (blockIdx = 0 mod 8) implies seven running workgroups of the other kind before it.
// tests/native/chain_items_check.cpp runs this very function under adversarial delays and with only a prefix of the workgroups ever running.
// The atomics are the caller's: A::take(list) = fetch_add 1; A::take_below(list, bound) = the same while the counter is below `bound`, else -1.
struct ChClaim {
  int have_pending = 0, rest_done = 0, crit_done = 0;
  ChItem pending{CH_NONE, 0, 0};
};
SGP_HD inline int ch_crit_needed(const ChItem& it, int nb) {   // critical tickets that must be out before this rest item may start
  const int c = it.kind == CH_INV ? it.i - 1 : (it.kind == CH_RHS ? nb : it.c);  // (CH_INV: row it.i of L^-1 reads tiles (i, p), p < i)
  const int need = 2 * c, all = ch_crit_items(nb);
  return need < 0 ? 0 : (need > all ? all : need);
}
template <class A>
SGP_HD inline ChItem ch_claim_next(ChClaim& st, bool split, bool crit_wg, int nb, bool want_inv, bool want_rhs, A& a) {
  const int extra = (want_inv ? ch_inv_items(nb) : 0) + (want_rhs ? 1 : 0);
  if (!split) {
    const int k = a.take(0);
    return k < ch_tile_items(nb) + extra ? ch_list_item(k, nb, want_inv, want_rhs, true) : ChItem{CH_NONE, 0, 0};
  }
  const int ncrit = ch_crit_items(nb), nrest = ch_rest_tile_items(nb) + extra;
  auto crit_item = [](int t) { return ChItem{(t & 1) ? CH_FUSED_S : CH_FUSED_D, t >> 1, (t >> 1) + 2}; };
  for (;;) {
    if (st.have_pending) {
      // ONE atomic step (a compare-and-swap loop): draw the next critical ticket only while the counter is below `need`.  A look followed
      // by a draw let other workgroups in between, and the ticket drawn could be the pending item's own column's -- FUSED_D(c) waits for
      // EARLY_D(c): round 6's first version ran it at once and the launch hung on itself (tools/potrf_trace_check.py under a budget of 10).
      const int t = a.take_below(1, ch_crit_needed(st.pending, nb));
      if (t < 0) { st.have_pending = 0; return st.pending; }   // every critical ticket of the earlier columns is out
      return crit_item(t);   // a critical item of an EARLIER column nobody had claimed: this workgroup runs it before its own
    }
    if (crit_wg && !st.crit_done) {
      const int t = a.take(1);
      if (t < ncrit) return crit_item(t);
      st.crit_done = 1;
    }
    if (!st.rest_done) {
      const int k = a.take(0);
      if (k < nrest) { st.pending = ch_list_item(k, nb, want_inv, want_rhs, false); st.have_pending = 1; continue; }
      st.rest_done = 1;
    }
    if (!st.crit_done) {
      const int t = a.take(1);
      if (t < ncrit) return crit_item(t);
      st.crit_done = 1;
    }
    return ChItem{CH_NONE, 0, 0};
  }
}

// =====================================================================================================================================
// The ACCESS TABLE of the launch (round 6): for every work item and for the chain workgroup's two roles, the ordered
// list of what it waits for, reads, writes and raises.  One description for
//   * the kernel: the flag words of its scratch are laid out by ch_flag_slot() below (sgp_potrf_chain.hpp: ch_scratch()), and a trace
//     build (-DSGP_CH_TRACE) logs every wait / raise it really performs for tools/potrf_trace_check.py to hold against this table;
//   * the host checker (tests/native/chain_items_check.cpp): deadlock freedom is simulated from the WAIT entries, read/write hazards are
//     checked on the happens-before graph the WAIT / RAISE entries span (rules there) -- the in-place read of FUSED_S that shipped in
//     round 5 (a 3e-3 error under a clean status word) is kept as `legacy_fused_s` and must be flagged.
// Granularity: a 64 x 64 tile of the matrix (diagonal tiles: their four 16-column panels, which are published one by one), one block
// of each scratch array.  "ORIG": the caller's entries of a tile, as opposed to the factor's.
// =====================================================================================================================================
enum ChFlagKind { CF_READY, CF_ABORT, CF_PRES, CF_PRED, CF_PREADY, CF_PRESE, CF_PREDE, CF_PREADY_L, CF_XREADY_L, CF_CWX, CF_IREADY, CF_LO2R, CF_LO2R_L, CF_TICKET, CF_TICKET_CRIT, CF_NKIND };
struct ChFlag { int kind, a, b; };  // READY(i, j) / IREADY(i, j): a = i, b = j; PREADY[_L](j, pb): a = j, b = pb; the others: a = their column / row index
SGP_HD inline int ch_tile_no(int ti, int tj, int nb) { return tj * nb - (tj * (tj - 1)) / 2 + (ti - tj); }
// first slot (one slot = DF_FLAG_STRIDE ints = one cache line) of each kind of flag: [ready: ntile | abort | preS: nb | preD: nb | pready: 4 nb |
// preSE: nb | preDE: nb | pready_l: 4 nb | xready_l: nb | cwx | iready: ntile | lo2r: nb | lo2r_l: nb | ticket | ticket_crit]
SGP_HD inline int ch_flag_base(int kind, int nb) {
  const int ntile = nb * (nb + 1) / 2;
  switch (kind) {
    case CF_READY: return 0;
    case CF_ABORT: return ntile;
    case CF_PRES: return ntile + 1;
    case CF_PRED: return ntile + 1 + nb;
    case CF_PREADY: return ntile + 1 + 2 * nb;
    case CF_PRESE: return ntile + 1 + 6 * nb;
    case CF_PREDE: return ntile + 1 + 7 * nb;
    case CF_PREADY_L: return ntile + 1 + 8 * nb;
    case CF_XREADY_L: return ntile + 1 + 12 * nb;
    case CF_CWX: return ntile + 1 + 13 * nb;
    case CF_IREADY: return ntile + 2 + 13 * nb;
    case CF_LO2R: return 2 * ntile + 2 + 13 * nb;
    case CF_LO2R_L: return 2 * ntile + 2 + 14 * nb;
    case CF_TICKET: return 2 * ntile + 2 + 15 * nb;   // the claim counters of the ticketed deal (not flags: nobody waits for them)
    case CF_TICKET_CRIT: return 2 * ntile + 3 + 15 * nb;
    default: return 2 * ntile + 4 + 15 * nb;  // = the number of slots
  }
}
SGP_HD inline int ch_flag_slot(const ChFlag& f, int nb) {
  switch (f.kind) {
    case CF_READY: case CF_IREADY: return ch_flag_base(f.kind, nb) + ch_tile_no(f.a, f.b, nb);
    case CF_PREADY: case CF_PREADY_L: return ch_flag_base(f.kind, nb) + 4 * f.a + f.b;
    case CF_ABORT: case CF_CWX: case CF_TICKET: case CF_TICKET_CRIT: return ch_flag_base(f.kind, nb);
    default: return ch_flag_base(f.kind, nb) + f.a;
  }
}
enum ChLocKind { CL_TILE, CL_DPANEL, CL_UPPER, CL_DINV, CL_UPRE, CL_DPRE, CL_LO2, CL_UPE, CL_DPE, CL_XT, CL_LINV, CL_SOL, CL_NKIND };
struct ChLoc { int kind, a, b; };  // TILE / UPPER / XT / LINV (i, j); DPANEL / DINV (j, pb); the scratch blocks: a = their index
struct ChProgramOptions {
  bool lite = false;            // this (fused) item shares the chain workgroup's XCD: light flags, light publications
  bool legacy_fused_s = false;  // round 5 before 59da5e9: FUSED_S reads the tile's original entries IN PLACE (the negative test)
};
// visitor: wait(ChFlag, count) | raise(ChFlag, bool light) | read(ChLoc, bool orig) | write(ChLoc) | barrier(int id)
template <class V>
SGP_HD inline void ch_prog_solve_panels(int jd, bool light, V& v) {  // solve_panels(): the panels of L(jd, jd) and their block inverses, as they appear
  for (int pb = 0; pb < 4; ++pb) {
    v.wait(ChFlag{light ? CF_PREADY_L : CF_PREADY, jd, pb}, 2);
    v.read(ChLoc{CL_DPANEL, jd, pb}, false);
    v.read(ChLoc{CL_DINV, jd, pb}, false);
  }
}
template <class V>
SGP_HD inline void ch_item_program(const ChItem& it, int nb, const ChProgramOptions& o, V& v) {
  const int c = it.c, i = it.i, jn = it.c + 1;
  switch (it.kind) {
    case CH_RHS:
      for (int jb = 0; jb < nb; ++jb) {
        for (int p = 0; p < jb; ++p) { v.wait(ChFlag{CF_READY, jb, p}, 1); v.read(ChLoc{CL_TILE, jb, p}, false); }
        v.wait(ChFlag{CF_READY, jb, jb}, 1);
        for (int pb = 0; pb < 4; ++pb) v.read(ChLoc{CL_DPANEL, jb, pb}, false);
        v.write(ChLoc{CL_SOL, jb, 0});
      }
      return;
    case CH_INV:  // block (i, c) of L^-1
      for (int p = c; p < i; ++p) {
        v.wait(ChFlag{CF_IREADY, p, c}, 1);
        v.wait(ChFlag{CF_READY, i, p}, 1);
        v.read(ChLoc{CL_XT, p, c}, false);
        v.read(ChLoc{CL_TILE, i, p}, false);
      }
      ch_prog_solve_panels(i, false, v);
      v.write(ChLoc{CL_XT, i, c});
      v.write(ChLoc{CL_LINV, i, c});
      v.raise(ChFlag{CF_IREADY, i, c}, false);
      return;
    case CH_EARLY_S:
    case CH_EARLY_D:
      if (c == 0) return;  // nothing to the left of column 0: no store, no flag (the FUSED items of column 0 do not ask)
      for (int p = 0; p < c; ++p) {
        v.wait(ChFlag{CF_READY, i, p}, 1);
        if (it.kind == CH_EARLY_S) { v.wait(ChFlag{CF_READY, jn, p}, 1); v.read(ChLoc{CL_TILE, jn, p}, false); }
        v.read(ChLoc{CL_TILE, i, p}, false);
      }
      if (it.kind == CH_EARLY_S) { v.write(ChLoc{CL_UPE, jn, 0}); v.raise(ChFlag{CF_PRESE, jn, 0}, false); }
      else { v.write(ChLoc{CL_DPE, i, 0}); v.raise(ChFlag{CF_PREDE, i, 0}, false); }
      return;
    case CH_TILE:
    case CH_FUSED_D:
    case CH_FUSED_S: {
      const bool fs = it.kind == CH_FUSED_S, fd = it.kind == CH_FUSED_D, lite = (fs || fd) && o.lite;
      if (fs || fd) v.wait(ChFlag{CF_CWX, 0, 0}, 1);   // ask_local(): where the chain workgroup runs
      if (!fs) v.write(ChLoc{CL_UPPER, c, i});          // the mirrored tile: zero
      if (!fs) v.read(ChLoc{CL_TILE, i, c}, true);      // the tile's original entries
      if (fs && o.legacy_fused_s) v.read(ChLoc{CL_TILE, i, c}, true);   // (round 5 before the fix: in place, with no order against FUSED_D's store)
      if (fd && !o.legacy_fused_s) {
        v.write(ChLoc{CL_LO2, c, 0});
        if (lite) v.raise(ChFlag{CF_LO2R_L, c, 0}, true);
      }
      for (int p = 0; p < c; ++p) {
        v.wait(ChFlag{CF_READY, i, p}, 1);
        v.wait(ChFlag{CF_READY, c, p}, 1);
        v.read(ChLoc{CL_TILE, i, p}, false);
        v.read(ChLoc{CL_TILE, c, p}, false);
      }
      if (fs && !o.legacy_fused_s) {
        v.wait(ChFlag{lite ? CF_LO2R_L : CF_LO2R, c, 0}, 1);
        v.read(ChLoc{CL_LO2, c, 0}, false);
      }
      ch_prog_solve_panels(c, lite, v);
      if (!fs) v.write(ChLoc{CL_TILE, i, c});
      if (it.kind == CH_TILE) { v.raise(ChFlag{CF_READY, i, c}, false); return; }
      if (fd) {
        v.raise(ChFlag{CF_READY, i, c}, false);
        if (!o.legacy_fused_s) {
          v.raise(ChFlag{CF_LO2R, c, 0}, false);
          if (!lite) v.raise(ChFlag{CF_LO2R_L, c, 0}, false);
        }
      }
      if (c > 0) {
        v.wait(ChFlag{fs ? CF_PRESE : CF_PREDE, fs ? jn : i, 0}, 1);
        v.read(ChLoc{fs ? CL_UPE : CL_DPE, fs ? jn : i, 0}, false);
      }
      if (fs) {
        if (lite) v.wait(ChFlag{CF_XREADY_L, c, 0}, 1);
        else v.wait(ChFlag{CF_READY, jn, c}, 1);
        v.read(ChLoc{CL_TILE, jn, c}, false);
        v.write(ChLoc{CL_UPRE, jn, 0});
        v.raise(ChFlag{CF_PRES, jn, 0}, lite);
      } else {
        v.write(ChLoc{CL_DPRE, i, 0});
        v.raise(ChFlag{CF_PRED, i, 0}, lite);
      }
      return;
    }
    default:
      return;
  }
}
// The chain workgroup, step j: the D-waves (0-3) and the S-waves (4-7) as two threads that meet at the step's three barriers
// (barrier ids 3 j + 0 / 1 / 2 = B1 / B2 / B3 of sgp_potrf_chain.hpp).
template <class V>
SGP_HD inline void ch_chain_d_program(int j, int nb, V& v) {
  if (j == 0) v.raise(ChFlag{CF_CWX, 0, 0}, false);
  v.barrier(3 * j + 0);
  v.barrier(3 * j + 1);
  for (int pb = 0; pb < 4; ++pb) {     // wave pb: the pivot chain of panel pb, the panel to global, its two flags
    v.write(ChLoc{CL_DPANEL, j, pb});
    v.raise(ChFlag{CF_PREADY_L, j, pb}, true);
    v.raise(ChFlag{CF_PREADY, j, pb}, false);
    if (pb + 1 < 4) {                  // the wave that has just finished panel pb forms the block inverse of panel pb + 1 behind the chain
      v.write(ChLoc{CL_DINV, j, pb + 1});
      v.raise(ChFlag{CF_PREADY_L, j, pb + 1}, true);
      v.raise(ChFlag{CF_PREADY, j, pb + 1}, false);
    }
  }
  v.barrier(3 * j + 2);
  v.raise(ChFlag{CF_READY, j, j}, false);                       // (every D-wave released its panel before B3)
  if (j + 1 < nb) v.raise(ChFlag{CF_XREADY_L, j, 0}, true);     // X = L(j+1, j): the S-waves' stores have completed (B3), not written back
}
template <class V>
SGP_HD inline void ch_chain_s_program(int j, int nb, V& v) {
  if (j == 0)
    for (int pb = 0; pb < 4; ++pb) v.read(ChLoc{CL_DPANEL, 0, pb}, true);    // load_diag_blocks(0)
  v.barrier(3 * j + 0);
  v.barrier(3 * j + 1);
  if (j > 0) v.raise(ChFlag{CF_READY, j, j - 1}, false);        // X of the previous step, written back beside this step's chain
  v.write(ChLoc{CL_DINV, j, 0});                                 // S-wave 0: the block inverse of panel 0
  v.raise(ChFlag{CF_PREADY_L, j, 0}, true);
  v.raise(ChFlag{CF_PREADY, j, 0}, false);
  if (j + 1 < nb) {
    v.read(ChLoc{CL_TILE, j + 1, j}, true);
    for (int pb = 0; pb < 4; ++pb) v.read(ChLoc{CL_DPANEL, j + 1, pb}, true);  // load_diag_blocks(j + 1)
    v.write(ChLoc{CL_UPPER, j, j + 1});
    if (j >= 1) { v.wait(ChFlag{CF_PRES, j, 0}, 1); v.read(ChLoc{CL_UPRE, j, 0}, false); }
    v.write(ChLoc{CL_TILE, j + 1, j});                           // panel by panel behind the chain (LDS flags: inside the workgroup)
    if (j >= 1) { v.wait(ChFlag{CF_PRED, j + 1, 0}, 1); v.read(ChLoc{CL_DPRE, j + 1, 0}, false); }
  }
  v.barrier(3 * j + 2);
}

}  // namespace sgp
