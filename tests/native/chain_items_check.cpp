// Host checker of the chain-workgroup Cholesky (csrc/sgp_potrf_chain.hpp), driven by the ACCESS TABLE of csrc/sgp_potrf_items.hpp:
// ch_item_program / ch_chain_{d,s}_program list, for every work item and for the chain workgroup's two roles, what is waited for, read,
// written and raised, in the kernel's order.  Nothing about the kernel is restated here.  Checks, for block counts nb = 2 ... 64, with /
// without the inverse and the right-hand side, 1 ... 255 workgroups, fused items with / without the light (same-XCD) protocol:
//
//   DEAL        every item of the launch is dealt exactly once (static deal), nothing else is.
//   PROGRESS    the launch completes: every thread (a workgroup running its items strictly in its own order; the chain workgroup's D- and
//               S-waves) gets past all its waits.  Flags only ever rise and waits are thresholds, so the set of operations that can
//               complete is the same under EVERY schedule (monotone dataflow: an execution is a least fixed point) -- one sweep decides
//               for all of them, however any workgroup is delayed.  Also under a TICKETED claim (items taken in list order by whichever
//               workgroup is free, any number of workgroups 1 ... 255): simulated with adversarial claim orders.
//   FLAGS       a wait for count k on a flag is matched by exactly k raises in the launch (so "the wait has returned" means "every
//               raise has happened"), all by one workgroup.
//   HAZARDS     on the happens-before order spanned by program order inside a thread, the chain workgroup's barriers and raise -> wait
//               -- i.e. for EVERY schedule and EVERY assignment of items to workgroups (items are taken as independent threads: no
//               hazard may depend on which workgroup runs what, or when):
//     H1 one writer: a location is written by one thread group only (an item; the chain workgroup).
//     H2 read-after-publication: a read of a location's FINAL contents by another group comes after a wait (same thread, earlier) on a
//        flag that the writer's group raises after the write -- a full raise (agent-scope release: L2 write-back), or a light one
//        (stores complete in the XCD's L2) when reader and writer share the XCD.
//     H3 no foreign read of ORIGINAL contents: only the group that overwrites a location may read what the caller put there (a
//        write-after-read hazard has no flag to order it; and a line fetched early would sit stale in the reader's L2 / vector cache when
//        it reads again behind the flag -- the consumer side takes no acquire).
//     H4 never touched before publication: a group's FIRST access to a location it does not write is a read that satisfies H2
//        (the invariant DESIGN 4h states for the acquire-free consumer side).
//   NEGATIVE    the item lists / programs of round 5 before 59da5e9 (FUSED_S reads tile (c+2, c) in place; a 3e-3 error with info = 0) must
//               be flagged by H3 for every nb >= 3 -- and a versioned-memory replay shows the failure needs a schedule in which FUSED_S starts
//               late (the default deal, both items starting together, computes the right factor: why a green GPU suite missed it).
// Plain C++ (g++ -fsanitize=address,undefined): the header is index arithmetic only.  `--dump nb inv rhs nout lite` prints the wait /
// raise sequence of every workgroup in the form the kernel's trace build logs them (tools/potrf_trace_check.py compares).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <tuple>
#include <vector>
#include "sgp_potrf_items.hpp"

using namespace sgp;

enum OpKind { OP_WAIT, OP_RAISE, OP_READ, OP_WRITE, OP_BARRIER };
struct Op {
  int kind;
  int key;     // flag slot / location key / barrier id
  int arg;     // wait: count; raise: light; read: orig
};
using LocKey = std::tuple<int, int, int>;

struct Thread {
  int group = 0;   // 0 = the chain workgroup; items: 1 + their number
  int xcd = 0;     // 0 = the chain workgroup's XCD; anything else: somewhere else (distinct per thread: the conservative reading)
  std::vector<Op> ops;
  std::vector<int> item_start;  // (deal threads) op index at which each item begins
};

struct Recorder {
  int nb;
  std::map<LocKey, int>* loc_ids;
  Thread* t;
  int loc(const ChLoc& l) {
    auto k = LocKey(l.kind, l.a, l.b);
    auto it = loc_ids->find(k);
    if (it != loc_ids->end()) return it->second;
    const int id = (int)loc_ids->size();
    (*loc_ids)[k] = id;
    return id;
  }
  void wait(const ChFlag& f, int count) { t->ops.push_back(Op{OP_WAIT, ch_flag_slot(f, nb), count}); }
  void raise(const ChFlag& f, bool light) { t->ops.push_back(Op{OP_RAISE, ch_flag_slot(f, nb), light ? 1 : 0}); }
  void read(const ChLoc& l, bool orig) { t->ops.push_back(Op{OP_READ, loc(l), orig ? 1 : 0}); }
  void write(const ChLoc& l) { t->ops.push_back(Op{OP_WRITE, loc(l), 0}); }
  void barrier(int id) { t->ops.push_back(Op{OP_BARRIER, id, 0}); }
};

static const char* kind_name(int k) {
  static const char* n[] = {"EARLY_S", "EARLY_D", "FUSED_S", "FUSED_D", "TILE", "INV", "RHS", "NONE"};
  return n[k];
}
static const char* loc_name(int k) {
  static const char* n[] = {"TILE", "DPANEL", "UPPER", "DINV", "UPRE", "DPRE", "LO2", "UPE", "DPE", "XT", "LINV", "SOL"};
  return n[k];
}

struct Launch {
  int nb;
  bool want_inv, want_rhs;
  std::vector<ChItem> items;  // every item of the launch, in the merged list order (single list with the fused items)
};

static std::vector<ChItem> all_items(int nb, bool want_inv, bool want_rhs, bool legacy) {
  std::vector<ChItem> v;
  for (int k = 0;; ++k) {
    const ChItem it = ch_list_item(k, nb, want_inv, want_rhs, true, legacy);
    if (it.kind == CH_NONE) break;
    v.push_back(it);
  }
  return v;
}

// lite policy: 0 none, 1 all fused items, 2 alternating by column, 3 FUSED_D only, 4 FUSED_S only
static bool lite_of(const ChItem& it, int policy) {
  if (it.kind != CH_FUSED_S && it.kind != CH_FUSED_D) return false;
  switch (policy) {
    case 1: return true;
    case 2: return (it.c & 1) != 0;
    case 3: return it.kind == CH_FUSED_D;
    case 4: return it.kind == CH_FUSED_S;
    default: return false;
  }
}

static void chain_threads(int nb, std::map<LocKey, int>& loc_ids, Thread& D, Thread& S) {
  D.group = S.group = 0;
  D.xcd = S.xcd = 0;
  Recorder rd{nb, &loc_ids, &D}, rs{nb, &loc_ids, &S};
  for (int j = 0; j < nb; ++j) {
    ch_chain_d_program(j, nb, rd);
    ch_chain_s_program(j, nb, rs);
  }
}

// ------------------------------------------------------------------------------------------------ PROGRESS (+ FLAGS)
// threads[0], [1] = chain D, S.  Returns the number of failures (0 or 1), prints the first stuck thread.
static int run_to_fixpoint(std::vector<Thread>& th, int nslots, const char* tag, std::vector<int>* order_out = nullptr) {
  std::vector<int> count(nslots, 0);
  std::vector<size_t> at(th.size(), 0);
  std::vector<int> barrier_at(2, -1);  // barrier id each chain thread is parked at
  bool moved = true;
  while (moved) {
    moved = false;
    for (size_t t = 0; t < th.size(); ++t) {
      while (at[t] < th[t].ops.size()) {
        const Op& o = th[t].ops[at[t]];
        if (o.kind == OP_WAIT && count[o.key] < o.arg) break;
        if (o.kind == OP_BARRIER) {
          const size_t other = 1 - t;  // only the chain threads hold barriers
          if (at[other] < th[other].ops.size() && th[other].ops[at[other]].kind == OP_BARRIER && th[other].ops[at[other]].key == o.key) {
            ++at[other];
            ++at[t];
            moved = true;
            continue;
          }
          break;
        }
        if (o.kind == OP_RAISE) ++count[o.key];
        ++at[t];
        moved = true;
      }
    }
  }
  for (size_t t = 0; t < th.size(); ++t)
    if (at[t] < th[t].ops.size()) {
      const Op& o = th[t].ops[at[t]];
      std::printf("%s: thread %zu stuck at op %zu (%s key %d count %d, flag at %d)\n", tag, t, at[t], o.kind == OP_WAIT ? "wait" : "barrier", o.key, o.arg,
                  o.kind == OP_WAIT ? count[o.key] : -1);
      return 1;
    }
  (void)order_out;
  return 0;
}

static int check_flags(const std::vector<Thread>& th, int nslots, const char* tag) {
  std::vector<int> raises(nslots, 0), raiser(nslots, -1);
  int failures = 0;
  for (const Thread& t : th)
    for (const Op& o : t.ops)
      if (o.kind == OP_RAISE) {
        ++raises[o.key];
        if (raiser[o.key] >= 0 && raiser[o.key] != t.group) { std::printf("%s: flag slot %d raised by two groups (%d, %d)\n", tag, o.key, raiser[o.key], t.group); ++failures; }
        raiser[o.key] = t.group;
      }
  for (const Thread& t : th)
    for (const Op& o : t.ops)
      if (o.kind == OP_WAIT && raises[o.key] != o.arg) {
        std::printf("%s: a wait for %d on flag slot %d, which is raised %d times in the launch\n", tag, o.arg, o.key, raises[o.key]);
        if (++failures > 5) return failures;
      }
  return failures;
}

// ------------------------------------------------------------------------------------------------ HAZARDS
struct HazardReport { int h1 = 0, h2 = 0, h3 = 0, h4 = 0; int total() const { return h1 + h2 + h3 + h4; } };

// th[0], th[1]: chain D, S (group 0); th[2 ...]: one thread per item.
static HazardReport check_hazards(const std::vector<Thread>& th, const std::map<LocKey, int>& loc_ids, const char* tag, bool verbose, bool quiet = false) {
  HazardReport rep;
  const int nloc = (int)loc_ids.size();
  std::vector<LocKey> loc_key(nloc);
  for (auto& kv : loc_ids) loc_key[kv.second] = kv.first;
  auto lname = [&](int id) {
    static char buf[64];
    std::snprintf(buf, sizeof buf, "%s(%d,%d)", loc_name(std::get<0>(loc_key[id])), std::get<1>(loc_key[id]), std::get<2>(loc_key[id]));
    return buf;
  };
  // chain threads: barriers passed before each op (order between the two threads of the chain workgroup)
  std::vector<std::vector<int>> bar(2);
  for (int t = 0; t < 2; ++t) {
    int b = 0;
    for (const Op& o : th[t].ops) {
      bar[t].push_back(b);
      if (o.kind == OP_BARRIER) ++b;
    }
  }
  // a (thread ta, op ia) happens before b (thread tb, op ib) inside ONE group?
  auto hb_in_group = [&](int ta, int ia, int tb, int ib) {
    if (ta == tb) return ia < ib;
    if (ta < 2 && tb < 2) return bar[tb][ib] > bar[ta][ia];  // a barrier lies between them (a before it, b behind it)
    return false;
  };
  // writers
  struct W { int thread, op; };
  std::vector<std::vector<W>> writes(nloc);
  for (size_t t = 0; t < th.size(); ++t)
    for (size_t k = 0; k < th[t].ops.size(); ++k)
      if (th[t].ops[k].kind == OP_WRITE) writes[th[t].ops[k].key].push_back(W{(int)t, (int)k});
  for (int l = 0; l < nloc; ++l)
    for (size_t a = 1; a < writes[l].size(); ++a)
      if (th[writes[l][a].thread].group != th[writes[l][0].thread].group) {
        if (!quiet && (verbose || rep.h1 < 3)) std::printf("%s: H1 %s written by groups %d and %d\n", tag, lname(l), th[writes[l][0].thread].group, th[writes[l][a].thread].group);
        ++rep.h1;
      }
  // raises of every group's threads, by thread
  std::vector<std::vector<int>> raise_ops(th.size());
  for (size_t t = 0; t < th.size(); ++t)
    for (size_t k = 0; k < th[t].ops.size(); ++k)
      if (th[t].ops[k].kind == OP_RAISE) raise_ops[t].push_back((int)k);
  std::vector<std::vector<int>> group_threads;  // group -> threads
  for (size_t t = 0; t < th.size(); ++t) {
    if ((int)group_threads.size() <= th[t].group) group_threads.resize(th[t].group + 1);
    group_threads[th[t].group].push_back((int)t);
  }
  for (size_t t = 0; t < th.size(); ++t) {
    std::map<int, int> waited;        // flag slot -> op index of this thread's (first) wait on it
    std::map<int, bool> touched;      // location -> accessed before by this thread
    for (size_t k = 0; k < th[t].ops.size(); ++k) {
      const Op& o = th[t].ops[k];
      if (o.kind == OP_WAIT && !waited.count(o.key)) waited[o.key] = (int)k;
      if (o.kind != OP_READ) {
        if (o.kind == OP_WRITE) touched[o.key] = true;
        continue;
      }
      const int l = o.key;
      const bool own = !writes[l].empty() && th[writes[l][0].thread].group == th[t].group;
      if (o.arg) {  // ORIGINAL contents
        if (!writes[l].empty() && !own) {
          if (!quiet && (verbose || rep.h3 < 3)) std::printf("%s: H3 group %d reads the original %s, which group %d overwrites\n", tag, th[t].group, lname(l), th[writes[l][0].thread].group);
          ++rep.h3;
        }
        // inside the writer's group: the read must come before the write
        if (own)
          for (const W& w : writes[l])
            if (!hb_in_group((int)t, (int)k, w.thread, w.op)) {
              if (!quiet && (verbose || rep.h3 < 3)) std::printf("%s: H3 %s: the writer's own read of the original contents is not ordered before its write\n", tag, lname(l));
              ++rep.h3;
            }
        touched[l] = true;
        continue;
      }
      if (writes[l].empty()) {
        if (!quiet && (verbose || rep.h2 < 3)) std::printf("%s: H2 group %d reads the final %s, which nobody writes\n", tag, th[t].group, lname(l));
        ++rep.h2;
        continue;
      }
      if (own) {  // the writer's group reads its own result: after the write
        bool ok = false;
        for (const W& w : writes[l]) ok = ok || hb_in_group(w.thread, w.op, (int)t, (int)k);
        if (!ok) { if (!quiet && (verbose || rep.h2 < 3)) std::printf("%s: H2 %s read by its own group before the write\n", tag, lname(l)); ++rep.h2; }
        continue;
      }
      // another group: a wait of THIS thread, earlier, on a flag the writer's group raises behind (every one of) its write(s)
      bool published = false;
      const int wg = th[writes[l][0].thread].group;
      for (int rt : group_threads[wg]) {
        for (int rk : raise_ops[rt]) {
          bool after_all = true;
          for (const W& w : writes[l]) after_all = after_all && hb_in_group(w.thread, w.op, rt, rk);
          if (!after_all) continue;
          const Op& r = th[rt].ops[rk];
          auto wt = waited.find(r.key);
          if (wt == waited.end() || wt->second > (int)k) continue;
          if (r.arg /* light */ && th[t].xcd != th[rt].xcd) continue;   // stores complete in the writer's L2 only: another XCD does not see them
          published = true;
          break;
        }
        if (published) break;
      }
      if (!published) {
        if (!quiet && (verbose || rep.h2 < 3)) std::printf("%s: H2 group %d (xcd %d) reads the final %s of group %d without a wait on a valid publication\n", tag, th[t].group, th[t].xcd, lname(l), wg);
        ++rep.h2;
      } else if (touched.count(l)) {
        if (!quiet && (verbose || rep.h4 < 3)) std::printf("%s: H4 group %d had touched %s before its publication\n", tag, th[t].group, lname(l));
        ++rep.h4;
      }
      touched[l] = true;
    }
  }
  return rep;
}

// ------------------------------------------------------------------------------------------------ builders
static int nslots_of(int nb) { return ch_flag_base(CF_NKIND, nb); }

// one thread per item (hazard analysis)
static void build_item_threads(int nb, bool want_inv, bool want_rhs, int lite_policy, bool legacy, std::vector<Thread>& th, std::map<LocKey, int>& loc_ids) {
  th.clear();
  loc_ids.clear();
  th.resize(2);
  chain_threads(nb, loc_ids, th[0], th[1]);
  const std::vector<ChItem> items = all_items(nb, want_inv, want_rhs, legacy);
  for (size_t n = 0; n < items.size(); ++n) {
    Thread t;
    t.group = 1 + (int)n;
    ChProgramOptions o;
    o.lite = lite_of(items[n], lite_policy);
    o.legacy_fused_s = legacy;
    t.xcd = o.lite ? 0 : 1 + (int)n;
    Recorder r{nb, &loc_ids, &t};
    ch_item_program(items[n], nb, o, r);
    th.push_back(t);
  }
}

// one thread per workgroup of the static deal (progress); fills `dealt` with how often each item was handed out
static int build_deal_threads(int nb, bool want_inv, bool want_rhs, int nout, int lite_policy, bool legacy, std::vector<Thread>& th,
                              std::map<LocKey, int>& loc_ids, const char* tag) {
  int failures = 0;
  th.clear();
  loc_ids.clear();
  th.resize(2);
  chain_threads(nb, loc_ids, th[0], th[1]);
  std::map<std::tuple<int, int, int>, int> expect;
  for (const ChItem& it : all_items(nb, want_inv, want_rhs, legacy)) expect[std::make_tuple(it.kind, it.c, it.i)] = 0;
  const int total = ch_tile_items(nb) + (want_inv ? ch_inv_items(nb) : 0) + (want_rhs ? 1 : 0);
  if ((int)expect.size() != total) { std::printf("%s: %zu items in the list, %d by the counts\n", tag, expect.size(), total); ++failures; }
  for (int ow = 0; ow < nout; ++ow) {
    Thread t;
    t.group = 1 + ow;
    t.xcd = 1 + ow;
    Recorder r{nb, &loc_ids, &t};
    const ChDeal d = ch_deal(ow, nout, nb, want_inv, want_rhs);
    for (int k = d.first; k < d.count; k += d.stride) {
      const ChItem it = ch_dealt_item(d, k, nb, want_inv, want_rhs, legacy);
      if (it.kind == CH_NONE) break;
      auto e = expect.find(std::make_tuple(it.kind, it.c, it.i));
      if (e == expect.end()) { std::printf("%s: unexpected item (%s, %d, %d)\n", tag, kind_name(it.kind), it.c, it.i); ++failures; continue; }
      ++e->second;
      ChProgramOptions o;
      o.lite = lite_of(it, lite_policy);
      o.legacy_fused_s = legacy;
      t.item_start.push_back((int)t.ops.size());
      ch_item_program(it, nb, o, r);
    }
    th.push_back(t);
  }
  for (auto& e : expect)
    if (e.second != 1) {
      std::printf("%s: item (%s, %d, %d) dealt %d times\n", tag, kind_name(std::get<0>(e.first)), std::get<1>(e.first), std::get<2>(e.first), e.second);
      ++failures;
    }
  return failures;
}

// Ticketed claim (sgp_potrf_items.hpp: ch_claim_next -- the very function the kernel runs, with host counters): `nwg` workgroups are
// launched, but only the first `running` of them ever run (workgroups are dispatched in index order: the resident set of a launch that
// shares the GPU is a prefix), and the adversary delays any of those -- and the chain workgroup -- at random.  Every item must be run
// exactly once and the launch must complete.
struct HostAtomics {
  std::vector<int>* ctr;
  int take(int list) { return (*ctr)[list]++; }
  int take_below(int list, int bound) { return (*ctr)[list] < bound ? (*ctr)[list]++ : -1; }   // (one atomic step on the GPU too: a CAS loop)
};
static int run_ticketed(int nb, bool want_inv, bool want_rhs, int nwg, int running, int lite_policy, unsigned seed, const char* tag) {
  std::map<LocKey, int> loc_ids;
  std::vector<Thread> chain(2);
  chain_threads(nb, loc_ids, chain[0], chain[1]);
  std::mt19937 rng(seed);
  std::vector<int> count(nslots_of(nb), 0), ctr(2, 0);
  HostAtomics at{&ctr};
  std::map<std::tuple<int, int, int>, int> ran;
  for (const ChItem& it : all_items(nb, want_inv, want_rhs, false)) ran[std::make_tuple(it.kind, it.c, it.i)] = 0;
  struct Wg { ChClaim claim; bool busy = false, finished = false; std::vector<Op> ops; size_t at = 0; ChDeal deal; };
  std::vector<Wg> wg(running);
  for (int w = 0; w < running; ++w) wg[w].deal = ch_deal(w, nwg, nb, want_inv, want_rhs);
  std::vector<size_t> cat(2, 0);
  auto step_chain = [&]() {
    bool moved = false;
    for (int t = 0; t < 2; ++t)
      while (cat[t] < chain[t].ops.size()) {
        const Op& o = chain[t].ops[cat[t]];
        if (o.kind == OP_WAIT && count[o.key] < o.arg) break;
        if (o.kind == OP_BARRIER) {
          const int other = 1 - t;
          if (cat[other] < chain[other].ops.size() && chain[other].ops[cat[other]].kind == OP_BARRIER && chain[other].ops[cat[other]].key == o.key) {
            ++cat[other]; ++cat[t]; moved = true; continue;
          }
          break;
        }
        if (o.kind == OP_RAISE) ++count[o.key];
        ++cat[t];
        moved = true;
      }
    return moved;
  };
  int failures = 0;
  auto step_wg = [&](Wg& g) {
    bool moved = false;
    for (;;) {
      if (!g.busy) {
        if (g.finished) return moved;
        const ChItem it = ch_claim_next(g.claim, g.deal.split, g.deal.crit_wg, nb, want_inv, want_rhs, at);
        moved = true;
        if (it.kind == CH_NONE) { g.finished = true; return moved; }
        auto e = ran.find(std::make_tuple(it.kind, it.c, it.i));
        if (e == ran.end()) { std::printf("%s: claimed an item that is not in the launch (%s, %d, %d)\n", tag, kind_name(it.kind), it.c, it.i); ++failures; g.finished = true; return moved; }
        ++e->second;
        Thread t;
        Recorder r{nb, &loc_ids, &t};
        ChProgramOptions o;
        o.lite = lite_of(it, lite_policy);
        ch_item_program(it, nb, o, r);
        g.ops = t.ops;
        g.at = 0;
        g.busy = true;
      }
      while (g.at < g.ops.size()) {
        const Op& o = g.ops[g.at];
        if (o.kind == OP_WAIT && count[o.key] < o.arg) return moved;
        if (o.kind == OP_RAISE) ++count[o.key];
        ++g.at;
        moved = true;
      }
      g.busy = false;
    }
  };
  for (int idle_rounds = 0;;) {
    bool moved = false;
    const bool everyone = idle_rounds > 0;  // nothing moved under the adversary's choice: one round without delays tells a deadlock from bad luck
    if (everyone || rng() % 4 != 0) moved = step_chain() || moved;
    std::vector<int> order(running);
    for (int w = 0; w < running; ++w) order[w] = w;
    std::shuffle(order.begin(), order.end(), rng);
    for (int w : order) {
      if (!everyone && rng() % 3 == 0) continue;  // delayed
      moved = step_wg(wg[w]) || moved;
    }
    bool all_done = cat[0] == chain[0].ops.size() && cat[1] == chain[1].ops.size();
    for (const Wg& g : wg) all_done = all_done && g.finished;
    if (all_done) break;
    if (moved) { idle_rounds = 0; continue; }
    if (++idle_rounds >= 2) {
      std::printf("%s: ticketed claim deadlocks, %d of %d workgroups running (seed %u); counters %d / %d\n", tag, running, nwg, seed, ctr[0], ctr[1]);
      return failures + 1;
    }
  }
  for (auto& e : ran)
    if (e.second != 1) {
      std::printf("%s: ticketed claim, %d of %d workgroups: item (%s, %d, %d) run %d times\n", tag, running, nwg, kind_name(std::get<0>(e.first)), std::get<1>(e.first),
                  std::get<2>(e.first), e.second);
      if (++failures > 3) break;
    }
  return failures;
}

// Versioned-memory replay of the static deal under a schedule: policy 0 = round-robin one op at a time (everything starts together, the
// default launch's timing), 1 = random with workgroup `late` held back until nothing else can move.  Returns the number of reads that
// saw the wrong version (ORIG wanted but overwritten, or FINAL wanted but not yet written).
static int replay_versions(std::vector<Thread>& th, int nslots, int nloc, int policy, int late, unsigned seed) {
  std::vector<int> count(nslots, 0), written(nloc, 0);
  std::vector<size_t> at(th.size(), 0);
  std::mt19937 rng(seed);
  int wrong = 0;
  auto runnable = [&](size_t t) {
    if (at[t] >= th[t].ops.size()) return false;
    const Op& o = th[t].ops[at[t]];
    if (o.kind == OP_WAIT) return count[o.key] >= o.arg;
    if (o.kind == OP_BARRIER) {
      const size_t other = 1 - t;
      return at[other] < th[other].ops.size() && th[other].ops[at[other]].kind == OP_BARRIER && th[other].ops[at[other]].key == o.key;
    }
    return true;
  };
  auto exec = [&](size_t t) {
    const Op& o = th[t].ops[at[t]];
    if (o.kind == OP_BARRIER) { ++at[1 - t]; }
    else if (o.kind == OP_RAISE) ++count[o.key];
    else if (o.kind == OP_WRITE) written[o.key] = 1;
    else if (o.kind == OP_READ) {
      if (o.arg && written[o.key]) ++wrong;
      // (FINAL reads of a location nobody has written yet are H2's business; here only the overwritten-original case is counted)
    }
    ++at[t];
  };
  for (;;) {
    std::vector<size_t> cand;
    for (size_t t = 0; t < th.size(); ++t)
      if ((int)t != late && runnable(t)) cand.push_back(t);
    if (cand.empty() && late >= 0 && runnable((size_t)late)) cand.push_back((size_t)late);
    if (cand.empty()) break;
    if (policy == 0) { for (size_t t : cand) if (runnable(t)) exec(t); }
    else exec(cand[rng() % cand.size()]);
  }
  return wrong;
}

static void dump(int nb, bool inv, bool rhs, int nout, int lite_policy) {
  std::vector<Thread> th;
  std::map<LocKey, int> loc_ids;
  build_deal_threads(nb, inv, rhs, nout, lite_policy, false, th, loc_ids, "dump");
  const char* names[] = {"chainD", "chainS"};
  for (size_t t = 0; t < th.size(); ++t) {
    if (t < 2) std::printf("%s:", names[t]); else std::printf("wg%zu:", t - 2);
    for (const Op& o : th[t].ops) {
      if (o.kind == OP_WAIT) std::printf(" w%d", o.key);
      if (o.kind == OP_RAISE) std::printf(" r%d", o.key);
    }
    std::printf("\n");
  }
}

// `--programs nb inv`: the wait / raise sequence of every item (fused items: with and without the light protocol) and of every step of the
// chain workgroup's two roles, in the access table's flag numbering -- what the kernel's trace build must log (tools/potrf_trace_check.py)
static void print_ops(const std::vector<Op>& ops) {
  for (const Op& o : ops) {
    if (o.kind == OP_WAIT) std::printf(" w%d", o.key);
    if (o.kind == OP_RAISE) std::printf(" r%d%s", o.key, o.arg ? "l" : "");
  }
  std::printf("\n");
}
static void programs(int nb, bool inv) {
  std::map<LocKey, int> loc_ids;
  for (const ChItem& it : all_items(nb, inv, false, false))
    for (int lite = 0; lite < ((it.kind == CH_FUSED_S || it.kind == CH_FUSED_D) ? 2 : 1); ++lite) {
      Thread t;
      Recorder r{nb, &loc_ids, &t};
      ChProgramOptions o;
      o.lite = lite != 0;
      ch_item_program(it, nb, o, r);
      std::printf("item %d %d %d %d:", it.kind, it.c, it.i, lite);
      print_ops(t.ops);
    }
  for (int j = 0; j < nb; ++j) {
    Thread d, sw;
    Recorder rd{nb, &loc_ids, &d}, rs{nb, &loc_ids, &sw};
    ch_chain_d_program(j, nb, rd);
    ch_chain_s_program(j, nb, rs);
    std::printf("chainD %d:", j);
    print_ops(d.ops);
    std::printf("chainS %d:", j);
    print_ops(sw.ops);
  }
  std::printf("slots cwx %d\n", ch_flag_base(CF_CWX, nb));
}

int main(int argc, char** argv) {
  if (argc >= 7 && std::strcmp(argv[1], "--dump") == 0) {
    dump(std::atoi(argv[2]), std::atoi(argv[3]) != 0, std::atoi(argv[4]) != 0, std::atoi(argv[5]), std::atoi(argv[6]));
    return 0;
  }
  if (argc >= 4 && std::strcmp(argv[1], "--programs") == 0) {
    programs(std::atoi(argv[2]), std::atoi(argv[3]) != 0);
    return 0;
  }
  int failures = 0, cases = 0, hazard_cases = 0, ticket_cases = 0;
  char tag[160];
  const bool quick = std::getenv("CHAIN_CHECK_QUICK") != nullptr;   // block counts up to 8 only (the mutation tests of tests/test_sanitizers.py)
  const int nb_max_deal = quick ? 8 : 64, nb_max_ticket = quick ? 6 : 24, nb_max_hazard = quick ? 8 : 20;
  const int nouts[] = {1, 2, 3, 7, 8, 9, 15, 16, 17, 39, 64, 255};
  // ---- DEAL + PROGRESS + FLAGS over the static deal
  for (int nb = 2; nb <= nb_max_deal; nb = nb < 20 ? nb + 1 : nb + 11)
    for (int inv = 0; inv < 2; ++inv)
      for (int rhs = 0; rhs < 2; ++rhs)
        for (int nout : nouts) {
          const int items = ch_tile_items(nb) + (inv ? ch_inv_items(nb) : 0) + rhs;
          if (items == 0) continue;                 // (nb = 2 without inverse / rhs: the launch has no other workgroups' items at all)
          if (nout > items && nout != 1) continue;  // potrf_lower never launches more workgroups than items
          for (int lite = 0; lite < (nb <= 12 ? 5 : 2); ++lite) {
            std::snprintf(tag, sizeof tag, "nb %d inv %d rhs %d nout %d lite %d", nb, inv, rhs, nout, lite);
            std::vector<Thread> th;
            std::map<LocKey, int> loc_ids;
            int f = build_deal_threads(nb, inv != 0, rhs != 0, nout, lite, false, th, loc_ids, tag);
            if (!f) f += run_to_fixpoint(th, nslots_of(nb), tag);
            if (!f && lite < 2 && nout == nouts[0]) f += check_flags(th, nslots_of(nb), tag);
            failures += f;
            ++cases;
          }
        }
  // ---- PROGRESS under a ticketed claim, adversarial claim / run orders
  for (int nb = 2; nb <= nb_max_ticket; nb = nb < 10 ? nb + 1 : nb + 7)
    for (int inv = 0; inv < 2; ++inv)
      for (int rhs = 0; rhs < 2; ++rhs) {
        if (ch_tile_items(nb) + (inv ? ch_inv_items(nb) : 0) + rhs == 0) continue;
        for (int nwg : {1, 2, 3, 7, 8, 9, 31, 64, 255}) {
          const int items = ch_tile_items(nb) + (inv ? ch_inv_items(nb) : 0) + rhs;
          if (nwg > items && nwg != 1) continue;
          for (int running : {1, 2, 7, 8, 9, 16, nwg}) {
            if (running > nwg) continue;
            for (unsigned seed = 1; seed <= 2; ++seed) {
              std::snprintf(tag, sizeof tag, "ticket nb %d inv %d rhs %d nwg %d", nb, inv, rhs, nwg);
              failures += run_ticketed(nb, inv != 0, rhs != 0, nwg, running, (int)(seed % 3), seed * 7919u + (unsigned)nb + 31u * (unsigned)running, tag);
              ++ticket_cases;
            }
          }
        }
      }
  // ---- HAZARDS on the happens-before order, items as independent threads (any deal, any schedule)
  for (int nb = 2; nb <= nb_max_hazard; nb = nb < 12 ? nb + 1 : nb + 4)
    for (int inv = 0; inv < 2; ++inv)
      for (int rhs = 0; rhs < 2; ++rhs)
        for (int lite = 0; lite < 5; ++lite) {
          if (ch_tile_items(nb) + (inv ? ch_inv_items(nb) : 0) + rhs == 0) continue;
          std::snprintf(tag, sizeof tag, "hazards nb %d inv %d rhs %d lite %d", nb, inv, rhs, lite);
          std::vector<Thread> th;
          std::map<LocKey, int> loc_ids;
          build_item_threads(nb, inv != 0, rhs != 0, lite, false, th, loc_ids);
          const HazardReport r = check_hazards(th, loc_ids, tag, false);
          failures += r.total();
          ++hazard_cases;
        }
  // ---- NEGATIVE: round 5's lists before 59da5e9 must be flagged (H3), for every nb >= 3; and the replay shows the schedule dependence
  int negative_flagged = 0, negative_cases = 0, replay_default_wrong = 0, replay_late_wrong = 0;
  for (int nb = 3; nb <= 12; ++nb)
    for (int lite = 0; lite < 2; ++lite) {
      std::vector<Thread> th;
      std::map<LocKey, int> loc_ids;
      build_item_threads(nb, true, false, lite, true, th, loc_ids);
      const HazardReport r = check_hazards(th, loc_ids, "legacy", false, true);   // (quiet: the expected findings are counted, not printed)
      ++negative_cases;
      if (r.h3 >= nb - 2) ++negative_flagged;   // one in-place read per column c <= nb - 3
      else std::printf("NEGATIVE nb %d lite %d: the legacy lists were NOT flagged (H3 = %d, expected >= %d)\n", nb, lite, r.h3, nb - 2);
    }
  if (negative_flagged != negative_cases) ++failures;
  for (int nb = 3; nb <= 8; ++nb)
    for (int nout : {3, 5, 7, 16, 39}) {
      const int items = ch_tile_items(nb) + ch_inv_items(nb);
      if (nout > items) continue;
      std::vector<Thread> th;
      std::map<LocKey, int> loc_ids;
      if (build_deal_threads(nb, true, false, nout, 0, true, th, loc_ids, "legacy replay")) { ++failures; continue; }
      std::vector<Thread> a = th;
      replay_default_wrong += replay_versions(a, nslots_of(nb), (int)loc_ids.size(), 0, -1, 1);
      for (int late = 2; late < (int)th.size(); ++late) {
        std::vector<Thread> b = th;
        replay_late_wrong += replay_versions(b, nslots_of(nb), (int)loc_ids.size(), 1, late, 17u * (unsigned)late + (unsigned)nb);
      }
      // ... and today's lists never see a wrong version, however a workgroup is held back
      std::vector<Thread> cur;
      std::map<LocKey, int> cur_ids;
      build_deal_threads(nb, true, false, nout, 0, false, cur, cur_ids, "replay");
      for (int late = -1; late < (int)cur.size(); ++late) {
        std::vector<Thread> c = cur;
        const int wrong = replay_versions(c, nslots_of(nb), (int)cur_ids.size(), late < 0 ? 0 : 1, late < 2 ? -1 : late, 23u * (unsigned)(late + 2));
        if (wrong) { std::printf("replay nb %d nout %d late %d: %d reads saw an overwritten original\n", nb, nout, late, wrong); ++failures; }
      }
    }
  if (replay_late_wrong == 0) { std::printf("NEGATIVE: holding a workgroup back never exposed the legacy in-place read\n"); ++failures; }
  std::printf("%d deal/progress cases, %d ticketed runs, %d hazard cases; legacy lists flagged in %d of %d cases (replay: %d wrong reads with everything "
              "starting together, %d with one workgroup held back); %d failures\n",
              cases, ticket_cases, hazard_cases, negative_flagged, negative_cases, replay_default_wrong, replay_late_wrong, failures);
  std::printf("%d cases, %d failures\n", cases + ticket_cases + hazard_cases + negative_cases, failures);
  return failures ? 1 : 0;
}
