#!/usr/bin/env python3
"""The chain-workgroup Cholesky's REAL synchronisation against the access table.

csrc/sgp_potrf_items.hpp lists, for every work item and for the chain workgroup's two roles, what is waited for and raised, in order
(ch_item_program, ch_chain_{d,s}_program); tests/native/chain_items_check.cpp proves progress and the absence of read / write hazards FROM
THAT TABLE.  This tool closes the loop on the GPU: a library built with -DSGP_CH_TRACE logs every wait that returned and every flag the
kernel raised, per workgroup and item, in the table's flag numbering; the log of every item must equal the table's program for it (the
chain workgroup's waves log concurrently: compared as multisets per step).

    bash tools/potrf_trace_check.sh          (builds the trace variant beside the product library, runs this, restores)
"""
import collections
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SGP_POTRF_CHAIN", "2")   # the chain kernel at every size (the product takes the round-1 kernel for nb <= 4 without inverse)
import ggp_amd  # noqa: E402

WG, LEN = 258, 8192
eng = ggp_amd.HipEngine()
lib = eng.lib
try:
    fn = lib.sgp_debug_potrf_trace
except AttributeError:
    sys.exit("this library was not built with -DSGP_CH_TRACE")
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
exe = "/tmp/chain_items_check_host"
subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "generalised-gaussian-processes_amd", "csrc"), "-o", exe,
                os.path.join(ROOT, "tests", "native", "chain_items_check.cpp")], check=True)


def table(nb, inv):
    out = subprocess.run([exe, "--programs", str(nb), str(int(inv))], capture_output=True, text=True, check=True).stdout
    items, chain, cwx = {}, {}, None
    for line in out.splitlines():
        head, _, rest = line.partition(":")
        f = head.split()
        if f[0] == "item":
            items[tuple(int(v) for v in f[1:5])] = rest.split()
        elif f[0] in ("chainD", "chainS"):
            chain[(f[0], int(f[1]))] = rest.split()
        elif f[0] == "slots":
            cwx = int(f[2])
    return items, chain, cwx


def fetch():
    counts = (C.c_int * WG)()
    entries = (C.c_int * (WG * LEN))()
    torch.cuda.synchronize()
    assert fn(C.cast(counts, C.c_void_p), C.cast(entries, C.c_void_p), 1) == 0
    return [list(entries[w * LEN:w * LEN + min(counts[w], LEN)]) for w in range(WG)], max(counts)


def decode(code):
    tag = (code >> 28) & 15
    if tag == 1:
        return "w%d" % (code & 0x7FFFFFF)
    if tag == 2:
        return "r%d%s" % (code & 0x7FFFFFF, "l" if code & (1 << 27) else "")
    return None


total_items = total_steps = bad = 0
BUDGET = int(os.environ.get("BUDGET", "0"))      # workgroups of the launch (sgp_set_cu_budget): few workgroups, many items each
CASES = ((192, False), (192, True), (384, True), (512, False), (512, True), (1024, True), (1024, False), (2048, True))
if os.environ.get("CASES"):
    CASES = tuple((int(c.split(":")[0]), c.split(":")[1] == "1") for c in os.environ["CASES"].split(","))
for M, inv in CASES:
    g = torch.Generator().manual_seed(M)
    R = torch.randn(M, M + 3, dtype=torch.float64, generator=g)
    K = (R @ R.T / M + torch.eye(M, dtype=torch.float64)).to(eng.device)
    fetch()  # clear
    lib.sgp_set_cu_budget(BUDGET)
    try:
        _, info = eng.kuu_factor(K) if inv else eng.chol_lower(K)
    finally:
        lib.sgp_set_cu_budget(0)
    logs, longest = fetch()
    if int(info.item()) != 0:
        bad += 1
        print("M %d inv %d budget %d: info %d" % (M, inv, BUDGET, int(info.item())))
        if os.environ.get("VERBOSE"):
            names = ["ES", "ED", "FS", "FD", "T", "INV", "RHS"]
            for w in range(0, 258):
                if logs[w]:
                    out = []
                    for code in logs[w]:
                        tag = (code >> 28) & 15
                        if tag == 4:
                            out.append("| %s(%d,%d)" % (names[(code >> 20) & 255], (code >> 10) & 1023, code & 1023))
                        elif tag == 6:
                            out.append("| step %d" % (code & 0xFFFF))
                        elif tag in (1, 2):
                            out.append(decode(code))
                    print("  wg %3d: %s" % (w, " ".join(out[-40:])))
    assert longest < LEN, "trace buffer too short"
    nb = (M + 127) // 128 * 2
    items, chain, cwx = table(nb, inv)
    seen = collections.Counter()
    lite_n = 0
    for w in range(1, 257):
        cur, seq, lite = None, [], 0

        def close():
            global bad, total_items
            if cur is None:
                return
            want = [t for t in items[cur + (lite,)] if t != "w%d" % cwx]
            got = [t for t in seq if t != "w%d" % cwx]
            total_items += 1
            seen[cur] += 1
            if want != got:
                bad += 1
                if bad <= 10:
                    print("M %d inv %d workgroup %d item %s lite %d:\n   table  %s\n   kernel %s" % (M, inv, w, cur, lite, " ".join(want), " ".join(got)))
        for code in logs[w]:
            tag = (code >> 28) & 15
            if tag == 4:
                close()
                cur, seq, lite = ((code >> 20) & 255, (code >> 10) & 1023, code & 1023), [], 0
            elif tag == 5:
                lite = code & 1
                lite_n += lite
            else:
                seq.append(decode(code))
        close()
    expect = {k[:3] for k in items if k[0] != 6}   # (the right-hand side item's waits are df_solve_rhs's own: not logged)
    if set(seen) != expect or any(v != 1 for v in seen.values()):
        bad += 1
        print("M %d inv %d: items run %d, expected %d; run more than once: %s" % (M, inv, len(seen), len(expect), [k for k, v in seen.items() if v != 1][:5]))
    for role, slot in (("chainD", 0), ("chainS", 257)):
        step, seq = None, []

        def close_step():
            global bad, total_steps
            if step is None:
                return
            total_steps += 1
            if sorted(chain[(role, step)]) != sorted(seq):
                bad += 1
                if bad <= 10:
                    print("M %d inv %d %s step %d:\n   table  %s\n   kernel %s" % (M, inv, role, step, sorted(chain[(role, step)]), sorted(seq)))
        pre = []
        for code in logs[slot]:
            if ((code >> 28) & 15) == 6:
                close_step()
                step, seq = code & 0xFFFF, list(pre)
                pre = []
            elif step is None:
                pre.append(decode(code))   # (the chain workgroup's XCC id goes out before its first step marker)
            else:
                seq.append(decode(code))
        close_step()
    print("M %4d inverse %d: nb %2d, %4d items on %3d workgroups (%d fused items took the light protocol), log of the longest workgroup %d entries"
          % (M, inv, nb, len(seen), sum(1 for w in range(1, 257) if logs[w]), lite_n, longest), flush=True)
print("%d items and %d chain steps compared with the access table: %d differences" % (total_items, total_steps, bad))
sys.exit(1 if bad else 0)
