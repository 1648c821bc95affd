#!/usr/bin/env python3
"""Fills the @@...@@ placeholders of DESIGN.md section 0 / 4i / 8 and of README.md from a digest-matched final set (gpurun_out/<tag>/ of tools/profile_final.sh)."""
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06_final"
O = "gpurun_out/%s/" % tag
r = json.load(open(O + "bench_default.json"))
g = r["config"]["streaming_guard"]
t = r["leapfrog_per_s_trained_theta"]
lo = json.loads([l for l in open(O + "lo_kernel_ms.json") if l.startswith("{")][-1])
suite = [l.strip() for l in open(O + "pytest_gpu.txt") if " passed" in l][-1]
cb = r["cpu_baseline"]
vals = {
    "EVALS": "%.1f" % r["value"], "EVALMS": "%.2f" % r["ms_per_step"], "LEAP": "%.2f" % r["leapfrog_per_s"], "LEAPMS": "%.1f" % r["ms_per_leapfrog"],
    "DIGEST": r["config"]["source_digest"], "EXTLEAP": "%.1f" % g["extended_order"]["ms_per_leapfrog"], "EXTEVAL": "%.1f" % g["extended_order"]["ms_per_evaluation"],
    "WHLEAP": "%.1f" % g["whitened_order"]["ms_per_leapfrog"], "WHEVAL": "%.1f" % g["whitened_order"]["ms_per_evaluation"],
    "PARITY": "%.1f" % t["parity"], "SAMPLER": "%.1f" % t["sampler"], "LOMS": "%.1f" % lo["suffstats_bwd_lo_ms"],
    "GPUSUITE": re.sub(r" in [0-9.]+s.*", "", suite.replace("=", "").strip()),
    "CPUSUITE": sys.argv[2] if len(sys.argv) > 2 else "139 passed",
    "PARITYMS": "%.1f" % (1000.0 / t["parity"]),
    "CPUBASE": "%s = %.4f evaluations / s on %d host threads" % (re.search(r"all \d+ rows: ([0-9.]+ s)", cb["sample"]).group(1), cb["value"], cb["cores"]),
}
import csv
for row in csv.reader(open(O + "lo_kernel_stats.csv")):
    if row and "kphi_lo3_kernel" in row[0]:
        ms = float(row[3]) / 1e6
        vals.update(LO3MS="%.2f" % ms, LO3TF="%.0f" % (2.0 * 1e6 * 1024 * 1024 / (ms * 1e-3) / 1e12), LO3FRAC="%.2f" % (2.0 * 1e6 * 1024 * 1024 / (ms * 1e-3) / 2.5e15))
for name in ("DESIGN.md", "README.md", "INTEGRATION.md"):
    s = open(name).read()
    for k, v in vals.items():
        s = s.replace("@@%s@@" % k, v)
    left = re.findall(r"@@[A-Z0-9]+@@", s)
    open(name, "w").write(s)
    print(name, "unfilled:", left)
print(vals)
