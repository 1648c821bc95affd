#!/usr/bin/env python3
"""Fills the @@...@@ placeholders of DESIGN.md section 0 / 4i / 8 from a digest-matched final set (gpurun_out/<tag>/ of tools/profile_final.sh)."""
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
    "CPUBASE": "%s = %.4f evaluations / s on %d host threads" % (re.search(r"all \d+ rows: ([0-9.]+ s)", cb["sample"]).group(1), cb["value"], cb["cores"]),
}
s = open("DESIGN.md").read()
for k, v in vals.items():
    s = s.replace("@@%s@@" % k, v)
left = re.findall(r"@@[A-Z]+@@", s)
open("DESIGN.md", "w").write(s)
print(vals, "unfilled:", left)
