#!/usr/bin/env python3
"""Kernel timeline of ONE steady-state evaluation from a rocprofv3 --kernel-trace CSV: every launch between two
consecutive launches of the marker kernel (default kfu_assemble), start relative to the first, duration, queue.
    python3 tools/one_eval_timeline.py <kernel_trace.csv> [marker substring]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "kfu_assemble"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
k = len(idx) * 3 // 4
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
busy = 0.0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - s) / 1e3
    print("%9.1f %8.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"].split("(")[0][-60:]))
print("period %.1f us, %d launches, busy %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, i1 - i0, busy))
