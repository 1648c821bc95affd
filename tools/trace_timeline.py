"""Print the kernel timeline of one evaluation from a rocprofv3 --kernel-trace CSV (diagnostic).
    python tools/trace_timeline.py <kernel_trace.csv> [index of the syrk launch to centre on]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "syrk_tile" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
i0, i1 = idx[k], idx[k + 1]
t_end = int(rows[i0]["End_Timestamp"])
for r in rows[max(0, i0 - 45):i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f  q%s %s" % ((s - t_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"][:70]))
