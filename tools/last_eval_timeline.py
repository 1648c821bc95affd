"""Kernel timeline of the LAST complete evaluation in a rocprofv3 kernel trace (csv): everything between the last two host read-backs."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
marks = [m for k, m in enumerate(marks) if k == 0 or m - marks[k - 1] > 1]  # first of each run of consecutive marker kernels
i0, i1 = marks[-2], marks[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"].split("(")[0][-60:]))
