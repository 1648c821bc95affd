import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "syrk_tile" in r["Kernel_Name"]]
k = len(idx) - 3
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(0, i0 - 12):i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"].split("(")[0][-50:]))
