set -u
export TMPDIR=/tmp
O=gpurun_out/r05_mid4
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trc3 -o run -- python3 tools/c3_trace.py > $O/c3.out 2> $O/c3.err
F=$(find $O/trc3 -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $O/c3_gaps.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# gap before every gemv_rows_kernel that follows a gemm64 on the same queue, last 12 evaluations
out = []
for i in range(1, len(rows)):
    r, p = rows[i], rows[i - 1]
    if "gemv_rows" in r["Kernel_Name"]:
        # previous kernel on the same queue
        j = i - 1
        while j >= 0 and rows[j]["Queue_Id"] != r["Queue_Id"]:
            j -= 1
        if j >= 0 and "gemm64" in rows[j]["Kernel_Name"]:
            out.append((int(r["Start_Timestamp"]) - int(rows[j]["End_Timestamp"])) / 1e3)
print("gap gemm64 -> gemv_rows (us), per evaluation:", [round(v, 1) for v in out[-25:]])
PY
python3 tools/last_eval_timeline.py $F kuu_kernel > $O/c3_timeline.txt 2>&1
rm -rf $O/trc3
cat $O/c3_gaps.txt
timeout 300 python3 tools/host_overhead.py 13279 grad > $O/host_overhead_13k.txt 2>&1; head -45 $O/host_overhead_13k.txt
