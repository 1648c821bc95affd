set -u
export TMPDIR=/tmp
O=gpurun_out/r05_inv6
mkdir -p $O
for i in 1 2 3; do
  echo "C3 $(timeout 300 python3 tools/c3_ab.py 2>/dev/null)"
done | tee $O/timings2.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trc3 -o run -- python3 tools/c3_trace.py > $O/c3.out 2> $O/c3.err
python3 tools/last_eval_timeline.py $(find $O/trc3 -name "*kernel_trace.csv" | head -1) kuu_kernel > $O/c3_timeline.txt 2>&1
rm -rf $O/trc3
tail -6 $O/c3_timeline.txt
