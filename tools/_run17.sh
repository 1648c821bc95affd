set -u
export TMPDIR=/tmp
O=gpurun_out/r05_tl
mkdir -p $O
for i in 1 2 3; do
  for v in 0 1; do
    echo "inline=$v $(SGP_SIDE_INLINE=$v timeout 300 python3 tools/c3_ab.py 2>/dev/null)"
  done
done | tee $O/c3_side_inline_ab.txt
SGP_SIDE_INLINE=1 rocprofv3 --kernel-trace --output-format csv -d $O/trc3 -o run -- python3 tools/c3_trace.py > $O/c3.out 2> $O/c3.err
python3 tools/last_eval_timeline.py $(find $O/trc3 -name "*kernel_trace.csv" | head -1) kuu_kernel > $O/c3_inline_timeline.txt 2>&1
rm -rf $O/trc3
