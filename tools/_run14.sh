set -u
export TMPDIR=/tmp
O=gpurun_out/r05_tl
mkdir -p $O
for mode in "" grad; do
  tag=${mode:-value}
  rocprofv3 --kernel-trace --output-format csv -d $O/trs_$tag -o run -- python3 tools/shard_trace.py 125000 $mode > $O/shard_$tag.out 2> $O/shard_$tag.err
  python3 tools/last_eval_timeline.py $(find $O/trs_$tag -name "*kernel_trace.csv" | head -1) kuu_kernel > $O/shard125k_${tag}_timeline.txt 2>&1
  rm -rf $O/trs_$tag
done
cat $O/shard125k_value_timeline.txt
