set -u
export TMPDIR=/tmp
O=gpurun_out/r05_inv2
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for i in 1 2; do
  for v in inline legacy; do
    echo "side=$v C3 $(SGP_SIDE_MODE=$v timeout 300 python3 tools/c3_ab.py 2>/dev/null)"
    echo "side=$v 125k $(SGP_SIDE_MODE=$v timeout 300 python3 tools/shard_trace.py 125000 2>/dev/null) $(SGP_SIDE_MODE=$v timeout 300 python3 tools/shard_trace.py 125000 grad 2>/dev/null)"
    echo "side=$v 1M $(SGP_SIDE_MODE=$v timeout 300 python3 tools/shard_trace.py 1000000 2>/dev/null)"
  done
done | tee $O/side_mode_ab.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trc3 -o run -- python3 tools/c3_trace.py > $O/c3.out 2> $O/c3.err
python3 tools/last_eval_timeline.py $(find $O/trc3 -name "*kernel_trace.csv" | head -1) kuu_kernel > $O/c3_timeline.txt 2>&1
rm -rf $O/trc3
