set -u
export TMPDIR=/tmp
bash tools/profile_round.sh r05_v5 > gpurun_out/r05_v5_round.log 2>&1
bash tools/profile_sq.sh r05_v5 > gpurun_out/r05_v5_sq.log 2>&1
O=gpurun_out/r05_v5
ARGS=""
for C in TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$O/tcc_$C" -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$O/tcc_$C.err"
  ARGS="$ARGS $C=$O/tcc_$C"
done
python3 tools/summarise_pmc.py "$O/pmc_tcc_counters.csv" $ARGS
rm -rf "$O"/tcc_TCC_*
ls -la $O
tail -c 600 $O/bench_default.json
timeout 300 python3 tools/potrf_bench.py > gpurun_out/r05_v5/potrf_bench.jsonl 2>/dev/null
SGP_POTRF_CHAIN=0 timeout 300 python3 tools/potrf_bench.py > gpurun_out/r05_v5/potrf_bench_dataflow_kernel.jsonl 2>/dev/null
timeout 300 python3 tools/bench_configs.py > gpurun_out/r05_v5/small_configs.jsonl 2>/dev/null
for r in 1000000 500000 250000 125000; do timeout 300 python3 tools/shard_trace.py $r >> gpurun_out/r05_v5/shard_sizes.jsonl 2>/dev/null; timeout 300 python3 tools/shard_trace.py $r grad >> gpurun_out/r05_v5/shard_sizes.jsonl 2>/dev/null; done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_v5/trc3 -o run -- python3 tools/c3_trace.py > gpurun_out/r05_v5/c3.out 2> gpurun_out/r05_v5/c3.err
python3 tools/last_eval_timeline.py $(find gpurun_out/r05_v5/trc3 -name "*kernel_trace.csv" | head -1) kuu_kernel > gpurun_out/r05_v5/c3_timeline.txt 2>&1
rm -rf gpurun_out/r05_v5/trc3
timeout 600 python3 tools/whitened_ms.py > gpurun_out/r05_v5/whitened_ms.json 2>/dev/null
cat gpurun_out/r05_v5/potrf_bench.jsonl gpurun_out/r05_v5/shard_sizes.jsonl
timeout 600 python3 tools/whitened_ms.py > gpurun_out/r05_v5/whitened_ms.json 2>/dev/null
cat gpurun_out/r05_v5/potrf_bench.jsonl gpurun_out/r05_v5/shard_sizes.jsonl
for mode in "" grad; do
  tag=${mode:-value}
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05_v5/trs_$tag -o run -- python3 tools/shard_trace.py 125000 $mode > /dev/null 2> gpurun_out/r05_v5/shard_$tag.err
  python3 tools/last_eval_timeline.py $(find gpurun_out/r05_v5/trs_$tag -name "*kernel_trace.csv" | head -1) kuu_kernel > gpurun_out/r05_v5/shard125k_${tag}_timeline.txt 2>&1
  rm -rf gpurun_out/r05_v5/trs_$tag
done
timeout 600 python3 tools/nuts_midsize.py 2>/dev/null > gpurun_out/r05_v5/nuts_midsize.jsonl
timeout 3000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r05_v5/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r05_v5/pytest_gpu.txt
