set -u
export TMPDIR=/tmp
O=gpurun_out/r05_base
mkdir -p $O
python3 tools/potrf_bench.py > $O/potrf_bench.jsonl 2> $O/potrf_bench.err
python3 tools/bench_configs.py > $O/small_configs.jsonl 2> $O/small_configs.err
for r in 125000 250000; do python3 tools/shard_trace.py $r >> $O/shard_sizes.jsonl 2>> $O/shard.err; python3 tools/shard_trace.py $r grad >> $O/shard_sizes.jsonl 2>> $O/shard.err; done
rocprofv3 --kernel-trace --output-format csv -d $O/tr125 -o run -- python3 tools/shard_trace.py 125000 > $O/tr125.out 2> $O/tr125.err
python3 tools/trace_timeline.py $(find $O/tr125 -name '*kernel_trace.csv' | head -1) > $O/shard125k_timeline.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tr125g -o run -- python3 tools/shard_trace.py 125000 grad > $O/tr125g.out 2> $O/tr125g.err
python3 tools/trace_timeline.py $(find $O/tr125g -name '*kernel_trace.csv' | head -1) > $O/shard125k_grad_timeline.txt 2>&1
rm -rf $O/tr125 $O/tr125g
timeout 1500 python3 -m pytest tests/test_c5_guarded_anchor.py tests/test_int8_theta_sweep.py -m gpu -q -x -s > $O/pytest_anchor.txt 2>&1
tail -5 $O/pytest_anchor.txt
