set -u
export TMPDIR=/tmp
O=gpurun_out/r05_mid5
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
timeout 300 python3 tools/bench_configs.py > $O/small_configs.jsonl 2> $O/small_configs.err; grep C3 $O/small_configs.jsonl
for r in 1000000 500000 250000 125000; do timeout 300 python3 tools/shard_trace.py $r >> $O/shard_sizes.jsonl 2>> $O/shard.err; timeout 300 python3 tools/shard_trace.py $r grad >> $O/shard_sizes.jsonl 2>> $O/shard.err; done
cat $O/shard_sizes.jsonl
