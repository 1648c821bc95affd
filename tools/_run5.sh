set -u
export TMPDIR=/tmp
O=gpurun_out/r05_chain17
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
timeout 300 python3 tools/bench_configs.py > $O/small_configs.jsonl 2> $O/small_configs.err; grep C3 $O/small_configs.jsonl
timeout 300 python3 tools/shard_trace.py 125000 >> $O/shard.jsonl 2>&1;  timeout 300 python3 tools/shard_trace.py 125000 grad >> $O/shard.jsonl 2>&1; cat $O/shard.jsonl
