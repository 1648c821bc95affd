set -u
export TMPDIR=/tmp
O=gpurun_out/r05_mid1
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
