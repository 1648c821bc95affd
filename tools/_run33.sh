set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_small
timeout 600 python3 tools/bench_configs.py 2>/dev/null | tee gpurun_out/r05_small/small_configs.jsonl
timeout 900 python3 -m pytest tests/test_small.py tests/test_composite.py tests/test_nuts_device_logic.py -x -q -m gpu 2>&1 | tail -3
