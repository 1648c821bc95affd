#!/bin/bash
# Re-creates the artefacts kept under profiles/ for one round tag (run on the GPU box, from the repo root):
#     bash tools/profile_round.sh r01_v4
# 1. default bench line  2. rocprofv3 --kernel-trace --stats of the same command  3. PMC passes (one counter each).
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py --cpu-sample 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/pmc_$C" -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$OUT/pmc_$C.err"
done
python3 tools/summarise_pmc.py "$OUT/pmc_hbm_traffic.csv" FETCH_SIZE="$OUT/pmc_FETCH_SIZE" WRITE_SIZE="$OUT/pmc_WRITE_SIZE"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv"
# keep the merge-back small: the raw traces are large
rm -rf "$OUT/stats" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
ls -la "$OUT"
