set -u
export TMPDIR=/tmp
bash tools/profile_round.sh r05_v1 > gpurun_out/r05_v1_round.log 2>&1
bash tools/profile_sq.sh r05_v1 > gpurun_out/r05_v1_sq.log 2>&1
O=gpurun_out/r05_v1
ARGS=""
for C in TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$O/tcc_$C" -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$O/tcc_$C.err"
  ARGS="$ARGS $C=$O/tcc_$C"
done
python3 tools/summarise_pmc.py "$O/pmc_tcc_counters.csv" $ARGS
rm -rf "$O"/tcc_TCC_*
ls -la $O
tail -c 600 $O/bench_default.json
