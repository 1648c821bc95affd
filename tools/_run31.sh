set -u
export TMPDIR=/tmp
O=gpurun_out/r05_kbar
mkdir -p $O
for rows in 125000 250000; do
for ns in 0 64 96 128 160 192 320 488; do
  echo "rows=$rows nsb=$ns $(SGP_KBAR_NSPLIT=$ns timeout 300 python3 tools/shard_trace.py $rows grad 2>/dev/null)"
done
done | tee $O/kbar_nsplit_sweep.txt
