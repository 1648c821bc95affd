#!/bin/bash
# MFMA / LDS / wait counters of the bench kernels (one counter per rocprofv3 pass; run on the GPU box from the repo root):
#     bash tools/profile_sq.sh r01_v12
# -> gpurun_out/<tag>/pmc_sq_counters.csv  (copy to profiles/<tag>_pmc_sq_counters.csv)
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=""
for C in SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/sq_$C" -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$OUT/sq_$C.err"
  ARGS="$ARGS $C=$OUT/sq_$C"
done
python3 tools/summarise_pmc.py "$OUT/pmc_sq_counters.csv" $ARGS
rm -rf "$OUT"/sq_SQ_*
head -40 "$OUT/pmc_sq_counters.csv"
