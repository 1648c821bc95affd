#!/bin/bash
# HBM traffic and matrix-pipe counters of tools/crt_syrk_proto (one counter per rocprofv3 pass; run on the GPU box from the repo root):
#     bash tools/profile_crt_proto.sh r04_crt
# -> gpurun_out/<tag>/crt_proto_pmc.csv
set -u
TAG=${1:-r04_crt}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=""
for C in FETCH_SIZE WRITE_SIZE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/pmc_$C" -o run -- tools/crt_syrk_proto 1048576 1024 16 2 > /dev/null 2> "$OUT/pmc_$C.err"
  ARGS="$ARGS $C=$OUT/pmc_$C"
done
python3 tools/summarise_pmc.py "$OUT/crt_proto_pmc.csv" $ARGS
rm -rf "$OUT"/pmc_*
cat "$OUT/crt_proto_pmc.csv"
