#!/bin/bash
# Counters of the trailing-word product's kernels (one counter per rocprofv3 pass over tools/lo_kernel_ms.py; from the repo root on the GPU box):
#     bash tools/profile_lo_pmc.sh r06_final   ->  gpurun_out/<tag>/lo_pmc_counters.csv  (copy to profiles/<tag>_lo_pmc_counters.csv)
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=""
for C in GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT/lopmc_$C" -o run -- python3 tools/lo_kernel_ms.py < /dev/null > /dev/null 2> "$OUT/lopmc_$C.err"
  ARGS="$ARGS $C=$OUT/lopmc_$C"
done
python3 tools/summarise_pmc.py "$OUT/lo_pmc_all.csv" $ARGS
grep -E "^#|counter|kphi_lo|lo_bx|lo_reduce|lo_prep|lo3_|kfu_digits" "$OUT/lo_pmc_all.csv" > "$OUT/lo_pmc_counters.csv"
rm -rf "$OUT"/lopmc_* "$OUT/lo_pmc_all.csv"
cat "$OUT/lo_pmc_counters.csv" | cut -c1-220
