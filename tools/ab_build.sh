#!/bin/bash
# Same-box A/B of a compile-time variant: builds the library twice on the GPU box (with and without the given hipcc
# flags) and runs a command with each, alternating.   bash tools/ab_build.sh "-DSGP_AB_LIBRARY_EXP" "python3 bench.py --cpu-sample 0 --steps 10"
set -u
FLAGS="$1"; shift
CMD="$*"
LIB=generalised-gaussian-processes_amd/csrc/libsgp_hip.so
python3 -c "import sys; sys.path.insert(0, 'generalised-gaussian-processes_amd'); import build; build.build_library(force=True)" && cp $LIB /tmp/lib_base.so
SGP_EXTRA_HIPCC_FLAGS="$FLAGS" python3 -c "import sys; sys.path.insert(0, 'generalised-gaussian-processes_amd'); import build; build.build_library(force=True)" && cp $LIB /tmp/lib_variant.so
for round in 1 2 3; do
  for v in base variant; do
    cp /tmp/lib_$v.so $LIB
    echo "== $v (round $round)"
    $CMD
  done
done
cp /tmp/lib_base.so $LIB
