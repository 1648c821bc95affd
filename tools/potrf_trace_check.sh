#!/bin/bash
# Builds the trace variant of the library (-DSGP_CH_TRACE), runs tools/potrf_trace_check.py with the ticketed and with the static deal,
# restores the product library.  Run on the GPU box from the repository root.
set -u
LIB=generalised-gaussian-processes_amd/csrc/libsgp_hip.so
cp $LIB /tmp/lib_product.so; cp $LIB.sha256 /tmp/lib_product.sha256
SGP_EXTRA_HIPCC_FLAGS="-DSGP_CH_TRACE" python3 -c "import sys; sys.path.insert(0, 'generalised-gaussian-processes_amd'); import build; build.build_library(force=True)" || exit 1
rc=0
python3 tools/potrf_trace_check.py || rc=1
echo "== static deal"; SGP_POTRF_TICKET=0 python3 tools/potrf_trace_check.py || rc=1
echo "== acquire mode, no light flags"; SGP_POTRF_ACQUIRE=1 SGP_POTRF_LIGHT=0 python3 tools/potrf_trace_check.py || rc=1
cp /tmp/lib_product.so $LIB; cp /tmp/lib_product.sha256 $LIB.sha256
exit $rc
