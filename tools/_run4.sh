set -u
export TMPDIR=/tmp
O=gpurun_out/r05_chain12
mkdir -p $O
SGP_EXTRA_HIPCC_FLAGS=-DSGP_POTRF_STAMPS python3 -c "import sys; sys.path.insert(0, 'generalised-gaussian-processes_amd'); import build; build.build_library(force=True)" > $O/stamps_build.txt 2>&1
timeout 120 python3 tools/potrf_chain_phases.py 1024 > $O/phases_1024.txt 2>&1; tail -40 $O/phases_1024.txt
