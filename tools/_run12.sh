set -u
export TMPDIR=/tmp
O=gpurun_out/r05_chain20
mkdir -p $O
for P in 0 1 3; do
SGP_EXTRA_HIPCC_FLAGS="-DSGP_POTRF_STAMPS -DSGP_CH_PRIO=$P" python3 -c "import sys; sys.path.insert(0, 'generalised-gaussian-processes_amd'); import build; build.build_library(force=True)" > $O/build_$P.txt 2>&1
timeout 120 python3 tools/potrf_chain_phases.py 1024 > $O/phases_1024_prio$P.txt 2>&1; echo "prio $P"; tail -4 $O/phases_1024_prio$P.txt | cut -c1-260
timeout 120 python3 tools/potrf_bench.py 2>/dev/null | grep -E '"M": (512|1024)' | cut -c1-120
done
