set -u
export TMPDIR=/tmp
O=gpurun_out/r05_chain19
mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chol or pivot or fuzz or potrf or logdiag" > $O/pytest_chol.txt 2>&1
tail -3 $O/pytest_chol.txt
timeout 300 python3 tools/potrf_bench.py > $O/potrf_bench.jsonl 2> $O/potrf_bench.err; cat $O/potrf_bench.jsonl
SGP_EXTRA_HIPCC_FLAGS=-DSGP_POTRF_STAMPS python3 -c "import sys; sys.path.insert(0, 'generalised-gaussian-processes_amd'); import build; build.build_library(force=True)" > $O/stamps_build.txt 2>&1
timeout 120 python3 tools/potrf_chain_phases.py 1024 > $O/phases_1024.txt 2>&1; tail -20 $O/phases_1024.txt
timeout 120 python3 tools/potrf_chain_phases.py 512 > $O/phases_512.txt 2>&1
