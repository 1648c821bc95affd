set -u
export TMPDIR=/tmp
O=gpurun_out/r05_soak
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chain_cholesky or chol" > $O/pytest_chol.txt 2>&1; tail -4 $O/pytest_chol.txt
timeout 1500 python3 tools/soak.py > $O/soak.txt 2>&1; tail -4 $O/soak.txt
