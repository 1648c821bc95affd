set -u
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kuu_factor_ex or inverse_inside or chain" 2>&1 | tail -3
