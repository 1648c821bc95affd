set -u
export TMPDIR=/tmp
O=gpurun_out/r05_inv6
mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/pytest_full.txt 2>&1
tail -4 $O/pytest_full.txt
timeout 600 python3 __graft_entry__.py > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 600 python3 tools/soak.py 2>&1 | tail -3
