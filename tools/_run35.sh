set -u
export TMPDIR=/tmp
for b in 0 24 40 64 128 0; do
  echo "cu_budget=$b $(CU_BUDGET=$b timeout 300 python3 tools/c3_ab.py 2>/dev/null)"
done
