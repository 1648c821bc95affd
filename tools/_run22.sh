set -u
export TMPDIR=/tmp
O=gpurun_out/r05_inv3
for ns in 0 16 24 32 40 48 56; do
  echo "nsplit=$ns $(SGP_SYRK_NSPLIT=$ns timeout 300 python3 tools/c3_ab.py 2>/dev/null)"
done | tee $O/c3_syrk_nsplit.txt
