set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_inv5
timeout 600 python3 tools/nuts_midsize.py 2>/dev/null | tee gpurun_out/r05_inv5/nuts_midsize.jsonl
