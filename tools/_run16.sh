set -u
export TMPDIR=/tmp
O=gpurun_out/r05_tl
mkdir -p $O
timeout 300 python3 tools/readback_latency.py > $O/readback_latency.txt 2>&1
timeout 300 python3 tools/host_gap.py 125000 > $O/host_gap.txt 2>&1
timeout 300 python3 tools/host_gap.py 125000 grad >> $O/host_gap.txt 2>&1
cat $O/readback_latency.txt $O/host_gap.txt
