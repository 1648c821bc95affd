set -u
export TMPDIR=/tmp
O=gpurun_out/r05_tl
mkdir -p $O
timeout 300 python3 tools/host_overhead.py 125000 > $O/host_overhead_125k_value.txt 2>&1
timeout 300 python3 tools/host_overhead.py 125000 grad > $O/host_overhead_125k_grad.txt 2>&1
head -70 $O/host_overhead_125k_value.txt
