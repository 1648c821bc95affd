set -u
export TMPDIR=/tmp
O=gpurun_out/r05_inv2
mkdir -p $O
timeout 300 python3 tools/host_profile.py grad > $O/host_profile_c3_grad.txt 2>&1
head -75 $O/host_profile_c3_grad.txt
