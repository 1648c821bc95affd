set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_inv3
timeout 300 python3 tools/host_steps.py grad 2>&1 | tee gpurun_out/r05_inv3/host_steps_c3_grad.txt
