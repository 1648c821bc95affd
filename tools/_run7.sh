set -u
export TMPDIR=/tmp
O=gpurun_out/r05_mid2
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_int8_theta_sweep.py tests/test_svgp.py tests/test_posterior_pin.py tests/test_int8_contraction.py tests/test_extended_order.py -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trc3 -o run -- python3 tools/c3_trace.py > $O/c3.out 2> $O/c3.err
python3 tools/last_eval_timeline.py $(find $O/trc3 -name "*kernel_trace.csv" | head -1) kuu_kernel > $O/c3_timeline.txt 2>&1
rm -rf $O/trc3
cat $O/c3.out
