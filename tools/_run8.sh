set -u
export TMPDIR=/tmp
O=gpurun_out/r05_mid3
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trc3 -o run -- python3 tools/c3_trace.py > $O/c3.out 2> $O/c3.err
python3 tools/last_eval_timeline.py $(find $O/trc3 -name "*kernel_trace.csv" | head -1) kuu_kernel > $O/c3_timeline.txt 2>&1
rm -rf $O/trc3
cat $O/c3.out
timeout 300 python3 tools/bench_configs.py > $O/small_configs.jsonl 2> $O/small_configs.err; grep C3 $O/small_configs.jsonl
timeout 900 python3 experiments/large_scale_regression.py --max_iters 30 --hmc_samples 30 --hmc_tune 30 --hmc_gradient sampler > $O/experiment_large_scale_sampler.json 2> $O/exp_s.err; tail -c 900 $O/experiment_large_scale_sampler.json
timeout 900 python3 experiments/large_scale_regression.py --max_iters 30 --hmc_samples 30 --hmc_tune 30 > $O/experiment_large_scale_parity.json 2> $O/exp_p.err; tail -c 900 $O/experiment_large_scale_parity.json
