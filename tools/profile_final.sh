#!/bin/bash
# The digest-matched final set of a round, in one go on the GPU box (from the repo root):   bash tools/profile_final.sh r05_v5
# bench line + rocprofv3 stats + PMC traffic (profile_round.sh), SQ counters (profile_sq.sh), TCC counters, potrf A/B against the round-1
# kernel, the smaller configs, the shard sizes of the multi-GPU job, C3 and 125 k-row-shard kernel timelines, the guarded orders' times,
# NUTS at the mid sizes, the chain Cholesky's budget sweep / soak / trace check, the trailing-word product's pieces, versions, accuracy sweep, race
# screen and counters, the experiment driver, smoke and the GPU suite.  Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/<tag>_*.
set -u
TAG=${1:-r05}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p "$O"
bash tools/profile_round.sh "$TAG" > "gpurun_out/${TAG}_round.log" 2>&1
bash tools/profile_sq.sh "$TAG" > "gpurun_out/${TAG}_sq.log" 2>&1
ARGS=""
for C in TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$O/tcc_$C" -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> "$O/tcc_$C.err"
  ARGS="$ARGS $C=$O/tcc_$C"
done
python3 tools/summarise_pmc.py "$O/pmc_tcc_counters.csv" $ARGS
rm -rf "$O"/tcc_TCC_*
timeout 300 python3 tools/potrf_bench.py > "$O/potrf_bench.jsonl" 2>/dev/null
SGP_POTRF_CHAIN=0 timeout 300 python3 tools/potrf_bench.py > "$O/potrf_bench_dataflow_kernel.jsonl" 2>/dev/null
timeout 300 python3 tools/bench_configs.py > "$O/small_configs.jsonl" 2>/dev/null
for r in 1000000 500000 250000 125000; do
  timeout 300 python3 tools/shard_trace.py $r >> "$O/shard_sizes.jsonl" 2>/dev/null
  timeout 300 python3 tools/shard_trace.py $r grad >> "$O/shard_sizes.jsonl" 2>/dev/null
done
rocprofv3 --kernel-trace --output-format csv -d "$O/trc3" -o run -- python3 tools/c3_trace.py > "$O/c3.out" 2> "$O/c3.err"
python3 tools/last_eval_timeline.py "$(find "$O/trc3" -name '*kernel_trace.csv' | head -1)" kuu_kernel > "$O/c3_timeline.txt" 2>&1
rm -rf "$O/trc3"
for mode in "" grad; do
  tag=${mode:-value}
  rocprofv3 --kernel-trace --output-format csv -d "$O/trs_$tag" -o run -- python3 tools/shard_trace.py 125000 $mode > /dev/null 2> "$O/shard_$tag.err"
  python3 tools/last_eval_timeline.py "$(find "$O/trs_$tag" -name '*kernel_trace.csv' | head -1)" kuu_kernel > "$O/shard125k_${tag}_timeline.txt" 2>&1
  rm -rf "$O/trs_$tag"
done
timeout 600 python3 tools/whitened_ms.py > "$O/whitened_ms.json" 2>/dev/null
# round 6: the chain Cholesky's evidence for THIS library (budgets x both claims, the claim's cost, a soak, the trace build against the access
# table) and the pieces of the extended order's two-word pass 2
timeout 600 python3 tools/potrf_budget_check.py < /dev/null > "$O/potrf_budget_check.txt" 2>&1
SGP_SHARED_DEVICE=1 timeout 300 python3 tools/potrf_bench.py < /dev/null > "$O/potrf_bench_ticketed_claim.jsonl" 2>/dev/null
SGP_POTRF_ACQUIRE=1 timeout 300 python3 tools/potrf_bench.py < /dev/null > "$O/potrf_bench_acquire.jsonl" 2>/dev/null
SGP_SHARED_DEVICE=1 timeout 600 python3 tools/bench_configs.py < /dev/null > "$O/small_configs_ticketed_claim.jsonl" 2>/dev/null
timeout 300 python3 tools/soak_potrf.py 60 < /dev/null > "$O/soak_potrf.txt" 2>&1
SGP_SHARED_DEVICE=1 timeout 300 python3 tools/soak_potrf.py 60 < /dev/null > "$O/soak_potrf_ticketed_claim.txt" 2>&1
timeout 900 bash tools/potrf_trace_check.sh < /dev/null > "$O/potrf_trace_check.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/lo_prof" -o run -- python3 tools/lo_kernel_ms.py < /dev/null > "$O/lo_kernel_ms.json" 2> "$O/lo.err"
f=$(find "$O/lo_prof" -name "*kernel_stats*" | head -1); if [ -n "$f" ]; then grep -E "Name|lo_|kphi|dd_gemm|phibar|kbar_contract" "$f" | sed 's/(.*)"/"/' > "$O/lo_kernel_stats.csv"; fi
rm -rf "$O/lo_prof"
# the trailing-word product: first version against the product; without the assembly kernel's fp16 image; the instruments (no contraction / two stages)
for c in "3 0 1" "1 0 1" "3 0 0" "3 1 1" "3 2 1" "3 0 1" "1 0 1" "3 0 0"; do
  set -- $c
  echo "SGP_LO_KERNEL=$1 SGP_LO_VARIANT=$2 LO_F16_IMAGE=$3"
  SGP_LO_KERNEL=$1 SGP_LO_VARIANT=$2 LO_F16_IMAGE=$3 timeout 300 python3 tools/lo_kernel_ms.py < /dev/null 2>/dev/null
done > "$O/lo_versions_ab.txt"
timeout 600 python3 tools/lo_v3_check.py < /dev/null > "$O/lo_v3_check.jsonl" 2>/dev/null
timeout 900 python3 tools/soak_lo.py 300 < /dev/null > "$O/soak_lo.txt" 2>&1
timeout 900 bash tools/profile_lo_pmc.sh "$TAG" > "$O/lo_pmc.log" 2>&1
GRADS=1 LEVEL=2 timeout 900 python3 tools/extended_check.py < /dev/null > "$O/extended_order_gradients_dd_phibar.jsonl" 2>/dev/null
timeout 600 python3 tools/extended_grad_check.py < /dev/null > "$O/extended_order_gradients_ard_dd_phibar.jsonl" 2>/dev/null
timeout 900 python3 tools/lo_threshold_probe.py < /dev/null > "$O/lo_threshold_probe.jsonl" 2>/dev/null
timeout 900 python3 experiments/large_scale_regression.py --max_iters 30 --hmc_samples 30 --hmc_tune 30 < /dev/null > "$O/experiment_large_scale_parity.json" 2> "$O/exp_p.err"
timeout 600 python3 tools/nuts_midsize.py 2>/dev/null > "$O/nuts_midsize.jsonl"
timeout 600 python3 __graft_entry__.py --smoke < /dev/null > "$O/smoke.txt" 2>&1
timeout 3000 python3 -m pytest tests -x -q -m gpu < /dev/null > "$O/pytest_gpu.txt" 2>&1
tail -3 "$O/pytest_gpu.txt"
cat "$O/potrf_bench.jsonl" "$O/shard_sizes.jsonl"
