#!/usr/bin/env python3
"""bench.py -- ELBO evaluations / s (and HMC leapfrogs / s) of the collapsed sparse-GP bound on MI355X.

Workload (BASELINE.json configs[4], SURVEY.md section 8d "C5"): synthetic regression, N = 1 000 000 rows,
d = 8, M = 1024 inducing points, RBF-ARD, fp64.  A "step" is ONE evaluation of the bound on all N rows
(pass 1 over the row shards + one all-reduce of [Phi|b|yy|kappa] + the O(M^3) tail).  With --gpus G the
N rows are split into G contiguous shards (strong scaling: the job is fixed, one rank per GPU over RCCL).

Prints ONE JSON line on rank 0 (see the driver contract in the task description).  Extra keys:
  leapfrog_per_s : value + gradient wrt (lengthscales, sig_f, sig_n) evaluations / s, same run, same data
  roofline       : dominant kernel (fused assembly + SYRK, `suffstats_fwd_kernel`) against the fp64 matrix peak
  cpu_baseline   : the oracle's PyMC3-op-order restatement timed on this box's host cores (rank 0, N=1 only)
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_TOTAL = 1_000_000
M_IND = 1024
DIM = 8
LS, SF, SN, JITTER = 2.0, 1.0, 0.3, 1e-6
FP64_MATRIX_PEAK_TFLOPS = 78.6  # MI355X datasheet fp64 matrix (= 256 CU x 2.4 GHz x 128 flop/clk/CU); see DESIGN.md
# HBM bytes one syrk_tile_kernel launch moved at the default config (N=1M, M=1024, 1 GPU), from the PMC passes
# committed under profiles/ (FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE); None for other configs.
SYRK_TRAFFIC_BYTES_PMC = 4.54e10


def synth(n_total, m, d):
    """X ~ N(0, I), y = sin(Xw) + 0.1 eps standardised, Z = X[randperm(N)[:M]]."""
    g = torch.Generator().manual_seed(0)
    X = torch.randn(n_total, d, dtype=torch.float64, generator=g)
    w = torch.randn(d, dtype=torch.float64, generator=g) / math.sqrt(d)
    y = torch.sin(X @ w) + 0.1 * torch.randn(n_total, dtype=torch.float64, generator=g)
    y = (y - y.mean()) / y.std()
    Z = X[torch.randperm(n_total, generator=g)[:m]].clone()
    return X, y, Z


def algorithmic_flops_fwd(n, m, d):
    """SYRK lower triangle (2 flop/MAC) + b + scaled distances."""
    return n * m * (m + 1) + 2.0 * n * m + n * m * (3 * d + 2)


def cpu_baseline(X, y, Z, sample_rows):
    from oracle import vfe_oracle as O
    torch.set_num_threads(os.cpu_count() or 1)
    Xs, ys = X[:sample_rows], y[:sample_rows]
    ls = torch.full((X.shape[1],), LS, dtype=torch.float64)
    t0 = time.perf_counter()
    O.vfe_pymc3_order_chunked(Xs, ys, Z, ls, SF, SN, JITTER, chunk=65536)
    dt = time.perf_counter() - t0
    full = dt * (X.shape[0] / float(sample_rows))
    return {"value": 1.0 / full, "unit": "ELBO evals/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle.vfe_pymc3_order_chunked (PyMC3 MarginalSparse op order, torch-CPU fp64) on the first %d of "
                      "%d rows, M=%d d=%d, value only: %.2f s, extrapolated linearly in N" % (sample_rows, X.shape[0], Z.shape[0], X.shape[1], dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="n", type=int, default=N_TOTAL)
    ap.add_argument("--inducing", dest="m", type=int, default=M_IND)
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--side-chain", choices=["graph", "launches", "off"], default="graph",
                    help="how chol(Kuu) is enqueued on the side stream (A/B knob; default = the product default)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # functional check of the multi-rank path on a 1-GPU box: SGP_BENCH_BACKEND=gloo SGP_BENCH_SHARE_GPU=1 lets
    # several ranks share cuda:0 (RCCL refuses duplicate devices); numbers from such a run mean nothing
    backend = os.environ.get("SGP_BENCH_BACKEND", "nccl")
    if os.environ.get("SGP_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the sparse-GP core has no CPU path")
    dev = torch.device("cuda", local_rank)

    import ggp_amd
    eng = ggp_amd.HipEngine(dev)
    X, y, Z = synth(args.n, args.m, DIM)
    lo, hi = ggp_amd.shard_rows(args.n, rank, world)
    Xd, yd, Zd = X[lo:hi].contiguous().to(dev), y[lo:hi].contiguous().to(dev), Z.to(dev)
    cb = ggp_amd.CollapsedBound(Xd, yd, kernel="rbf", jitter=JITTER, engine=eng)
    cb.use_graph = args.side_chain == "graph"
    cb.overlap_tail = args.side_chain != "off"
    ls = [LS] * DIM
    sf2, s2 = SF * SF, SN * SN

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    last = {}

    def step_value():
        last["F"], _ = cb.value(Zd, ls, sf2, s2)

    def step_grad():
        last["Fg"], last["g"] = cb.value_and_grad(Zd, ls, sf2, s2, want_gz=False)

    dt_val = timed(step_value, args.steps, args.warmup)
    dt_grad = timed(step_grad, max(2, args.steps // 2), 1)
    evals_per_s = args.steps / dt_val
    leap_per_s = max(2, args.steps // 2) / dt_grad

    # dominant kernel (the SYRK contraction of pass 1 on this rank's shard), timed alone: the library
    # records HIP events on the launch stream right around that kernel (include/sgp.h: sgp_timing_*)
    import ctypes
    eng.lib.sgp_timing_enable(1)
    packed = eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf")
    torch.cuda.synchronize(dev)
    reps = max(3, min(10, args.steps))
    syrk_ms, asm_ms = [], []
    for _ in range(reps):
        eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", out=packed)
        t = ctypes.c_float(0.0)
        assert eng.lib.sgp_timing_last_ms(1, ctypes.byref(t)) == 0
        syrk_ms.append(t.value)
        assert eng.lib.sgp_timing_last_ms(0, ctypes.byref(t)) == 0
        asm_ms.append(t.value)
    eng.lib.sgp_timing_enable(0)
    syrk_ms.sort()
    asm_ms.sort()
    pass1_ms = syrk_ms[len(syrk_ms) // 2]
    assemble_ms = asm_ms[len(asm_ms) // 2]
    n_local = hi - lo
    achieved = algorithmic_flops_fwd(n_local, args.m, DIM) / (pass1_ms * 1e-3) / 1e12
    kfu_bytes = 8.0 * ((n_local + 255) // 256 * 256) * ((args.m + 127) // 128 * 128)

    res = {
        "metric": "ELBO evals/sec", "value": evals_per_s, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt_val / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C5 synthetic regression N=%d d=%d M=%d RBF-ARD, collapsed VFE bound, rows sharded over %d GPU(s)"
                               % (args.n, DIM, args.m, world), "N": args.n, "M": args.m, "d": DIM, "jitter": JITTER,
                   "theta": {"ls": LS, "sig_f": SF, "sig_n": SN}},
        "leapfrog_per_s": leap_per_s, "ms_per_leapfrog": 1e3 / leap_per_s,
        "F": last["F"], "F_per_datum": last["F"] / args.n,
        "roofline": {"bound": "mfma", "kernel": "sgp::syrk_tile_kernel (pass-1 contraction, this rank's shard)",
                     "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": SYRK_TRAFFIC_BYTES_PMC if (args.n, args.m, world) == (N_TOTAL, M_IND, 1) else None, "ms": pass1_ms,
                     "algorithmic_flops": algorithmic_flops_fwd(n_local, args.m, DIM),
                     "traffic_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, "
                                     "profiles/r01_v14_pmc_hbm_traffic.csv (tools/profile_round.sh); collected in separate passes, not in this run"},
        "assembly": {"bound": "hbm", "kernel": "sgp::kfu_assemble_kernel<8,0>", "ms": assemble_ms,
                     "achieved": kfu_bytes / (assemble_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": kfu_bytes / (assemble_ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes": kfu_bytes},
    }
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        res["cpu_baseline"] = cpu_baseline(X, y, Z, min(args.cpu_sample, args.n))
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
