#!/usr/bin/env python3
"""bench.py -- ELBO evaluations / s (and HMC leapfrogs / s) of the collapsed sparse-GP bound on MI355X.

Workload (BASELINE.json configs[4], SURVEY.md section 8d "C5"): synthetic regression, N = 1 000 000 rows,
d = 8, M = 1024 inducing points, RBF-ARD, fp64.  A "step" is ONE evaluation of the bound on all N rows
(pass 1 over the row shards + one all-reduce of [Phi|b|yy|kappa] + the O(M^3) tail).  With --gpus G the
N rows are split into G contiguous shards (strong scaling: the job is fixed, one rank per GPU over RCCL).

Prints ONE JSON line on rank 0 (see the driver contract in the task description).  Extra keys:
  leapfrog_per_s    : value + gradient wrt (lengthscales, sig_f, sig_n) evaluations / s, same run, same data
  roofline          : the dominant kernel of the timed evaluations: `i8_syrk_tile_kernel` (pass-1 contraction on the integer
                      matrix cores, what evaluations of big shards run) against the dense int8 peak, with the
                      fp64-equivalent rate beside it; `roofline_fp64_contraction` is `syrk_tile_kernel` (the fp64 contraction:
                      sgp_set_contraction(0), timed for the record) against the fp64 matrix peak.  achieved = the kernel's OWN algorithmic
                      work (N M (M + 1) flop; x 28 int8 op for the digit-pair products) / its HIP-event time
                      (events recorded by the library on the launch stream right around the kernel: sgp_timing_*).
                      `pass1` inside it times kernel assembly + contraction together against SURVEY section 8d's
                      whole-evaluation W_fwd; traffic = HBM bytes of the kernel from the newest profiles/*_pmc_hbm_traffic.csv
                      (PMC passes, tools/profile_round.sh), traffic_ratio = traffic / the kernel's algorithmic bytes;
                      mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) from the newest profiles/*_pmc_sq_counters.csv
  roofline_leapfrog : the same for the dominant kernel of the reverse pass, `kbar_contract_kernel` (2 N M^2 flop)
  cpu_baseline      : the oracle's PyMC3-op-order restatement timed on this box's host cores (rank 0, N=1 only): value
                      only on ALL N rows in 65 536-row chunks (--cpu-fit: two sample sizes, fixed + per-row
                      cost fitted and evaluated at N), and value + autograd gradient fitted from two samples (its graph does
                      not fit the host at N = 1M) next to leapfrog_per_s
"""
import csv
import glob
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
INT8_MATRIX_PEAK_TOPS = 5033.0  # dense int8 (= 2 x the dense bf16 peak: 256 CU x 4 SIMD x 2.4 GHz x 2048 op/clk), MI355X_MICROARCH.md
INT8_SUSTAINED_TOPS = 3528.0    # measured: back-to-back 32x32x32 int8 MFMAs on full-entropy operands (profiles/r03_i8_rates.txt)


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel at the default config (N = 1M, M = 1024, 1 GPU) from the newest
    profiles/*_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as
    MI355X_MICROARCH.md prescribes: tools/summarise_pmc.py).  Returns (bytes, file) or (None, None)."""
    import re
    # newest = highest (round, version) in the file name r<round>_v<version>_...: a checkout gives every file the same mtime
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.csv")),
                   key=lambda f: [int(t) for t in re.findall(r"\d+", os.path.basename(f))])
    for f in reversed(files):
        tot, seen = 0.0, set()
        with open(f, newline="") as fh:
            for row in csv.reader(r for r in fh if not r.startswith("#")):
                if len(row) == 5 and row[0] in ("FETCH_SIZE", "WRITE_SIZE") and kernel_substr in row[1] and row[0] not in seen:
                    tot += float(row[4])
                    seen.add(row[0])
        if len(seen) == 2:
            return tot, os.path.relpath(f, ROOT)
    return None, None


def synth(n_total, m, d):
    """X ~ N(0, I), y = sin(Xw) + 0.1 eps standardised, Z = X[randperm(N)[:M]]."""
    g = torch.Generator().manual_seed(0)
    X = torch.randn(n_total, d, dtype=torch.float64, generator=g)
    w = torch.randn(d, dtype=torch.float64, generator=g) / math.sqrt(d)
    y = torch.sin(X @ w) + 0.1 * torch.randn(n_total, dtype=torch.float64, generator=g)
    y = (y - y.mean()) / y.std()
    Z = X[torch.randperm(n_total, generator=g)[:m]].clone()
    return X, y, Z


def pmc_mfma_busy(kernel_substr):
    """Matrix-pipe utilisation of a kernel from the newest profiles/*_pmc_sq_counters.csv (tools/profile_sq.sh; one counter per rocprofv3
    pass): SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) -- MFMA-busy cycles per SIMD over the cycles its CU was busy.
    Returns (fraction, file) or (None, None)."""
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_sq_counters.csv")),
                   key=lambda f: [int(t) for t in re.findall(r"\d+", os.path.basename(f))])
    for f in reversed(files):
        val = {}
        with open(f, newline="") as fh:
            for row in csv.reader(r for r in fh if not r.startswith("#")):
                if len(row) == 5 and row[0] in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES") and kernel_substr in row[1] and row[0] not in val:
                    val[row[0]] = float(row[4])
        if len(val) == 2 and val["SQ_BUSY_CU_CYCLES"] > 0:
            return val["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * val["SQ_BUSY_CU_CYCLES"]), os.path.relpath(f, ROOT)
    return None, None


def algorithmic_flops_fwd(n, m, d):
    """SYRK lower triangle (2 flop/MAC) + b + scaled distances."""
    return n * m * (m + 1) + 2.0 * n * m + n * m * (3 * d + 2)


def cpu_baseline(X, y, Z, sample_rows, full):
    """The timed CPU path: the oracle's PyMC3-op-order form, torch-CPU fp64, every host thread."""
    from oracle import vfe_oracle as O
    torch.set_num_threads(os.cpu_count() or 1)
    N, d = X.shape
    M = Z.shape[0]
    ls = torch.full((d,), LS, dtype=torch.float64)

    def t_value(rows):
        t0 = time.perf_counter()
        O.vfe_pymc3_order_chunked(X[:rows], y[:rows], Z, ls, SF, SN, JITTER, chunk=65536)
        return time.perf_counter() - t0

    def t_grad(rows):
        t0 = time.perf_counter()
        O.grads_autograd(X[:rows], y[:rows], Z, ls, SF * SF, SN * SN, JITTER)
        return time.perf_counter() - t0

    t_value(min(8192, N))  # thread pool / allocator warm-up
    if full:
        full_s = t_value(N)
        how = "all %d rows: %.2f s" % (N, full_s)
    else:
        n1, n2 = max(1, sample_rows // 2), sample_rows
        t1, t2 = t_value(n1), t_value(n2)
        per_row = max((t2 - t1) / max(1, n2 - n1), 0.0)
        fixed = max(t1 - per_row * n1, 0.0)
        full_s = fixed + per_row * N
        how = ("%d rows: %.2f s, %d rows: %.2f s; fixed %.2f s + %.3g s/row evaluated at N = %d -> %.1f s"
               % (n1, t1, n2, t2, fixed, per_row, N, full_s))
    g1, g2 = max(1, min(N, sample_rows) // 8), max(2, min(N, sample_rows) // 4)
    tg1, tg2 = t_grad(g1), t_grad(g2)
    gper = max((tg2 - tg1) / max(1, g2 - g1), 0.0)
    gfull = max(tg1 - gper * g1, 0.0) + gper * N
    return {"value": 1.0 / full_s, "unit": "ELBO evals/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle.vfe_pymc3_order_chunked (PyMC3 MarginalSparse op order, torch-CPU fp64), M=%d d=%d, value only, %s"
                      % (M, d, how),
            "leapfrog": {"value": 1.0 / gfull, "unit": "value+gradient evals/s",
                         "sample": "oracle.grads_autograd (torch autograd through the same graph) on %d rows: %.2f s and %d rows: "
                                   "%.2f s, fitted and evaluated at N = %d -> %.1f s" % (g1, tg1, g2, tg2, N, gfull)}}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment): run the N ranks as fresh child processes
    under torch.distributed.run -- one rank per GPU, RCCL over xGMI -- relay their output (rank 0 prints the JSON
    line) and return the launcher's exit code.  Nothing here initialises HIP: `torch.cuda.device_count()` only counts."""
    import socket
    import subprocess
    share = os.environ.get("SGP_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < n and not share:
        print("bench.py: --gpus %d asks for %d ranks but this node shows %d device(s); refusing to run several ranks on "
              "one GPU (functional check only: SGP_BENCH_SHARE_GPU=1 SGP_BENCH_BACKEND=gloo)" % (n, n, have), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    global LS, SN  # (--ls / --sig-n: cpu_baseline() reads the module-level theta too)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="n", type=int, default=N_TOTAL)
    ap.add_argument("--inducing", dest="m", type=int, default=M_IND)
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="rows of the larger CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-full", dest="cpu_full", action="store_true", default=True,
                    help="time the CPU baseline's value-only evaluation on ALL N rows in 65 536-row chunks (SURVEY section 8d; ~80-110 s "
                         "at C5; the default since round 4)")
    ap.add_argument("--cpu-fit", dest="cpu_full", action="store_false",
                    help="fit the CPU baseline from two samples (--cpu-sample rows and half of it) instead of timing all N rows")
    ap.add_argument("--ls", type=float, default=LS, help="lengthscale of the timed theta (default: SURVEY section 8d's 2.0; tests use a long "
                                                         "one to drive the streaming-order guard on several ranks)")
    ap.add_argument("--sig-n", type=float, default=SN, help="noise standard deviation of the timed theta (default 0.3)")
    ap.add_argument("--side-chain", choices=["on", "off", "graph", "launches"], default="on",
                    help="chol(Kuu) on a side stream beside pass 1 (A/B knob; default = the product default; 'graph' / 'launches' are "
                         "round-4 spellings of 'on': the chain is six plain launches now)")
    args = ap.parse_args()
    LS, SN = float(args.ls), float(args.sig_n)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, BEFORE anything in this process
        # touches the GPU (a process that has initialised HIP must never exec / fork GPU workers on this pool)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # functional check of the multi-rank path on a 1-GPU box: SGP_BENCH_BACKEND=gloo SGP_BENCH_SHARE_GPU=1 lets
    # several ranks share cuda:0 (RCCL refuses duplicate devices); numbers from such a run mean nothing
    backend = os.environ.get("SGP_BENCH_BACKEND", "nccl")
    if os.environ.get("SGP_BENCH_SHARE_GPU") == "1":
        local_rank = 0
        # several processes on one device: the single-launch Cholesky claims its work items by ticket (include/sgp.h: SGP_OPT_SHARED_DEVICE;
        # read when the engine's context is created) -- two statically dealt spinning launches can starve each other into a time-out
        os.environ.setdefault("SGP_SHARED_DEVICE", "1")
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
    coll_per_eval = cb.n_collectives / max(1, cb.n_evals)  # 1 with several ranks (the packed statistics), 0 with one
    dt_grad = timed(step_grad, max(2, args.steps // 2), 1)
    evals_per_s = args.steps / dt_val
    leap_per_s = max(2, args.steps // 2) / dt_grad

    # dominant kernels, timed alone: the library records HIP events on the launch stream right around them
    # (include/sgp.h: sgp_timing_*; slot 0 kernel assembly, 1 pass-1 contraction, 2 pass-2 contraction)
    import ctypes
    n_local = hi - lo
    eng.lib.sgp_timing_enable(1)
    kfu = eng.kfu_buffer(n_local, args.m)
    prev_mode = eng.lib.sgp_set_contraction(0)  # the fp64 contraction and the fp64 assembly, timed for the record (roofline_fp64_contraction)
    packed = eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", kfu=kfu)
    Kuu = eng.kuu(Zd, ls, sf2, JITTER, "rbf")
    adj = eng.bound(Kuu, packed, s2, args.n, with_adjoints=True)
    torch.cuda.synchronize(dev)
    reps = max(3, min(10, args.steps))
    ms = {0: [], 1: [], 2: []}
    for _ in range(reps):
        eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", out=packed, kfu=kfu)
        eng.suffstats_bwd(Xd, yd, Zd, ls, sf2, adj["Phibar"], adj["bbar"], -1.0 / (2.0 * s2), "rbf", kfu=kfu)
        for slot in ms:
            t = ctypes.c_float(0.0)
            assert eng.lib.sgp_timing_last_ms(slot, ctypes.byref(t)) == 0
            ms[slot].append(t.value)
    eng.lib.sgp_set_contraction(prev_mode)
    # pass 1 of a value + gradient evaluation as the timed leapfrogs run it: on the integer cores the assembly writes the fp64 block too
    mg = {0: [], 1: []}
    for _ in range(reps):
        eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", out=packed, kfu=kfu)
        for slot in mg:
            t = ctypes.c_float(0.0)
            assert eng.lib.sgp_timing_last_ms(slot, ctypes.byref(t)) == 0
            mg[slot].append(t.value)
    int8_grad = eng.lib.sgp_contraction_last() == 1
    both_ms, i8g_ms = (sorted(v)[len(v) // 2] for v in (mg[0], mg[1]))
    del kfu
    med = {k: sorted(v)[len(v) // 2] for k, v in ms.items()}
    assemble_ms, syrk_ms, kbar_ms = med[0], med[1], med[2]
    # pass 1 as the timed evaluations (value only, nobody keeps K'_fu) run it: on big shards the library contracts on the INTEGER
    # matrix cores (include/sgp.h: sgp_set_contraction; csrc/sgp_suffstats_i8.hip) -- same statistics, different dominant kernel
    mv = {0: [], 1: []}
    for _ in range(reps):
        eng.suffstats(Xd, yd, Zd, ls, sf2, "rbf", out=packed)
        for slot in mv:
            t = ctypes.c_float(0.0)
            assert eng.lib.sgp_timing_last_ms(slot, ctypes.byref(t)) == 0
            mv[slot].append(t.value)
    int8_eval = eng.lib.sgp_contraction_last() == 1
    eng.lib.sgp_timing_enable(0)
    digits_ms, i8_ms = (sorted(v)[len(v) // 2] for v in (mv[0], mv[1]))
    Mp = (args.m + 127) // 128 * 128
    kfu_bytes = 8.0 * ((n_local + 255) // 256 * 256) * Mp
    syrk_flops = float(n_local) * args.m * (args.m + 1)            # the contraction's own work: lower triangle, 2 flop / MAC
    kbar_flops = 2.0 * float(n_local) * args.m * args.m             # Kbar_uf = 2 Phibar K_uf
    wfwd = algorithmic_flops_fwd(n_local, args.m, DIM)
    default_cfg = (args.n, args.m, world) == (N_TOTAL, M_IND, 1)
    syrk_traffic, syrk_file = pmc_traffic("sgp::syrk_tile_kernel<") if default_cfg else (None, None)
    kbar_traffic, kbar_file = pmc_traffic("kbar_contract_kernel") if default_cfg else (None, None)
    syrk_tf = syrk_flops / (syrk_ms * 1e-3) / 1e12
    i8_traffic, i8_file = pmc_traffic("sgp::i8_syrk_tile_kernel<6, false>(") if default_cfg else (None, None)
    i8_busy, i8_busy_file = pmc_mfma_busy("sgp::i8_syrk_tile_kernel<6, false>(") if default_cfg else (None, None)
    syrk_busy, _ = pmc_mfma_busy("sgp::syrk_tile_kernel<") if default_cfg else (None, None)
    kbar_busy, _ = pmc_mfma_busy("sgp::kbar_contract_kernel<") if default_cfg else (None, None)
    q_bytes = 7.0 * ((n_local + 255) // 256 * 256) * Mp          # digit planes: 7 bytes per element of K'_fu
    i8_ops = 28.0 * syrk_flops                                   # 28 digit-pair products per fp64 product, 2 op per MAC
    i8_tops = i8_ops / (i8_ms * 1e-3) / 1e12
    kbar_tf = kbar_flops / (kbar_ms * 1e-3) / 1e12
    note = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE in separate passes "
            "(tools/profile_round.sh), read from %s; not collected in this run")
    # the binary these numbers come from: sha256 over every kernel source / header / linker script (build.py), so a traffic
    # figure read from profiles/ can be matched to the build it was collected on
    try:
        with open(os.path.join(ROOT, "generalised-gaussian-processes_amd", "csrc", "libsgp_hip.so.sha256")) as fh:
            source_digest = fh.read().strip()[:16]
    except OSError:
        source_digest = None

    devices = [torch.cuda.get_device_properties(dev).name + " #%d" % local_rank]
    device_check = "single rank"
    allreduce_ms = None
    if world > 1:
        share = os.environ.get("SGP_BENCH_SHARE_GPU") == "1"
        props = torch.cuda.get_device_properties(dev)
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, local_rank, torch.cuda.current_device(), str(getattr(props, "uuid", "")),
                                         dist.get_backend()))
        devices = ["rank %d: cuda:%d %s" % (r, cur, u) for r, lr, cur, u, _ in gathered]
        # what "N GPUs" must mean: N ranks in the group, the collective library is RCCL, every rank on its own device
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)
        distinct = len({(cur, u) for _, _, cur, u, _ in gathered}) == world
        backends = sorted({b for *_, b in gathered})
        if share:
            device_check = "SGP_BENCH_SHARE_GPU=1: ranks share cuda:0 (functional check, numbers meaningless), backend %s" % backends
        else:
            if not distinct:
                raise SystemExit("bench.py: %d ranks but not %d distinct devices: %r" % (world, world, gathered))
            if backends != ["nccl"]:
                raise SystemExit("bench.py: collectives must run over RCCL (backend 'nccl'), got %r" % (backends,))
            device_check = "distinct device per rank, backend nccl (RCCL), world size %d" % world
        # THE exchange of an evaluation, timed alone: all-reduce of [lower triangle of Phi | b | yy | kappa]
        tri = torch.zeros(args.m * (args.m + 1) // 2 + args.m + 2, dtype=torch.float64, device=dev)
        for _ in range(3):
            dist.all_reduce(tri)
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(tri)
        torch.cuda.synchronize(dev)
        t = torch.tensor([(time.perf_counter() - t0) / 20], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allreduce_ms = 1e3 * float(t.item())
        allreduce_bytes = tri.numel() * 8

    res = {
        "metric": "ELBO evals/sec", "value": evals_per_s, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt_val / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C5 synthetic regression N=%d d=%d M=%d RBF-ARD, collapsed VFE bound, rows sharded over %d GPU(s)"
                               % (args.n, DIM, args.m, world), "N": args.n, "M": args.m, "d": DIM, "jitter": JITTER,
                   "theta": {"ls": LS, "sig_f": SF, "sig_n": SN}, "ranks": world, "collective_backend": backend if world > 1 else None,
                   "devices": devices, "device_check": device_check, "rows_per_rank": n_local,
                   "allreduce_ms": allreduce_ms, "allreduce_bytes": allreduce_bytes if world > 1 else None,
                   "collectives_per_eval": coll_per_eval, "source_digest": source_digest,
                   "evaluation_order": ("single launch (M <= 128)" if cb._small_ok(args.m) else
                                        "whitened (A = L^-1 K_uf materialised, B = I + A A^T / s2; pass 2 from the factored adjoint)"
                                        if cb._whitened(args.m) else
                                        "streaming (Phi = K_uf K_fu over the row shards, W = L^-1 Phi L^-T in the replicated tail)")},
        "leapfrog_per_s": leap_per_s, "ms_per_leapfrog": 1e3 / leap_per_s,
        "F": last["F"], "F_per_datum": last["F"] / args.n,
        "roofline": None,
        "roofline_fp64_contraction": {
                     "bound": "mfma", "kernel": "sgp::syrk_tile_kernel (pass-1 contraction on the fp64 matrix cores: sgp_set_contraction(0), and shards below the "
                                                "integer path's threshold)",
                     "achieved": syrk_tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": syrk_tf / FP64_MATRIX_PEAK_TFLOPS,
                     "ms": syrk_ms, "mfma_pipe_busy": syrk_busy, "algorithmic_flops": syrk_flops, "algorithmic_flops_formula": "N M (M + 1): lower triangle of Phi, 2 flop per MAC",
                     "algorithmic_bytes": kfu_bytes, "traffic": syrk_traffic,
                     "traffic_ratio": (syrk_traffic / kfu_bytes) if syrk_traffic else None,
                     "traffic_note": (note % syrk_file) if syrk_file else None,
                     "pass1": {"kernels": "kfu_assemble_kernel + syrk_tile_kernel", "ms": assemble_ms + syrk_ms, "algorithmic_flops": wfwd,
                               "achieved": wfwd / ((assemble_ms + syrk_ms) * 1e-3) / 1e12, "frac": wfwd / ((assemble_ms + syrk_ms) * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                               "note": "SURVEY section 8d W_fwd (contraction + b + scaled distances) over both kernels of pass 1"},
                     "materialised_note": "K'_fu is written once (HBM-write bound assembly) and read back by the contraction: a fused kernel would "
                                          "move 80.5 MB per evaluation instead of ~55 GB, but measured 53 ms (fp64 VALU exp() and MFMA share the datapath)"},
        "roofline_leapfrog": {"bound": "mfma", "kernel": "sgp::kbar_contract_kernel (pass-2 contraction + derivative epilogue)",
                              "achieved": kbar_tf, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": kbar_tf / FP64_MATRIX_PEAK_TFLOPS,
                              "ms": kbar_ms, "mfma_pipe_busy": kbar_busy, "algorithmic_flops": kbar_flops, "algorithmic_flops_formula": "2 N M^2 (Kbar_uf = 2 Phibar K_uf)",
                              "algorithmic_bytes": kfu_bytes, "traffic": kbar_traffic,
                              "traffic_ratio": (kbar_traffic / kfu_bytes) if kbar_traffic else None,
                              "traffic_note": (note % kbar_file) if kbar_file else None},
        "assembly": {"bound": "hbm", "kernel": "sgp::kfu_assemble_kernel<8,0>", "ms": assemble_ms,
                     "achieved": kfu_bytes / (assemble_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": kfu_bytes / (assemble_ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes": kfu_bytes},
    }
    if int8_eval:
        # the dominant kernel of the timed evaluations: the contraction on the integer matrix cores.  Its roofline is the int8
        # one; the fp64-equivalent rate (the work it REPLACES over its time) is what compares with the fp64 contraction above.
        res["roofline"] = {
            "bound": "mfma", "kernel": "sgp::i8_syrk_tile_kernel (pass-1 contraction of the timed evaluations: K'_fu as 7 balanced 8-bit digit "
                                       "planes, 28 exact int32 digit-pair products on v_mfma_i32_32x32x32_i8, one fp64 fold per split)",
            "achieved": i8_tops, "peak": INT8_MATRIX_PEAK_TOPS, "unit": "TOP/s", "frac": i8_tops / INT8_MATRIX_PEAK_TOPS, "ms": i8_ms,
            "algorithmic_ops": i8_ops, "algorithmic_ops_formula": "28 N M (M + 1): 28 of the 49 digit pairs of every fp64 product, 2 op per MAC",
            "sustained_peak": INT8_SUSTAINED_TOPS, "frac_of_sustained_peak": i8_tops / INT8_SUSTAINED_TOPS,
            "sustained_peak_note": "back-to-back v_mfma_i32_32x32x32_i8 on full-entropy operands, no memory traffic: the chip clocks down to "
                                   "~1.78 GHz (profiles/r03_i8_rates.txt) -- power, not issue slots, bounds this kernel",
            "fp64_equivalent": {"achieved": syrk_flops / (i8_ms * 1e-3) / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": syrk_flops / (i8_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                                "note": "N M (M + 1) flop of the fp64 contraction it replaces over its own time; > 1 = faster than the fp64 matrix peak allows"},
            "mfma_pipe_busy": i8_busy, "mfma_pipe_busy_note": ("SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) from %s (PMC passes, tools/profile_sq.sh); "
                                                                 "not collected in this run" % i8_busy_file) if i8_busy_file else None,
            "algorithmic_bytes": q_bytes, "traffic": i8_traffic, "traffic_ratio": (i8_traffic / q_bytes) if i8_traffic else None,
            "traffic_note": (note % i8_file) if i8_file else None,
            "pass1": {"kernels": "kfu_digits_kernel + i8_syrk_tile_kernel", "ms": digits_ms + i8_ms, "fp64_pass1_ms": assemble_ms + syrk_ms},
            "pass1_of_a_leapfrog": ({"kernels": "kfu_digits_kernel<.., true> (fp64 block + digit planes) + i8_syrk_tile_kernel", "assembly_ms": both_ms,
                                     "contraction_ms": i8g_ms, "ms": both_ms + i8g_ms} if int8_grad else None),
            "accuracy": "as the fp64 contraction: |K' - q 2^-54| <= 2^-55, digit products exact, dropped pairs < 6 x 2^-54 per product and "
                        "zero-mean; 2.4-2.8e-16 of max |Phi| against long double (tests/test_int8_contraction.py, tools/i8_syrk_proto.hip)"}
        res["config"]["contraction"] = "int8 digit planes on the integer matrix cores (error-free split of K'_fu, fp64 result)"
        res["dtype_note"] = ("fp64 kernel values, statistics, factorizations and gradients; the pass-1 contraction multiplies exact 8-bit digits of the "
                             "fp64 kernel values on the integer matrix cores (int32 sums, folded to fp64): as accurate as the fp64 contraction, not a reduced precision")
        res["assembly_digits"] = {"bound": "valu", "kernel": "sgp::kfu_digits_kernel<8,0>", "ms": digits_ms,
                                  "achieved": q_bytes / (digits_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                  "frac": q_bytes / (digits_ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes": q_bytes,
                                  "note": "exp + 53-bit fixed-point conversion + byte transposition per element: fp64 / integer VALU bound, not HBM bound"}
    else:
        res["roofline"] = dict(res["roofline_fp64_contraction"])
        res["config"]["contraction"] = "fp64 matrix cores"
    res["config"]["streaming_guard"] = {
        "tolerance_per_datum": cb.streaming_tol, "estimate_per_datum": cb.last_estimate, "repeats_in_whitened_order": cb.n_guard_reruns,
        "direct_whitened_evaluations": cb.n_direct_whitened, "extended_order_evaluations": cb.n_extended,
        "extended_order_level": cb.extended_level, "extended_order_reach_values": cb.extended_range,
        "extended_order_reach_gradients": (cb.extended_grad_range_lo if cb._bwd_lo_ok(args.m) else cb.extended_grad_range),
        "extended_order_gradient_check": "both words of a double-double Phibar in pass 2 (sgp_phibar_dd, sgp_suffstats_bwd_lo); accepted while the "
                                         "trailing word's correction is <= %g of the gradient, else repeated in the whitened order" % cb.extended_lo_max_correction,
        "note": "first-order estimate of |dF| / N of the streaming order (2^-53 max Phi_ii tr(Kuu^-1) / (s2 N), include/sgp.h: "
                "sgp_streaming_error_report), read back with every evaluation; above the tolerance the evaluation is repeated in the "
                "extended streaming order (level %d: %d digit pairs; values while the estimate is <= %g x the tolerance, value + gradient "
                "while <= %g x) or in the whitened (PyMC3) order beyond, and the evaluations that follow START there until the estimate is "
                "below half the tolerance -- 0, 0 and 0 = every timed step ran the streaming design"
                % (cb.extended_level, 39 if cb.extended_level >= 2 else 34, cb.extended_range,
                   cb.extended_grad_range_lo if cb._bwd_lo_ok(args.m) else cb.extended_grad_range)}
    if world == 1:
        # what an evaluation costs where the guard sends it: the whitened (PyMC3) order on the same shard and theta, a few repetitions
        # outside the timed region (engine.suffstats_whitened_rows + suffstats_bwd_factored)
        wb = ggp_amd.CollapsedBound(Xd, yd, kernel="rbf", jitter=JITTER, engine=eng, form="whitened")
        wb._kfu = cb._kfu  # the same N x M block: K'_fu in the streaming order, K'_fu L^-T in this one
        wv = lambda: wb.value(Zd, ls, sf2, s2)  # noqa: E731
        wg = lambda: wb.value_and_grad(Zd, ls, sf2, s2, want_gz=False)  # noqa: E731
        Fw = wv()[0]
        wg()
        torch.cuda.synchronize(dev)
        tw = []
        for fn in (wv, wg):
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
            tw.append((time.perf_counter() - t0) / 3 * 1e3)
        res["config"]["streaming_guard"]["whitened_order"] = {
            "ms_per_evaluation": tw[0], "ms_per_leapfrog": tw[1], "F_minus_streaming_F_per_datum": (Fw - last["F"]) / args.n,
            "note": "not part of `value`: the cost of one evaluation / one value+gradient in the order the guard falls back to beyond "
                    "the extended order's reach (%g x the tolerance for values, %g x for gradients that must hold 1e-6 -- and wherever "
                    "the trailing-word check rejects the extended order's gradient)"
                    % (cb.extended_range, cb.extended_grad_range_lo if cb._bwd_lo_ok(args.m) else cb.extended_grad_range)}
        del wb
        # ... and in the tier between: the extended streaming order (engine.suffstats_extended)
        xb = ggp_amd.CollapsedBound(Xd, yd, kernel="rbf", jitter=JITTER, engine=eng, form="extended")
        xb._kfu = cb._kfu
        xv = lambda: xb.value(Zd, ls, sf2, s2)  # noqa: E731
        xg = lambda: xb.value_and_grad(Zd, ls, sf2, s2, want_gz=False)  # noqa: E731
        Fx = xv()[0]
        xg()
        torch.cuda.synchronize(dev)
        tx = []
        for fn in (xv, xg):
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize(dev)
            tx.append((time.perf_counter() - t0) / 3 * 1e3)
        res["config"]["streaming_guard"]["extended_order"] = {
            "ms_per_evaluation": tx[0], "ms_per_leapfrog": tx[1], "F_minus_whitened_F_per_datum": (Fx - Fw) / args.n,
            "note": "not part of `value`: Phi on the integer cores with %d digit pairs and a double-double fold, W = L^-1 Phi L^-T in "
                    "double-double, pass 2 from the explicit Phibar -- formed in double-double, its leading word on the fp64 matrix cores, "
                    "its trailing word on the fp16 ones (round 6)" % (39 if xb.extended_level >= 2 else 34)}
        del xb
        # ... and what NUTS gets where the reference samples after training (train_fixed_model, models/bayesian_sgpr_hmc.py:160-180):
        # the trained ARD theta of profiles/r04_experiment_large_scale.json is in the guarded regime.  Leapfrogs / s of the NUTS target
        # there in the default mode (every gradient holds 1e-6: whitened order beyond 3 x the tolerance) and with
        # HmcTarget(gradient="sampler") (the extended order's gradient as far as its value holds; same energy, same posterior).
        ls_tr = [4.870895252562722, 2.274348615181124, 7.035384773166531, 6.388168428424034, 7.176420862837876, 3.3523772450641136,
                 2.314383327914714, 6.492694463809999][:DIM] + [2.0] * max(0, DIM - 8)
        th_tr = [math.log(v) for v in ls_tr] + [0.0, math.log(0.14415221312756948)]
        trained = {}
        for mode in ("parity", "sampler"):
            tb = ggp_amd.CollapsedBound(Xd, yd, kernel="rbf", jitter=JITTER, engine=eng)
            tb._kfu = cb._kfu
            tgt = ggp_amd.HmcTarget(tb, Zd, gradient=mode)
            for _ in range(2):
                lp, _g = tgt.logp_and_grad(th_tr)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(4):
                tgt.logp_and_grad(th_tr)
            torch.cuda.synchronize(dev)
            trained[mode] = {"leapfrog_per_s": 4.0 / (time.perf_counter() - t0), "tier": tb.last_tier, "estimate_per_datum": tb.last_estimate,
                             "logp": lp, "lo_correction": tb.last_lo_correction, "lo_rejections": tb.n_lo_rejections}
            del tgt, tb
        res["leapfrog_per_s_trained_theta"] = {
            "parity": trained["parity"]["leapfrog_per_s"], "sampler": trained["sampler"]["leapfrog_per_s"],
            "tier": {k: ("streaming", "extended", "whitened")[v["tier"]] for k, v in trained.items()},
            "estimate_per_datum": trained["parity"]["estimate_per_datum"],
            "logp_difference_per_datum": (trained["sampler"]["logp"] - trained["parity"]["logp"]) / args.n,
            "parity_trailing_word_correction_over_gradient": trained["parity"]["lo_correction"],
            "parity_trailing_word_rejections": trained["parity"]["lo_rejections"],
            "note": "outside `value`: value + gradient of the NUTS target at the trained ARD theta of C5 (lengthscales 2.3 .. 7.2, sig_n 0.144), "
                    "four evaluations each; `parity` (default): every gradient holds 1e-6 -- since round 6 in the extended order with both words "
                    "of a double-double Phibar while the trailing word's correction stays small (13.4 / s in the whitened order through round 5); "
                    "`sampler` = HmcTarget(gradient='sampler'): the leading word only"}
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        res["cpu_baseline"] = cpu_baseline(X, y, Z, min(args.cpu_sample, args.n), args.cpu_full)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
