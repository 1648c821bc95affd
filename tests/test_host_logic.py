"""CPU: the host layer above the C ABI (CollapsedBound, HmcTarget, sharding helpers) driven through the
oracle-backed test double, against the golden vectors.  Also the world_size-2 gloo run of the sharded path."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, golden_names, load_golden
from fake_engine import OracleEngine

import ggp_amd

KNAME = {0: "rbf", 1: "matern32", 2: "matern52"}


def T(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


@pytest.mark.parametrize("form", ["streaming", "whitened"])
@pytest.mark.parametrize("name", golden_names())
def test_collapsed_bound_value_and_grad(name, form):
    G = load_golden(name)
    eng = OracleEngine()
    cb = ggp_amd.CollapsedBound(T(G["X"]), T(G["y"]), kernel=KNAME[int(G["kernel_id"])], jitter=float(G["jitter"]), engine=eng,
                                form=form)
    F, parts = cb.value(T(G["Z"]), G["ls"], float(G["sf2"]), float(G["s2"]))
    tol = 1e-9 * max(1.0, abs(float(G["F"])))
    rt, rz = float(G["grad_rtol"]), float(G["gz_rtol"])
    ptol = tol * (1000.0 if rt > 1e-6 else 1.0)  # the two parts cancel to F; ill-conditioned fixtures pin them loosely
    assert abs(F - float(G["F"])) < tol and abs(parts["trace_term"] - float(G["trace_term"])) < ptol
    F2, g = cb.value_and_grad(T(G["Z"]), G["ls"], float(G["sf2"]), float(G["s2"]), want_gz=True)
    assert abs(F2 - F) < tol
    assert float((g["ls"] - T(G["g_ls"])).abs().max()) < rt * max(1.0, float(T(G["g_ls"]).abs().max()))
    assert abs(g["sf2"] - float(G["g_sf2"])) < rt * max(1.0, abs(float(G["g_sf2"])))
    assert abs(g["s2"] - float(G["g_s2"])) < rt * max(1.0, abs(float(G["g_s2"])))
    assert float((g["Z"] - T(G["g_Z"])).abs().max()) < rz * max(1.0, float(T(G["g_Z"]).abs().max()))
    assert eng.calls["suffstats" if form == "streaming" else "suffstats_whitened"] == 2
    assert eng.calls["suffstats_whitened" if form == "streaming" else "suffstats"] == 0
    assert eng.calls["suffstats_bwd"] == 1 and eng.calls["kuu_bwd"] == 1
    mean, var, _ = cb.predict(T(G["Xs"]), T(G["Z"]), G["ls"], float(G["sf2"]), float(G["s2"]))
    assert float((mean - T(G["pred_mean"])).abs().max()) < 1e-8 and float((var - T(G["pred_var"])).abs().max()) < 1e-8


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("rbf")])
def test_hmc_target(name):
    G = load_golden(name)
    cb = ggp_amd.CollapsedBound(T(G["X"]), T(G["y"]), jitter=1e-6, engine=OracleEngine())
    tgt = ggp_amd.HmcTarget(cb, T(G["Z"]))
    assert tgt.ndim == G["X"].shape[1] + 2
    rt = float(G["grad_rtol"])
    for th, lp_ref, g_ref in zip(G["hmc_theta"], G["hmc_logp"], G["hmc_grad"]):
        lp, gr = tgt.logp_and_grad(th)
        ftol = (1e-9 if rt <= 1e-6 else 1e-7) * max(1.0, abs(lp_ref))  # duplicate-Z fixture: cond(Kuu) ~ 1e6
        assert abs(lp - lp_ref) < ftol
        assert float(np.abs(np.array(gr) - g_ref).max()) < rt * max(1.0, float(np.abs(g_ref).max()))
        assert abs(tgt.logp(th) - lp_ref) < ftol


def test_failed_cholesky_paths():
    X = torch.randn(30, 2, dtype=torch.float64)
    y = torch.randn(30, dtype=torch.float64)
    Z = torch.zeros(5, 2, dtype=torch.float64)
    cb = ggp_amd.CollapsedBound(X, y, jitter=0.0, engine=OracleEngine())
    with pytest.raises(ggp_amd.NotPositiveDefiniteError):
        cb.value(Z, [1.0, 1.0], 1.0, 0.1)
    F, parts = cb.value(Z, [1.0, 1.0], 1.0, 0.1, raise_on_fail=False)
    assert math.isnan(F) and parts["info"] > 0
    tgt = ggp_amd.HmcTarget(cb, Z)  # PyMC3 semantics: non-finite energy, never an exception
    lp, gr = tgt.logp_and_grad([0.0, 0.0, 0.0, -1.0])
    assert lp == -math.inf and all(v == 0.0 for v in gr)
    # a leapfrog that flies off to where exp() overflows is a divergence too, not an OverflowError
    assert tgt.logp_and_grad([1e4, 0.0, 0.0, 0.0])[0] == -math.inf and tgt.logp([0.0, float("nan"), 0.0, 0.0]) == -math.inf


def test_shard_rows_partition():
    for N, W in ((10, 3), (1_000_000, 8), (7, 8), (0, 2)):
        cuts = [ggp_amd.shard_rows(N, r, W) for r in range(W)]
        assert cuts[0][0] == 0 and cuts[-1][1] == N
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


def test_1d_inputs_are_accepted_like_the_reference_demo():
    # experiments/demo_1d_regression.py:72 passes Z_init = torch.randn(25) (1-D) and X as N x 1
    g = torch.Generator().manual_seed(3)
    X = torch.randn(80, dtype=torch.float64, generator=g) * 2
    y = torch.sin(3 * X)
    Z = torch.linspace(-3, 3, 9, dtype=torch.float64)
    cb = ggp_amd.CollapsedBound(X, y, jitter=1e-6, engine=OracleEngine())
    F, _ = cb.value(Z, [0.7], 1.0, 0.1)
    from oracle import vfe_oracle as O
    assert abs(F - O.vfe_streaming(X[:, None], y, Z[:, None], torch.tensor([0.7], dtype=torch.float64), 1.0, 0.1, 1e-6)["F"]) < 1e-9


# ---------------------------------------------------------------------------------------------
# world_size = 2 over gloo: row shards + one all-reduce reproduce the single-process result
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ggp_amd as pkg
    from fake_engine import OracleEngine as Eng
    from conftest import load_golden as lg
    G = lg(name)
    X, y, Z = T(G["X"]), T(G["y"]), T(G["Z"])
    lo, hi = pkg.shard_rows(X.shape[0], rank, world)
    cb = pkg.CollapsedBound(X[lo:hi], y[lo:hi], jitter=float(G["jitter"]), engine=Eng())
    assert cb.N == X.shape[0]
    F, g = cb.value_and_grad(Z, G["ls"], float(G["sf2"]), float(G["s2"]), want_gz=True)
    mean, var, _ = cb.predict(T(G["Xs"]), Z, G["ls"], float(G["sf2"]), float(G["s2"]))
    q.put((rank, F, g["ls"].numpy(), g["sf2"], g["s2"], g["Z"].numpy(), mean.numpy(), var.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["rbf_d3_small", "rbf_d8_nojit"])
def test_two_rank_gloo_matches_single_process(name):
    G = load_golden(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    outs.sort(key=lambda t: t[0])
    tol = 1e-9 * max(1.0, abs(float(G["F"])))
    for rank, F, gls, gsf2, gs2, gZ, mean, var in outs:
        assert abs(F - float(G["F"])) < tol
        assert np.abs(gls - G["g_ls"]).max() < 1e-6 * max(1.0, np.abs(G["g_ls"]).max())
        assert abs(gsf2 - float(G["g_sf2"])) < 1e-6 * max(1.0, abs(float(G["g_sf2"])))
        assert abs(gs2 - float(G["g_s2"])) < 1e-6 * max(1.0, abs(float(G["g_s2"])))
        assert np.abs(gZ - G["g_Z"]).max() < 1e-6 * max(1.0, np.abs(G["g_Z"]).max())
        assert np.abs(mean - G["pred_mean"]).max() < 1e-8 and np.abs(var - G["pred_var"]).max() < 1e-8
    # replicated tail: both ranks hold bit-identical results
    assert outs[0][1] == outs[1][1] and np.array_equal(outs[0][5], outs[1][5])


def _uneven_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ggp_amd as pkg
    from fake_engine import OracleEngine as Eng
    from conftest import load_golden as lg
    G = lg("rbf_d3_small")
    X, y, Z = T(G["X"]), T(G["y"]), T(G["Z"])
    lo, hi = (0, 150) if rank == 0 else (150, X.shape[0])  # 150 and 250 rows, M = 30
    pkg.CollapsedBound.WHITENED_MAX_WORK = 200 * 30        # ... straddle the threshold of form="auto"
    cb = pkg.CollapsedBound(X[lo:hi], y[lo:hi], jitter=float(G["jitter"]), engine=Eng())
    choice = cb._whitened(Z.shape[0])
    F, g = cb.value_and_grad(Z, G["ls"], float(G["sf2"]), float(G["s2"]), want_gz=True)
    q.put((rank, choice, F, g["ls"].numpy(), g["Z"].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_uneven_shards_agree_on_the_evaluation_order():
    """form="auto" picks whitened / streaming from the shard size: with uneven shards either side of the threshold every rank
    must still make the SAME choice (they all-reduce W = L^-1 Phi L^-T or the raw Phi into one buffer) -- ADVICE r2, core.py:226."""
    G = load_golden("rbf_d3_small")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[0][1] == outs[1][1] is False  # the largest shard (250 x 30) is above the threshold: streaming on both
    for _, _, F, gls, gZ in outs:
        assert abs(F - float(G["F"])) < 1e-9 * max(1.0, abs(float(G["F"])))
        assert np.abs(gls - G["g_ls"]).max() < 1e-6 * max(1.0, np.abs(G["g_ls"]).max())
        assert np.abs(gZ - G["g_Z"]).max() < 1e-6 * max(1.0, np.abs(G["g_Z"]).max())
    assert outs[0][2] == outs[1][2]


# ---------------------------------------------------------------------------------------------
# world_size = 2: NUTS with the DEFAULT seed -- every rank must build the same trees (each leapfrog issues
# collectives: ranks that draw different momenta would issue different numbers of all-reduces and hang)
# ---------------------------------------------------------------------------------------------
def _nuts_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ggp_amd as pkg
    from fake_engine import OracleEngine as Eng
    from conftest import load_golden as lg
    G = lg("rbf_d3_small")
    X, y, Z = T(G["X"]), T(G["y"]), T(G["Z"])
    lo, hi = pkg.shard_rows(X.shape[0], rank, world)
    cb = pkg.CollapsedBound(X[lo:hi], y[lo:hi], jitter=1e-6, engine=Eng())
    tgt = pkg.HmcTarget(cb, Z)
    np.random.seed(1234 + rank)  # whatever the ranks' global RNG state is, the chains must coincide
    tr = pkg.sample_nuts(tgt, n_samples=5, tune=5, seed=None)
    thetas = np.stack([row["theta_unc"] for row in tr])
    q.put((rank, thetas, int(tr.n_leapfrog), int(cb.n_collectives), tr.get_sampler_stats("tree_size")))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_nuts_default_seed_builds_identical_trees():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nuts_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    outs.sort(key=lambda t: t[0])
    (_, th0, nl0, nc0, ts0), (_, th1, nl1, nc1, ts1) = outs
    assert np.array_equal(th0, th1), "ranks drew different chains"
    assert nl0 == nl1 and np.array_equal(ts0, ts1)
    assert nc0 == nc1 and nc0 == 2 * nl0  # one statistics + one gradient all-reduce per leapfrog
    assert th0.shape == (5, 5) and np.all(np.isfinite(th0))


def test_lower_triangle_exchange_round_trip():
    eng = OracleEngine()
    M = 7
    g = torch.Generator().manual_seed(0)
    A = torch.randn(M, M, dtype=torch.float64, generator=g)
    stats = torch.cat([(A + A.T).reshape(-1), torch.randn(M + 2, dtype=torch.float64, generator=g)])
    tri = eng.pack_lower(stats, M)
    assert tri.numel() == M * (M + 1) // 2 + M + 2
    back = eng.unpack_lower(tri, M, torch.zeros_like(stats))
    assert torch.equal(back, stats)


def test_few_host_threads_caps_and_restores_the_intra_op_pool():
    """core.few_host_threads: the host-driven loops run with at most 4 torch intra-op threads (a 128-thread pool made a
    BayesianSVGP minibatch step 10x slower on the GPU box) and the caller's setting comes back, also after an exception."""
    import torch
    from ggp_amd.core import few_host_threads
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(8)
        seen = []

        @few_host_threads
        def loop(fail):
            seen.append(torch.get_num_threads())
            if fail:
                raise RuntimeError("boom")
            return 7

        assert loop(False) == 7 and seen[-1] == 4 and torch.get_num_threads() == 8
        try:
            loop(True)
        except RuntimeError:
            pass
        assert seen[-1] == 4 and torch.get_num_threads() == 8
        torch.set_num_threads(2)
        assert loop(False) == 7 and seen[-1] == 2 and torch.get_num_threads() == 2
    finally:
        torch.set_num_threads(before)


def test_bench_gpus_n_refuses_to_run_n_ranks_on_fewer_devices():
    """`python bench.py --gpus 2` is the driver's SCALE command: without WORLD_SIZE it must start its own ranks, and on a node with
    fewer than 2 devices fail loudly ("2 ranks ... 1 device") rather than print a one-GPU number labelled n_gpus: 1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SGP_BENCH_SHARE_GPU")}
    if torch.cuda.device_count() >= 2:
        pytest.skip("node has two devices: the launcher would start the job")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "2 ranks" in r.stderr and "device" in r.stderr
    assert "{" not in r.stdout
    # a launcher that started a different number of ranks than --gpus says is refused as well
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=env, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_device_sampler_run_length_guard():
    """The persistent sampler's sync counters are cumulative ints: a run whose worst case could pass 2^31 is not sent to it
    (HmcTarget.device_sampler_ok(n) answers False and the host-driven sampler takes over; sample_nuts_device itself raises)."""
    from ggp_amd.core import device_run_fits
    assert device_run_fits(500, 3000, 10) and device_run_fits(13279, 3000, 10)      # the reference's runs, Elevator included
    assert not device_run_fits(13279, 12000, 10) and device_run_fits(13279, 12000, 7)
    assert device_run_fits(500, 100000, 10) and not device_run_fits(500, 1000000, 10)


# ---------------------------------------------------------------------------------------------
# the streaming-order guard (core.CollapsedBound.streaming_tol) on the CPU double, one process and two gloo ranks
# ---------------------------------------------------------------------------------------------
def _guard_problem():
    g = torch.Generator().manual_seed(4)
    N, M, d = 600, 24, 3
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.sin(X[:, 0]) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    Z = X[:M].clone()
    return X, y, Z


def test_streaming_guard_repeats_in_the_whitened_order_on_the_cpu_double():
    """Host logic of round 4's guard: a streaming-order evaluation whose error estimate (2^-53 max Phi_ii tr(Kuu^-1) / (s2 N)) exceeds
    `streaming_tol` is repeated in the whitened order -- value, value + gradient and the predictive factors; the evaluations that
    follow go to the whitened order directly (no wasted streaming attempt) until the predicted estimate (the whitened order's upper
    bound x the ratio seen at the trip) falls below half the tolerance; a benign theta is streamed; form="streaming" and
    streaming_tol = 0 never repeat."""
    import ggp_amd as pkg
    from fake_engine import GuardedOracleEngine
    X, y, Z = _guard_problem()
    old = pkg.CollapsedBound.WHITENED_MAX_WORK
    pkg.CollapsedBound.WHITENED_MAX_WORK = 0   # form="auto" would take the whitened order for a problem this small
    try:
        eng = GuardedOracleEngine()
        cb = pkg.CollapsedBound(X, y, jitter=1e-6, engine=eng)
        F0, _ = cb.value(Z, [0.8] * 3, 1.0, 0.3)                       # benign: short lengthscale, moderate noise
        assert cb.n_guard_reruns == 0 and cb.last_estimate < 1e-9 and eng.calls["suffstats_whitened"] == 0
        hard = ([25.0] * 3, 1.0, 1e-5)                                  # K_uu at its jitter floor, tiny noise
        F1, _ = cb.value(Z, *hard)
        assert cb.n_guard_reruns == 1 and cb.last_estimate > 1e-9 and eng.calls["suffstats_whitened"] == 1
        assert cb._prefer_whitened and 0.0 < cb.guard.ratio <= 1.0
        cw = pkg.CollapsedBound(X, y, jitter=1e-6, engine=GuardedOracleEngine(), form="whitened")
        assert F1 == cw.value(Z, *hard)[0]                              # the repeat IS the whitened evaluation
        n_stream = eng.calls["suffstats"]
        F2, g2 = cb.value_and_grad(Z, *hard, want_gz=True)              # straight to the whitened order: no streaming attempt
        Fw, gw = cw.value_and_grad(Z, *hard, want_gz=True)
        assert cb.n_guard_reruns == 1 and cb.n_direct_whitened == 1 and eng.calls["suffstats"] == n_stream
        assert F2 == Fw and torch.equal(g2["ls"], gw["ls"]) and torch.equal(g2["Z"], gw["Z"])
        cb.factors(Z, *hard)
        assert cb.n_direct_whitened == 2 and cb._prefer_whitened
        Fb, _ = cb.value(Z, [0.8] * 3, 1.0, 0.3)                       # benign again: still whitened (it cannot know yet) ...
        assert cb.n_direct_whitened == 3 and not cb._prefer_whitened     # ... but the bound it reports ends the episode
        assert abs(Fb - F0) < 1e-9 * abs(F0)
        Fc, _ = cb.value(Z, [0.8] * 3, 1.0, 0.3)
        assert Fc == F0 and cb.n_direct_whitened == 3 and cb.n_guard_reruns == 1   # streamed again, bit for bit as before
        cb._prefer_whitened = True
        cb.streaming_tol = 0.0
        cb.value(Z, *hard, raise_on_fail=False)
        cs = pkg.CollapsedBound(X, y, jitter=1e-6, engine=GuardedOracleEngine(), form="streaming")
        cs.value(Z, *hard, raise_on_fail=False)
        assert cb.n_guard_reruns == 1 and cb.n_direct_whitened == 3 and cs.n_guard_reruns == 0
    finally:
        pkg.CollapsedBound.WHITENED_MAX_WORK = old


def _guard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ggp_amd as pkg
    from fake_engine import GuardedOracleEngine
    X, y, Z = _guard_problem()
    pkg.CollapsedBound.WHITENED_MAX_WORK = 0
    lo, hi = pkg.shard_rows(X.shape[0], rank, world)
    cb = pkg.CollapsedBound(X[lo:hi], y[lo:hi], jitter=1e-6, engine=GuardedOracleEngine())
    out = []
    for ls, s2 in ((0.8, 0.3), (25.0, 1e-5), (3.0, 1e-3), (25.0, 1e-5)):
        F, g = cb.value_and_grad(Z, [ls] * 3, 1.0, s2, want_gz=False)
        out.append((F, g["ls"].numpy().tolist(), cb.n_guard_reruns + cb.n_direct_whitened, cb.last_estimate, cb.n_collectives,
                    cb._prefer_whitened, cb.guard.ratio))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_guard_decisions_are_identical_on_every_rank():
    """The estimate is a function of the replicated tail and the all-reduced statistics: both ranks must compute the same bits, repeat
    the same evaluations (a rank deciding alone would wait in its all-reduce for ever) and issue the same number of collectives."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = outs[0][1], outs[1][1]
    assert a == b                                        # F, gradients, whitened counts, estimates, ratios, collective counts: bit for bit
    assert a[0][2] == 0 and a[1][2] == 1 and a[-1][2] >= 2


def test_whitened_order_in_the_streaming_layout_hands_T_to_pass_2_on_the_cpu_double():
    """Host logic of the large-shard whitened order: above `whitened_rows_min_work` local rows x inducing points the
    statistics come from suffstats_whitened_rows, T = K'_fu L^-T stays in the bound's K'_fu block and pass 2 receives exactly that
    block (the double asserts its contents); value-only evaluations keep nothing; below the threshold the chunked routine runs.  F and
    the gradients agree with the chunked order to rounding (the factored adjoint is re-expanded in the double)."""
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    X, y, Z = _guard_problem()
    theta = ([1.5] * 3, 1.2, 0.05)
    eng = FactoredOracleEngine()
    cb = pkg.CollapsedBound(X, y, jitter=1e-6, engine=eng, form="whitened")
    cb.whitened_rows_min_work = 0
    F, _ = cb.value(Z, *theta)
    assert eng.calls["suffstats_whitened_rows"] == 1 and eng.calls["suffstats_whitened"] == 0 and cb._kfu is None
    F2, g = cb.value_and_grad(Z, *theta, want_gz=True)
    assert eng.calls["suffstats_whitened_rows"] == 2 and eng.calls["suffstats_bwd_factored"] == 1 and eng.calls["t_handed_over"] == 1
    assert cb._kfu is not None and not bool(torch.isnan(cb._kfu[: X.shape[0] * Z.shape[0]]).any())
    ref_eng = FactoredOracleEngine()
    ref = pkg.CollapsedBound(X, y, jitter=1e-6, engine=ref_eng, form="whitened")   # default threshold: this problem is far below it
    Fr, gr = ref.value_and_grad(Z, *theta, want_gz=True)
    assert ref_eng.calls["suffstats_whitened_rows"] == 0 and ref_eng.calls["suffstats_whitened"] == 1 and ref_eng.calls["t_handed_over"] == 0
    assert F == F2 == Fr
    for k in ("ls", "Z"):
        assert torch.equal(torch.as_tensor(g[k]), torch.as_tensor(gr[k]))
    assert g["sf2"] == gr["sf2"] and g["s2"] == gr["s2"]
    # the guard's fallback takes the same route: a streaming bound that trips hands T over on the repeat
    old = pkg.CollapsedBound.WHITENED_MAX_WORK
    pkg.CollapsedBound.WHITENED_MAX_WORK = 0
    try:
        e2 = FactoredOracleEngine()
        c2 = pkg.CollapsedBound(X, y, jitter=1e-6, engine=e2)
        c2.whitened_rows_min_work = 0
        c2.value_and_grad(Z, [25.0] * 3, 1.0, 1e-5, want_gz=False)
        assert c2.n_guard_reruns == 1 and e2.calls["suffstats_whitened_rows"] == 1 and e2.calls["t_handed_over"] == 1
    finally:
        pkg.CollapsedBound.WHITENED_MAX_WORK = old


def _rows_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    X, y, Z = _guard_problem()
    lo, hi = pkg.shard_rows(X.shape[0], rank, world)
    eng = FactoredOracleEngine()
    cb = pkg.CollapsedBound(X[lo:hi], y[lo:hi], jitter=1e-6, engine=eng, form="whitened")
    # rank 0 takes the streaming layout, rank 1 the chunked routine: the all-reduced statistics mean the same either way
    cb.whitened_rows_min_work = 0 if rank == 0 else 1 << 40
    F, g = cb.value_and_grad(Z, [1.5] * 3, 1.2, 0.05, want_gz=True)
    q.put((rank, F, g["ls"].numpy().tolist(), torch.as_tensor(g["Z"]).numpy().tolist(), eng.calls["suffstats_whitened_rows"], eng.calls["t_handed_over"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_whitened_layouts_may_differ_between_ranks():
    """A rank picks the whitened layout from its LOCAL shard size: ranks with different choices still all-reduce the same quantities
    ([W | u | yy | kappa], then the packed gradients) and agree with the one-process evaluation."""
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[0][1:4] == outs[1][1:4]                                   # F and the gradients: the same bits on both ranks
    assert outs[0][4:] == (1, 1) and outs[1][4:] == (0, 0)
    X, y, Z = _guard_problem()
    one = pkg.CollapsedBound(X, y, jitter=1e-6, engine=FactoredOracleEngine(), form="whitened")
    F1, g1 = one.value_and_grad(Z, [1.5] * 3, 1.2, 0.05, want_gz=True)
    assert abs(outs[0][1] - F1) < 1e-9 * abs(F1)
    assert np.max(np.abs(np.asarray(outs[0][2]) - g1["ls"].numpy())) < 1e-8 * np.max(np.abs(g1["ls"].numpy()))


def test_three_tier_guard_on_the_cpu_double():
    """Round 4's second tier: a streaming evaluation whose estimate exceeds the tolerance is repeated in the EXTENDED streaming order
    (engine.suffstats_extended) while the estimate is within `extended_range` (values) / `extended_grad_range` (value + gradient: pass 2
    takes the explicit Phibar on the K'_fu it kept) times the tolerance, beyond that in the whitened order; the evaluations that follow
    start in the tier the prediction asks for; an extended evaluation whose own theta turns out to be beyond the range is repeated
    whitened; the episode ends as before."""
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    X, y, Z = _guard_problem()
    old = pkg.CollapsedBound.WHITENED_MAX_WORK
    pkg.CollapsedBound.WHITENED_MAX_WORK = 0
    try:
        eng = FactoredOracleEngine()
        cb = pkg.CollapsedBound(X, y, jitter=1e-6, engine=eng)
        cb.whitened_rows_min_work = 0
        cb.extended_lo = False   # (the gradient range of the leading word alone -- another kernel, d > 8: the trailing-word path has its
        #                           own test, test_guard_logic.py::test_extended_gradient_is_trusted_by_its_own_trailing_word_correction)
        # estimates: 1.9e-9 (inside the gradient range 3e-9), 3.8e-9 (outside it, inside the value range 1.6e-5), 1.6e-4 (outside both), ~1e-12
        gmid, mid, far, benign = ([3.0] * 3, 1.0, 2e-2), ([3.0] * 3, 1.0, 1e-2), ([25.0] * 3, 1.0, 1e-5), ([0.8] * 3, 1.0, 0.3)
        ref = pkg.CollapsedBound(X, y, jitter=1e-6, engine=FactoredOracleEngine(), form="whitened")
        F, g = cb.value_and_grad(Z, *gmid, want_gz=True)                                       # trips -> repeated in the extended order
        assert cb.n_guard_reruns == 1 and cb.n_extended == 1 and eng.calls["suffstats_extended"] == 1 and eng.calls["extended_level_2"] == 1
        assert eng.calls["suffstats_whitened_rows"] == 0 and eng.calls["suffstats_bwd_factored"] == 0 and eng.calls["suffstats_bwd"] == 2
        Fr, gr = ref.value_and_grad(Z, *gmid, want_gz=True)
        # explicit against factored adjoint (the double re-expands the factored one through L and L^-1: rounding of that round trip)
        assert F == Fr and (g["ls"] - gr["ls"]).abs().max() < 1e-7 * gr["ls"].abs().max()
        n_stream = eng.calls["suffstats"]
        cb.value(Z, *mid)                                                                       # the episode continues: extended, directly
        assert cb.n_direct_whitened == 1 and cb.n_extended == 2 and eng.calls["suffstats"] == n_stream
        cb.value_and_grad(Z, *mid, want_gz=False)        # a gradient at an estimate of 3.8e-9: beyond the gradient range -> whitened, T handed over
        assert cb.n_extended == 2 and eng.calls["suffstats_whitened_rows"] == 1 and eng.calls["t_handed_over"] == 1
        cb.value(Z, *mid)
        assert cb.n_extended == 3
        F2, _ = cb.value(Z, *far, raise_on_fail=False)           # starts extended (it cannot know yet); its bound says: beyond the range
        assert cb.n_extended == 4 and cb.n_guard_reruns == 2 and eng.calls["suffstats_whitened_rows"] == 2
        assert F2 == ref.value(Z, *far, raise_on_fail=False)[0]
        cb.value(Z, *far, raise_on_fail=False)                                                  # now predicted: whitened directly
        assert cb.n_extended == 4 and eng.calls["suffstats_whitened_rows"] == 3 and cb.n_guard_reruns == 2
        cb.value(Z, *mid)                                        # whitened (the prediction is still `far`), which re-predicts: mid
        assert eng.calls["suffstats_whitened_rows"] == 4
        cb.value(Z, *mid)
        assert cb.n_extended == 5
        cb.value(Z, *benign)                                                                    # extended once more, then the episode ends
        assert not cb._prefer_whitened
        n_stream = eng.calls["suffstats"]
        cb.value(Z, *benign)
        assert eng.calls["suffstats"] == n_stream + 1 and cb.n_extended == 6
        # form="extended" pins the tier (tools, tests); extended_range = 1 switches it off
        ce = pkg.CollapsedBound(X, y, jitter=1e-6, engine=FactoredOracleEngine(), form="extended")
        assert ce.value(Z, *mid)[0] == ref.value(Z, *mid)[0] and ce.engine.calls["suffstats_extended"] == 1
        e3 = FactoredOracleEngine()
        c3 = pkg.CollapsedBound(X, y, jitter=1e-6, engine=e3)
        c3.whitened_rows_min_work = 0
        c3.extended_range = 1.0
        c3.value(Z, *mid)
        assert e3.calls.get("suffstats_extended", 0) == 0 and e3.calls["suffstats_whitened_rows"] == 1
    finally:
        pkg.CollapsedBound.WHITENED_MAX_WORK = old


def _tier_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ggp_amd as pkg
    from fake_engine import FactoredOracleEngine
    X, y, Z = _guard_problem()
    pkg.CollapsedBound.WHITENED_MAX_WORK = 0
    lo, hi = pkg.shard_rows(X.shape[0], rank, world)
    eng = FactoredOracleEngine()
    cb = pkg.CollapsedBound(X[lo:hi], y[lo:hi], jitter=1e-6, engine=eng)
    cb.whitened_rows_min_work = 0
    out = []
    for ls, s2 in ((0.8, 0.3), (3.0, 2e-2), (3.0, 2e-2), (25.0, 1e-5), (25.0, 1e-5), (3.0, 2e-2), (0.8, 0.3), (0.8, 0.3)):   # (far: beyond the gradient range of the double-double Phibar too)
        F, g = cb.value_and_grad(Z, [ls] * 3, 1.0, s2, want_gz=False)
        out.append((F, g["ls"].numpy().tolist(), cb.n_guard_reruns, cb.n_direct_whitened, cb.n_extended, cb.n_collectives,
                    cb._prefer_whitened, cb.guard.predicted))
    q.put((rank, out, eng.calls.get("suffstats_extended", 0), eng.calls["suffstats_whitened_rows"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_three_tier_decisions_are_identical_on_every_rank():
    """The tier of every evaluation (streaming, extended, whitened), the repeats and the collective counts are functions of replicated
    numbers: two ranks walking benign -> mid -> far -> mid -> benign theta take them in lockstep and return the same bits."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tier_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[0][1] == outs[1][1] and outs[0][2:] == outs[1][2:]
    a = outs[0][1]
    assert a[0][2:5] == (0, 0, 0)                      # benign: streamed
    assert a[1][4] == 1 and a[2][4] == 2               # mid: the trip's repeat and the next evaluation ran in the extended order
    assert a[4][4] == a[3][4] and outs[0][3] >= 2      # far: whitened (the extended attempt of the first one was repeated)
    assert not a[-1][6]                                # the episode has ended


def test_reciprocal_lengthscale_array_is_cached_by_value_not_by_identity():
    """HipEngine._inv_ls keeps the last C array (an evaluation passes the same list to four or five library calls): a list changed IN PLACE
    between two evaluations -- what a sampler's loop does -- must give a new array, a tensor argument is never cached, and the composite
    kernel's parameter block keeps its own length check.  (The method needs no device: it is called on a bare object here.)"""
    from ggp_amd import engine as E

    class Bare:
        pass

    b = Bare()
    ls = [2.0, 4.0]
    a1 = E.HipEngine._inv_ls(b, ls, 2)
    assert list(a1) == [0.5, 0.25]
    assert E.HipEngine._inv_ls(b, ls, 2) is a1                       # same values: the kept array
    assert E.HipEngine._inv_ls(b, [2.0, 4.0], 2) is a1               # another list object with the same values
    ls[1] = 8.0                                                      # changed in place
    a2 = E.HipEngine._inv_ls(b, ls, 2)
    assert a2 is not a1 and list(a2) == [0.5, 0.125]
    assert list(E.HipEngine._inv_ls(b, [2.0], 3)) == [0.5, 0.5, 0.5]  # one lengthscale for every dimension
    assert list(E.HipEngine._inv_ls(b, [2.0, 8.0], 2, "matern52")) == [0.5, 0.125]
    t = torch.tensor([2.0, 4.0], dtype=torch.float64)
    a3 = E.HipEngine._inv_ls(b, t, 2)
    assert list(a3) == [0.5, 0.25] and E.HipEngine._inv_ls(b, t, 2) is not a3
    with pytest.raises(ValueError):
        E.HipEngine._inv_ls(b, [1.0, 2.0, 3.0], 2)
    with pytest.raises(ValueError):
        E.HipEngine._inv_ls(b, [1.0, 2.0], 2, "composite")
