"""CPU only: the host-compilable headers of the product (csrc/sgp_nuts.hpp -- the sampler the persistent kernel runs --
and csrc/sgp_stream.hpp -- the split / taper plan of both streaming passes) built with AddressSanitizer + UBSan and run here.
SURVEY section 5: sanitizers run on the CPU build only; the GPU pool refuses sanitizer builds, so this file is listed in
.gpurunignore and never travels to a GPU box (nothing in it needs one)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "generalised-gaussian-processes_amd", "csrc")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]


def _env():
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    return {k: v for k, v in env.items() if not k.startswith("SGP_")}


def test_sampler_header_is_clean_under_asan_ubsan(tmp_path):
    """The tree stack, the adaptation windows and the draw buffers of sgp_nuts.hpp are indexed by hand -- on the GPU an
    out-of-bounds write would corrupt LDS silently.  Gaussians in 1 ... NUTS_MAXD dimensions, a zero-density wall (divergences),
    tree-depth limits 1 and 10 (tests/native/nuts_host.cpp, -DNUTS_HOST_MAIN)."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "nuts_asan")
    subprocess.run([gxx, "-O1", "-std=c++17", "-DNUTS_HOST_MAIN"] + SAN + ["-I", INC, "-o", exe,
                    os.path.join(ROOT, "tests", "native", "nuts_host.cpp")], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0 and "sanitized sampler ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_stream_plan_is_clean_under_asan_ubsan(tmp_path):
    """make_stream_plan() fills its split tables by hand-written index arithmetic for every shard / inducing-set size
    (tests/native/plan_check.cpp walks them): host-only build of the HIP header under the sanitizers."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "plan_check_asan")
    subprocess.run([hipcc, "-x", "hip", "--cuda-host-only", "-std=c++17", "-O1", "-w"] + SAN + ["-I", INC, "-o", exe,
                    os.path.join(ROOT, "tests", "native", "plan_check.cpp")], check=True, timeout=600)
    for knobs in ({}, {"SGP_SYRK_NSPLIT": "24", "SGP_KBAR_NSPLIT": "40"}):
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(_env(), **knobs))
        assert out.returncode == 0 and "0 failures" in out.stdout, (knobs, out.stdout[-2000:], out.stderr[-3000:])


def _build_chain_check(tmp_path, inc, name="chain_items_asan"):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / name)
    subprocess.run([gxx, "-O1", "-std=c++17"] + SAN + ["-I", inc, "-o", exe, os.path.join(ROOT, "tests", "native", "chain_items_check.cpp")],
                   check=True, timeout=300)
    return exe


def test_chain_cholesky_items_progress_and_hazards_from_the_access_table(tmp_path):
    """csrc/sgp_potrf_items.hpp holds the ACCESS TABLE of the chain-workgroup Cholesky: per work item and per role of the chain workgroup
    what is waited for, read, written and raised (the kernel's scratch flags are laid out by the same header, and tools/potrf_trace_check.py
    holds the kernel's logged waits / raises against it on the GPU).  tests/native/chain_items_check.cpp derives from it, for block counts
    2 ... 64, with / without inverse and right-hand side, 1 ... 255 workgroups, light / ordinary hand-overs: every item dealt once;
    progress under the static deal and under a ticketed claim with adversarially delayed workgroups; flag counts; and -- on the
    happens-before order, i.e. for every schedule and every assignment of items to workgroups -- one writer per location, every foreign
    read behind a valid publication, no foreign read of original contents, nothing touched before its publication.  The item lists of
    round 5 before 59da5e9 (FUSED_S reading tile (c+2, c) in place: factors off by 3e-3 with info = 0) must be flagged,
    and a versioned replay must show that they only fail when a workgroup is held back."""
    exe = _build_chain_check(tmp_path, INC)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0 and " 0 failures" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "legacy lists flagged in 20 of 20 cases (replay: 0 wrong reads with everything starting together" in r.stdout, r.stdout[-600:]


MUTATIONS = [
    # (what is broken, text in the header, its replacement, a finding that must be printed)
    ("a tile item does not wait for tile (c, p) of its own column's row", "        v.wait(ChFlag{CF_READY, c, p}, 1);\n", "", "H2"),
    ("a non-local FUSED_S reads X behind the LIGHT flag", "        else v.wait(ChFlag{CF_READY, jn, c}, 1);",
     "        else v.wait(ChFlag{CF_XREADY_L, c, 0}, 1);", "H2"),
    ("the S-waves read UD without waiting for its flag", "    if (j >= 1) { v.wait(ChFlag{CF_PRED, j + 1, 0}, 1); v.read(ChLoc{CL_DPRE, j + 1, 0}, false); }",
     "    if (j >= 1) { v.read(ChLoc{CL_DPRE, j + 1, 0}, false); }", "H2"),
    ("FUSED_S reads the tile's original entries in place (the round-5 race)", "      if (fs && o.legacy_fused_s) v.read(",
     "      if (fs) v.read(", "H3"),
    ("a block of L^-1 waits for a LATER block of its column (deadlock)", "        v.wait(ChFlag{CF_IREADY, p, c}, 1);",
     "        v.wait(ChFlag{CF_IREADY, p + 1 <= i ? p + 1 : p, c}, 1);", "stuck"),
    ("the ticketed claim starts a rest item without the critical tickets of the columns before it (starvation)",
     "      if (t < 0) { st.have_pending = 0; return st.pending; }", "      if (t < 0 || true) { st.have_pending = 0; return st.pending; }", "ticketed claim deadlocks"),
    ("a helper runs ANY critical ticket it draws, also its own column's (round 6's first version: a look, then a draw -- hung on the GPU)",
     "      const int t = a.take_below(1, ch_crit_needed(st.pending, nb));", "      const int t = a.take_below(1, ch_crit_needed(st.pending, nb) + 1);",
     "ticketed claim deadlocks"),
    ("EARLY_D publishes twice", "      else { v.write(ChLoc{CL_DPE, i, 0}); v.raise(ChFlag{CF_PREDE, i, 0}, false); }",
     "      else { v.write(ChLoc{CL_DPE, i, 0}); v.raise(ChFlag{CF_PREDE, i, 0}, false); v.raise(ChFlag{CF_PREDE, i, 0}, false); }", "raised 2 times"),
]


@pytest.mark.parametrize("what,old,new,finding", MUTATIONS, ids=[m[0].split(" (")[0][:50] for m in MUTATIONS])
def test_chain_cholesky_checker_flags_a_broken_table(tmp_path, what, old, new, finding):
    """The checker's own sensitivity: one entry of the access table broken at a time (a copy of the header, textually) -- each must be found."""
    src = open(os.path.join(INC, "sgp_potrf_items.hpp")).read()
    assert src.count(old) == 1, (what, src.count(old))
    (tmp_path / "sgp_potrf_items.hpp").write_text(src.replace(old, new))
    exe = _build_chain_check(tmp_path, str(tmp_path), "chain_items_mutant")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(_env(), CHAIN_CHECK_QUICK="1"))
    assert r.returncode != 0 and finding in r.stdout, (what, r.stdout[-1500:])
