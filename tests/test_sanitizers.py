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


def test_chain_cholesky_items_are_dealt_once_and_cannot_deadlock(tmp_path):
    """csrc/sgp_potrf_items.hpp: the work items of the chain-workgroup Cholesky (tiles, the early / fused partial sums, the blocks of
    L^-1, the right-hand side) and their static deal to the workgroups.  tests/native/chain_items_check.cpp walks block counts 2 ... 64,
    with / without the inverse and the right-hand side, 1 ... 255 workgroups: every item dealt exactly once, and a simulation of the
    dataflow (workgroups take their items strictly in order; an item completes only behind the flags the kernel waits for) completes."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "chain_items_asan")
    subprocess.run([gxx, "-O1", "-std=c++17"] + SAN + ["-I", INC, "-o", exe, os.path.join(ROOT, "tests", "native", "chain_items_check.cpp")],
                   check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0 and " 0 failures" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
