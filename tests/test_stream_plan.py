"""Host logic of the streaming geometry (csrc/sgp_stream.hpp), compiled for the host only and run on the CPU:
the (tapered) split ranges of pass 1 and pass 2 must tile their chunk / block range exactly once for every shard
size and inducing-set size -- a hole or an overlap there would silently drop or double-count data rows."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_ranges_tile_the_row_range(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "plan_check")
    src = os.path.join(ROOT, "tests", "native", "plan_check.cpp")
    inc = os.path.join(ROOT, "generalised-gaussian-processes_amd", "csrc")
    subprocess.run([hipcc, "-x", "hip", "--cuda-host-only", "-std=c++17", "-O1", "-w", "-I", inc, "-o", exe, src], check=True, timeout=600)
    base = {k: v for k, v in os.environ.items() if not k.startswith("SGP_")}  # the tuning knobs change the plan
    for knobs in ({}, {"SGP_SYRK_TAPER": "0", "SGP_KBAR_TAPER": "0"}, {"SGP_SYRK_NSPLIT": "24", "SGP_KBAR_NSPLIT": "40"},
                  {"SGP_TARGET_WGS": "512"}):
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(base, **knobs))
        assert out.returncode == 0 and "0 failures" in out.stdout, (knobs, out.stdout[-2000:])
