"""Builds libsgp_hip.so (the C-ABI library declared in include/sgp.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
.so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libsgp_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
SOURCES = ["sgp_ctx.hip", "sgp_suffstats_fwd.hip", "sgp_suffstats_i8.hip", "sgp_suffstats_bwd.hip", "sgp_suffstats_bwd_lo.hip", "sgp_dense.hip", "sgp_tail.hip", "sgp_svgp.hip", "sgp_composite.hip", "sgp_small.hip"]
VERSION_SCRIPT = "libsgp.map"
PUBLIC_HEADER = os.path.join("..", "..", "include", "sgp.h")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]  # only the SGP_API symbols of include/sgp.h leave the library


def _headers():
    """Every header a translation unit can reach: all of csrc/*.hpp (globbed, so a new #include cannot be forgotten --
    sgp_nuts.hpp once was and the sampler could be edited without a rebuild) plus the public C header."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [PUBLIC_HEADER]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the sparse-GP core has no CPU fallback and cannot be built without ROCm")


def _digest() -> str:
    h = hashlib.sha256()
    h.update((" ".join(FLAGS) + " " + os.environ.get("SGP_EXTRA_HIPCC_FLAGS", "")).encode())
    for f in SOURCES + _headers() + [VERSION_SCRIPT]:
        h.update(os.path.basename(f).encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 and link them into csrc/libsgp_hip.so."""
    stamp = LIB_PATH + ".sha256"
    digest = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == digest:
                return LIB_PATH
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=" + ARCH] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        cmd += os.environ.get("SGP_EXTRA_HIPCC_FLAGS", "").split()  # A/B builds (tools/ab_build.sh); empty for the product
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, VERSION_SCRIPT),
           "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(stamp, "w") as fh:
        fh.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
