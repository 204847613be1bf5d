"""Build ``libaogym.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is split into translation units that compile in parallel: ``aogym.hip`` (C-ABI, small kernels) and
``fused_inst.hip`` once per padded mode count (the fused-kernel template instantiations, the expensive part)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
DEPS = [os.path.join(CSRC, f) for f in ("aogym.hip", "fused_inst.hip", "aogym_kernels.h", "aogym_internal.h")] + \
       [os.path.join(HERE, "..", "include", "aogym.h")]
OUT = os.path.join(HERE, "libaogym.so")
APADS = (16, 32, 64, 128)
# -fno-slp-vectorize: no kernel gets scalar fp32 pairs re-packed into v_pk_* behind its back (packed-FMA trap)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-fno-slp-vectorize"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True, fast: bool = False, dev: bool = False) -> str:
    """fast=True builds only the 8-table kernels (developer iteration); dev=True (or AOG_DEV=1 in the environment at build time)
    compiles the developer switches in (-DAOG_DEV: placement / skew overrides read from the environment, timing read-outs); the
    default product build has none of them."""
    if not force and not needs_build():
        return OUT
    hipcc = hipcc_path()
    os.makedirs(OBJ, exist_ok=True)
    extra = ["-DAOG_FAST_BUILD"] if fast else []
    if dev or os.environ.get("AOG_DEV") == "1":
        extra.append("-DAOG_DEV")
    jobs = [([hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, "aogym.hip"), "-o", os.path.join(OBJ, "aogym.o")])]
    for a in APADS:
        jobs.append([hipcc, *FLAGS, *extra, f"-DAOG_INST_APAD={a}", "-c", os.path.join(CSRC, "fused_inst.hip"), "-o",
                     os.path.join(OBJ, f"fused_apad{a}.o")])

    def run(cmd):
        if verbose:
            print("[aogym build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, "aogym.o")] + [os.path.join(OBJ, f"fused_apad{a}.o") for a in APADS]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-lhipfft"])
    return OUT


if __name__ == "__main__":
    build(force=True, fast="--fast" in sys.argv, dev="--dev" in sys.argv)
    print(OUT)
