"""Build ``libaogym.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "aogym.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "aogym_kernels.h"), os.path.join(HERE, "..", "include", "aogym.h")]
OUT = os.path.join(HERE, "libaogym.so")


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


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", OUT, SRC]
    if verbose:
        print("[aogym build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=os.path.join(HERE, "csrc"))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
