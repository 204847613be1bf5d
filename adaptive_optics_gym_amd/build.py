"""Build ``libaogym.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One translation unit per kernel family, compiled in parallel: ``aogym.hip`` (C-ABI core: handle, tables, step / reset), ``atmosphere.hip``
(dynamic atmosphere), ``screens.hip`` (screen synthesis), ``shack.hip`` (Shack-Hartmann chain), ``focal.hip`` (K4), ``actor.hip`` (policy query)
and ``fused_inst.hip`` once per padded mode count (the fused-kernel template instantiations).

The library is tied to its sources: ``source_id()`` is a SHA-256 over ``csrc/*.{h,hip}`` + ``include/aogym.h``; it is compiled into the core
unit (``aog_build_id()``), ``needs_build()`` compares it with the id the existing library reports (not file times), every object carries the
hash of what it was compiled from (``build/<name>.o.id``) so an edit recompiles only the units it touches, and ``_lib.load()`` refuses a
library whose id differs from the sources beside it."""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
ABI_HEADER = os.path.join(HERE, "..", "include", "aogym.h")
OUT = os.path.join(HERE, "libaogym.so")
APADS = (16, 32, 64, 128)
UNITS = ("aogym", "atmosphere", "screens", "shack", "focal", "actor")
# -fno-slp-vectorize: no kernel gets scalar fp32 pairs re-packed into v_pk_* behind its back (packed-FMA trap)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-fno-slp-vectorize"]
# headers each unit includes besides the shared ones (an edit of a family header recompiles that family only)
SHARED = ("aogym_internal.h", "host_common.h", "k_common.h")
FAMILY = {"aogym": ("k_pack.h", "k_step.h"), "atmosphere": ("k_pack.h", "k_extrude.h", "k_extrude_i8.h"), "screens": ("k_fft.h", "k_screens.h"),
          "shack": ("k_fft.h", "k_shack.h"), "focal": ("k_focal.h",), "actor": ("k_actor.h",), "fused_inst": ("k_fused.h",)}


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _sha(paths, extra="") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_files():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip"))) + [ABI_HEADER]


def source_id() -> str:
    """Identity of the sources the library must have been built from (first 32 hex digits of their SHA-256)."""
    return _sha(source_files())[:32]


def library_id(path: str = OUT):
    """The build id compiled into an existing library (``aog_build_id()``), read from the file without loading it; None if absent."""
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    i = data.find(b"AOG_BUILD_ID=")
    if i < 0:
        return None
    j = data.find(b"\0", i)
    return data[i + 13:j].decode("ascii", "replace")


def needs_build() -> bool:
    return library_id() != source_id()


def _unit_deps(unit: str):
    fam = [os.path.join(CSRC, h) for h in SHARED + FAMILY[unit] if os.path.exists(os.path.join(CSRC, h))]
    return [os.path.join(CSRC, unit + ".hip")] + fam + [ABI_HEADER]


def build(force: bool = False, verbose: bool = True, fast: bool = False, dev: bool = False) -> str:
    """fast=True builds only the 8-table kernels (developer iteration); dev=True (or AOG_DEV=1 in the environment at build time)
    compiles the developer switches in (-DAOG_DEV: placement / skew overrides read from the environment, timing read-outs); the
    default product build has none of them.  Such builds carry a build id that never matches ``source_id()``'s plain form: the flags
    are part of it."""
    extra = ["-DAOG_FAST_BUILD"] if fast else []
    if dev or os.environ.get("AOG_DEV") == "1":
        extra.append("-DAOG_DEV")
    sid = source_id()
    if not force and not extra and library_id() == sid:
        return OUT
    hipcc = hipcc_path()
    os.makedirs(OBJ, exist_ok=True)
    flag_key = " ".join(FLAGS + extra)
    jobs = []   # (object, id file, id, command)
    for u in UNITS:
        obj = os.path.join(OBJ, u + ".o")
        uid = _sha(_unit_deps(u), flag_key + (sid if u == "aogym" else ""))
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, u + ".hip"), "-o", obj]
        if u == "aogym":
            cmd.insert(-4, f'-DAOG_BUILD_ID="{sid}{"+" + "+".join(x[2:] for x in extra) if extra else ""}"')
        jobs.append((obj, uid, cmd))
    for a in APADS:
        obj = os.path.join(OBJ, f"fused_apad{a}.o")
        uid = _sha(_unit_deps("fused_inst"), flag_key + f" apad{a}")
        jobs.append((obj, uid, [hipcc, *FLAGS, *extra, f"-DAOG_INST_APAD={a}", "-c", os.path.join(CSRC, "fused_inst.hip"), "-o", obj]))

    def stale(obj, uid):
        try:
            return force or not os.path.exists(obj) or open(obj + ".id").read().strip() != uid
        except OSError:
            return True

    def run(job):
        obj, uid, cmd = job
        if verbose:
            print("[aogym build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        with open(obj + ".id", "w") as f:
            f.write(uid)

    todo = [j for j in jobs if stale(j[0], j[1])]
    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 4))) as ex:
        list(ex.map(run, todo))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *[j[0] for j in jobs], "-lhipfft"]
    if verbose:
        print("[aogym build]", " ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, fast="--fast" in sys.argv, dev="--dev" in sys.argv)
    print(OUT, library_id())
