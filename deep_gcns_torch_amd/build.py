"""Build libdgcn.so (hipcc, gfx950 only) in-tree.

    python -m deep_gcns_torch_amd.build [--force] [--verbose]

Each csrc/*.hip is compiled to an object (in parallel, only when stale) and the objects are
linked into csrc/libdgcn.so.  hipcc cross-compiles without a GPU, so this also runs in the
CPU-only container; the built .so is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
OBJ = CSRC / "_obj"
LIB = CSRC / "libdgcn.so"
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _newest_header_mtime() -> float:
    hdrs = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    return max(h.stat().st_mtime for h in hdrs)


def build_debug_ids() -> Path:
    """libdgcn_dbg.so: the library with -DDGCN_DEBUG_IDS (investigation builds: out-of-range arg-max ids are counted
    and recorded, dgcn_debug_bad_ids).  Loaded through DGCN_LIB_PATH; never the shipped library."""
    out = CSRC / "libdgcn_dbg.so"
    srcs = sorted(CSRC.glob("*.hip"))
    objs = []
    dbg = CSRC / "_obj_dbg"
    dbg.mkdir(exist_ok=True)
    for src in srcs:
        obj = dbg / (src.stem + ".o")
        reuse = OBJ / (src.stem + ".o")
        if src.name != "gen_aggr_bwd.hip" and reuse.exists():
            objs.append(reuse)
            continue
        cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-DDGCN_DEBUG_IDS", f"-I{INCLUDE}",
               f"-I{CSRC}", "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        objs.append(obj)
    r = subprocess.run([hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(out)] + [str(o) for o in objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


def _compile_one(src: Path, force: bool, verbose: bool) -> Path:
    obj = OBJ / (src.stem + ".o")
    if (not force and obj.exists() and obj.stat().st_mtime > src.stat().st_mtime
            and obj.stat().st_mtime > _newest_header_mtime()):
        return obj
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
           "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}",
           "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError(f"no HIP sources under {CSRC}")
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, force, verbose), srcs))
    stale = force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs)
    if stale:
        # No rpath to /opt/rocm: at run time the HIP runtime must be the one the host process
        # (PyTorch-ROCm) already loaded, so that streams and device pointers are shared.
        cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__" and "--debug-ids" in sys.argv:
    build()
    print(build_debug_ids())
    sys.exit(0)
if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
