"""In-tree build of libdeephisto_hip.so (hipcc, gfx950 only).

`python -m deephisto_amd.build` or `deephisto_amd.build.build_library()`.  The
shared object is written next to this file so that it travels with the repo
snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
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
LIB = PKG / "libdeephisto_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libdeephisto_hip.so cannot be built")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = (sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc")) + sorted((PKG.parent / "include").glob("*.h"))
            + [Path(__file__)])
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(srcs) or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
