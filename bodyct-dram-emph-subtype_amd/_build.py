"""Build libdram_hip.so (hand-written HIP kernels, gfx950) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  Objects go to ``build/`` and the
shared library to ``lib/libdram_hip.so`` inside this package directory, so the
built library travels with the source tree (nothing is installed anywhere).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_DIR, "include")
BUILD = os.path.join(PKG_DIR, "build")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdram_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libdram_hip.so")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header() -> float:
    hs = [os.path.join(INCLUDE, "dram_hip.h"), os.path.join(CSRC, "common.h")]
    return max(os.path.getmtime(h) for h in hs)


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdr = _newest_header()
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr):
            jobs.append([hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-I", INCLUDE, "-I", CSRC,
                         "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    need_link = force or bool(jobs) or not os.path.exists(LIB_PATH) or \
        any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs)
    if need_link:
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
