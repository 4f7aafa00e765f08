"""Build libdram_hip.so (hand-written HIP kernels, gfx950) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  Objects go to ``build/`` and the
shared library to ``lib/libdram_hip.so`` inside this package directory, so the
built library travels with the source tree (nothing is installed anywhere).
"""
from __future__ import annotations

import hashlib
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


def _read(path: str) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def abi_hash() -> str:
    """Fingerprint of the C ABI = sha1 of include/dram_hip.h.  Compiled into the library
    (dram_abi_hash()) and compared by _lib.load(): a library built from another header is
    refused instead of being called with shifted arguments."""
    return hashlib.sha1(_read(os.path.join(INCLUDE, "dram_hip.h"))).hexdigest()[:16]


def source_hash() -> str:
    """Fingerprint of everything the library is compiled from (kernels + headers)."""
    h = hashlib.sha1()
    for f in [os.path.join(INCLUDE, "dram_hip.h")] + [os.path.join(CSRC, s) for s in sorted(os.listdir(CSRC))
                                                      if s.endswith((".hip", ".h"))]:
        h.update(os.path.basename(f).encode())
        h.update(_read(f))
    return h.hexdigest()[:16]


def _flags():
    # DRAM_EXTRA_HIPCC_FLAGS: ablation builds of the tuning tools (e.g. -DDRAM_BF16_ABL=3); part of the object keys
    # (honoured under DRAM_TUNING=1 only: the ablation macros produce garbage results by design)
    extra = os.environ.get("DRAM_EXTRA_HIPCC_FLAGS", "").split() if os.environ.get("DRAM_TUNING", "0") == "1" else []
    return ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", f'-DDRAM_ABI_HASH="{abi_hash()}"',
            "-I", INCLUDE, "-I", CSRC] + extra


def have_hipcc() -> bool:
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Incremental by CONTENT (not mtime: the tree is copied to the GPU box): an object is
    rebuilt when the sha1 of its source + every header + the flags differs from the key
    stored beside it, the library is relinked when any object key changed."""
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(BUILD, ".lock"), "w") as lock:     # ranks of one job may all get here at once
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(force, verbose)


def _build_locked(force: bool, verbose: bool) -> str:
    hipcc = _hipcc()
    flags = _flags()
    hdrs = b"".join(_read(os.path.join(d, f)) for d in (INCLUDE, CSRC) for f in sorted(os.listdir(d))
                    if f.endswith(".h"))
    jobs, objs, keys = [], [], []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src[:-4] + ".o")
        key = hashlib.sha1(_read(s) + hdrs + " ".join(flags).encode()).hexdigest()
        kf = o + ".key"
        objs.append(o)
        keys.append(key)
        old = _read(kf).decode() if os.path.exists(kf) else ""
        if force or not os.path.exists(o) or old != key:
            jobs.append((key, kf, [hipcc] + flags + ["-c", s, "-o", o]))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    def compile_one(job):
        key, kf, cmd = job
        if os.path.exists(kf):
            os.remove(kf)
        run(cmd)
        with open(kf, "w") as f:
            f.write(key)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    link_key = hashlib.sha1("".join(keys).encode()).hexdigest()
    lk = LIB_PATH + ".key"
    old = _read(lk).decode() if os.path.exists(lk) else ""
    if force or bool(jobs) or not os.path.exists(LIB_PATH) or old != link_key:
        if os.path.exists(lk):
            os.remove(lk)
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs)
        with open(lk, "w") as f:
            f.write(link_key)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
