"""Builds libdevo_hip.so (the C-ABI core, include/devo_hip.h) in-tree with hipcc for gfx950.

    python -m devo_amd.build            # rebuild if sources are newer than the library

The library is cross-compiled without a GPU (hipcc --offload-arch=gfx950) and travels to the GPU box
as a built artefact inside the repo snapshot; nothing is JIT-compiled at import time.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdevo_hip.so")
SOURCES = ["lie.hip", "corr.hip", "ba.hip", "update.hip", "linear.hip", "linear_dw.hip", "events.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "-fno-slp-vectorize"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libdevo_hip.so)")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "devo_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
