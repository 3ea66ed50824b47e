"""Builds libdevo_hip.so (the C-ABI core, include/devo_hip.h) in-tree with hipcc for gfx950.

    python -m devo_amd.build            # rebuild if sources are newer than the library

The library is cross-compiled without a GPU (hipcc --offload-arch=gfx950) and travels to the GPU box
as a built artefact inside the repo snapshot; nothing is JIT-compiled at import time.

build_binding(): devo_amd/_C*.so — csrc/bind.cpp, the compiled (pybind11 / torch::Tensor) binding with the reference's three module
interfaces + torch.ops.devo_hip, host-only C++ (g++ against torch's headers), linked against libdevo_hip.so ($ORIGIN/lib).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdevo_hip.so")
SOURCES = ["lie.hip", "corr.hip", "ba.hip", "update.hip", "linear.hip", "linear_dw.hip", "mlp2.hip", "gemm_rs.hip", "events.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "-fno-slp-vectorize"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libdevo_hip.so)")


HEADER = os.path.join(os.path.dirname(HERE), "include", "devo_hip.h")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [HEADER]
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


def binding_path():
    import sysconfig
    return os.path.join(HERE, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_binding(force=False, verbose=True):
    """devo_amd/_C*.so from csrc/bind.cpp (needs libdevo_hip.so: build_lib() first).  ~40 s of g++ on torch/extension.h."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as C
    out = binding_path()
    src = os.path.join(CSRC, "bind.cpp")
    lib = build_lib(verbose=verbose)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(p) for p in (src, HEADER, lib)):
        return out
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no C++ compiler for the compiled binding (the ctypes binding needs none)")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", "-DTORCH_EXTENSION_NAME=_C",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += ["-I" + p for p in C.include_paths()] + ["-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]]
    cmd += [src, "-o", out, "-L" + LIBDIR, "-ldevo_hip", "-Wl,-rpath,$ORIGIN/lib", "-L" + tlib, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip", "-ltorch_hip",
            "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB)
    if "--no-binding" not in sys.argv:
        print(build_binding(force="--force" in sys.argv))
