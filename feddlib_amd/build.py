"""Build libfedd_hip.so (gfx950 only) in-tree with hipcc.  No CPU fallback exists: if hipcc is
missing this raises."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfedd_hip.so")
SOURCES = ["abi.hip", "scan.hip", "symbolic.hip", "assemble.hip", "blocks.hip", "spmv.hip", "schwarz.hip", "schwarz_big.hip", "invert_mfma.hip", "dense.hip", "coarse.hip", "gmres.hip", "multi.hip",
           "halo.hip", "mesh_structured.cpp", "mesh_io.cpp", "mesh_partition.cpp", "fe_tables.cpp"]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-I" + os.path.join(ROCM, "include")]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built and there is no CPU fallback")
    return exe


def _stale(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src, os.path.join(CSRC, "fedd_internal.hpp"), os.path.join(HERE, "..", "include", "fedd_hip.h"),
            os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or _stale(obj, src):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + \
               ["-L" + os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        run(link)
    return LIB


def build_driver(verbose: bool = True, which: str = "laplace") -> str:
    """The example drivers under examples/drivers (test harness, not product: they mirror feddlib/problems/tests/laplace/main.cpp and
    steadyLinElas_Perf/main.cpp), g++ against the C ABI."""
    gxx = shutil.which("g++")
    if gxx is None:
        raise RuntimeError("g++ not found")
    host = os.path.join(HERE, "host")
    ex = os.path.join(HERE, "..", "examples")
    out = os.path.join(ex, "bin", which + "_driver")
    src = os.path.join(ex, "drivers", which + "_main.cpp")
    deps = [src, os.path.join(host, "feddlib", "fedd_facade.hpp"), os.path.join(host, "Teuchos_shim.hpp"), LIB]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [gxx, "-std=c++17", "-O2", "-pthread", "-I", host, src, "-o", out, "-L", LIBDIR, "-lfedd_hip",
           "-Wl,-rpath,$ORIGIN/../../feddlib_amd/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n" + r.stdout + r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_driver())
    print(build_driver(which="linelas"))
