"""ctypes loader/builder for oracle/oracle.c (the C/OpenMP CPU restatement).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__ and bench.py's cpu_baseline leg import this."""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "oracle.c")
LIB = os.path.join(HERE, "_build", "liboracle.so")
_lib = None


def build(verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found")
    cmd = [gcc, "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, capture_output=not verbose)
    return LIB


def available() -> bool:
    try:
        _load()
        return True
    except Exception:
        return False


def _load():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.oracle_laplace3d.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, dp, ip, dp, ip,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64), ip, dp, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int32), dp, dp]
        L.oracle_laplace3d.restype = C.c_int
        _lib = L
    return _lib


def run_laplace3d(M, target=27, rtol=1e-8, restart=100, max_it=2000, use_prec=True, want_system=False, threads=0):
    """Whole laplace driver on the CPU.  Returns timings (s), iteration count, threads, and -- with
    want_system -- the Dirichlet-modified CSR matrix, rhs and solution."""
    L = _load()
    n = (M + 1) ** 3
    times = np.zeros(8)
    its, thr, maxn = C.c_int(), C.c_int(int(threads)), C.c_int()
    rel = C.c_double()
    nnz, nsub = C.c_int64(), C.c_int64()
    dp = C.POINTER(C.c_double)
    x = np.zeros(n) if want_system else None
    rowptr = col = val = rhs = None
    if want_system:
        cap = 15 * n
        rowptr = np.zeros(n + 1, dtype=np.int64)
        col = np.zeros(cap, dtype=np.int32)
        val = np.zeros(cap)
        rhs = np.zeros(n)

    def p(a, ty):
        return None if a is None else a.ctypes.data_as(ty)

    L.oracle_laplace3d(M, target, rtol, restart, max_it, int(use_prec), p(times, dp), C.byref(its), C.byref(rel),
                       C.byref(thr), C.byref(nnz), C.byref(nsub), C.byref(maxn), p(x, dp), p(rowptr, C.POINTER(C.c_int64)),
                       p(col, C.POINTER(C.c_int32)), p(val, dp), p(rhs, dp))
    out = dict(dofs=n, its=its.value, relres=rel.value, threads=thr.value, nnz=nnz.value, n_subdomains=nsub.value,
               max_size=maxn.value, t_mesh=times[0], t_assemble=times[1], t_bc=times[2], t_prec=times[3],
               t_gmres=times[4], t_gmres_operator=times[5], t_gmres_ortho=times[6], seconds=float(times[1] + times[2] + times[3] + times[4]))
    if want_system:
        out.update(x=x, rowptr=rowptr, col=col[:nnz.value], val=val[:nnz.value], rhs=rhs)
    return out
