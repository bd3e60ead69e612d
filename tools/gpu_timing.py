"""Phase timings of the 3D P1 Laplace path on one GPU (development aid; bench.py is the contract)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=100)
    ap.add_argument("--target", type=int, default=27)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--restart", type=int, default=100)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--noprec", action="store_true")
    ap.add_argument("--coarse", type=float, default=-1.0, help="two-level: lattice cells target (0 = default)")
    a = ap.parse_args()
    t0 = time.time()
    m = capi.structured_mesh(3, 1, a.M)
    t_mesh = time.time() - t0
    c = capi.Context(device=0)
    t0 = time.time()
    c.mesh_set_dict(m)
    t_up = time.time() - t0
    c.timing_enable(True)
    out = []
    for rep in range(a.reps):
        c.timing_reset()
        c.sync()
        w = {}
        t = time.perf_counter()
        nnz = c.pattern_build(1, capi.BLOCK_SCALAR); c.sync()
        w["symbolic"] = time.perf_counter() - t; t = time.perf_counter()
        c.assemble(capi.FORM_LAPLACE); c.sync()
        w["assemble"] = time.perf_counter() - t; t = time.perf_counter()
        c.assemble_rhs([1.0]); c.sync()
        w["rhs"] = time.perf_counter() - t; t = time.perf_counter()
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0]); c.sync()
        w["dirichlet"] = time.perf_counter() - t; t = time.perf_counter()
        if not a.noprec:
            c.schwarz_set_target(a.target, a.scale)
            if a.coarse >= 0:
                c.schwarz_set_coarse(a.coarse)
                c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)
            else:
                c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
            c.sync()
        w["schwarz_setup"] = time.perf_counter() - t; t = time.perf_counter()
        _, its, rel = c.gmres(None, rtol=a.rtol, max_it=2000, restart=a.restart, use_prec=not a.noprec, want_x=False)
        c.sync()
        w["gmres"] = time.perf_counter() - t
        tm = c.timing_get()
        total = sum(w.values())
        rec = dict(rep=rep, M=a.M, dofs=m["gid_uni"].shape[0], nnz=nnz, its=its, relres=rel,
                   wall_ms={k: round(v * 1e3, 3) for k, v in w.items()}, total_ms=round(total * 1e3, 3),
                   dof_per_s=m["gid_uni"].shape[0] / total,
                   dev_ms={k: (round(v[0], 3), v[1]) for k, v in tm.items()},
                   schwarz=None if a.noprec else c.schwarz_info())
        out.append(rec)
        print(json.dumps(rec), flush=True)
    # kernel micro-timings on resident data
    for kind in (1, 0):
        c.set_option("asm_kind", kind)
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE); c.sync()
        c.timing_reset()
        for _ in range(5):
            c.assemble(capi.FORM_LAPLACE)
        c.sync()
        tk = c.timing_get()["assemble"]
        ab = 4.0 * m["conn"].size + 24.0 * m["xyz"].shape[0] + 12.0 * nnz + 4.0 * (m["gid_uni"].shape[0] + 1)
        print(json.dumps({"asm_kind": kind, "ms": tk[0] / tk[1], "GBs": ab / (tk[0] / tk[1]) / 1e6}), flush=True)
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    nr = m["gid_uni"].shape[0]
    for kind in (1, 0):
        c.set_option("spmv_kind", kind)
        c.spmv_device(5); c.sync()
        c.timing_reset()
        c.spmv_device(50); c.sync()
        tk = c.timing_get()["spmv"]
        print(json.dumps({"spmv_kind": kind, "ms": tk[0] / tk[1], "GBs": (12 * nnz + 20 * nr) / (tk[0] / tk[1]) / 1e6}), flush=True)
    if not a.noprec:
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)     # pattern_build above dropped the operator
    c.timing_reset()
    c.spmv_device(50); c.sync()
    if not a.noprec:
        c.schwarz_apply_device(50); c.sync()
    tm = c.timing_get()
    nr = m["gid_uni"].shape[0]
    sp_bytes = 12 * nnz + 20 * nr
    sp_ms = tm["spmv"][0] / max(1, tm["spmv"][1])
    res = dict(spmv_ms=sp_ms, spmv_GBs=sp_bytes / sp_ms / 1e6, mesh_s=t_mesh, upload_s=t_up)
    if not a.noprec:
        info = c.schwarz_info()
        ap_ms = tm["schwarz_apply"][0] / max(1, tm["schwarz_apply"][1])
        res.update(apply_ms=ap_ms, apply_GBs=(info["inverse_bytes"] + 16 * nr) / ap_ms / 1e6)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
