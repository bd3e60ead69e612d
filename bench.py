#!/usr/bin/env python
"""bench.py -- DoF/s of the assemble + solve hot path on the 3D P1-Laplace cube (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full pass of the hot path over the resident mesh: symbolic CSR pattern ->
element assembly (matrix + rhs) -> Dirichlet rows -> one-level Schwarz setup -> right-preconditioned
GMRES(100) to 1e-8 (feddlib/problems/tests/laplace/main.cpp:199-208 with 3D/P1/structured
parameters).  The mesh (connectivity, coordinates, flags) is generated on the host and uploaded
BEFORE the timed region (metric definition).

Workload: N = 1 -> BASELINE.json configs[1]: unit cube, 100^3 cells, 1 030 301 dofs.
          N > 1 -> weak scaling: every GPU owns a 100^3-cell block (1x1x2, 1x2x2, 2x2x2 blocks);
                   the reference generator has no 2/4-rank cube (laplace/main.cpp:132), the slab /
                   pencil splits of the same lattice are this repo's extension.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DECOMP = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=100, help="cells per direction and GPU (M = H/h)")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--restart", type=int, default=100)
    ap.add_argument("--max-it", type=int, default=2000)
    ap.add_argument("--target", type=int, default=27, help="nodes per Schwarz subdomain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-level", action="store_true", help="skip the extra two-level measurement")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="development only: run the N > 1 path with all ranks on GPU 0, gloo + the library's "
                         "host-staged transport (RCCL refuses several ranks on one device); numbers are not a measurement")
    ap.add_argument("--coarse", type=float, default=0.0, help="two-level variant: lattice cells (0 = library default)")
    ap.add_argument("--cpu-cells", type=int, default=0, help="cells per direction of the CPU-baseline sample (0 = auto)")
    return ap.parse_args()


def one_step(c, capi, a, two_level=False):
    c.pattern_build(1, capi.BLOCK_SCALAR)
    c.assemble(capi.FORM_LAPLACE)
    c.assemble_rhs([1.0])
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    c.schwarz_set_target(a.target, 1.0)
    if two_level:
        c.schwarz_set_coarse(a.coarse)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)
    else:
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
    _, its, rel = c.gmres(None, rtol=a.rtol, max_it=a.max_it, restart=a.restart, use_prec=True, want_x=False)
    return its, rel


def cpu_baseline(a):
    """The CPU restatement (oracle, kind 'port') timed on this box's host cores on a bounded
    sample of the same workload: same driver sequence, same preconditioner definition, same
    tolerance, smaller cube.  Reported beside the GPU number; never the thing measured as `value`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import oracle_c
        have_c = oracle_c.available()
    except Exception:
        have_c = False
    if have_c:
        M = a.cpu_cells or a.cells
        # the GPU box gives a 1-GPU job a 16-core share; more OpenMP threads only oversubscribe it
        nthr = int(os.environ.get("FEDD_CPU_THREADS", "0")) or min(16, os.cpu_count() or 1)
        r = oracle_c.run_laplace3d(M, a.target, a.rtol, a.restart, a.max_it, threads=nthr)
        return {"value": r["dofs"] / r["seconds"], "unit": "DoF/s", "cores": r["threads"], "kind": "port",
                "sample": "%s workload, %d^3-cell cube (%d dofs), one pass of the same path (assemble %.2f s, "
                          "Dirichlet %.2f s, Schwarz setup %.2f s, GMRES %.2f s / %d its): oracle/oracle.c "
                          "(C/OpenMP restatement, not Trilinos), one-level RAS with %d-node subdomains"
                          % ("the full" if M == a.cells else "reduced", M, r["dofs"], r["t_assemble"], r["t_bc"],
                             r["t_prec"], r["t_gmres"], r["its"], a.target)}
    import fedd_oracle as fo
    M = a.cpu_cells or 32
    t0 = time.perf_counter()
    m = fo.build_mesh_structured(3, 1, M)
    t_mesh = time.perf_counter() - t0
    t0 = time.perf_counter()
    A_bc, rhs_bc, _, _, _ = fo.laplace_problem(m)
    nb_, nb, _ = fo.schwarz_bins(m.xyz_uni, a.target)
    ras = fo.RAS(A_bc, nb_, nb)
    x, its, hist = fo.gmres_right(A_bc, rhs_bc, ras.apply, rtol=a.rtol, max_it=a.max_it, restart=a.restart)
    dt = time.perf_counter() - t0
    return {"value": m.n_global / dt, "unit": "DoF/s", "cores": 1, "kind": "port",
            "sample": "same path on a %d^3-cell cube (%d dofs, %d GMRES its, %.1f s): numpy/scipy oracle"
                      % (M, m.n_global, its, dt)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = a.gpus
    if world != N:
        if world == 1 and N > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (N, N))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, N))
    if N not in DECOMP:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")
    import torch
    import torch.distributed as dist
    from feddlib_amd import capi
    rehearse = a.rehearse_one_gpu and N > 1
    dev = 0 if rehearse else local_rank
    torch.cuda.set_device(dev)
    nccl_id = None
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(capi.nccl_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            nccl_id = bytes(idt.cpu().numpy().tobytes())

    def max_over_ranks(v):
        if N == 1:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def barrier():
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dec = DECOMP[N]
    cells = [a.cells] * 3
    t0 = time.perf_counter()
    m = capi.structured_mesh(3, dec, cells, rank, ghosts=N > 1)
    t_mesh = time.perf_counter() - t0
    c = capi.Context(device=dev, rank=rank, nranks=N, nccl_id=nccl_id)
    t0 = time.perf_counter()
    c.mesh_set_dict(m)
    if N > 1:
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, cells, m["gid_rep"]))
        if rehearse:
            c.comm_set_torch_dist(dist)
        else:
            c.halo_exchange_setup()
    c.sync()
    t_upload = time.perf_counter() - t0
    n_global = m["n_global"]

    for _ in range(a.warmup):
        one_step(c, capi, a)
    c.sync()
    # HIP events on the library's stream, live in the timed region; the per-iteration kernels are
    # sampled every 8th launch (an event pair around every launch costs 3-6 % of the step)
    c.timing_enable(8)
    c.timing_reset()
    barrier()
    t0 = time.perf_counter()
    its = rel = None
    for _ in range(a.steps):
        its, rel = one_step(c, capi, a)
    c.sync()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    tm = c.timing_get()
    nr, ncol, nnz = c.csr_sizes()
    info = c.schwarz_info()

    # ---- extra, outside the headline number: the same step with the coarse level switched on ----
    two = None
    if not a.no_two_level:
        one_step(c, capi, a, two_level=True)
        c.sync()
        c.timing_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            its2, rel2 = one_step(c, capi, a, two_level=True)
        c.sync()
        barrier()
        dt2 = max_over_ranks(time.perf_counter() - t0)
        g2, _ = c.schwarz_coarse_sizes()
        tm2 = c.timing_get()
        two = {"value": n_global * a.steps / dt2, "unit": "DoF/s", "ms_per_step": dt2 / a.steps * 1e3,
               "gmres_iterations": its2, "relres": rel2, "coarse_cells": [int(v) for v in g2],
               "coarse_dofs": int(c.schwarz_coarse_sizes()[1]),
               "phases_device_ms_per_step": {k: round(v[0] / a.steps, 4) for k, v in tm2.items()},
               "note": "same step with fedd_schwarz_setup(two_level=1, FEDD_COARSE_Q1); not the headline config"}

    # ---- extra: SpMV launched back to back on resident vectors (single GPU; no halo in the loop) ----
    spmv_b2b = None
    if N == 1:
        c.timing_enable(1)
        c.spmv_device(5)
        c.timing_reset()
        c.spmv_device(50)
        c.sync()
        ms, nl = c.timing_get()["spmv"]
        spmv_b2b = {"ms_per_launch": ms / nl, "GBs": (12.0 * nnz + 20.0 * nr) / (ms / nl) / 1e6,
                    "frac_hbm_peak": (12.0 * nnz + 20.0 * nr) / (ms / nl) / 1e6 / HBM_PEAK_GBS,
                    "note": "50 launches back to back: the 203 MB matrix stays in the 256 MB Infinity Cache between "
                            "launches, so this is not a pure HBM figure; spmv_frac_hbm_peak is the in-solver one"}

    if rank == 0:
        # algorithmic bytes per launch (SURVEY.md 8d / DESIGN.md), this rank's share
        models = {
            "spmv": 12.0 * nnz + 20.0 * nr,
            "schwarz_apply": info["inverse_bytes"] + 3 * 8.0 * nr,
            "assemble": 4.0 * m["conn"].size + 8.0 * 3 * m["xyz"].shape[0] + 12.0 * nnz + 4.0 * (nr + 1),
        }
        kern = {}
        for k, b in models.items():
            ms, nl = tm[k]
            if nl:
                kern[k] = dict(ms_per_launch=ms / nl, launches=nl, total_ms=ms, GBs=b / (ms / nl) / 1e6, bytes=b)
        # GMRES orthogonalisation: 2 passes over the basis per DGKS pass; total bytes over the solve
        ms, nl = tm["ortho"]
        dominant = max(kern, key=lambda k: kern[k]["total_ms"])
        d = kern[dominant]
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": d["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d["GBs"] / HBM_PEAK_GBS, "traffic": None,
                    "ms_per_launch": d["ms_per_launch"], "algorithmic_bytes_per_launch": d["bytes"]}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                roofline["traffic"] = json.load(open(pmc)).get(dominant)
            except Exception:
                pass
        out = {
            "metric": "DoF/s assemble+solve, 3D P1-Laplace cube",
            "value": n_global * a.steps / dt,
            "unit": "DoF/s",
            "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (structured unit cube, f=1, homogeneous Dirichlet; reference generator semantics)",
            "config": {"workload": "3D P1 Laplace, structured cube, %s blocks x %d^3 cells, %d dofs, nnz %d/GPU; "
                                   "GMRES(%d) rtol %g + one-level RAS (overlap 1, %d-node subdomains, exact local solves)"
                                   % ("x".join(map(str, dec)), a.cells, n_global, nnz, a.restart, a.rtol, a.target),
                       "dofs": n_global, "gmres_iterations": its, "relres": rel,
                       "subdomains_per_gpu": info["n_subdomains"], "max_subdomain_size": info["max_size"]},
            "roofline": roofline,
            "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kern.items()},
            "phases_device_ms_per_step": {k: round(v[0] / a.steps, 4) for k, v in tm.items()},
            "spmv_frac_hbm_peak": kern["spmv"]["GBs"] / HBM_PEAK_GBS if "spmv" in kern else None,
            "mesh_generation_s": t_mesh, "mesh_upload_s": t_upload,
        }
        if spmv_b2b is not None:
            out["spmv_back_to_back"] = spmv_b2b
        if two is not None:
            out["two_level_variant"] = two
        if N == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a)
        print(json.dumps(out), flush=True)
    c.close()
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
