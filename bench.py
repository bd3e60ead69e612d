#!/usr/bin/env python
"""bench.py -- DoF/s of the assemble + solve hot path on the 3D P1-Laplace cube (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full pass of the hot path over the resident mesh: symbolic CSR pattern ->
element assembly (matrix + rhs) -> Dirichlet rows -> one-level Schwarz setup -> right-preconditioned
GMRES(100) to 1e-8 (feddlib/problems/tests/laplace/main.cpp:199-208 with 3D/P1/structured
parameters).  The mesh (connectivity, coordinates, flags) is generated on the host and uploaded
BEFORE the timed region (metric definition).

Workload (default): the grid of BASELINE.md's headline cfg 3 at every N -- unit cube, 214^3 cells,
          9 938 375 dofs, 147 968 803 nonzeros -- split into 1x1x1, 1x1x2, 1x2x2, 2x2x2 blocks
          (strong scaling; N = 8 is the reference's N = 2, M = 107 decomposition, the 2- and 4-block
          splits of the same grid are BASELINE.md section 2's stated extension: the reference
          generator needs N^3 ranks, laplace/main.cpp:132).  At N = 1 the line also carries
          `cfg2_one_gpu`: the same path on BASELINE.json configs[1] (100^3 cells, 1 030 301 dofs).
          --cells M switches to M^3 cells per GPU (weak scaling; development and rehearsals).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: without this RCCL's peer-to-peer setup fails (hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

DECOMP = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}
MFMA_F64_PEAK_TFLOPS = 78.6    # dense f64 matrix rate: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--global-cells", type=int, default=214, help="cells per direction of the whole cube (strong scaling)")
    ap.add_argument("--cells", type=int, default=0,
                    help="cells per direction and GPU (M = H/h): weak scaling instead of the fixed global grid")
    ap.add_argument("--no-cfg2", action="store_true", help="N = 1: skip the extra configs[1] (100^3 cells) measurement")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--restart", type=int, default=100)
    ap.add_argument("--max-it", type=int, default=2000)
    # 64-node boxes: the fastest one-level decomposition on this grid since the local inverses are shared
    # (profiles/r02_one_level_sweep_214_shared.txt: 248 ms against 312 ms for 27-node boxes; 145 against 178 iterations)
    ap.add_argument("--target", type=int, default=64, help="nodes per Schwarz subdomain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: skip the extra entries per_gpu_share_n8, cfg5_share_one_gpu and cfg4_one_gpu")
    ap.add_argument("--no-two-level", action="store_true", help="skip the extra two-level measurement")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="development only: run the N > 1 path with all ranks on GPU 0, gloo + the library's "
                         "host-staged transport (RCCL refuses several ranks on one device); numbers are not a measurement")
    ap.add_argument("--coarse", type=float, default=0.0, help="two-level variant: lattice cells (0 = library default)")
    ap.add_argument("--cpu-cells", type=int, default=0, help="cells per direction of the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-multi-ab", action="store_true",
                    help="N > 1: skip the A/B of the multi-GPU switches that follows the result line (stderr only)")
    ap.add_argument("--cpu-full", action="store_true",
                    help="time the CPU restatement on the full grid of the GPU run instead of the bounded sample (minutes)")
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


def _cycle_cols(its, restart):
    """sum over the iterations of GMRES(restart) of the number of basis columns orthogonalised against"""
    full, rest = divmod(its, restart)
    return full * restart * (restart + 1) // 2 + rest * (rest + 1) // 2


def _recorded_full_cpu_run(full_cells, target):
    """the full-grid CPU run of this round (bench.py --cpu-full on a GPU box, committed as profiles/cpu_full.json): the
    measured partner of the extrapolated figure, quoted when it is of the same grid"""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "cpu_full.json")))
        if pj.get("cells") == full_cells and pj.get("target", 27) == target:
            cb = pj["cpu_baseline"]
            return {"value": cb["value"], "seconds": cb["seconds"], "cores": cb["cores"], "sample": cb["sample"],
                    "source": "profiles/cpu_full.json (" + pj.get("command", "") + ")"}
    except Exception:
        pass
    return None


def host_cores():
    """the cores this process may run on (the box's CPU share; os.cpu_count() is the whole host)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    # a container's CPU share is a cgroup quota, not an affinity mask: 256 OpenMP threads on a 16-core share run 13x slower
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                n = max(1, min(n, int(int(q) / int(per) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return n


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_run(a, M, nthr):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    return oracle_c.run_laplace3d(M, a.target, a.rtol, a.restart, a.max_it, threads=nthr)


def cpu_baseline(a, full_cells, gpu_its, full=False, nthr=0):
    """The CPU restatement (oracle/oracle.c, kind 'port': C/OpenMP, the same driver sequence, preconditioner
    definition and tolerance as the GPU run -- not Trilinos) timed on this box's host cores.  The headline grid
    (214^3 cells) takes minutes on the CPU, so by default a bounded sample of THE SAME workload is run -- the same
    path on a smaller cube of the same grid family -- and extrapolated explicitly to the headline grid
    (BASELINE.md 4.3-4.4): assembly, Dirichlet and Schwarz setup scale with the dofs; the solve is
    iterations x (operator time per iteration and dof + Gram-Schmidt time per basis column and dof), with the
    iteration count of the full grid taken from the GPU run (same algorithm; the tests hold the two counts within
    +-2).  --cpu-full runs the full grid instead (recorded once per round under profiles/)."""
    nthr = nthr or int(os.environ.get("FEDD_CPU_THREADS", "0")) or host_cores()
    n_full = (full_cells + 1) ** 3
    M = full_cells if full else min(a.cpu_cells, full_cells)
    r = cpu_run(a, M, nthr)
    setup = r["t_assemble"] + r["t_bc"] + r["t_prec"]
    desc = ("assemble %.2f s, Dirichlet %.2f s, Schwarz setup %.2f s, GMRES %.2f s / %d its (operator %.2f s, "
            "Gram-Schmidt %.2f s)" % (r["t_assemble"], r["t_bc"], r["t_prec"], r["t_gmres"], r["its"],
                                      r["t_gmres_operator"], r["t_gmres_ortho"]))
    if M == full_cells:
        return {"value": r["dofs"] / r["seconds"], "unit": "DoF/s", "cores": r["threads"], "kind": "port",
                "sample": "the full workload, %d^3-cell cube (%d dofs), one pass of the same path: %s; oracle/oracle.c "
                          "(C/OpenMP restatement, not Trilinos), one-level RAS with %d-node subdomains"
                          % (M, r["dofs"], desc, a.target),
                "seconds": r["seconds"], "extrapolated": False}
    scale = n_full / r["dofs"]
    cols_s, cols_f = _cycle_cols(r["its"], a.restart), _cycle_cols(gpu_its, a.restart)
    other = max(r["t_gmres"] - r["t_gmres_operator"] - r["t_gmres_ortho"], 0.0)   # basis scaling, x update: per iteration
    t_full = scale * (setup + (r["t_gmres_operator"] + other) * gpu_its / max(r["its"], 1)
                      + r["t_gmres_ortho"] * cols_f / max(cols_s, 1))
    return {"value": n_full / t_full, "unit": "DoF/s", "cores": r["threads"], "kind": "port",
            "sample": "bounded sample of the same workload: the same path on a %d^3-cell cube (%d dofs): %s, %.1f s in all "
                      "= %.0f DoF/s measured; EXTRAPOLATED to the %d^3-cell headline grid: setup x dofs ratio %.2f, operator "
                      "time x dofs ratio x iterations %d/%d (count of the GPU run on the full grid), Gram-Schmidt time x "
                      "dofs ratio x basis columns %d/%d (GMRES(%d) cycles) = %.1f s; oracle/oracle.c (C/OpenMP "
                      "restatement, not Trilinos), one-level RAS with %d-node subdomains; a full-grid CPU run is "
                      "recorded under profiles/ (bench.py --cpu-full)"
                      % (M, r["dofs"], desc, r["seconds"], r["dofs"] / r["seconds"], full_cells, scale, gpu_its, r["its"],
                         cols_f, cols_s, a.restart, t_full, a.target),
            "seconds": t_full, "extrapolated": True, "full_grid_measured": _recorded_full_cpu_run(full_cells, a.target),
            "sample_measured": {"cells": M, "dofs": r["dofs"], "seconds": r["seconds"], "value": r["dofs"] / r["seconds"],
                                "gmres_iterations": r["its"]}}


def env_options(c):
    """development: FEDD_OPTIONS=key=value,... sets library options on every context of the run"""
    for kv in filter(None, os.environ.get("FEDD_OPTIONS", "").split(",")):
        c.set_option(kv.split("=")[0], float(kv.split("=")[1]))


def timed_passes(c, step, steps, warmup, stride=1, wall_without_timers=False):
    """`warmup` untimed and `steps` timed passes of step() on one context; wall ms per pass, the last pass' return value and the
    device-time table.  wall_without_timers: the wall time comes from `steps` passes with the device timers OFF (an event pair
    around every launch of a launch-bound step costs 10 % and more of it), the table from `steps` further passes with them on"""
    for _ in range(warmup):
        step()
    c.sync()
    wall = None
    if wall_without_timers:
        c.timing_enable(0)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        c.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
    c.timing_enable(stride)
    c.timing_reset()
    t0 = time.perf_counter()
    ret = None
    for _ in range(steps):
        ret = step()
    c.sync()
    wall_t = (time.perf_counter() - t0) / steps * 1e3
    tm = c.timing_get()
    c.timing_enable(0)
    if wall is None:
        wall = wall_t
    else:
        tm["_wall_with_timers_ms"] = wall_t
    return wall, ret, tm


def phases(tm, steps):
    return {k: round(v[0] / steps, 4) for k, v in tm.items() if not k.startswith("_")}


def spmv_bytes_model(si, nr):
    """bytes one solver SpMV streams: values (8) and column ids (4) per entry + row pointer, x and y per row; with the
    column patterns (fedd_spmv_patterns) a 2-byte pattern id per row replaces the column ids of the rows that have one"""
    if si.get("row_classes"):
        # row classes (k_spmv_cls): a classed row reads its 4-byte (class, pattern) word and x (8) and writes y (8) -- its values come
        # from the class table (66 bytes per class, cache resident); the other rows stream their entries as before
        nc, rest = si["rows_in_classes"], si["nnz_streamed_outside_classes"]
        return 20.0 * nc + 12.0 * rest + 22.0 * (nr - nc) + 66.0 * si["row_classes"]
    if si.get("column_patterns"):
        return 8.0 * si["nnz_streamed"] + 22.0 * nr + 4.0 * si["nnz_streamed"] * si["rows_with_explicit_columns"] / max(nr, 1)
    # (16-bit column offsets: 10 B per entry, 12 for the entries of the windows that keep 32-bit indices)
    wide = si.get("entries_with_32bit_columns", 0) if si.get("column_index_bytes", 4) == 2 else 0
    return (8.0 + si.get("column_index_bytes", 4)) * si["nnz_streamed"] + 2.0 * wide + 20.0 * nr


def extra_per_gpu_share(capi, dev, a, its_headline):
    """The share of ONE GPU of the 8-GPU run (BASELINE cfg 3: 2 x 2 x 2 blocks of 107^3 cells), on one GPU: the same step on a
    107^3-cell cube (1 259 712 dofs) held to the iteration count of the full grid (rtol off, max_it = that count), because at
    N = 8 every rank iterates as long as the whole problem does.  No communication in it: what it prices is the fixed cost
    per iteration at the per-GPU vector length -- device time by kernel class and the wall-minus-device gap (launch latency,
    host round trips) -- which bounds the 8-GPU step from below."""
    m = capi.structured_mesh(3, (1, 1, 1), [107] * 3, 0)
    c = capi.Context(device=dev)
    env_options(c)
    # (the solve is stopped by max_it with the tolerance switched off: the block length must not follow that tolerance)
    c.set_option("gmres_tol_blocks", 0)
    c.mesh_set_dict(m)

    def step():
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        c.schwarz_set_target(a.target, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
        return c.gmres(None, rtol=1e-300, max_it=its_headline, restart=a.restart, use_prec=True, want_x=False)[1:]

    wall, (its, rel), tm = timed_passes(c, step, 5, 2, wall_without_timers=True)
    wall_timed = tm.pop("_wall_with_timers_ms")
    per_it = ("spmv", "schwarz_apply", "ortho", "gs_dot", "gs_update")
    dev_total = sum(v[0] for k, v in tm.items() if k not in ("gs_dot", "gs_update", "gs_fused", "allreduce")) / 5
    out = {"workload": "107^3-cell cube, %d dofs, held to the %d iterations of the 214^3 grid" % (m["n_global"], its),
           "ms_per_step": wall, "ms_per_step_with_event_timers": wall_timed,
           "note": "ms_per_step: 5 steps with the device timers off (round 3 quoted the step WITH an event pair around every "
                   "launch: 21.1 ms); the device table below is from 5 further steps with the timers on",
           "gmres_iterations": its, "phases_device_ms_per_step": phases(tm, 5),
           "device_us_per_iteration": {k: round(tm[k][0] / 5 / max(its, 1) * 1e3, 2) for k in per_it},
           "device_ms_per_step": dev_total, "wall_minus_device_ms": wall - dev_total,
           "solve_device_ms_per_iteration": round(sum(tm[k][0] for k in ("spmv", "schwarz_apply", "ortho")) / 5 / max(its, 1), 4),
           "gmres": c.gmres_info(), "schwarz": c.schwarz_info()}
    c.close()
    return out


def extra_cfg5_share(capi, dev, a, only=None):
    """BASELINE cfg 5 (3D P1 linear elasticity, 189^3 nodes on 8 GPUs, two-level Schwarz) as the share of one GPU: a
    94^3-cell cube, 2 571 375 dofs, steadyLinElas_Perf parameters (mu 2e6, nu 0.4, f = (0, 1, 0), Dirichlet on flag 2, rtol 1e-6,
    restart 100; parametersProblem.xml:5-11, parametersSolver.xml), 8-node boxes; coarse level RGDSW as the XML names it, the Q1
    lattice space and the one-level operator beside it.  Its matrix (113 M entries, 1.4 GB, no repeated column patterns per
    3-dof row block) is the HBM-sized pattern-free SpMV case."""
    M = 94
    m = capi.structured_mesh(3, (1, 1, 1), [M] * 3, 0)
    c = capi.Context(device=dev)
    env_options(c)
    c.mesh_set_dict(m)
    mu, nu = 2.0e6, 0.4
    lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
    n = 3 * m["n_global"]
    out = {"workload": "3D P1 linear elasticity, 94^3 cells, %d dofs (one GPU's share of cfg 5), rtol 1e-6, 8-node boxes" % n}
    # (the one-level operator alone needs > 2000 iterations on this problem with 8-node boxes: not a configuration anyone runs)
    # rgdsw_rotations: option gdsw_rotations, what FROSch builds from steadyLinElas/parametersPrec.xml:6, 100 (node lists and
    # "Rotations" = true; the Perf file's coarse block has "Rotations" = false)
    kinds = (("q1", capi.COARSE_Q1), ("rgdsw", capi.COARSE_RGDSW), ("rgdsw_rotations", capi.COARSE_RGDSW), ("gdsw", capi.COARSE_GDSW))
    for name, kind in kinds:
        if only is not None and name not in only:
            continue
        c.set_option("gdsw_rotations", 1 if name.endswith("_rotations") else 0)

        def step():
            c.pattern_build(3, capi.BLOCK_FULL)
            c.assemble(capi.FORM_LINELAS, [lam, mu])
            c.assemble_rhs([0.0, 1.0, 0.0])
            c.dirichlet([2], [0.0, 0.0, 0.0])
            c.schwarz_set_target(8, 1.0)
            if kind is None:
                c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
            else:
                c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=kind)
            return c.gmres(None, rtol=1e-6, max_it=2000, restart=100, use_prec=True, want_x=False)[1:]
        npass = 1 if name == "gdsw" else 2      # (GDSW: 78 extension columns, over a second per step)
        wall, (its, rel), tm = timed_passes(c, step, npass, 1)
        x, b = c.solution_get(), c.rhs_get()
        e = {"ms_per_step": wall, "value": n / wall * 1e3, "unit": "DoF/s", "gmres_iterations": its, "relres": rel,
             "true_relres": float(np.linalg.norm(b - c.spmv(x)) / np.linalg.norm(b)),
             "phases_device_ms_per_step": phases(tm, npass)}
        if kind is not None:
            e["coarse_dofs"] = int(c.schwarz_coarse_sizes()[1])
        if name == "q1":
            nr, _, nnz = c.csr_sizes()
            si = c.spmv_info()
            ms, nl = tm["spmv"]
            byt = spmv_bytes_model(si, nr)
            e["spmv"] = {"ms_per_launch": ms / max(nl, 1), "bytes_streamed": byt, "GBs": byt / (ms / max(nl, 1)) / 1e6,
                         "frac_hbm_peak": byt / (ms / max(nl, 1)) / 1e6 / HBM_PEAK_GBS, "nnz": si,
                         "parity_csr_bytes": 12.0 * nnz + 20.0 * nr,
                         "note": "the solver's kernel; on this structured grid the rows repeat (row classes): it streams a word per row "
                                 "and is bound by the gathers of x, not by HBM -- the general kernels on the same matrix are below"}
            # the same matrix through the kernel a mesh WITHOUT repeated rows gets (an unstructured elasticity matrix of this size):
            # row classes off -- the 45-entry rows are longer than the pattern kernel takes, so this is the per-entry stream, values +
            # 16-bit column offsets --, 50 launches back to back; y is the same bit for bit (tests/test_gpu_edge_cases.py)
            c.set_option("spmv_classes", 0)
            c.timing_enable(1)
            c.spmv_device(5)
            c.timing_reset()
            c.spmv_device(50)
            c.sync()
            msg, nlg = c.timing_get()["spmv"]
            c.timing_enable(0)
            sig = c.spmv_info()
            bg = spmv_bytes_model(sig, nr)
            c.set_option("spmv_classes", 1)
            e["spmv_general_kernel_back_to_back"] = {
                "ms_per_launch": msg / max(nlg, 1), "bytes_streamed": bg, "GBs": bg / (msg / max(nlg, 1)) / 1e6,
                "frac_hbm_peak": bg / (msg / max(nlg, 1)) / 1e6 / HBM_PEAK_GBS, "column_patterns": sig.get("column_patterns", 0),
                "row_classes": sig.get("row_classes", 0), "column_index_bytes": sig.get("column_index_bytes", 4),
                "note": "option spmv_classes 0: the HBM-sized, value-free case (what an unstructured mesh of this size runs)"}
            ms, nl = tm["assemble"]
            ab = 4.0 * m["conn"].size + 8.0 * 3 * m["xyz"].shape[0] + 12.0 * nnz + 4.0 * (nr + 1)
            e["assemble"] = {"ms_per_launch": ms / max(nl, 1), "bytes": ab, "frac_hbm_peak": ab / (ms / max(nl, 1)) / 1e6 / HBM_PEAK_GBS}
        out[name] = e
    c.close()
    return out


def extra_p2_cube(capi, dev, M=64):
    """P2 assembly at an HBM-relevant size (the only P2 figure was the 27 618-tet cylinder, latency-bound): the
    P2 mesh of an M^3-cell structured cube (fedd_mesh_p2_build: edge mid-points), pattern and FE::assemblyLaplace with the
    5-point rule and 10 x 10 element matrices (FE_def.hpp:604-667 with :6262-6293, :5661-5713), plus the P2 mass matrix; device
    ms per launch, algorithmic bytes 4 nen E + 8 dim N + 12 nnz + 4 (N + 1), fraction of the HBM peak."""
    m1 = capi.structured_mesh(3, (1, 1, 1), [M] * 3, 0)
    t0 = time.perf_counter()
    mv = capi.p2_of_p1(m1, volume_id=0)
    t_p2 = time.perf_counter() - t0
    c = capi.Context(device=dev)
    env_options(c)
    c.mesh_set_dict(mv)
    out = {"workload": "P2 tetrahedra on a %d^3-cell structured cube: %d nodes, %d elements" % (M, mv["n_global"], mv["conn"].shape[0]),
           "p2_mesh_build_host_s": t_p2}
    for name, form in (("laplace", capi.FORM_LAPLACE), ("mass", capi.FORM_MASS)):
        def step():
            c.pattern_build(1, capi.BLOCK_SCALAR)
            c.assemble(form)
        wall, _, tm = timed_passes(c, step, 3, 2)
        nr, _, nnz = c.csr_sizes()
        byt = 4.0 * mv["conn"].size + 8.0 * 3 * mv["xyz"].shape[0] + 12.0 * nnz + 4.0 * (nr + 1)
        ms = tm["assemble"][0] / 3          # per step (two launches: element matrices, row sums)
        setup = c.mesh_setup_info()
        out[name] = {"assemble_ms": ms, "symbolic_ms": tm["symbolic"][0] / 3, "nnz": int(nnz),
                     "algorithmic_bytes": byt, "GBs": byt / ms / 1e6, "frac_hbm_peak": byt / ms / 1e6 / HBM_PEAK_GBS,
                     "kernel": "k_elem_matrix (one element per wavefront, 10 x 10 element matrices) + k_p2_gather (row sums by gather lists)"}
    out["per_mesh_setup"] = setup          # adjacency + the gather lists of the row sums, once per mesh
    # the same Laplace assembly with the round-3 path (every (row, element) pair re-derives its row)
    c2 = capi.Context(device=dev)
    c2.set_option("asm_p2_elem", 0)
    c2.mesh_set_dict(mv)

    def step_old():
        c2.pattern_build(1, capi.BLOCK_SCALAR)
        c2.assemble(capi.FORM_LAPLACE)
    _, _, tm = timed_passes(c2, step_old, 2, 1)
    out["laplace"]["assemble_ms_pair_kernels_alone"] = tm["assemble"][0] / 2
    c2.close()
    c.close()
    return out


def extra_cfg4(capi, dev):
    """BASELINE cfg 4: P2 / P1 Stokes on DFG3DCylinder_6k.mesh (141 742 dofs), block assembly, merge, the driver's boundary
    conditions (no-slip on flags 1 and 4, parabolic inflow on flag 2; stokes/main.cpp:80-88, 267-296), monolithic one-level
    Schwarz on subdomains of several hundred dofs, GMRES to 1e-6 (the reference's stokes XML tolerance)."""
    path = os.path.join(ROOT, "tests", "golden", "DFG3DCylinder_6k.mesh")
    if not os.path.exists(path):
        return None
    m1 = capi.read_mesh(path, 3)
    mv = capi.p2_of_p1(m1, volume_id=0)
    n_p, nv = m1["xyz"].shape[0], mv["xyz"].shape[0]
    n = 3 * nv + n_p
    c = capi.Context(device=dev)
    env_options(c)
    c.mesh_set_dict(mv)
    X, flag, H = mv["xyz"], mv["flag_uni"], 0.41
    nodes = np.nonzero(np.isin(flag, (1, 2, 4)))[0]
    rows = (3 * nodes[:, None] + np.arange(3)[None, :]).ravel()
    vals = np.zeros((nodes.shape[0], 3))
    inflow = flag[nodes] == 2
    vals[inflow, 0] = (16.0 * X[nodes, 1] * (H - X[nodes, 1]) * X[nodes, 2] * (H - X[nodes, 2]) / H ** 4)[inflow]

    def step():
        c.pattern_build(3, capi.BLOCK_DIAG)
        c.assemble(capi.FORM_LAPLACE_VEC)
        c.matrix_store(0)
        c.assemble_div(n_p, 1, 2)
        c.matrix_scale(1, -1.0)
        c.matrix_scale(2, -1.0)
        c.block_merge(0, 2, 1, -1)
        c.rhs_set(np.zeros(n))
        c.dirichlet_rows(rows, vals.ravel())
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
        return c.gmres(None, rtol=1e-6, max_it=3000, restart=300, use_prec=True, want_x=False)[1:]

    wall, (its, rel), tm = timed_passes(c, step, 2, 1)
    x, b = c.solution_get(), c.rhs_get()
    out = {"workload": "P2/P1 Stokes, DFG3DCylinder_6k.mesh, %d dofs, merged system, monolithic one-level Schwarz, rtol 1e-6" % n,
           "ms_per_step": wall, "value": n / wall * 1e3, "unit": "DoF/s", "gmres_iterations": its, "relres": rel,
           "true_relres": float(np.linalg.norm(b - c.spmv(x)) / np.linalg.norm(b)),
           "phases_device_ms_per_step": phases(tm, 2), "schwarz": c.schwarz_info()}
    c.close()
    return out


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

    def nccl_id_ab():
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.nccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        return bytes(idt.cpu().numpy().tobytes())

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
    weak = a.cells > 0
    if weak:
        cells = [a.cells] * 3
        # cells stay cubes for every decomposition: the domain is dec / max(dec) of the unit cube
        dom = [d / float(max(dec)) for d in dec]
    else:
        if any(a.global_cells % d for d in dec):
            raise SystemExit("--global-cells %d is not divisible by the %s decomposition" % (a.global_cells, dec))
        cells = [a.global_cells // d for d in dec]
        dom = [1.0, 1.0, 1.0]
    if not a.cpu_cells:
        a.cpu_cells = min(128, max(cells))

    cold = {}

    def measure(c, n_global, two_level, extra=True):
        """W warm-up steps, then exactly K timed steps between barriers; max over ranks."""
        # the headline takes exactly the W warm-up steps of the contract; the extra measurements (two-level variant,
        # cfg 2) follow host-side checks during which the GPU idles and clocks down -- a 12 ms step measured 25 ms
        # right after one -- so they warm up for at least W steps AND a quarter of a second
        t_w = time.perf_counter()
        n_w = 0
        while n_w < a.warmup or (extra and time.perf_counter() - t_w < 0.25 and n_w < 50):
            one_step(c, capi, a, two_level)
            if n_w == 0 and not extra and "first_step_ms" not in cold:
                # the very first step on this mesh: the per-mesh structures (node -> element adjacency, tile structures of
                # the assembly kernel), first-touch allocations and everything a steady-state step does
                c.sync()
                cold["first_step_ms"] = (time.perf_counter() - t_w) * 1e3
                cold["per_mesh_setup"] = c.mesh_setup_info()
            n_w += 1
        c.sync()
        # HIP events on the library's stream, live in the timed region; the per-iteration kernels are
        # sampled every 8th launch (an event pair around every launch costs 3-6 % of the step)
        c.timing_enable(TIMING_STRIDE)
        c.timing_reset()
        barrier()
        t0 = time.perf_counter()
        its = rel = None
        for _ in range(a.steps):
            its, rel = one_step(c, capi, a, two_level)
        c.sync()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        tm_ = c.timing_get()
        # the Gram-Schmidt sweeps: device time and algorithmic bytes of exactly the launches that were timed
        smp = c.timing_get_sampled()
        tm_["_sampled"] = {k: smp[k] for k in ("gs_dot", "gs_update", "gs_fused") if k in smp}
        tm_["_gmres"] = c.gmres_info()
        return dt, its, rel, tm_

    def true_relres(ctx):
        """||b - A x|| / ||b|| of the solution the last solve left on the device, formed with fedd_spmv
        (outside every timed region); the solver itself only sees its implicit (recurrence) residual."""
        x, b = ctx.solution_get(), ctx.rhs_get()
        r = b - ctx.spmv(x)
        if N == 1:
            return float(np.linalg.norm(r) / np.linalg.norm(b))
        tt = torch.tensor([float(r @ r), float(b @ b)], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        return float((tt[0] / tt[1]) ** 0.5)

    def checked(ctx, label):
        tr = true_relres(ctx)
        if not tr <= 10.0 * a.rtol:
            raise SystemExit("bench.py: %s: true residual %.3e exceeds 10 x rtol = %.1e" % (label, tr, 10.0 * a.rtol))
        return tr

    TIMING_STRIDE = 8       # the per-iteration kernel classes are timed every 8th launch (measure() below)

    def kernel_table(tm, m, nr, nnz, info):
        # algorithmic bytes per launch (SURVEY.md 8d / DESIGN.md section 6), this rank's share
        shared = info["n_unique"] * 4 <= info["n_subdomains"]
        models = {
            "spmv": spmv_bytes_model(info["spmv"], nr),
            # every stored inverse once, the dof lists once, r in and z out once (r is gathered ~4x over through the
            # caches: cache traffic, not counted)
            # (the dof lists only of the subdomains whose ids the matrix-core kernel does not compute)
            "schwarz_apply": info["inverse_bytes"] + 2 * 8.0 * nr + 4.0 * info.get("sum_sizes", 0) *
                             (1.0 - (info.get("n_conforming", 0) / max(info["n_subdomains"], 1)
                                     if info["n_unique"] * 4 <= info["n_subdomains"] and info["n_subdomains"] >= 4096 else 0.0)),
            "assemble": 4.0 * m["conn"].size + 8.0 * 3 * m["xyz"].shape[0] + 12.0 * nnz + 4.0 * (nr + 1),
        }
        kern = {}
        # the Newton block basis of the s-step solver rides in the SpMV's store (y = A z - theta v): those launches read one
        # more vector.  All but the applications of the monomial blocks that come before the shifts exist (2 x 8) carry it.
        gk0 = tm.get("_gmres", {})
        if gk0.get("kind") == 2 and gk0.get("s", 0) > 8 and tm["spmv"][1]:
            per_step = tm["spmv"][1] / a.steps
            models["spmv"] += 8.0 * nr * max(0.0, per_step - 16 - 4) / per_step
        for k, b in models.items():
            ms, nl = tm[k]
            if nl:
                kern[k] = dict(ms_per_launch=ms / nl, launches=nl, total_ms=ms, GBs=b / (ms / nl) / 1e6, bytes=b)
        if "schwarz_apply" in kern:
            kern["schwarz_apply"]["distinct_inverses"] = info["n_unique"]
            bt = info.get("n_conforming") == info["n_subdomains"]     # every box conforms: the batch-table kernel runs
            kern["schwarz_apply"]["kernel"] = (("k_apply_bt (batch table, shared inverses)" if bt else "k_apply_mfma (shared inverses)")
                                               if shared and info["n_subdomains"] >= 4096 else "k_apply_flat")
        # the sweeps of the Gram-Schmidt process over the Krylov basis (gmres.hip): the launch sites state their algorithmic
        # bytes -- DCGS2 (gmres_kind 0): sweep 1 reads the k final columns, u and B u, sweep 2 reads them again and writes
        # v_{k+1} and the next u; s-step (gmres_kind 2): a dot sweep reads the k final columns and the s block columns, an
        # update sweep reads both and writes the block -- and the figure is bytes / time over exactly the timed launches
        gk = tm.get("_gmres", {"kind": 0})
        names = {0: ("k_multidot2", "k_axpy2"), 1: ("k_multidot", "k_multiaxpy"), 2: ("k_blockdot", "k_blockaxpy")}[gk["kind"]]
        classes = ("gs_dot", "gs_update")
        if gk.get("fused_blocks"):
            # three sweeps per block: the second pass' dot products ride in the first pass' update sweep (k_blockfuse, class gs_fused).
            # Its algorithmic bytes are those of the update alone (basis and block read once, block written): the second read of the
            # workgroup's rows comes from the Infinity Cache
            names = ("k_blockdot (first pass)", "k_blockaxpy (second pass)", "k_blockfuse (first pass' update + second pass' dot in one sweep)")
            classes = ("gs_dot", "gs_update", "gs_fused")
        for name, kname in zip(classes, names):
            ms, nl = tm.get(name, (0.0, 0))
            sms, snl, sb = tm.get("_sampled", {}).get(name, (0.0, 0, 0.0))
            if nl and snl and sms > 0 and sb > 0:
                kern[name] = dict(ms_per_launch=sms / snl, launches=nl, total_ms=ms, GBs=sb / sms / 1e6, bytes=sb / snl,
                                  timed_launches=snl, mean_columns_per_launch=sb / snl / (8.0 * nr), kernel=kname)
        return kern

    def rounded(kern):
        return {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kern.items()}

    t0 = time.perf_counter()
    # N > 1: as many ghost-element layers as a Schwarz box is wide plus one of overlap (27-node boxes: 3 + 1, 64-node
    # boxes: 4 + 1), so that the rows of the ghost nodes within one layer less are complete on the rank
    # (fedd_mesh_set_rows): every box that a rank boundary crosses is then built whole on both sides and the
    # preconditioner is the one-rank preconditioner at any N
    ghost_layers = int(math.ceil(round(a.target ** (1.0 / 3.0), 6))) + 1
    m = capi.structured_mesh(3, dec, cells, rank, size=dom, ghosts=ghost_layers if N > 1 else 0)
    t_mesh = time.perf_counter() - t0
    c = capi.Context(device=dev, rank=rank, nranks=N, nccl_id=nccl_id)
    for kv in filter(None, os.environ.get("FEDD_OPTIONS", "").split(",")):   # development: key=value,...
        c.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    preflight = None
    if N > 1 and not rehearse:
        # pre-flight: the solver's collectives on the library's own communicator (all-reduce, grouped send / receive with every
        # rank, ring shift) before any work depends on them, under a watchdog -- a misuse of RCCL ends the run within a minute
        # and with a message instead of hanging the scaling run
        import threading

        def _hung():
            sys.stderr.write("bench.py: rank %d: the RCCL pre-flight (fedd_comm_selftest) did not return within 90 s\n" % rank)
            sys.stderr.flush()
            os._exit(3)

        wd = threading.Timer(90.0, _hung)
        wd.daemon = True
        wd.start()
        t_pf = time.perf_counter()
        pf_err = c.comm_selftest(65536)
        wd.cancel()
        pf_err = max_over_ranks(pf_err)
        preflight = {"max_abs_err": pf_err, "seconds": time.perf_counter() - t_pf, "doubles_per_message": 65536}
        if not pf_err <= 1e-9:
            raise SystemExit("bench.py: RCCL pre-flight failed: max deviation %r" % pf_err)
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

    dt, its, rel, tm = measure(c, n_global, False, extra=False)
    true_rel = checked(c, "one-level headline")
    nr, ncol, nnz = c.csr_sizes()
    info = c.schwarz_info()
    info["spmv"] = c.spmv_info()

    # ---- extra, outside the headline number: the same step with the coarse level switched on ----
    two = None
    if not a.no_two_level:
        dt2, its2, rel2, tm2 = measure(c, n_global, True)
        true_rel2 = checked(c, "two-level variant")
        g2, n02 = c.schwarz_coarse_sizes()
        two = {"value": n_global * a.steps / dt2, "unit": "DoF/s", "ms_per_step": dt2 / a.steps * 1e3,
               "gmres_iterations": its2, "relres": rel2, "true_relres": true_rel2, "coarse_cells": [int(v) for v in g2],
               "coarse_dofs": int(n02),
               "phases_device_ms_per_step": {k: round(v[0] / a.steps, 4) for k, v in tm2.items() if not k.startswith("_")},
               "note": "same step with fedd_schwarz_setup(two_level=1, FEDD_COARSE_Q1); not the headline config"}

    # ---- extra: SpMV launched back to back on resident vectors (single GPU; no halo in the loop) ----
    spmv_b2b = None
    read_ceiling = None
    if N == 1:
        c.timing_enable(1)
        c.spmv_device(5)
        c.timing_reset()
        c.spmv_device(50)
        c.sync()
        ms, nl = c.timing_get()["spmv"]
        b = spmv_bytes_model(info["spmv"], nr)
        spmv_b2b = {"ms_per_launch": ms / nl, "GBs": b / (ms / nl) / 1e6, "frac_hbm_peak": b / (ms / nl) / 1e6 / HBM_PEAK_GBS,
                    "note": "50 launches back to back on resident vectors" +
                            ("; the matrix (%.0f MB) fits the 256 MB Infinity Cache, so this is not a pure HBM figure"
                             % (b / 1e6) if b < 256e6 else "")}
        read_ceiling = c.read_bandwidth(2 << 30, 10)     # GB/s of a pure read stream on this box, outside timed regions
    kern = kernel_table(tm, m, nr, nnz, info) if rank == 0 else None

    # ---- N > 1, outside every timed region: the communication checked against communication-free references ----
    self_check = None
    if N > 1:
        import scipy.sparse as sp

        def sum_over_ranks(v):
            tt = torch.tensor([v], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            return float(tt.item())

        # (1) halo exchange: distributed SpMV of x(gid) against the same product with every column value computed
        #     locally from its global id (no exchange involved)
        rowptr, col, val, col_gid = c.csr_get()
        fgid = lambda g: np.sin(1.0e-3 * g) + 1.0e-7 * g
        y = c.spmv(fgid(m["gid_uni"].astype(np.float64)))
        A_loc = sp.csr_matrix((val, col, rowptr), shape=(rowptr.shape[0] - 1, col_gid.shape[0]))
        y_ref = A_loc @ fgid(col_gid.astype(np.float64))
        halo_err = max_over_ranks(float(np.abs(y - y_ref).max() / max(np.abs(y_ref).max(), 1e-300)))
        # (2) all-reduce: the residual GMRES reports (built from the library's all-reduced dot products) against
        #     ||b - A x|| / ||b|| formed with torch.distributed
        _, rel_chk = one_step(c, capi, a, False)
        c.sync()
        x, b = c.solution_get(), c.rhs_get()
        r = b - c.spmv(x)
        true_rel_sc = (sum_over_ranks(float(r @ r)) / sum_over_ranks(float(b @ b))) ** 0.5
        self_check = {"halo_spmv_max_rel_err": halo_err, "true_relres": true_rel_sc, "reported_relres": rel_chk}
        if not (halo_err <= 1e-12 and true_rel_sc <= 10.0 * a.rtol):
            raise SystemExit("bench.py: communication self-check failed: %r" % (self_check,))
    c.close()
    del m

    # ---- extra (N = 1, fixed-grid mode): the same path on BASELINE.json configs[1], 100^3 cells ----
    cfg2 = None
    if N == 1 and not weak and not a.no_cfg2:
        m2 = capi.structured_mesh(3, dec, [100] * 3, 0)
        c2 = capi.Context(device=dev, rank=0, nranks=1, nccl_id=None)
        for kv in filter(None, os.environ.get("FEDD_OPTIONS", "").split(",")):
            c2.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        c2.mesh_set_dict(m2)
        c2.sync()
        d1, i1, r1, t1 = measure(c2, m2["n_global"], False)
        tr1 = checked(c2, "cfg 2")
        nr2, _, nnz2 = c2.csr_sizes()
        info2 = c2.schwarz_info()
        info2["spmv"] = c2.spmv_info()
        k2 = kernel_table(t1, m2, nr2, nnz2, info2)
        cfg2 = {"workload": "BASELINE.json configs[1]: unit cube, 100^3 cells, %d dofs, nnz %d, same solver settings"
                            % (m2["n_global"], nnz2),
                "value": m2["n_global"] * a.steps / d1, "unit": "DoF/s", "ms_per_step": d1 / a.steps * 1e3,
                "gmres_iterations": i1, "relres": r1, "true_relres": tr1, "kernels": rounded(k2),
                "spmv_nnz": info2["spmv"],
                "spmv_frac_hbm_peak": k2["spmv"]["GBs"] / HBM_PEAK_GBS,
                "schwarz_apply_frac_hbm_peak": k2["schwarz_apply"]["GBs"] / HBM_PEAK_GBS,
                "phases_device_ms_per_step": {k: round(v[0] / a.steps, 4) for k, v in t1.items() if not k.startswith("_")}}
        if not a.no_two_level:
            d2, i2, r2, t2 = measure(c2, m2["n_global"], True)
            cfg2["two_level_variant"] = {"value": m2["n_global"] * a.steps / d2, "unit": "DoF/s",
                                         "ms_per_step": d2 / a.steps * 1e3, "gmres_iterations": i2, "relres": r2,
                                         "true_relres": checked(c2, "cfg 2 two-level"),
                                         "phases_device_ms_per_step": {k: round(v[0] / a.steps, 4) for k, v in t2.items() if not k.startswith("_")}}
        c2.close()
        del m2
        if not a.no_cpu_baseline:
            # the same grid on the host cores: a same-workload partner for cfg2_one_gpu (no extrapolation)
            cfg2["cpu_baseline"] = cpu_baseline(a, 100, i1, full=True)
            cfg2["gpu_over_cpu"] = cfg2["value"] / cfg2["cpu_baseline"]["value"]

    extras = {}
    if N == 1 and not weak and not a.no_extras:
        extras["per_gpu_share_n8"] = extra_per_gpu_share(capi, dev, a, its)
        extras["cfg5_share_one_gpu"] = extra_cfg5_share(capi, dev, a)
        e4 = extra_cfg4(capi, dev)
        if e4 is not None:
            extras["cfg4_one_gpu"] = e4
        extras["p2_cube"] = extra_p2_cube(capi, dev)

    if rank == 0:
        dominant = max(kern, key=lambda k: kern[k]["total_ms"])
        d = kern[dominant]
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": d["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": d["GBs"] / HBM_PEAK_GBS, "traffic": None,
                    "ms_per_launch": d["ms_per_launch"], "algorithmic_bytes_per_launch": d["bytes"]}
        if read_ceiling:
            roofline["measured_read_ceiling_GBs"] = read_ceiling
            roofline["frac_of_measured_ceiling"] = d["GBs"] / read_ceiling
        if "schwarz_apply" in kern and info.get("sum_sizes") and info.get("sum_owned"):
            # the apply's own roof is the f64 matrix rate (below), whichever class is the dominant one: its entry carries the figures.
            # 78.6 TFLOP/s is the spec; bare loops of the instruction sustain 66-74 on this chip (tools/microbench/mfma_f64.hip,
            # profiles/r04_mfma_f64_microbench.txt)
            fl = 2.0 * info["sum_sizes"] * info["sum_owned"] / max(info["n_subdomains"], 1)
            tf_ = fl / (kern["schwarz_apply"]["ms_per_launch"] * 1e-3) / 1e12
            kern["schwarz_apply"].update({"algorithmic_flops": fl, "TFLOPs": tf_, "frac_f64_matrix_peak": tf_ / MFMA_F64_PEAK_TFLOPS,
                                          "f64_matrix_rate_sustained_by_the_bare_instruction_TFLOPs": [66.0, 74.0]})
        if dominant == "schwarz_apply" and info.get("sum_sizes") and info.get("sum_owned"):
            # the Schwarz apply multiplies every subdomain's restriction of r with its (shared, cache-resident) inverse on the f64
            # matrix cores: 2 n_i rows_i flops per subdomain (boxes of one size: sum n_i x mean rows) against inverse + r + z bytes.
            # Whichever roof takes longer bounds the launch: on this workload the matrix cores (78.6 TFLOP/s dense f64, the figure
            # the review uses), not HBM
            flops = 2.0 * info["sum_sizes"] * info["sum_owned"] / max(info["n_subdomains"], 1)
            t_mfma, t_hbm = flops / (MFMA_F64_PEAK_TFLOPS * 1e12), d["bytes"] / (HBM_PEAK_GBS * 1e9)
            roofline["hbm_bound_figures"] = {"achieved": d["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["GBs"] / HBM_PEAK_GBS}
            if t_mfma > t_hbm:
                tf = flops / (d["ms_per_launch"] * 1e-3) / 1e12
                roofline.update({"bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tf / MFMA_F64_PEAK_TFLOPS, "algorithmic_flops_per_launch": flops})
        # HBM bytes per launch from the committed PMC passes of this same workload (tools/pmc_summary.py);
        # only quoted when the passes were taken on the grid this run used
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if pj.get("cells_per_gpu") == cells and pj.get("n_gpus", 1) == N:
                    roofline["traffic"] = pj.get(dominant)
                    # not measured in this run: the committed PMC passes of the same command (separate --pmc runs, as the
                    # guide prescribes); the stamp says which source tree they were taken on
                    roofline["traffic_source"] = {"file": "profiles/pmc_traffic.json", "git_sha": pj.get("git_sha"),
                                                  "tag": pj.get("tag"), "kernel": pj.get("detail", {}).get(dominant, {}).get("kernel")}
            except Exception:
                pass
        split = "x".join(map(str, dec))
        if weak:
            wl = ("3D P1 Laplace, structured box %s (cubic cells), %s blocks x %d^3 cells per GPU, %d dofs"
                  % ("x".join("%g" % v for v in dom), split, a.cells, n_global))
        else:
            wl = ("3D P1 Laplace, unit cube, %d^3 cells = %d dofs (BASELINE cfg 3 grid), %s blocks of %s cells"
                  % (a.global_cells, n_global, split, "x".join(map(str, cells))))
        out = {
            "metric": "DoF/s assemble+solve, 3D P1-Laplace cube",
            "value": n_global * a.steps / dt,
            "unit": "DoF/s",
            "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (structured unit cube, f=1, homogeneous Dirichlet; reference generator semantics)",
            "config": {"workload": wl + ", nnz %d/GPU; GMRES(%d) rtol %g + one-level RAS (overlap 1, %d-node subdomains, "
                                        "exact local solves)" % (nnz, a.restart, a.rtol, a.target),
                       "dofs": n_global, "gmres_iterations": its, "relres": rel, "true_relres": true_rel,
                       "subdomains_per_gpu": info["n_subdomains"], "max_subdomain_size": info["max_size"]},
            "roofline": roofline,
            "kernels": rounded(kern),
            "phases_device_ms_per_step": {k: round(v[0] / a.steps, 4) for k, v in tm.items() if not k.startswith("_")},
            "gmres": tm["_gmres"],
            "spmv_frac_hbm_peak": kern["spmv"]["GBs"] / HBM_PEAK_GBS if "spmv" in kern else None,
            "spmv_bytes": None if "spmv" not in kern else {
                "nnz_pattern": info["spmv"]["nnz_pattern"], "nnz_streamed": info["spmv"]["nnz_streamed"],
                "column_patterns": info["spmv"].get("column_patterns", 0),
                "row_classes": info["spmv"].get("row_classes", 0), "rows_in_classes": info["spmv"].get("rows_in_classes", 0),
                "rows_with_explicit_columns": info["spmv"].get("rows_with_explicit_columns"),
                "streamed_bytes_per_launch": kern["spmv"]["bytes"],
                "parity_csr_bytes_per_launch": 12.0 * nnz + 20.0 * nr,
                "frac_hbm_peak_on_streamed_bytes": kern["spmv"]["GBs"] / HBM_PEAK_GBS,
                "frac_measured_ceiling_on_streamed_bytes": (kern["spmv"]["GBs"] / read_ceiling) if read_ceiling else None,
                "effective_frac_hbm_peak_on_parity_csr_bytes": (12.0 * nnz + 20.0 * nr) / kern["spmv"]["ms_per_launch"] / 1e6 / HBM_PEAK_GBS,
                # the same launch priced on the compacted (value, column) stream it replaces when column patterns are in use
                "effective_frac_hbm_peak_on_compacted_csr_bytes": (12.0 * info["spmv"]["nnz_streamed"] + 20.0 * nr)
                                                                   / kern["spmv"]["ms_per_launch"] / 1e6 / HBM_PEAK_GBS,
                "note": "the solver streams a compacted copy of the owned rows: entries with |a_ij| <= 2^-52 x the row's "
                        "largest magnitude are left out (option spmv_drop_tol; that is the exact zeros -- 8 of the 15 pattern "
                        "entries of an interior Kuhn-cube row, all but the 1 of a Dirichlet row -- and cancellation noise "
                        "below one ulp of the row's largest entry); fedd_spmv, true_relres and the solver's own acceptance "
                        "check multiply with the parity CSR (every stored entry); fractions are quoted on "
                        "the bytes actually streamed, the parity-CSR figure (SURVEY 8d model) is the effective rate; "
                        "with column_patterns > 0 the rows repeat their column offsets and the stream carries a 2-byte "
                        "pattern id per row instead of 4 bytes per entry (values per row, y bit for bit the same); with "
                        "row_classes > 0 the rows also repeat their VALUES bit for bit (a few thousand distinct rows among the "
                        "millions of a structured grid) and the stream carries a 4-byte (class, pattern) word per row, the values come from a "
                        "cache-resident table of at most 16384 classes (fedd_spmv_classes): the kernel is then bound by the "
                        "caches and the gathers of x, not by the HBM stream, and its fraction of the HBM peak on the bytes it "
                        "still streams says little -- the time per launch is the figure"},
            "mesh_generation_s": t_mesh, "mesh_upload_s": t_upload,
            # what a driver that assembles and solves ONCE pays (the reference's drivers do): the cold first step, and inside it
            # the structures built once per mesh by device kernels (wall ms each, device synchronised around them)
            "first_step_ms": cold.get("first_step_ms"),
            "per_mesh_setup_ms": None if "per_mesh_setup" not in cold else
                round(cold["per_mesh_setup"]["adjacency_ms"] + cold["per_mesh_setup"]["tiles_ms"], 3),
            "per_mesh_setup": cold.get("per_mesh_setup"),
        }
        if N > 1:
            out["rccl_preflight"] = preflight
            out["self_check"] = self_check
            out["communication_note"] = ("phases 'halo' (pack kernel, grouped RCCL send/receive, unpack kernel; one before every SpMV "
                                         "and every Schwarz apply) and 'allreduce' are HIP-event times on the library's stream, rank 0, "
                                         "sampled every 8th call; 'allreduce' is also contained in 'ortho' and 'coarse_*'")
        if spmv_b2b is not None:
            out["spmv_back_to_back"] = spmv_b2b
        if two is not None:
            out["two_level_variant"] = two
        if cfg2 is not None:
            out["cfg2_one_gpu"] = cfg2
        out.update(extras)
        if N == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, cells[0], its, full=a.cpu_full)
            out["cpu_baseline"]["host"] = {"cpu_model": host_cpu_model(), "os_cpu_count": os.cpu_count(),
                                           "cores_available_to_this_process": host_cores()}
            if host_cores() > 16 and not a.cpu_full:
                c16 = cpu_baseline(a, cells[0], its, nthr=16)
                out["cpu_baseline"]["with_16_threads"] = {"value": c16["value"], "seconds": c16["seconds"], "cores": c16["cores"]}
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if N > 1 and not rehearse and not a.no_multi_ab:
        # ---- after the one JSON line, to STDERR, under a watchdog: A/B of the multi-GPU switches that cannot be measured on the
        # one-GPU development boxes (ghost import of r behind the interior subdomains, operator applications issued ahead of a
        # block's host round trip, the one-vector solver).  Whatever happens here, the result above has been printed.
        import threading

        def _ab_timeout():
            sys.stderr.write("[bench ab] rank %d: watchdog after 150 s, leaving\n" % rank)
            sys.stderr.flush()
            os._exit(0)

        wd = threading.Timer(150.0, _ab_timeout)
        wd.daemon = True
        wd.start()
        try:
            m = capi.structured_mesh(3, dec, cells, rank, size=dom, ghosts=ghost_layers)
            ab = {}
            for name, opts in (("default", {}), ("halo_overlap=1", {"halo_overlap": 1}), ("gmres_spec=2", {"gmres_spec": 2}),
                               ("halo_overlap=1,gmres_spec=2", {"halo_overlap": 1, "gmres_spec": 2}), ("gmres_kind=0", {"gmres_kind": 0})):
                c = capi.Context(device=dev, rank=rank, nranks=N, nccl_id=nccl_id_ab())
                for k, v in opts.items():
                    c.set_option(k, v)
                c.mesh_set_dict(m)
                c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, cells, m["gid_rep"]))
                c.halo_exchange_setup()
                for _ in range(2):
                    one_step(c, capi, a, False)
                c.sync()
                c.timing_enable(TIMING_STRIDE)
                c.timing_reset()
                barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    its_ab, rel_ab = one_step(c, capi, a, False)
                c.sync()
                barrier()
                dt_ab = max_over_ranks(time.perf_counter() - t0) / 3 * 1e3
                tm_ab = c.timing_get()
                ab[name] = {"ms_per_step": round(dt_ab, 3), "its": its_ab, "relres": rel_ab,
                            "device_ms": {k: round(v[0] / 3, 3) for k, v in tm_ab.items() if v[1]}}
                c.close()
            if rank == 0:
                sys.stderr.write("[bench ab] " + json.dumps({"n_gpus": N, "ab": ab}) + "\n")
                sys.stderr.flush()
        except BaseException as e:      # (never let the A/B change the exit status of a finished measurement)
            sys.stderr.write("[bench ab] rank %d: %r\n" % (rank, e))
            sys.stderr.flush()
            sys.stdout.flush()
            # the other ranks may be waiting in a collective this rank will never join: leave without the orderly shutdown of the
            # process group (which would wait for them); their own watchdogs end them
            os._exit(0)
        wd.cancel()
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
