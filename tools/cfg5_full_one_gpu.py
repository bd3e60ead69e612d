"""BASELINE cfg 5 in full on ONE MI355X: 3D P1 linear elasticity on the 189^3-node cube (188^3 cells, 20 253 807 dofs, 904 M matrix
entries), steadyLinElas_Perf parameters (mu 2e6, nu 0.4, f = (0, 1, 0), Dirichlet on flag 2, rtol 1e-6, restart 100), 8-node boxes,
two-level Schwarz with the Q1 lattice space and with RGDSW.  usage: cfg5_full_one_gpu.py [q1|rgdsw|both] [cells]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from feddlib_amd import capi  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 188
t0 = time.perf_counter()
m = capi.structured_mesh(3, (1, 1, 1), [M] * 3, 0)
print("mesh %.1f s" % (time.perf_counter() - t0), flush=True)
c = capi.Context(device=0)
bench.env_options(c)
c.mesh_set_dict(m)
mu, nu = 2.0e6, 0.4
lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
n = 3 * m["n_global"]
kinds = [("q1", capi.COARSE_Q1), ("rgdsw", capi.COARSE_RGDSW)]
for name, kind in kinds:
    if which not in ("both", name):
        continue

    def step():
        c.pattern_build(3, capi.BLOCK_FULL)
        c.assemble(capi.FORM_LINELAS, [lam, mu])
        c.assemble_rhs([0.0, 1.0, 0.0])
        c.dirichlet([2], [0.0, 0.0, 0.0])
        c.schwarz_set_target(8, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=kind)
        return c.gmres(None, rtol=1e-6, max_it=2000, restart=100, use_prec=True, want_x=False)[1:]

    wall, (its, rel), tm = bench.timed_passes(c, step, 1, 1)
    x, b = c.solution_get(), c.rhs_get()
    true_rel = float(np.linalg.norm(b - c.spmv(x)) / np.linalg.norm(b))
    free, total = c.mem_info() if hasattr(c, "mem_info") else (None, None)
    print(json.dumps({"coarse": name, "cells": M, "dofs": n, "nnz": int(c.csr_sizes()[2]), "ms_per_step": wall, "MDoF_per_s": n / wall * 1e-3,
                      "gmres_iterations": its, "relres": rel, "true_relres": true_rel, "coarse_dofs": int(c.schwarz_coarse_sizes()[1]),
                      "phases_device_ms": bench.phases(tm, 1)}), flush=True)
