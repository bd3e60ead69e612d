"""A/B of the SpMV kernels (`spmv_kind`) on the structured 3D P1 Laplace matrix (development aid).
usage: ab_spmv.py [cells per direction] [kinds ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kinds = [int(v) for v in sys.argv[2:]] or [0, 1, 2]
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.pattern_build(1, capi.BLOCK_SCALAR)
c.assemble(capi.FORM_LAPLACE)
c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
nr, _, nnz = c.csr_sizes()
x = np.random.default_rng(0).standard_normal(nr)
ys = {}
c.timing_enable(True)
for rep in range(2):
    for kind in kinds:
        c.set_option("spmv_kind", kind)
        ys[kind] = c.spmv(x)
        c.spmv_device(5)
        c.timing_reset()
        c.spmv_device(50)
        c.sync()
        t = c.timing_get()["spmv"]
        ms = t[0] / t[1]
        err = np.abs(ys[kind] - ys[kinds[0]]).max() / np.abs(ys[kinds[0]]).max()
        print("M %d spmv_kind %d  %.2f us  %.0f GB/s  (%.3f of 8 TB/s)  diff vs kind %d: %.1e"
              % (M, kind, ms * 1e3, (12.0 * nnz + 20.0 * nr) / ms / 1e6, (12.0 * nnz + 20.0 * nr) / ms / 8e9, kinds[0], err),
              flush=True)
c.close()
