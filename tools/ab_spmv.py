"""A/B of SpMV kernel variants on the structured 3D P1 Laplace matrix (development aid).
usage: ab_spmv.py [cells per direction] [key=value,key=value ...]   (each argument = one configuration of options)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
configs = sys.argv[2:] or ["spmv_var=0"]
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.pattern_build(1, capi.BLOCK_SCALAR)
c.assemble(capi.FORM_LAPLACE)
c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
nr, _, nnz = c.csr_sizes()
x = np.random.default_rng(0).standard_normal(nr)
ceiling = c.read_bandwidth(2 << 30, 10)
print("read ceiling %.0f GB/s" % ceiling, flush=True)
y0 = None
c.timing_enable(True)
for rep in range(2):
    for cfg in configs:
        for kv in cfg.split(","):
            k, v = kv.split("=")
            c.set_option(k, float(v))
        y = c.spmv(x)
        if y0 is None:
            y0 = y
        c.spmv_device(5)
        c.timing_reset()
        c.spmv_device(50)
        c.sync()
        t = c.timing_get()["spmv"]
        ms = t[0] / t[1]
        si = c.spmv_info()
        if si["column_patterns"]:    # values + pattern id and row pointer per row + explicit columns of the rows without a pattern
            b = 8.0 * si["nnz_streamed"] + 22.0 * nr + 4.0 * si["nnz_streamed"] * si["rows_with_explicit_columns"] / nr
        else:
            b = 12.0 * si["nnz_streamed"] + 20.0 * nr
        err = np.abs(y - y0).max() / np.abs(y0).max()
        print("M %d %-40s %.2f us  %.0f GB/s streamed (%.3f of 8 TB/s, %.3f of ceiling)  diff vs first: %.1e  patterns %d explicit rows %d  classes %d rows in classes %d"
              % (M, cfg, ms * 1e3, b / ms / 1e6, b / ms / 8e9, b / ms / 1e6 / ceiling, err, si["column_patterns"],
                 si["rows_with_explicit_columns"], si.get("row_classes", 0), si.get("rows_in_classes", 0)), flush=True)
c.close()
