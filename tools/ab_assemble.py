"""A/B of the matrix assembly kernels on the structured 3D P1 cube (development aid).
usage: ab_assemble.py [cells] [laplace|linelas] [key=value,... per configuration]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prob = sys.argv[2] if len(sys.argv) > 2 else "laplace"
configs = sys.argv[3:] or ["asm_kind=0"]
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
if prob == "laplace":
    nnz = c.pattern_build(1, capi.BLOCK_SCALAR)
    form, par = capi.FORM_LAPLACE, None
else:
    nnz = c.pattern_build(3, capi.BLOCK_FULL)
    form, par = capi.FORM_LINELAS, [2.0e6 * 2 * 0.4 / 0.2, 2.0e6]
nr = c.csr_sizes()[0]
bytes_alg = 4.0 * m["conn"].size + 8.0 * 3 * m["xyz"].shape[0] + 12.0 * nnz + 4.0 * (nr + 1)
ref = None
c.timing_enable(True)
for rep in range(2):
    for cfg in configs:
        for kv in cfg.split(","):
            k, v = kv.split("=")
            c.set_option(k, float(v))
        c.assemble(form, par)
        val = c.csr_get()[2]
        if ref is None:
            ref = val
        c.timing_reset()
        for _ in range(10):
            c.assemble(form, par)
        c.sync()
        t = c.timing_get()["assemble"]
        ms = t[0] / t[1]
        print("M %d %s %-32s %.3f ms  %.0f GB/s algorithmic (%.3f of 8 TB/s)  max diff vs first %.2e  bitwise %s"
              % (M, prob, cfg, ms, bytes_alg / ms / 1e6, bytes_alg / ms / 8e9, np.abs(val - ref).max() / np.abs(ref).max(),
                 np.array_equal(val, ref)), flush=True)
c.close()
