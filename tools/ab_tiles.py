"""Element-major tile assembly (asm_kind 4) against the pair kernels (asm_kind 0 with asm_tiles 0): values, reproducibility, time.
usage: ab_tiles.py [cells of the big Laplace cube] [cells of the big elasticity cube]   (development aid)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def assemble(c, form, dofs, mode, params=None, kind=0):
    c.set_option("asm_kind", kind)
    c.pattern_build(dofs, mode)
    c.assemble(form, params)
    return c.csr_get()[2].copy()


def compare(name, m, form, dofs, mode, params=None):
    c = capi.Context(device=0)
    c.mesh_set_dict(m)
    v0 = assemble(c, form, dofs, mode, params, 0)
    v4 = assemble(c, form, dofs, mode, params, 4)
    v4b = assemble(c, form, dofs, mode, params, 4)
    scale = np.abs(v0).max()
    print("%-44s nnz %9d  max |tiles - pairs| / max|a| %.2e  identical %s  reproducible %s" % (
        name, v0.shape[0], np.abs(v4 - v0).max() / scale, bool(np.array_equal(v4, v0)), bool(np.array_equal(v4, v4b))), flush=True)
    c.close()


for dim, M in ((3, 7), (3, 12), (2, 20)):
    m = capi.structured_mesh(dim, 1, M)
    compare("Laplace %dD M=%d" % (dim, M), m, capi.FORM_LAPLACE, 1, capi.BLOCK_SCALAR)
    compare("Laplace vec %dD M=%d" % (dim, M), m, capi.FORM_LAPLACE_VEC, dim, capi.BLOCK_DIAG)
    compare("elasticity %dD M=%d" % (dim, M), m, capi.FORM_LINELAS, dim, capi.BLOCK_FULL, [1.5, 1.0])
mc = capi.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)
compare("Laplace cylinder 1k", mc, capi.FORM_LAPLACE, 1, capi.BLOCK_SCALAR)
compare("elasticity cylinder 1k", mc, capi.FORM_LINELAS, 3, capi.BLOCK_FULL, [1.5, 1.0])

big = [(int(sys.argv[1]) if len(sys.argv) > 1 else 100, capi.FORM_LAPLACE, 1, capi.BLOCK_SCALAR, None, "Laplace"),
       (int(sys.argv[2]) if len(sys.argv) > 2 else 48, capi.FORM_LINELAS, 3, capi.BLOCK_FULL, [8.0e6, 2.0e6], "elasticity")]
for M, form, dofs, mode, params, name in big:
    m = capi.structured_mesh(3, 1, M)
    c = capi.Context(device=0)
    c.mesh_set_dict(m)
    for kind in (0, 4, 0, 4):
        c.set_option("asm_kind", kind)
        t0 = time.perf_counter()
        c.pattern_build(dofs, mode)
        c.assemble(form, params)
        c.sync()
        first = time.perf_counter() - t0
        c.timing_enable(1)
        c.timing_reset()
        for _ in range(3):
            c.assemble(form, params)
        c.sync()
        ms = c.timing_get()["assemble"][0] / 3
        c.timing_enable(0)
        print("%s %d^3 cells asm_kind %d: %.3f ms per assembly (first call incl. pattern and structures %.2f s)" % (name, M, kind, ms, first), flush=True)
    c.close()
