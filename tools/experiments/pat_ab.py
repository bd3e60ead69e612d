"""node-pattern merge: hashed kernel against the ordered-insertion kernel (option "pat_hash"): same CSR pattern, symbolic time.
usage: pat_ab.py [M]   (development aid)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 214
c = capi.Context(device=0)
out = {}
for name, m in (("cube", capi.structured_mesh(3, 1, M)), ("cyl", capi.read_mesh("tests/golden/DFG3DCylinder_6k.mesh", 3)),
                ("square", capi.structured_mesh(2, 1, 300))):
    c.mesh_set_dict(m)
    for h in (0, 1, 0, 1):
        c.set_option("pat_hash", h)
        c.timing_reset()
        c.sync()
        t0 = time.perf_counter()
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.sync()
        dt = (time.perf_counter() - t0) * 1e3
        rp, ci, _, _ = c.csr_get()
        out[(name, h)] = (rp.copy(), ci.copy())
        print(name, "pat_hash", h, "pattern_build %.2f ms" % dt, "nnz", ci.shape[0], flush=True)
    assert np.array_equal(out[(name, 0)][0], out[(name, 1)][0]) and np.array_equal(out[(name, 0)][1], out[(name, 1)][1]), name
print("patterns identical")
