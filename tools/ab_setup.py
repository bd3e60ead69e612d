"""Setup-phase kernels of cfg 2 in isolation: local-inverse kernels A/B (development aid)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.pattern_build(1, capi.BLOCK_SCALAR)
c.assemble(capi.FORM_LAPLACE)
c.assemble_rhs([1.0])
c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
c.timing_enable(True)
r = np.random.default_rng(0).standard_normal(m["gid_uni"].shape[0])
z = {}
for kind in (2, 0, 2, 0):
    c.set_option("inv_kind", kind)
    c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
    c.timing_reset()
    for _ in range(3):
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
    c.sync()
    t = c.timing_get()["schwarz_setup"]
    print("inv_kind", kind, "schwarz_setup ms", t[0] / t[1], flush=True)
    z[kind] = c.schwarz_apply(r)
print("max |z0 - z1| / max |z|", np.abs(z[0] - z[2]).max() / np.abs(z[2]).max())
c.close()
