"""A/B of the restricted Schwarz apply kernels on cfg 2 (development aid)."""
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
c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
info = c.schwarz_info()
r = np.random.default_rng(0).standard_normal(m["gid_uni"].shape[0])
zs = {}
c.timing_enable(True)
for kind in (1, 0, 1, 0):
    c.set_option("apply_kind", kind)
    zs[kind] = c.schwarz_apply(r)
    c.schwarz_apply_device(5)
    c.timing_reset()
    c.schwarz_apply_device(50)
    c.sync()
    t = c.timing_get()["schwarz_apply"]
    ms = t[0] / t[1]
    print("apply_kind", kind, "ms", ms, "GB/s", (info["inverse_bytes"] + 24 * r.shape[0]) / ms / 1e6, flush=True)
print("max |z0 - z1| / max|z|", np.abs(zs[0] - zs[1]).max() / np.abs(zs[1]).max())
c.close()
