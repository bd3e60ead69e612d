"""Setup-phase kernels of cfg 2 in isolation, for rocprofv3 (development aid)."""
import os
import sys

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
c.timing_reset()
for _ in range(3):
    c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
c.sync()
t = c.timing_get()["schwarz_setup"]
print("schwarz_setup ms", t[0] / t[1], flush=True)
c.timing_reset()
for _ in range(5):
    c.assemble_rhs([1.0])
c.sync()
t = c.timing_get()["rhs"]
print("rhs ms", t[0] / t[1], flush=True)
c.close()
