"""A few bench steps at a given grid (for kernel traces).  usage: steps214.py [cells] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 214
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
for _ in range(steps):
    c.pattern_build(1, capi.BLOCK_SCALAR)
    c.assemble(capi.FORM_LAPLACE)
    c.assemble_rhs([1.0])
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    c.schwarz_set_target(64, 1.0)
    c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
    print(c.gmres(None, rtol=1e-8, max_it=2000, restart=100, use_prec=True, want_x=False)[1:], flush=True)
c.close()
