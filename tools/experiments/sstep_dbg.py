import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from feddlib_amd import capi
for M, target in ((12, 27), (16, 27), (48, 64)):
    m = capi.structured_mesh(3, 1, M)
    c = capi.Context(device=0)
    c.mesh_set_dict(m); c.pattern_build(1, capi.BLOCK_SCALAR); c.assemble(capi.FORM_LAPLACE); c.assemble_rhs([1.0])
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0]); c.schwarz_set_target(target, 1.0); c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
    for s in (1, 4, 8):
        c.set_option("gmres_s", s)
        print("M", M, "s", s, c.gmres(None, rtol=1e-13, max_it=600, restart=200, use_prec=True, want_x=False)[1:], flush=True)
    c.close()
