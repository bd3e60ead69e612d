import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from feddlib_amd import capi
m = capi.structured_mesh(3, 1, 12)
c = capi.Context(device=0)
c.mesh_set_dict(m); c.pattern_build(1, capi.BLOCK_SCALAR); c.assemble(capi.FORM_LAPLACE); c.assemble_rhs([1.0])
c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0]); c.schwarz_set_target(8, 1.0); c.schwarz_set_coarse(27)
for kind in (0, 2):
    for ck in (capi.COARSE_GDSW, capi.COARSE_RGDSW):
        c.set_option("gmres_kind", kind)
        try:
            c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=ck)
            print(kind, ck, "setup ok", c.gmres(None, rtol=1e-8, max_it=200, restart=100, use_prec=True, want_x=False)[1:])
        except capi.FeddError as e:
            print(kind, ck, "FAILED", str(e)[:100])
