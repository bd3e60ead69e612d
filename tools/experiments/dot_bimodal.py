"""is the time of the dot sweep tied to where the Krylov basis lies?  several contexts in one process: sweep time + basis address
(FEDD_GMRES_DEBUG=1 prints it).  (development aid)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 214
m = capi.structured_mesh(3, 1, M)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    c = capi.Context(device=0)
    c.mesh_set_dict(m)
    c.pattern_build(1, capi.BLOCK_SCALAR)
    c.assemble(capi.FORM_LAPLACE)
    c.assemble_rhs([1.0])
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    c.schwarz_set_target(64, 1.0)
    c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
    for it in range(2):
        c.timing_enable(8)
        c.timing_reset()
        c.gmres(None, rtol=1e-8, max_it=1000, restart=100, use_prec=True, want_x=False)
        tm = c.timing_get()
        c.timing_enable(0)
    print("context", rep, "gs_dot %.4f ms  gs_update %.4f ms  spmv %.4f" % tuple(tm[k][0] / max(tm[k][1], 1) for k in ("gs_dot", "gs_update", "spmv")), flush=True)
    c.close()
