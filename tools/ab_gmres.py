"""A/B of solver options on the structured 3D P1 Laplace cube: one full solve per configuration, device ms per kernel
class (development aid).  usage: ab_gmres.py [cells] [key=value,... per configuration]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
configs = sys.argv[2:] or ["md2_gy=0"]
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.pattern_build(1, capi.BLOCK_SCALAR)
c.assemble(capi.FORM_LAPLACE)
c.assemble_rhs([1.0])
c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
c.schwarz_set_target(27, 1.0)
c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
for rep in range(2):
    for cfg in configs:
        for kv in cfg.split(","):
            k, v = kv.split("=")
            c.set_option(k, float(v))
        c.gmres(None, rtol=1e-8, max_it=1000, restart=100, use_prec=True, want_x=False)
        c.timing_enable(1)
        c.timing_reset()
        _, its, rel = c.gmres(None, rtol=1e-8, max_it=1000, restart=100, use_prec=True, want_x=False)
        c.sync()
        t = c.timing_get()
        print("M %d %-28s its %d  ortho %.2f ms  gs_dot %.2f (%.1f us/launch)  gs_update %.2f (%.1f us/launch)  spmv %.2f  apply %.2f"
              % (M, cfg, its, t["ortho"][0], t["gs_dot"][0], 1e3 * t["gs_dot"][0] / max(1, t["gs_dot"][1]), t["gs_update"][0],
                 1e3 * t["gs_update"][0] / max(1, t["gs_update"][1]), t["spmv"][0], t["schwarz_apply"][0]), flush=True)
c.close()
