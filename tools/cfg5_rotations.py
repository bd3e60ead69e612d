"""cfg 5's share of one GPU (94^3-cell elasticity, steadyLinElas_Perf parameters) with and without the rotations in the coarse
space: iterations, ms per step (assembly + setup + solve), coarse dofs.  usage: python tools/cfg5_rotations.py [M] [kinds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from feddlib_amd import capi

M = int(sys.argv[1]) if len(sys.argv) > 1 else 94
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["rgdsw", "rgdsw+rot", "gdsw", "gdsw+rot"]
cells = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
m = capi.structured_mesh(3, (1, 1, 1), [M] * 3, 0)
c = capi.Context(device=0)
c.mesh_set_dict(m)
mu, nu = 2.0e6, 0.4
lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
for name in kinds:
    kind = capi.COARSE_RGDSW if name.startswith("rgdsw") else capi.COARSE_GDSW
    c.set_option("gdsw_rotations", 1 if name.endswith("+rot") else 0)
    c.schwarz_set_coarse(cells)
    def step():
        c.pattern_build(3, capi.BLOCK_FULL)
        c.assemble(capi.FORM_LINELAS, [lam, mu])
        c.assemble_rhs([0.0, 1.0, 0.0])
        c.dirichlet([2], [0.0, 0.0, 0.0])
        c.schwarz_set_target(8, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=kind)
        return c.gmres(None, rtol=1e-6, max_it=2000, restart=100, use_prec=True, want_x=False)[1:]
    step()
    c.sync()
    c.timing_enable(1)
    c.timing_reset()
    t0 = time.perf_counter()
    its, rel = step()
    c.sync()
    wall = (time.perf_counter() - t0) * 1e3
    tm = c.timing_get()
    c.timing_enable(0)
    g, n0 = c.schwarz_coarse_sizes()
    print("%-10s cells %s coarse dofs %5d  iterations %4d relres %.2e  step %8.1f ms  coarse_setup %.1f coarse_apply %.1f ortho %.1f apply %.1f spmv %.1f"
          % (name, tuple(int(v) for v in g), n0, its, rel, wall, tm["coarse_setup"][0], tm["coarse_apply"][0], tm["ortho"][0],
             tm["schwarz_apply"][0], tm["spmv"][0]), flush=True)
c.close()
