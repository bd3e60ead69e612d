"""Tolerance of GDSW's interior extension solves against setup time and outer iteration count (cfg 5's share: 3D P1 linear
elasticity, steadyLinElas_Perf parameters).  usage: gdsw_tol_sweep.py [cells] [kind 2=GDSW 3=RGDSW] [tol,tol,...] [coarse cells]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 48
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tols = [float(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1e-10, 1e-6, 1e-4, 1e-3]
cells = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0        # coarse lattice cells (0 = library default)
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
mu, nu = 2.0e6, 0.4
lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
c.pattern_build(3, capi.BLOCK_FULL)
c.assemble(capi.FORM_LINELAS, [lam, mu])
c.assemble_rhs([0.0, 1.0, 0.0])
c.dirichlet([2], [0.0, 0.0, 0.0])
c.schwarz_set_target(8, 1.0)
c.timing_enable(True)
for tol in tols:
    c.set_option("gdsw_tol", tol)
    c.schwarz_set_coarse(cells)
    c.timing_reset()
    c.sync()
    t0 = time.perf_counter()
    c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=kind)
    c.sync()
    t1 = time.perf_counter()
    _, its, rel = c.gmres(None, rtol=1e-6, max_it=1000, restart=100, use_prec=True, want_x=False)
    c.sync()
    t2 = time.perf_counter()
    print("M %d kind %d gdsw_tol %.0e: setup %.1f ms, solve %.1f ms, outer iterations %d, relres %.2e, coarse dofs %d"
          % (M, kind, tol, (t1 - t0) * 1e3, (t2 - t1) * 1e3, its, rel, int(c.schwarz_coarse_sizes()[1])), flush=True)
c.close()
