"""s-step GMRES (gmres_kind 2) against DCGS2 (0) on the GPU: iteration counts, true residuals, solution difference."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from feddlib_amd import capi


def problem(M, target=64):
    m = capi.structured_mesh(3, 1, M)
    c = capi.Context(device=0)
    c.mesh_set_dict(m)
    c.pattern_build(1, capi.BLOCK_SCALAR)
    c.assemble(capi.FORM_LAPLACE)
    c.assemble_rhs([1.0])
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    c.schwarz_set_target(target, 1.0)
    c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
    return c


def true_rel(c):
    x, b = c.solution_get(), c.rhs_get()
    return float(np.linalg.norm(b - c.spmv(x)) / np.linalg.norm(b)), x


for M in [int(v) for v in (sys.argv[1:] or ["24", "48"])]:
    c = problem(M)
    for rtol, restart in ((1e-8, 100), (1e-13, 100), (1e-10, 10), (1e-8, 7)):
        ref = None
        for kind, s in ((0, 0), (2, 1), (2, 2), (2, 3), (2, 4), (2, 5), (2, 8)):
            c.set_option("gmres_kind", kind)
            if s:
                c.set_option("gmres_s", s)
            t0 = time.perf_counter()
            _, its, rel = c.gmres(None, rtol=rtol, max_it=2000, restart=restart, use_prec=True, want_x=False)
            c.sync()
            dt = time.perf_counter() - t0
            tr, x = true_rel(c)
            if ref is None:
                ref = x
            print("M=%d rtol %g restart %d kind %d s %d: its %d relres %.3e true %.3e |x - x_dcgs2|/|x| %.2e  %.1f ms %s"
                  % (M, rtol, restart, kind, s, its, rel, tr, np.abs(x - ref).max() / np.abs(ref).max(), dt * 1e3,
                     c.gmres_info() if kind == 2 else ""), flush=True)
    c.close()
