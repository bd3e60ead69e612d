"""iteration counts: one level vs Q1 vs GDSW coarse level at fixed H/h (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feddlib_amd import capi
c = capi.Context(device=0)
for per_dir, hh in ((2, 8), (4, 8), (6, 8), (8, 8), (4, 16), (6, 16)):
    M = per_dir * hh
    m = capi.structured_mesh(3, 1, M)
    c.mesh_set_dict(m)
    c.pattern_build(1, capi.BLOCK_SCALAR)
    c.assemble(capi.FORM_LAPLACE)
    c.assemble_rhs([1.0])
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    c.schwarz_set_target(27, 1.0)
    c.schwarz_set_coarse(per_dir ** 3)
    out = {}
    for name, kw in (("one", dict()), ("q1", dict(two_level=1, coarse_kind=capi.COARSE_Q1)), ("gdsw", dict(two_level=1, coarse_kind=capi.COARSE_GDSW)), ("rgdsw", dict(two_level=1, coarse_kind=capi.COARSE_RGDSW))):
        t0 = time.time()
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, **kw)
        c.sync()
        ts = time.time() - t0
        _, its, rel = c.gmres(None, rtol=1e-8, max_it=500, restart=100, use_prec=True)
        out[name] = (its, round(ts, 3))
    print("M", M, "cells/dir", per_dir, "H/h", hh, out, flush=True)
