"""cfg-5 share (94^3-cell elasticity) with the RGDSW / GDSW coarse level: setup only, for a kernel trace of the setup.
usage: gdsw_prof.py [rgdsw|gdsw] [M] [steps]   (development aid)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feddlib_amd import capi  # noqa: E402

kind = {"gdsw": capi.COARSE_GDSW, "q1": capi.COARSE_Q1}.get(sys.argv[1] if len(sys.argv) > 1 else "", capi.COARSE_RGDSW)
M = int(sys.argv[2]) if len(sys.argv) > 2 else 94
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m = capi.structured_mesh(3, (1, 1, 1), [M] * 3, 0)
c = capi.Context(device=0)
for kv in os.environ.get("FEDD_OPTIONS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        c.set_option(k, float(v))
c.mesh_set_dict(m)
mu, nu = 2.0e6, 0.4
lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
c.pattern_build(3, capi.BLOCK_FULL)
c.assemble(capi.FORM_LINELAS, [lam, mu])
c.assemble_rhs([0.0, 1.0, 0.0])
c.dirichlet([2], [0.0, 0.0, 0.0])
c.schwarz_set_target(8, 1.0)
if os.environ.get("GD_CELLS"):
    c.schwarz_set_coarse(int(os.environ["GD_CELLS"]))
for s in range(steps):
    c.timing_reset()
    t0 = time.perf_counter()
    c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=kind)
    c.sync()
    t1 = time.perf_counter()
    print("setup %.1f ms" % ((t1 - t0) * 1e3), c.schwarz_coarse_sizes(), flush=True)
c.sync()
t0 = time.perf_counter()
its, rel = c.gmres(None, rtol=1e-6, max_it=2000, restart=100, use_prec=True, want_x=False)[1:]
c.sync()
print("solve %.1f ms" % ((time.perf_counter() - t0) * 1e3), its, rel, {k: round(v[0], 2) for k, v in c.timing_get().items() if v[0] > 0})
