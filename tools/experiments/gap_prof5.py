"""cfg 5's share with the Q1 coarse level: wall against device time per API call (where the host waits).  (development aid)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 94
m = capi.structured_mesh(3, (1, 1, 1), [M] * 3, 0)
c = capi.Context(device=0)
c.mesh_set_dict(m)
mu, nu = 2.0e6, 0.4
lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
c.timing_enable(1)


def step(report):
    calls = [("pattern_build", lambda: c.pattern_build(3, capi.BLOCK_FULL)), ("assemble", lambda: c.assemble(capi.FORM_LINELAS, [lam, mu])),
             ("assemble_rhs", lambda: c.assemble_rhs([0.0, 1.0, 0.0])), ("dirichlet", lambda: c.dirichlet([2], [0.0, 0.0, 0.0])),
             ("set_target", lambda: c.schwarz_set_target(8, 1.0)),
             ("schwarz_setup", lambda: c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)),
             ("gmres", lambda: c.gmres(None, rtol=1e-6, max_it=2000, restart=100, use_prec=True, want_x=False))]
    tw = td = 0.0
    for name, f in calls:
        c.timing_reset()
        c.sync()
        t0 = time.perf_counter()
        f()
        c.sync()
        w = (time.perf_counter() - t0) * 1e3
        tm = c.timing_get()
        d = sum(v[0] for k, v in tm.items() if k not in ("gs_dot", "gs_update") and not k.startswith("_"))
        tw += w
        td += d
        if report:
            print("%-14s wall %7.3f ms  device %7.3f ms  gap %6.3f   %s" % (name, w, d, w - d, {k: round(v[0], 2) for k, v in tm.items() if v[0] > 0.005 and not k.startswith("_")}), flush=True)
    if report:
        print("%-14s wall %7.3f ms  device %7.3f ms  gap %6.3f" % ("step", tw, td, tw - td))


for i in range(3):
    step(i == 2)
