"""wall time against device time per API call of one step (where the host waits): usage gap_prof.py [M] [target]  (development aid)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 107
target = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)


def step(report):
    calls = [("pattern_build", lambda: c.pattern_build(1, capi.BLOCK_SCALAR)), ("assemble", lambda: c.assemble(capi.FORM_LAPLACE)),
             ("assemble_rhs", lambda: c.assemble_rhs([1.0])), ("dirichlet", lambda: c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])),
             ("set_target", lambda: c.schwarz_set_target(target, 1.0)),
             ("schwarz_setup", lambda: c.schwarz_setup(1, capi.COMBINE_RESTRICTED)),
             ("gmres", lambda: c.gmres(None, rtol=1e-8, max_it=1000, restart=100, use_prec=True, want_x=False))]
    tot_w = tot_d = 0.0
    for name, f in calls:
        c.timing_reset()
        c.sync()
        t0 = time.perf_counter()
        f()
        c.sync()
        w = (time.perf_counter() - t0) * 1e3
        tm = c.timing_get()
        d = sum(v[0] for k, v in tm.items() if k not in ("gs_dot", "gs_update"))
        tot_w += w
        tot_d += d
        if report:
            print("%-14s wall %7.3f ms  device %7.3f ms  gap %6.3f" % (name, w, d, w - d), flush=True)
    if report:
        print("%-14s wall %7.3f ms  device %7.3f ms  gap %6.3f" % ("step", tot_w, tot_d, tot_w - tot_d))


for i in range(3):
    step(i == 2)
