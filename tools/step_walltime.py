"""Wall time of every C-ABI call of one bench step (host clock, device synchronised after each call): shows where a step
spends time that the device timers do not see (development aid).  usage: step_walltime.py [cells] [two_level 0|1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
two = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.sync()


def timed(name, fn, acc):
    t0 = time.perf_counter()
    r = fn()
    t1 = time.perf_counter()
    c.sync()
    t2 = time.perf_counter()
    acc.append((name, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
    return r


for rep in range(4):
    acc = []
    c.timing_enable(8)
    c.timing_reset()
    t0 = time.perf_counter()
    timed("pattern_build", lambda: c.pattern_build(1, capi.BLOCK_SCALAR), acc)
    timed("assemble", lambda: c.assemble(capi.FORM_LAPLACE), acc)
    timed("assemble_rhs", lambda: c.assemble_rhs([1.0]), acc)
    timed("dirichlet", lambda: c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0]), acc)
    c.schwarz_set_target(27, 1.0)
    if two:
        c.schwarz_set_coarse(0.0)
        timed("schwarz_setup(two_level)", lambda: c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1), acc)
    else:
        timed("schwarz_setup", lambda: c.schwarz_setup(1, capi.COMBINE_RESTRICTED), acc)
    r = timed("gmres", lambda: c.gmres(None, rtol=1e-8, max_it=1000, restart=100, use_prec=True, want_x=False), acc)
    tot = (time.perf_counter() - t0) * 1e3
    if rep >= 2:
        print("step %.2f ms, %d its" % (tot, r[1]))
        for name, call, done in acc:
            print("   %-26s call returned after %8.3f ms, device idle after %8.3f ms" % (name, call, done))
        tm = c.timing_get()
        print("   device timers:", {k: round(v[0], 3) for k, v in tm.items() if v[1]})
c.close()
