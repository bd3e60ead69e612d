"""Host wall time of every C-ABI call of one step at the per-GPU share of the 8-GPU run (107^3 cells, 64-node boxes, held to 145
iterations), device synchronised after each call, against the device timers: where the share's wall-minus-device gap lies
(development aid).  usage: share_walltime.py [key=value,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(os.environ.get("FEDD_SHARE_CELLS", "107"))
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
for kv in filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")):
    c.set_option(kv.split("=")[0], float(kv.split("=")[1]))
c.set_option("gmres_tol_blocks", 0)
c.mesh_set_dict(m)
c.sync()


def timed(name, fn, acc, sync=True):
    t0 = time.perf_counter()
    r = fn()
    t1 = time.perf_counter()
    if sync:
        c.sync()
    t2 = time.perf_counter()
    acc.append((name, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
    return r


for rep in range(6):
    acc = []
    sync = rep < 5          # last repetition: no synchronisation between the calls (the bench's way)
    c.timing_enable(8 if sync else 0)
    c.timing_reset()
    c.sync()
    t0 = time.perf_counter()
    timed("pattern_build", lambda: c.pattern_build(1, capi.BLOCK_SCALAR), acc, sync)
    timed("assemble", lambda: c.assemble(capi.FORM_LAPLACE), acc, sync)
    timed("assemble_rhs", lambda: c.assemble_rhs([1.0]), acc, sync)
    timed("dirichlet", lambda: c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0]), acc, sync)
    c.schwarz_set_target(64, 1.0)
    timed("schwarz_setup", lambda: c.schwarz_setup(1, capi.COMBINE_RESTRICTED), acc, sync)
    r = timed("gmres", lambda: c.gmres(None, rtol=1e-300, max_it=145, restart=100, use_prec=True, want_x=False), acc, sync)
    c.sync()
    tot = (time.perf_counter() - t0) * 1e3
    if rep >= 3:
        print("step %.3f ms, %d its (%s)" % (tot, r[1], "synchronised after every call" if sync else "no synchronisation between calls, timers off"))
        for name, call, done in acc:
            print("   %-26s call returned after %8.3f ms, device idle after %8.3f ms" % (name, call, done))
        if sync:
            tm = c.timing_get()
            print("   device timers:", {k: round(v[0], 3) for k, v in tm.items() if v[1]})
c.close()
