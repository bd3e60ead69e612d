"""One-level Schwarz on the headline grid: overlap x box size against ms per assemble + solve step.
usage: sweep_one_level.py [cells per direction] [overlap:box:big[:restart],...] ; writes one line per configuration"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 214
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
n = m["n_global"]
del m


def step(overlap, target, big, restart=100):
    c.set_option("schwarz_big", 1 if big else 0)
    if big:
        c.set_option("schwarz_big_target", target)
    c.pattern_build(1, capi.BLOCK_SCALAR)
    c.assemble(capi.FORM_LAPLACE)
    c.assemble_rhs([1.0])
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    c.schwarz_set_target(target, 1.0)
    c.schwarz_setup(overlap, capi.COMBINE_RESTRICTED)
    return c.gmres(None, rtol=1e-8, max_it=2000, restart=restart, use_prec=True, want_x=False)


CONFIGS = ((1, 27, 0), (1, 8, 0), (1, 64, 0), (2, 8, 0), (2, 27, 0), (1, 125, 1), (1, 343, 1), (2, 125, 1))
if len(sys.argv) > 2:       # "overlap:box:big,..." picks other configurations
    CONFIGS = tuple(tuple(int(v) for v in t.split(":")) for t in sys.argv[2].split(","))
for cfg in CONFIGS:
    overlap, target, big = cfg[:3]
    restart = cfg[3] if len(cfg) > 3 else 100
    try:
        step(overlap, target, big, restart)                  # warm-up (allocations)
        c.sync()
        c.timing_enable(8)
        c.timing_reset()
        t0 = time.perf_counter()
        _, its, rel = step(overlap, target, big, restart)
        c.sync()
        dt = time.perf_counter() - t0
        tm = c.timing_get()
        info = c.schwarz_info()
        print("cells %d overlap %d box %4d %s restart %3d: %7.1f ms/step  %4d its  relres %.1e  subdomains %7d  largest %4d  slabs %6.2f GB | "
              "device ms: setup %.1f apply %.1f spmv %.1f ortho %.1f"
              % (M, overlap, target, "bisection" if big else "lattice  ", restart, dt * 1e3, its, rel, info["n_subdomains"], info["max_size"],
                 info["inverse_bytes"] / 1e9, tm["schwarz_setup"][0], tm["schwarz_apply"][0], tm["spmv"][0], tm["ortho"][0]), flush=True)
    except capi.FeddError as e:
        print("cells %d overlap %d box %d %s: refused: %s" % (M, overlap, target, "bisection" if big else "lattice", str(e)[:160]), flush=True)
c.close()
