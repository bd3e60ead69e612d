"""Wall time per step of cfg 2 with the library's HIP-event accounting on and off (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from feddlib_amd import capi  # noqa: E402


class A:
    target = 27
    rtol = 1e-8
    max_it = 2000
    restart = 100
    coarse = 0.0


m = capi.structured_mesh(3, 1, 100)
c = capi.Context(device=0)
c.mesh_set_dict(m)
for two in (False, True):
    for on in (0, 1, 8, 0, 1, 8):
        c.timing_enable(on)
        bench.one_step(c, capi, A, two)
        c.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            its, rel = bench.one_step(c, capi, A, two)
        c.sync()
        print("two_level", two, "timing", on, "ms/step", (time.perf_counter() - t0) / 5 * 1e3, "its", its, flush=True)
c.close()
