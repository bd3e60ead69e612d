"""The cold first step on a mesh (per-mesh structures: adjacency, tile structures) -- for a kernel trace of the build kernels.
usage: first_step.py [cells]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 214
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.sync()
for rep in range(2):
    t0 = time.perf_counter()
    c.pattern_build(1, capi.BLOCK_SCALAR)
    c.sync()
    t1 = time.perf_counter()
    c.assemble(capi.FORM_LAPLACE)
    c.sync()
    t2 = time.perf_counter()
    print("pass %d: pattern_build %.2f ms, assemble %.2f ms, %r" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, c.mesh_setup_info()), flush=True)
    if rep == 0:
        c.mesh_set_dict(m)      # the same mesh again: the structures are rebuilt, the buffers are there already
        c.sync()
c.close()
