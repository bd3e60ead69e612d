import sys, time
sys.path.insert(0, "/root/repo")
from feddlib_amd import capi
m = capi.structured_mesh(3, 1, 214)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.set_option("asm_kind", 4)
c.pattern_build(1, capi.BLOCK_SCALAR)
for dbg in (64, 1, 2, 3):
    c.set_option("asm_dbg", dbg)
    c.assemble(capi.FORM_LAPLACE)
    c.sync()
    c.timing_enable(1); c.timing_reset()
    for _ in range(3): c.assemble(capi.FORM_LAPLACE)
    c.sync()
    print("dbg", dbg, "%.3f ms" % (c.timing_get()["assemble"][0] / 3), flush=True)
    c.timing_enable(0)
