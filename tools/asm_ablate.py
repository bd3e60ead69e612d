"""Tile assembly kernel at cfg 3 (214^3 cells): time per launch with phases switched off (option asm_dbg: 1 = no element phase,
2 = no gathers, 3 = neither: blob stream + stores alone).  The results of the ablated runs are wrong by construction."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from feddlib_amd import capi
M = int(sys.argv[1]) if len(sys.argv) > 1 else 214
m = capi.structured_mesh(3, (1, 1, 1), [M] * 3, 0)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.pattern_build(1, capi.BLOCK_SCALAR)
c.assemble(capi.FORM_LAPLACE)
for dbg in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "64,0,1,2,3".split(","))]:
    c.set_option("asm_dbg", dbg)
    c.assemble(capi.FORM_LAPLACE)
    c.sync()
    c.timing_enable(1)
    c.timing_reset()
    for _ in range(5):
        c.assemble(capi.FORM_LAPLACE)
    c.sync()
    ms, nl = c.timing_get()["assemble"]
    c.timing_enable(0)
    print("asm_dbg %2d: %.3f ms per launch" % (dbg, ms / nl), flush=True)
c.close()
