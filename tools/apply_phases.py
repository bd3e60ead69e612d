import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from feddlib_amd import capi
M = 214
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.pattern_build(1, capi.BLOCK_SCALAR)
c.assemble(capi.FORM_LAPLACE)
c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
nr = c.csr_sizes()[0]
r = np.random.default_rng(0).standard_normal(nr)
c.schwarz_set_target(64, 1.0)
c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
for span in (0,):
    c.set_option("apply_span", span)
    c.set_option("apply_dbg", 0)
    c.schwarz_apply_device(3)
    c.sync()
    print("span", span, flush=True)
    c.set_option("apply_dbg", -1)
    c.schwarz_apply_device(1)
    c.sync()
c.close()
