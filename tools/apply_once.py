"""One Schwarz setup and a few applies on the structured cube with the bench's 64-node boxes (for rocprofv3 --pmc passes of the
apply kernels).  usage: apply_once.py [cells] [apply_kind] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 214
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.pattern_build(1, capi.BLOCK_SCALAR)
c.assemble(capi.FORM_LAPLACE)
c.assemble_rhs([1.0])
c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
c.set_option("apply_kind", kind)
c.schwarz_set_target(64, 1.0)
c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
c.schwarz_apply_device(reps)
c.sync()
c.close()
