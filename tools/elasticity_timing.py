"""cfg 5's problem (3D P1 linear elasticity, steadyLinElas_Perf parameters) on one GPU: one- and two-level
solve, iteration counts and phase times (development aid)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 48
target = int(sys.argv[2]) if len(sys.argv) > 2 else 8
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
mu, nu = 2.0e6, 0.4
lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
c.timing_enable(True)
KINDS = {0: None, 1: capi.COARSE_Q1, 2: capi.COARSE_GDSW, 3: capi.COARSE_RGDSW}
NAMES = {0: "one level", 1: "Q1", 2: "GDSW", 3: "RGDSW"}
for two in ([int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else (1, 1, 3, 3, 2, 2)):
    c.timing_reset()
    c.sync()
    t0 = time.perf_counter()
    nnz = c.pattern_build(3, capi.BLOCK_FULL)
    c.assemble(capi.FORM_LINELAS, [lam, mu])
    c.assemble_rhs([0.0, 1.0, 0.0])
    c.dirichlet([2], [0.0, 0.0, 0.0])
    c.schwarz_set_target(target, 1.0)
    if two:
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=KINDS[two])
    else:
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
    _, its, rel = c.gmres(None, rtol=1e-6, max_it=1000, restart=100, use_prec=True, want_x=False)
    c.sync()
    dt = time.perf_counter() - t0
    tm = c.timing_get()
    n = 3 * m["gid_uni"].shape[0]
    print(json.dumps(dict(M=M, dofs=n, nnz=nnz, coarse=NAMES[two], coarse_dofs=(int(c.schwarz_coarse_sizes()[1]) if two else 0), its=its, relres=rel, ms=dt * 1e3, MDoFs=n / dt / 1e6,
                          dev_ms={k: round(v[0], 3) for k, v in tm.items()}, schwarz=c.schwarz_info())), flush=True)
c.close()
