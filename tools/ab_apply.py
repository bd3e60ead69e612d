"""A/B of the restricted Schwarz apply on the structured 3D P1 Laplace cube (development aid).
usage: ab_apply.py [cells] [key=value,... per configuration]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
configs = sys.argv[2:] or ["schwarz_dedupe=1"]
m = capi.structured_mesh(3, 1, M)
c = capi.Context(device=0)
c.mesh_set_dict(m)
c.pattern_build(1, capi.BLOCK_SCALAR)
c.assemble(capi.FORM_LAPLACE)
c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
nr = c.csr_sizes()[0]
r = np.random.default_rng(0).standard_normal(nr)
z0 = None
c.timing_enable(True)
for rep in range(2):
    for cfg in configs:
        target = 27
        for kv in cfg.split(","):
            k, v = kv.split("=")
            if k == "target":       # (not a library option: nodes per box)
                target = int(v)
            else:
                c.set_option(k, float(v))
        c.schwarz_set_target(target, 1.0)
        c.timing_reset()
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
        c.sync()
        ts = c.timing_get()["schwarz_setup"][0]
        z = c.schwarz_apply(r)
        if z0 is None:
            z0 = z
        c.schwarz_apply_device(5)
        c.timing_reset()
        c.schwarz_apply_device(40)
        c.sync()
        t = c.timing_get()["schwarz_apply"]
        info = c.schwarz_info()
        print("M %d %-44s apply %.1f us  setup %.2f ms  unique %d of %d  slabs %.1f MB  diff vs first %.1e"
              % (M, cfg, t[0] / t[1] * 1e3, ts, info["n_unique"], info["n_subdomains"], info["inverse_bytes"] / 1e6,
                 np.abs(z - z0).max() / np.abs(z0).max()), flush=True)
c.close()
