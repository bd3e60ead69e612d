"""How many bitwise-distinct rows (column offsets + values, sub-ulp entries dropped as the solver's compaction does) does the
assembled Laplace matrix of the structured cube have?  (development: sizing of a value dictionary for the SpMV)
usage: row_classes.py cells..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feddlib_amd import capi  # noqa: E402

for M in [int(a) for a in sys.argv[1:]] or [107]:
    m = capi.structured_mesh(3, 1, M)
    c = capi.Context(device=0)
    c.mesh_set_dict(m)
    c.pattern_build(1, capi.BLOCK_SCALAR)
    c.assemble(capi.FORM_LAPLACE)
    c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    rowptr, col, val, gid = c.csr_get()
    c.close()
    n = rowptr.shape[0] - 1
    lens = np.diff(rowptr)
    full = np.nonzero(lens == 15)[0]
    idx = rowptr[full][:, None] + np.arange(15)[None, :]
    V = val[idx]
    Cc = col[idx] - full[:, None]
    mx = np.abs(V).max(axis=1, keepdims=True)
    V = np.where(np.abs(V) <= 2.220446049250313e-16 * mx, 0.0, V)
    key = np.concatenate([V.view(np.int64), Cc.astype(np.int64)], axis=1)
    # hash rows to 64 bits for counting (collisions negligible for the purpose)
    h = (key * np.arange(1, 31, dtype=np.int64)[None, :] * np.int64(0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF)).sum(axis=1)
    u, cnt = np.unique(h, return_counts=True)
    order = np.argsort(-cnt)
    print("M %d: %d rows, %d with 15 entries, %d distinct (offsets, values) rows among them; the 10 most frequent cover %.1f %%, "
          "the 1024 most frequent %.1f %%" % (M, n, full.shape[0], u.shape[0], 100.0 * cnt[order[:10]].sum() / full.shape[0],
                                             100.0 * cnt[order[:1024]].sum() / full.shape[0]), flush=True)
