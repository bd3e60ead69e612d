"""Iteration counts of the fixed-grid (strong-scaling) split used by bench.py, with the N ranks as threads of
one process on one GPU (capi.ThreadGroup; development aid -- times are meaningless, counts are not).
usage: threads_scaling.py [global cells per direction] [ghost element layers: 1, 2 (overlap rows), box width + 1 (whole boxes)] [box_kind] [nodes per box]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feddlib_amd import capi  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
GHOSTS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
BOX = int(sys.argv[3]) if len(sys.argv) > 3 else 0
TARGET = int(sys.argv[4]) if len(sys.argv) > 4 else 27
DECOMP = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}


def rank_main(group, rank, dec, cells, out):
    world = group.world
    m = capi.structured_mesh(3, dec, cells, rank, ghosts=GHOSTS if world > 1 else 0)
    c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
    c.mesh_set_dict(m)
    c.set_option("box_kind", BOX)
    if world > 1:
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, cells, m["gid_rep"]))
        c.comm_set_thread_group(group)
    res = {}
    for two in (False, True):
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        c.schwarz_set_target(TARGET, 1.0)
        if two:
            c.schwarz_set_coarse(0.0)
            c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)
        else:
            c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
        _, its, rel = c.gmres(None, rtol=1e-8, max_it=2000, restart=100, use_prec=True, want_x=False)
        res["two" if two else "one"] = (its, rel)
    out[rank] = res
    c.close()


for N in (1, 2, 4, 8):
    dec = DECOMP[N]
    cells = [G // d for d in dec]
    group = capi.ThreadGroup(N, timeout=600.0)
    out = [None] * N
    th = [threading.Thread(target=rank_main, args=(group, r, dec, cells, out)) for r in range(N)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print("ghost layers %d box_kind %d nodes per box %d: N %d blocks %s of %s cells: one level %d its (%.1e), two levels %d its (%.1e)"
          % (GHOSTS, BOX, TARGET, N, "x".join(map(str, dec)), "x".join(map(str, cells)), out[0]["one"][0], out[0]["one"][1],
             out[0]["two"][0], out[0]["two"][1]), flush=True)
