#!/usr/bin/env python
"""Writes tests/golden/stokes_6k_direct.npz: the sparse DIRECT solution (scipy SuperLU + iterative refinement, CPU
oracle) of BASELINE cfg 4 -- P2/P1 Stokes on DFG3DCylinder_6k.mesh, nu = 1, no-slip on flags 1 and 4,
parabolic_benchmark inflow on flag 2 (height 0.41, max velocity 1), flag 3 natural -- assembled by the oracle
(oracle/fedd_oracle.py: stokes_blocks + block_merge + set_dirichlet).  141 742 doubles (1.1 MB), stored as float64.
The factorisation takes minutes, which is why the GPU test (tests/test_gpu_stokes.py) compares against this file
instead of factorising at test time; it first checks the stored vector against the DEVICE matrix (residual <= 1e-12).
Run in the build container:  python tests/golden/make_stokes_fixture.py"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import fedd_oracle as fo  # noqa: E402

t0 = time.time()
m1 = fo.read_mesh_file(os.path.join(HERE, "DFG3DCylinder_6k.mesh"), 3, volume_id=0)
mv = fo.build_p2_of_p1(m1)
nv, n_p = mv.xyz.shape[0], m1.xyz.shape[0]
A, BT, B = fo.stokes_blocks(mv, m1, 1.0)
Mo = fo.block_merge(A, BT, B).tocsr()
n = 3 * nv + n_p
X, flag, H = mv.xyz, mv.flag_uni, 0.41
nodes = np.nonzero(np.isin(flag, (1, 2, 4)))[0]
rows = (3 * nodes[:, None] + np.arange(3)[None, :]).ravel()
vals = np.zeros((nodes.shape[0], 3))
inflow = flag[nodes] == 2
y, z = X[nodes, 1], X[nodes, 2]
vals[inflow, 0] = (16.0 * y * (H - y) * z * (H - z) / H ** 4)[inflow]
is_dir = np.zeros(n, bool)
is_dir[rows] = True
g = np.zeros(n)
g[rows] = vals.ravel()
M, rhs = fo.set_dirichlet(Mo, np.zeros(n), is_dir, g)
print("assembled: n %d nnz %d (%.0f s)" % (n, M.nnz, time.time() - t0), flush=True)
x = fo.direct_solve(M.tocsr(), rhs, refine=3)
res = np.linalg.norm(rhs - M @ x) / np.linalg.norm(rhs)
print("direct solve done (%.0f s), relative residual %.2e" % (time.time() - t0, res), flush=True)
np.savez_compressed(os.path.join(HERE, "stokes_6k_direct.npz"), x=x, n=n, nv=nv, n_p=n_p, relres=res)
print("wrote stokes_6k_direct.npz")
