"""Conditioning of the s-step block basis: monomial against Newton (Leja-ordered Ritz shifts taken from the first s Arnoldi steps)
for the RAS-preconditioned Laplace operator (numpy prototype; development tool)."""
import math, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import fedd_oracle as fo
from sstep_proto import leja

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
om = fo.build_mesh_structured(3, 1, M)
A, rhs, _, _, _ = fo.laplace_problem(om)
bins, nb, _ = fo.schwarz_bins(om.xyz_uni, 64)
ras = fo.RAS(A, bins, nb)
B = lambda v: A @ ras.apply(v)
n = rhs.shape[0]
# Arnoldi to get Ritz values
def arnoldi(v0, m):
    V = np.zeros((n, m + 1)); H = np.zeros((m + 1, m)); V[:, 0] = v0 / np.linalg.norm(v0)
    for j in range(m):
        w = B(V[:, j])
        for _ in range(2):
            h = V[:, :j + 1].T @ w; w -= V[:, :j + 1] @ h; H[:j + 1, j] += h
        H[j + 1, j] = np.linalg.norm(w); V[:, j + 1] = w / H[j + 1, j]
    return V, H
for s in (8, 12, 16, 20):
    V, H = arnoldi(rhs, max(s, 40))
    ritz = np.linalg.eigvals(H[:s, :s])
    th = np.array([z.real for z in leja(list(ritz))])
    for start in (0, 8, 24, 39):
        q = V[:, start]
        for name, shifts in (("monomial", np.zeros(s)), ("newton", th)):
            W = np.zeros((n, s)); w = q
            for i in range(s):
                w = B(w) - shifts[i] * w
                W[:, i] = w
            # project out the previous basis (as the block orthogonalisation does), then condition of the scaled block
            Q = V[:, :start + 1]
            Wp = W - Q @ (Q.T @ W)
            Wp = Wp - Q @ (Q.T @ Wp)
            Wn = Wp / np.linalg.norm(Wp, axis=0)
            print("M=%d s=%2d start %2d %-8s: cond(scaled projected block) %.2e  max|imag ritz| %.1e  ritz range [%.3f, %.3f]"
                  % (M, s, start, name, np.linalg.cond(Wn), np.abs(ritz.imag).max(), ritz.real.min(), ritz.real.max()), flush=True)
