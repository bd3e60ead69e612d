"""CPU experiment (oracle side, scipy): does monolithic one-level RAS with box subdomains converge on the merged
P2/P1 Stokes system of the DFG cylinder?  usage: stokes_ras_cpu.py [1k|6k] [target dofs per box] [combine]"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fedd_oracle as fo

which = sys.argv[1] if len(sys.argv) > 1 else "1k"
target = int(sys.argv[2]) if len(sys.argv) > 2 else 200
combine = sys.argv[3] if len(sys.argv) > 3 else "restricted"
nu = 1.0
m1 = fo.read_mesh_file(os.path.join(ROOT, "tests", "golden", "DFG3DCylinder_%s.mesh" % which), 3, volume_id=0)
mv = fo.build_p2_of_p1(m1)
nv, n_p = mv.xyz.shape[0], m1.xyz.shape[0]
A, BT, B = fo.stokes_blocks(mv, m1, nu)
Mo = fo.block_merge(A, BT, B).tocsr()
n = 3 * nv + n_p
X = mv.xyz
flag = mv.flag_uni
H = 0.41
rows, vals = [], []
for node in np.nonzero(np.isin(flag, (1, 2, 4)))[0]:
    for d in range(3):
        rows.append(3 * node + d)
        v = 0.0
        if flag[node] == 2 and d == 0:
            y, z = X[node, 1], X[node, 2]
            v = 16.0 * 1.0 * y * (H - y) * z * (H - z) / H ** 4
        vals.append(v)
rows = np.array(rows); vals = np.array(vals)
is_dir = np.zeros(n, bool); is_dir[rows] = True
g = np.zeros(n); g[rows] = vals
M, rhs = fo.set_dirichlet(Mo, np.zeros(n), is_dir, g)
M = M.tocsr()
print("n", n, "nnz", M.nnz, "dirichlet", rows.shape[0], flush=True)
# dof -> carrying node coordinates
xyz_dof = np.concatenate([np.repeat(X, 3, axis=0), m1.xyz], axis=0)
def rcb(xyz, target):
    """recursive coordinate bisection: split the longest axis at the median until a bin holds <= target points"""
    bins = np.zeros(xyz.shape[0], dtype=np.int64)
    stack = [np.arange(xyz.shape[0])]
    out = []
    while stack:
        idx = stack.pop()
        if idx.shape[0] <= target:
            out.append(idx); continue
        ext = xyz[idx].max(axis=0) - xyz[idx].min(axis=0)
        d = int(np.argmax(ext))
        order = idx[np.argsort(xyz[idx, d], kind="stable")]
        h = order.shape[0] // 2
        stack.append(order[h:]); stack.append(order[:h])
    for k, idx in enumerate(out):
        bins[idx] = k
    return bins, len(out)
if os.environ.get("RCB"):
    bins, nb = rcb(xyz_dof, target); gg = "rcb"
else:
    bins, nb, gg = fo.schwarz_bins(xyz_dof, target)
print("boxes", nb, "lattice", gg, flush=True)
t0 = time.time()
G = M.copy(); G.data[:] = 1.0
P0 = sp.csr_matrix((np.ones(n), (np.arange(n), bins)), shape=(n, nb))
Pk = (G @ P0 + P0); Pk.data[:] = 1.0
Pk = Pk.tocsc(); P0c = P0.tocsc()
subs = []
mult = np.zeros(n)
mx = 0
nA = 3 * nv
for i in range(nb):
    own = np.sort(P0c.indices[P0c.indptr[i]:P0c.indptr[i + 1]])
    allr = np.sort(Pk.indices[Pk.indptr[i]:Pk.indptr[i + 1]])
    ext = np.setdiff1d(allr, own, assume_unique=True)
    idx = np.concatenate([own, ext])
    Ai = M[idx][:, idx].tocsc()
    lu = spla.splu(Ai)
    subs.append((idx, own.shape[0], lu))
    mult[idx] += 1
    mx = max(mx, idx.shape[0])
print("setup %.1f s, max subdomain %d, mean %.0f" % (time.time() - t0, mx, np.mean([s[0].shape[0] for s in subs])), flush=True)

def apply(r):
    z = np.zeros_like(r)
    for idx, no, lu in subs:
        y = lu.solve(r[idx])
        if combine == "restricted":
            z[idx[:no]] += y[:no]
        else:
            z[idx] += y
    if combine == "averaging":
        z /= mult
    return z

t0 = time.time()
x, its, hist = fo.gmres_right(M, rhs, apply, rtol=1e-8, max_it=600, restart=200)
print("gmres its", its, "relres", hist[-1] if len(hist) else None, "true", np.linalg.norm(rhs - M @ x) / np.linalg.norm(rhs), "%.1f s" % (time.time() - t0))
