"""Prototype (numpy) of the s-step GMRES with BCGS-PIP2 block orthogonalisation that gmres.hip's gmres_kind 2
implements: validates the Hessenberg recovery formulas and the conditioning of the monomial block basis.
Development tool; reads only oracle/ (test infrastructure)."""
import math
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import fedd_oracle as fo


def chol_scaled(S):
    """upper R with R^T R = S via scaled Cholesky; returns (R, ncols_ok)"""
    s = S.shape[0]
    d = np.sqrt(np.maximum(np.diag(S), 0.0))
    R = np.zeros_like(S)
    ok = s
    Sn = S / np.outer(np.where(d > 0, d, 1), np.where(d > 0, d, 1))
    L = np.zeros_like(S)
    for j in range(s):
        v = Sn[j, j] - L[j, :j] @ L[j, :j]
        if not (d[j] > 0) or v <= 1e-14 * 100:
            ok = j
            break
        L[j, j] = math.sqrt(v)
        for i in range(j + 1, s):
            L[i, j] = (Sn[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
    R = (L * d[:, None]).T
    return R, ok


def leja(vals):
    vals = list(vals)
    out = [max(vals, key=abs)]
    vals.remove(out[0])
    while vals:
        nx = max(vals, key=lambda v: sum(math.log(max(abs(v - o), 1e-300)) for o in out))
        out.append(nx); vals.remove(nx)
    return out


def sstep_gmres(A, b, Mop, rtol, max_it, restart, s, verbose=False, newton=False):
    n = b.shape[0]
    x = np.zeros(n)
    r = b.copy()
    beta0 = np.linalg.norm(r)
    its = 0
    B = lambda v: A @ Mop(v)
    conds = []
    theta = np.zeros(s)
    while its < max_it:
        beta = np.linalg.norm(r)
        m = min(restart, max_it - its)
        V = np.zeros((n, m + 1))
        Hraw = np.zeros((m + 1, m))
        V[:, 0] = r / beta
        k = 1
        s_cur = s
        while k - 1 < m:
            sa = min(s_cur, m - (k - 1))
            for i in range(sa):
                V[:, k + i] = B(V[:, k - 1 + i]) - theta[i] * V[:, k - 1 + i]
            Q = V[:, :k]
            W = V[:, k:k + sa]
            # pass 1
            P = V[:, :k + sa].T @ W
            C1, G = P[:k], P[k:]
            R1, ok = chol_scaled(G - C1.T @ C1)
            if ok < sa:
                if verbose:
                    print("truncate block at k=%d: %d of %d" % (k, ok, sa))
                if ok == 0:
                    raise RuntimeError("breakdown")
                sa = ok
                s_cur = max(1, ok)
                W = V[:, k:k + sa]
                C1, R1 = C1[:, :sa], R1[:sa, :sa]
            W[:] = (W - Q @ C1) @ np.linalg.inv(R1)
            # pass 2
            P = V[:, :k + sa].T @ W
            C2, G2 = P[:k], P[k:]
            R2, ok2 = chol_scaled(G2 - C2.T @ C2)
            assert ok2 == sa
            W[:] = (W - Q @ C2) @ np.linalg.inv(R2)
            C = C1 + C2 @ R1
            R = R2 @ R1
            conds.append(np.linalg.cond(R))
            # Hessenberg columns k-1 .. k-2+sa
            Hraw[:k, k - 1] = C[:, 0]
            Hraw[k, k - 1] = R[0, 0]
            Hraw[k - 1, k - 1] += theta[0]
            if sa > 1:
                X = np.vstack([C[:, 1:sa], R[:, 1:sa]])
                X += np.vstack([C[:, :sa - 1], R[:, :sa - 1]]) * theta[1:sa][None, :]
                X[:k + 1] -= Hraw[:k + 1, :k] @ C[:, :sa - 1]
                Hn = X @ np.linalg.inv(R[:sa - 1, :sa - 1])
                Hraw[:k + sa, k:k + sa - 1] = Hn
            k += sa
            if newton and k - 1 == sa and not theta.any():
                ev = np.linalg.eigvalsh(0.5 * (Hraw[:sa, :sa] + Hraw[:sa, :sa].T)) if os.environ.get('SYM') else np.linalg.eigvals(Hraw[:sa, :sa])
                theta = np.array([e.real for e in leja(ev)] + [0.0] * (s - sa))
                if verbose: print('shifts', theta)
            # residual from the least-squares problem (prototype: dense lstsq)
            kk = k - 1
            e1 = np.zeros(kk + 1); e1[0] = beta
            res = []
            for c in range(kk - sa + 1, kk + 1):
                y, *_ = np.linalg.lstsq(Hraw[:c + 1, :c], e1[:c + 1], rcond=None)
                res.append(np.linalg.norm(e1[:c + 1] - Hraw[:c + 1, :c] @ y) / beta0)
            hit = [c for c, rr in zip(range(kk - sa + 1, kk + 1), res) if rr <= rtol]
            if hit:
                kk = hit[0]
                its += kk - (k - 1 - sa)
                break
            its += sa
        else:
            kk = k - 1
        kk = min(kk, k - 1)
        e1 = np.zeros(kk + 1); e1[0] = beta
        y, *_ = np.linalg.lstsq(Hraw[:kk + 1, :kk], e1, rcond=None)
        x = x + Mop(V[:, :kk] @ y)
        r = b - A @ x
        tr = np.linalg.norm(r) / beta0
        orth = np.abs(V[:, :kk + 1].T @ V[:, :kk + 1] - np.eye(kk + 1)).max()
        if verbose:
            print("cycle end: its %d true relres %.3e orth %.2e max cond(R) %.2e" % (its, tr, orth, max(conds)))
        if tr <= rtol:
            break
    return x, its, max(conds)


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    target = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    om = fo.build_mesh_structured(3, 1, M)
    A, rhs, _, _, _ = fo.laplace_problem(om)
    bins, nb, _ = fo.schwarz_bins(om.xyz_uni, target)
    ras = fo.RAS(A, bins, nb)
    xd = fo.direct_solve(A, rhs)
    for rtol in (1e-8, 1e-13):
        x0, it0, _ = fo.gmres_right(A, rhs, ras.apply, rtol=rtol, max_it=1000, restart=100)
        print("M=%d rtol %g: reference GMRES its %d err %.2e" % (M, rtol, it0, np.abs(x0 - xd).max() / np.abs(xd).max()))
        for s, nw in ((8, 0), (8, 1)):
            x, it, cmax = sstep_gmres(A, rhs, ras.apply, rtol, 1000, 100, s, verbose=False, newton=bool(nw))
            print("  s=%2d newton=%d:" % (s, nw) + " its %d err vs direct %.2e max cond(R) %.2e true relres %.2e" % (
                it, np.abs(x - xd).max() / np.abs(xd).max(), cmax, np.linalg.norm(rhs - A @ x) / np.linalg.norm(rhs)))
