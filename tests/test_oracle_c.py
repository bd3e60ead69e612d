"""The C/OpenMP restatement (oracle/oracle.c, the timed CPU baseline) against the numpy oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import fedd_oracle as fo
import oracle_c


@pytest.mark.parametrize("M", [3, 8])
def test_c_oracle_matches_numpy_oracle(M):
    r = oracle_c.run_laplace3d(M, target=27, rtol=1e-13, restart=100, max_it=400, want_system=True)
    m = fo.build_mesh_structured(3, 1, M)
    A_bc, rhs_bc, _, _, _ = fo.laplace_problem(m)
    A = sp.csr_matrix((r["val"], r["col"], r["rowptr"]), shape=A_bc.shape)
    assert A.nnz == A_bc.nnz
    assert abs(A - A_bc).max() < 1e-14
    np.testing.assert_allclose(r["rhs"], rhs_bc, atol=1e-16)
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(r["x"], xd, atol=1e-11 * np.abs(xd).max())
    nb_, nb, _ = fo.schwarz_bins(m.xyz_uni, 27)
    ras = fo.RAS(A_bc, nb_, nb)
    assert r["n_subdomains"] == nb and r["max_size"] == ras.max_size
    _, its, _ = fo.gmres_right(A_bc, rhs_bc, ras.apply, rtol=1e-13, max_it=400, restart=100)
    assert abs(its - r["its"]) <= 1


def test_c_oracle_unpreconditioned_and_restart():
    r = oracle_c.run_laplace3d(10, rtol=1e-12, restart=7, max_it=500, use_prec=False, want_system=True)
    A = sp.csr_matrix((r["val"], r["col"], r["rowptr"]), shape=(1331, 1331))
    assert np.linalg.norm(r["rhs"] - A @ r["x"]) / np.linalg.norm(r["rhs"]) < 1e-10
    assert r["its"] > 7
