"""Unstructured inputs through the GPU path: cfg 1 of BASELINE.json (2D P1 Laplace on the reference's
square.mesh, one rank) against the committed golden fixture, the single reference tetrahedron
against its closed-form element matrices, and a 3D unstructured solve on the DFG cylinder."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import fedd_oracle as fo
from test_gpu_parity import assert_matrix_close, csr_global, oracle_mesh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx(fedd_lib):
    c = fedd_lib.Context(device=0)
    yield c
    c.close()


def test_cfg1_square_mesh_against_golden_fixture(fedd_lib, ctx):
    d = np.load(os.path.join(GOLD, "laplace_square_mesh.npz"))
    m = fedd_lib.read_mesh(os.path.join(GOLD, "square.mesh"), 2)
    assert m["xyz"].shape == (29, 2) and m["conn"].shape == (40, 3)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.assemble_rhs([1.0])
    A, _ = csr_global(ctx, 29)
    assert_matrix_close(A, sp.csr_matrix((d["A_data"], d["A_indices"], d["A_indptr"]), shape=(29, 29)))
    np.testing.assert_allclose(ctx.rhs_get(), d["rhs"], rtol=1e-10, atol=0)
    # the laplace driver registers Dirichlet for flags 1, 2, 3 only: the three flag-4 nodes stay natural
    ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    A, _ = csr_global(ctx, 29)
    assert_matrix_close(A, sp.csr_matrix((d["Abc_data"], d["Abc_indices"], d["Abc_indptr"]), shape=(29, 29)))
    np.testing.assert_allclose(ctx.rhs_get(), d["rhs_bc"], rtol=1e-10, atol=1e-300)
    ctx.schwarz_set_target(6, 1.0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    x, its, rel = ctx.gmres(None, rtol=1e-13, max_it=100, restart=50, use_prec=True)
    np.testing.assert_allclose(x, d["x"], rtol=0, atol=1e-10 * np.abs(d["x"]).max())
    # also unpreconditioned and with one subdomain (direct)
    x2, _, _ = ctx.gmres(None, rtol=1e-13, max_it=100, restart=50, use_prec=False)
    np.testing.assert_allclose(x2, d["x"], rtol=0, atol=1e-10 * np.abs(d["x"]).max())
    ctx.schwarz_set_target(1000, 1.0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    x3, its3, _ = ctx.gmres(None, rtol=1e-13, max_it=10, restart=10, use_prec=True)
    assert its3 == 1
    np.testing.assert_allclose(x3, d["x"], rtol=0, atol=1e-10 * np.abs(d["x"]).max())


def test_single_reference_tetrahedron(fedd_lib, ctx):
    m = fedd_lib.read_mesh(os.path.join(GOLD, "tetrahedron.mesh"), 3)
    assert m["conn"].shape == (1, 4)
    ctx.mesh_set_dict(m)
    assert ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR) == 16
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    A, _ = csr_global(ctx, 4)
    Kexp = np.array([[3, -1, -1, -1], [-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]]) / 6.0
    np.testing.assert_allclose(A.toarray(), Kexp, atol=1e-15)
    ctx.assemble(fedd_lib.FORM_MASS)
    A, _ = csr_global(ctx, 4)
    np.testing.assert_allclose(A.toarray(), (1 + np.eye(4)) / 120.0, rtol=1e-13)
    ctx.assemble_rhs([1.0])
    np.testing.assert_allclose(ctx.rhs_get(), np.full(4, 1 / 24.0), rtol=1e-13)


def test_unstructured_3d_laplace_solve(fedd_lib, ctx):
    m = fedd_lib.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)
    om = oracle_mesh(m)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.assemble_rhs([1.0])
    ctx.dirichlet([1, 2, 4], [0.0, 0.0, 0.0])
    A_bc, rhs_bc, _, _, _ = fo.laplace_problem(om, bc_flags=(1, 2, 4))
    A, _ = csr_global(ctx, om.n_global)
    assert_matrix_close(A, A_bc)
    ctx.schwarz_set_target(8, 1.0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    info = ctx.schwarz_info()
    node_bin, nb, g = fo.schwarz_bins(m["xyz"], 8)
    ras = fo.RAS(A_bc, node_bin, nb)
    assert info["n_subdomains"] == nb and info["max_size"] == ras.max_size
    x, its, rel = ctx.gmres(None, rtol=1e-13, max_it=500, restart=200, use_prec=True)
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-10 * np.abs(xd).max())
    _, its_o, _ = fo.gmres_right(A_bc, rhs_bc, ras.apply, rtol=1e-13, max_it=500, restart=200)
    assert abs(its - its_o) <= 2


def test_api_misuse_is_reported(fedd_lib, ctx):
    m = fedd_lib.structured_mesh(3, 1, 3)
    ctx.mesh_set_dict(m)
    with pytest.raises(fedd_lib.FeddError, match="fedd_pattern_build first"):
        ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    with pytest.raises(fedd_lib.FeddError, match="FULL pattern"):
        ctx.assemble(fedd_lib.FORM_LINELAS, [1.0, 1.0])
    with pytest.raises(fedd_lib.FeddError, match="unknown form"):
        ctx.assemble(99)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    with pytest.raises(fedd_lib.FeddError, match="preconditioner requested"):
        ctx.gmres(None, use_prec=True)
    with pytest.raises(fedd_lib.FeddError, match="coarse_kind 0 is not built"):
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1)
    ctx.schwarz_set_target(100000, 1.0)          # one 64-node box + overlap would exceed nothing; 4^3 = 64 dofs
    ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    m = fedd_lib.structured_mesh(3, 1, 8)        # 729 dofs in one box: above the dense solver's limit
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    with pytest.raises(fedd_lib.FeddError, match="at most 256"):
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    with pytest.raises(fedd_lib.FeddError, match="unknown key"):
        ctx.set_option("nope", 1)
