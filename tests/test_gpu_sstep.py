"""GPU tests of the s-step GMRES (fedd_set_option "gmres_kind" 2; gmres.hip): in exact arithmetic its iterates are those of
the one-vector-at-a-time solver, so it is held against that solver (same iteration count at the reference's tolerance), against
the oracle's GMRES on the same preconditioner definition, and against a direct solve with both sides driven to 1e-13
(north_star: solution within 1e-10).  Stands in for Belos "Block GMRES", parametersSolver.xml:5-15."""
import numpy as np
import pytest

import fedd_oracle as fo
from test_gpu_parity import _setup_laplace, RTOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(fedd_lib):
    c = fedd_lib.Context(device=0)
    yield c
    c.set_option("gmres_kind", 2)
    c.set_option("gmres_s", 0)
    c.close()


def _true_relres(A, b, x):
    return float(np.linalg.norm(b - A @ x) / np.linalg.norm(b))


@pytest.mark.parametrize("s", [1, 2, 3, 4, 5, 7, 8, 11, 12, 16])
def test_same_iterations_as_the_one_vector_solver(fedd_lib, ctx, s):
    """rtol 1e-8 (laplace/parametersSolver.xml): identical iteration count, true residual below the tolerance.  Blocks longer
    than 8 (Newton basis) on the unpreconditioned operator, which takes enough iterations for several of them."""
    prec = s <= 8
    m, om, A_bc, rhs_bc = _setup_laplace(fedd_lib, ctx, 3, 20 if prec else 16)
    if prec:
        ctx.schwarz_set_target(27, 1.0)
        ctx.schwarz_setup(overlap=1, combine=fedd_lib.COMBINE_RESTRICTED)
    ctx.set_option("gmres_kind", 0)
    x0, its0, rel0 = ctx.gmres(None, rtol=1e-8, max_it=400, restart=100, use_prec=prec)
    ctx.set_option("gmres_kind", 2)
    ctx.set_option("gmres_s", s)
    x, its, rel = ctx.gmres(None, rtol=1e-8, max_it=400, restart=100, use_prec=prec)
    info = ctx.gmres_info()
    assert info["kind"] == 2 and info["s"] == s and info["blocks"] >= (its + s - 1) // s
    if s > 8:      # the long blocks were used: two monomial 8-blocks at most before the shifts exist
        assert its >= 2 * s and info["blocks"] <= 2 + (its - s + s - 1) // s + info["cut_blocks"], (its, info)
    else:
        assert info["cut_blocks"] == 0
    assert abs(its - its0) <= 1, (its, its0)
    tr = _true_relres(A_bc, rhs_bc, x)
    assert tr <= 1e-8 and abs(rel - tr) <= 1e-3 * tr       # the reported residual IS the true one
    np.testing.assert_allclose(x, x0, rtol=0, atol=1e-7 * np.abs(x0).max())


@pytest.mark.parametrize("dim,M,use_prec,s", [(3, 12, True, 8), (3, 12, False, 8), (2, 32, True, 8), (3, 16, True, 4),
                                              (3, 16, True, 8)])
def test_solution_matches_direct_solve(fedd_lib, ctx, dim, M, use_prec, s):
    m, om, A_bc, rhs_bc = _setup_laplace(fedd_lib, ctx, dim, M)
    if use_prec:
        ctx.schwarz_set_target(27 if dim == 3 else 16, 1.0)
        ctx.schwarz_setup(overlap=1, combine=fedd_lib.COMBINE_RESTRICTED)
    ctx.set_option("gmres_kind", 2)
    ctx.set_option("gmres_s", s)
    x, its, rel = ctx.gmres(None, rtol=1e-13, max_it=600, restart=200, use_prec=use_prec)
    assert rel <= 1e-13
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=RTOL * np.abs(xd).max())
    assert _true_relres(A_bc, rhs_bc, x) <= 1.05e-13
    if use_prec:
        # the oracle's GMRES stops on its recurrence residual, this solver on the true one: a few iterations more at 1e-13
        node_bin, nb, g = fo.schwarz_bins(m["xyz"], 27 if dim == 3 else 16)
        ras = fo.RAS(A_bc, node_bin, nb)
        xo, its_o, hist = fo.gmres_right(A_bc, rhs_bc, ras.apply, rtol=1e-13, max_it=600, restart=200)
        assert its_o - 2 <= its <= 1.25 * its_o + 2, (its, its_o)


@pytest.mark.parametrize("restart,s", [(7, 4), (7, 8), (10, 3), (9, 8)])
def test_restart_and_iteration_cap(fedd_lib, ctx, restart, s):
    """restart lengths that are no multiple of s (short last block), the iteration limit inside a block"""
    m, om, A_bc, rhs_bc = _setup_laplace(fedd_lib, ctx, 3, 10)
    ctx.set_option("gmres_kind", 0)
    _, its0, _ = ctx.gmres(None, rtol=1e-10, max_it=400, restart=restart, use_prec=False)
    ctx.set_option("gmres_kind", 2)
    ctx.set_option("gmres_s", s)
    x, its, rel = ctx.gmres(None, rtol=1e-10, max_it=400, restart=restart, use_prec=False)
    assert rel <= 1e-10 and its > restart and abs(its - its0) <= 2, (its, its0)
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-7 * np.abs(xd).max())
    x, its, rel = ctx.gmres(None, rtol=1e-30, max_it=5, restart=50, use_prec=False)
    assert its == 5            # convergence failure is not an error (LinearSolver_def.hpp:124-125)
    assert rel == pytest.approx(_true_relres(A_bc, rhs_bc, x), rel=1e-6)


def test_one_subdomain_is_one_iteration(fedd_lib, ctx):
    """M^-1 = A^-1: B = I, the first block breaks down after one vector"""
    m, om, A_bc, rhs_bc = _setup_laplace(fedd_lib, ctx, 3, 4)
    ctx.schwarz_set_target(10 ** 6, 1.0)
    ctx.schwarz_setup(overlap=1, combine=fedd_lib.COMBINE_RESTRICTED)
    ctx.set_option("gmres_kind", 2)
    ctx.set_option("gmres_s", 8)
    x, its, rel = ctx.gmres(None, rtol=1e-12, max_it=20, restart=20, use_prec=True)
    assert its == 1 and rel <= 1e-12
    assert ctx.gmres_info()["cut_blocks"] >= 1
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=RTOL * np.abs(xd).max())


def test_dependent_block_is_cut_not_trusted(fedd_lib, ctx):
    """an absurd threshold cuts every block after its first vector: the solver degrades to s = 1, not to a wrong answer"""
    m, om, A_bc, rhs_bc = _setup_laplace(fedd_lib, ctx, 3, 12)
    ctx.schwarz_set_target(27, 1.0)
    ctx.schwarz_setup(overlap=1, combine=fedd_lib.COMBINE_RESTRICTED)
    ctx.set_option("gmres_kind", 2)
    ctx.set_option("gmres_s", 8)
    ctx.set_option("gmres_chol_tol", 0.5)
    try:
        x, its, rel = ctx.gmres(None, rtol=1e-10, max_it=300, restart=100, use_prec=True)
    finally:
        ctx.set_option("gmres_chol_tol", 1e-13)
    # one-column blocks lean on the Pythagorean norm alone (no second vector to re-orthogonalise against): the recurrence
    # drifts from the true residual near 1e-8.  The solver notices (its claims are checked against b - A x), restarts once
    # from the true residual and then stops; what it returns is the TRUE residual, with the status saying why it stopped
    st = ctx.gmres_status()
    assert ctx.gmres_info()["cut_blocks"] >= 1
    assert rel == pytest.approx(_true_relres(A_bc, rhs_bc, x), rel=1e-3)
    assert rel <= 1e-10 or (st["floor_reached"] == 1 and st["recurrence_relres"] <= 1e-10 and rel <= 1e-8), (rel, st)
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-7 * np.abs(xd).max())


@pytest.mark.parametrize("s", [8, 16])
def test_bitwise_reproducible(fedd_lib, ctx, s):
    m, om, A_bc, rhs_bc = _setup_laplace(fedd_lib, ctx, 3, 14)
    ctx.schwarz_set_target(8, 1.0)
    ctx.schwarz_setup(overlap=1, combine=fedd_lib.COMBINE_RESTRICTED)
    ctx.set_option("gmres_kind", 2)
    ctx.set_option("gmres_s", s)
    x1, its1, _ = ctx.gmres(None, rtol=1e-9, max_it=300, restart=100, use_prec=True)
    x2, its2, _ = ctx.gmres(None, rtol=1e-9, max_it=300, restart=100, use_prec=True)
    assert its1 == its2 and np.array_equal(x1, x2)


def test_newton_basis_is_what_makes_long_blocks_hold(fedd_lib, ctx):
    """s = 16 on the monomial basis runs in blocks of at most 8 (by construction); on the Newton basis the 16-blocks go through"""
    m, om, A_bc, rhs_bc = _setup_laplace(fedd_lib, ctx, 3, 16)
    ctx.set_option("gmres_kind", 2)
    ctx.set_option("gmres_s", 16)
    out = {}
    for newton in (0, 1):
        ctx.set_option("gmres_newton", newton)
        x, its, rel = ctx.gmres(None, rtol=1e-8, max_it=400, restart=100, use_prec=False)
        out[newton] = (x, its, ctx.gmres_info()["blocks"])
        assert rel <= 1e-8 and _true_relres(A_bc, rhs_bc, x) <= 1e-8
    ctx.set_option("gmres_newton", 1)
    assert abs(out[0][1] - out[1][1]) <= 1 and out[1][2] < out[0][2], [(o[1], o[2]) for o in out.values()]
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=1e-7 * np.abs(out[0][0]).max())


def test_elasticity_and_two_level(fedd_lib, ctx):
    """3 dofs per node (cfg 5 family) with the coarse level on: same counts as the one-vector solver"""
    M = 10
    m = fedd_lib.structured_mesh(3, 1, M)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(3, fedd_lib.BLOCK_FULL)
    mu, nu = 2.0e6, 0.4
    lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
    ctx.assemble(fedd_lib.FORM_LINELAS, [lam, mu])
    ctx.assemble_rhs([0.0, 1.0, 0.0])
    ctx.dirichlet([2], [0.0, 0.0, 0.0])
    ctx.schwarz_set_target(9, 1.0)
    out = {}
    for two in (0, 1):
        ctx.schwarz_setup(overlap=1, combine=fedd_lib.COMBINE_RESTRICTED, two_level=two, coarse_kind=fedd_lib.COARSE_Q1)
        for kind in (0, 2):
            ctx.set_option("gmres_kind", kind)
            ctx.set_option("gmres_s", 16)
            x, its, rel = ctx.gmres(None, rtol=1e-8, max_it=500, restart=100, use_prec=True)
            out[(two, kind)] = (x, its, rel)
        x0, i0, _ = out[(two, 0)]
        x2, i2, r2 = out[(two, 2)]
        assert abs(i2 - i0) <= max(2, i0 // 20), (two, i0, i2)
        b = ctx.rhs_get()
        assert np.linalg.norm(b - ctx.spmv(x2)) <= 1e-8 * np.linalg.norm(b)
        np.testing.assert_allclose(x2, x0, rtol=0, atol=1e-6 * np.abs(x0).max())


@pytest.mark.parametrize("restart", [24, 20, 100])
def test_three_sweeps_per_block_and_the_folded_solution_update(fedd_lib, ctx, restart):
    """Option "gmres_fuse" (blocks of 16): the first update and the second dot of a block as one sweep (k_blockfuse), and the
    solution update of a restart cycle formed by the last update of the block that fills the cycle (k_blockaxpy<16, true>;
    restart 24 = 8 + 8 + 8 columns, 20: a last block of 4, 100: no cycle fills).  Same iterations as the four-sweep
    form, same solution to rounding, true residual below the tolerance, and the direct solve at 1e-10."""
    m, om, A_bc, rhs_bc = _setup_laplace(fedd_lib, ctx, 3, 20)
    ctx.set_option("gmres_kind", 2)
    ctx.set_option("gmres_s", 16)
    res = {}
    for fuse in (0, 1):
        ctx.set_option("gmres_fuse", fuse)
        x, its, rel = ctx.gmres(None, rtol=1e-9, max_it=600, restart=restart, use_prec=False)
        info = ctx.gmres_info()
        res[fuse] = (x, its, rel, info)
        assert _true_relres(A_bc, rhs_bc, x) <= 1e-9
    ctx.set_option("gmres_fuse", -1)
    (x0, its0, _, info0), (x1, its1, _, info1) = res[0], res[1]
    assert info0["fused_blocks"] == 0 and info1["fused_blocks"] == info1["blocks"] >= 2
    assert abs(its0 - its1) <= 1 and (its1 > restart or restart == 100)
    np.testing.assert_allclose(x1, x0, rtol=0, atol=1e-9 * np.abs(x0).max())
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x1, xd, rtol=0, atol=1e-7 * np.abs(xd).max())      # tolerance-limited (1e-9 residual)
