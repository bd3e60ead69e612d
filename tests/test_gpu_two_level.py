"""Two-level Schwarz on the GPU against the oracle's CoarseQ1 (normative definition of the coarse
level that stands in for FROSch's GDSWCoarseOperator, parametersPrec.xml:13, 62-122): lattice,
dense K0^-1 (Galerkin product + blocked Gauss-Jordan on the f64 matrix cores), operator apply,
iteration counts, elasticity, 2D, and the misuse messages."""
import os

import numpy as np
import pytest

import fedd_oracle as fo
from test_gpu_parity import oracle_mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(fedd_lib):
    c = fedd_lib.Context(device=0)
    yield c
    c.close()


def laplace_setup(fedd_lib, ctx, dim, M, bc=(1, 2, 3)):
    m = fedd_lib.structured_mesh(dim, 1, M)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.assemble_rhs([1.0])
    ctx.dirichlet(list(bc), [0.0] * len(bc))
    om = oracle_mesh(m)
    A_bc, rhs_bc, _, _, flags = fo.laplace_problem(om, bc_flags=bc)
    return m, om, A_bc, rhs_bc, np.isin(flags, bc)


@pytest.mark.parametrize("dim,M,cells", [(3, 12, 27), (3, 16, 200), (2, 24, 36), (3, 10, 1)])
def test_coarse_matrix_and_apply(fedd_lib, ctx, dim, M, cells):
    m, om, A_bc, rhs_bc, is_dir = laplace_setup(fedd_lib, ctx, dim, M)
    ctx.schwarz_set_target(27 if dim == 3 else 9, 1.0)
    ctx.schwarz_set_coarse(cells)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
    g, Kinv = ctx.schwarz_coarse()
    co = fo.CoarseQ1(A_bc, m["xyz"], is_dir, 1, cells_target=cells)
    np.testing.assert_array_equal(g[:dim], co.g)
    assert Kinv.shape == (co.n0, co.n0)
    np.testing.assert_allclose(Kinv, co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
    node_bin, nb, _ = fo.schwarz_bins(m["xyz"], 27 if dim == 3 else 9)
    ras = fo.RAS(A_bc, node_bin, nb)
    rng = np.random.default_rng(5)
    for _ in range(2):
        r = rng.standard_normal(A_bc.shape[0])
        z = ctx.schwarz_apply(r)
        zo = ras.apply(r) + co.apply(r)
        np.testing.assert_allclose(z, zo, rtol=0, atol=1e-10 * np.abs(zo).max())
    # same operator twice: the apply is deterministic
    r = rng.standard_normal(A_bc.shape[0])
    np.testing.assert_array_equal(ctx.schwarz_apply(r), ctx.schwarz_apply(r))


def test_two_level_solve_iterations(fedd_lib, ctx):
    m, om, A_bc, rhs_bc, is_dir = laplace_setup(fedd_lib, ctx, 3, 24)
    ctx.schwarz_set_target(27, 1.0)
    ctx.schwarz_set_coarse(216)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    x1, its1, _ = ctx.gmres(None, rtol=1e-8, max_it=300, restart=100, use_prec=True)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
    x2, its2, rel2 = ctx.gmres(None, rtol=1e-8, max_it=300, restart=100, use_prec=True)
    node_bin, nb, _ = fo.schwarz_bins(m["xyz"], 27)
    ras = fo.RAS(A_bc, node_bin, nb)
    co = fo.CoarseQ1(A_bc, m["xyz"], is_dir, 1, cells_target=216)
    _, its_o, _ = fo.gmres_right(A_bc, rhs_bc, lambda r: ras.apply(r) + co.apply(r), rtol=1e-8, max_it=300, restart=100)
    assert abs(its2 - its_o) <= 1
    assert its2 < its1
    xd = fo.direct_solve(A_bc, rhs_bc)
    assert np.abs(x2 - xd).max() <= 1e-6 * np.abs(xd).max()
    # back to one level: the coarse part is gone
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    _, its3, _ = ctx.gmres(None, rtol=1e-8, max_it=300, restart=100, use_prec=True)
    assert its3 == its1
    with pytest.raises(fedd_lib.FeddError, match="no coarse level"):
        ctx.schwarz_coarse()


def test_two_level_elasticity(fedd_lib, ctx):
    """3 dofs per node, FULL pattern: the coarse space carries one hat function per component."""
    M = 8
    m = fedd_lib.structured_mesh(3, 1, M)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(3, fedd_lib.BLOCK_FULL)
    mu, nu = 1.0, 0.3
    lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
    ctx.assemble(fedd_lib.FORM_LINELAS, [lam, mu])
    ctx.assemble_rhs([0.0, 1.0, 0.0])
    ctx.dirichlet([2], [0.0, 0.0, 0.0])
    om = oracle_mesh(m)
    A_bc, rhs_bc, _, _, flags = fo.linelas_problem(om, mu, nu)
    is_dir = np.repeat(np.isin(flags, (2,)), 3)
    ctx.schwarz_set_target(8, 1.0)
    ctx.schwarz_set_coarse(27)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
    g, Kinv = ctx.schwarz_coarse()
    co = fo.CoarseQ1(A_bc, m["xyz"], is_dir, 3, cells_target=27)
    assert Kinv.shape == (co.n0, co.n0)
    np.testing.assert_allclose(Kinv, co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
    node_bin, nb, _ = fo.schwarz_bins(m["xyz"], 8)
    ras = fo.RAS(A_bc, node_bin, nb, dofs=3)
    r = np.random.default_rng(2).standard_normal(A_bc.shape[0])
    zo = ras.apply(r) + co.apply(r)
    np.testing.assert_allclose(ctx.schwarz_apply(r), zo, rtol=0, atol=1e-10 * np.abs(zo).max())
    x, its, rel = ctx.gmres(None, rtol=1e-10, max_it=300, restart=150, use_prec=True)
    xd = fo.direct_solve(A_bc, rhs_bc)
    assert np.abs(x - xd).max() <= 1e-7 * np.abs(xd).max()
    # default subdomain size for a 3-dof problem: 27 / 3 nodes per box, which the dense local solver takes
    ctx.schwarz_set_target(0, 1.0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    node_bin, nb, _ = fo.schwarz_bins(m["xyz"], 9)
    info = ctx.schwarz_info()
    assert info["n_subdomains"] == nb and info["max_size"] <= 256


def test_two_level_misuse(fedd_lib, ctx):
    laplace_setup(fedd_lib, ctx, 3, 6)
    with pytest.raises(fedd_lib.FeddError, match="coarse_kind 7 is not built"):
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=7)
    ctx.schwarz_set_coarse(20000)
    with pytest.raises(fedd_lib.FeddError, match="at most 8192|finer than the mesh|not positive definite"):
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
    ctx.schwarz_set_coarse(0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
    g, Kinv = ctx.schwarz_coarse()
    assert tuple(g) == (1, 1, 1) and Kinv.shape == (8, 8)    # default: 343 nodes / 500 -> one cell


@pytest.mark.parametrize("dim,M,cells,kind", [(3, 12, 8, "gdsw"), (3, 12, 27, "gdsw"), (2, 24, 16, "gdsw"), (3, 9, 1, "gdsw"),
                                              (3, 12, 8, "rgdsw"), (3, 12, 27, "rgdsw"), (3, 16, 64, "rgdsw"),
                                              (2, 24, 16, "rgdsw"), (3, 9, 1, "rgdsw"), (3, 12, 2, "rgdsw")])
def test_gdsw_coarse_matrix_and_apply(fedd_lib, dim, M, cells, kind):
    """FEDD_COARSE_GDSW / FEDD_COARSE_RGDSW against the oracle's CoarseGDSW (exact sparse interior solves): interface
    classification, coarse nodes and weights of the reduced space, harmonic extensions (device GMRES on the constrained
    operator, solved to 1e-13 here), K0^-1 and the operator.  (cells = 2: a slab decomposition, whose coarse nodes are
    faces.)"""
    c = fedd_lib.Context(device=0)
    try:
        m, om, A_bc, rhs_bc, is_dir = laplace_setup(fedd_lib, c, dim, M)
        tgt = 27 if dim == 3 else 9
        c.schwarz_set_target(tgt, 1.0)
        c.schwarz_set_coarse(cells)
        c.set_option("gdsw_tol", 1e-13)
        reduced = kind == "rgdsw"
        c.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1,
                        coarse_kind=fedd_lib.COARSE_RGDSW if reduced else fedd_lib.COARSE_GDSW)
        g, Kinv = c.schwarz_coarse()
        co = fo.CoarseGDSW(A_bc, m["conn"], m["xyz"], is_dir, 1, cells_target=cells, reduced=reduced)
        np.testing.assert_array_equal(g[:dim], co.g)
        assert Kinv.shape == (co.n0, co.n0)
        assert co.n0 == (int(np.prod(np.where(co.g >= 2, co.g - 1, 1))) if reduced else int(np.prod(2 * co.g - 1)))
        np.testing.assert_allclose(Kinv, co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
        node_bin, nb, _ = fo.schwarz_bins(m["xyz"], tgt)
        ras = fo.RAS(A_bc, node_bin, nb)
        rng = np.random.default_rng(5)
        r = rng.standard_normal(A_bc.shape[0])
        z = c.schwarz_apply(r)
        zo = ras.apply(r) + co.apply(r)
        np.testing.assert_allclose(z, zo, rtol=0, atol=1e-10 * np.abs(zo).max())
        np.testing.assert_array_equal(c.schwarz_apply(r), c.schwarz_apply(r))
        x, its, rel = c.gmres(None, rtol=1e-12, max_it=300, restart=100, use_prec=True)
        xd = fo.direct_solve(A_bc, rhs_bc)
        assert rel <= 1e-12
        np.testing.assert_allclose(x, xd, rtol=0, atol=1e-10 * np.abs(xd).max())
    finally:
        c.close()


def test_gdsw_elasticity_and_iteration_counts(fedd_lib):
    """3-dof elasticity (FULL blocks, steadyLinElas_Perf parameters, Dirichlet on flag 2): GDSW with translations per
    interface component against the oracle; and the point of a coarse level: iteration counts that stay put when the
    mesh is refined at fixed H / h (more and more coarse cells), where one level keeps growing."""
    c = fedd_lib.Context(device=0)
    try:
        mu, nu = 2.0e6, 0.4
        lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
        M = 8
        m = fedd_lib.structured_mesh(3, 1, M)
        c.mesh_set_dict(m)
        c.pattern_build(3, fedd_lib.BLOCK_FULL)
        c.assemble(fedd_lib.FORM_LINELAS, [lam, mu])
        c.assemble_rhs([0.0, 1.0, 0.0])
        c.dirichlet([2], [0.0, 0.0, 0.0])
        om = oracle_mesh(m)
        A_bc, rhs_bc, _, _, flags = fo.linelas_problem(om, mu, nu)
        is_dir = fo.dirichlet_rows(flags, (2,), dofs=3)
        c.schwarz_set_coarse(8)
        c.set_option("gdsw_tol", 1e-13)
        c.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_GDSW)
        g, Kinv = c.schwarz_coarse()
        co = fo.CoarseGDSW(A_bc, m["conn"], m["xyz"], is_dir, 3, cells_target=8)
        assert Kinv.shape == (co.n0, co.n0) and co.n0 == 27 * 3
        np.testing.assert_allclose(Kinv, co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
        x, its, rel = c.gmres(None, rtol=1e-12, max_it=400, restart=100, use_prec=True)
        xd = fo.direct_solve(A_bc, rhs_bc)
        np.testing.assert_allclose(x, xd, rtol=0, atol=1e-10 * np.abs(xd).max())
        # Laplace, H / h = 8 fixed: 4^3, 6^3, 8^3 coarse cells (32^3 ... 64^3 fine cells)
        c.set_option("gdsw_tol", 1e-10)
        counts = {}
        for cells_per_dir in (4, 6, 8):
            m = fedd_lib.structured_mesh(3, 1, 8 * cells_per_dir)
            c.mesh_set_dict(m)
            c.pattern_build(1, fedd_lib.BLOCK_SCALAR)
            c.assemble(fedd_lib.FORM_LAPLACE)
            c.assemble_rhs([1.0])
            c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
            c.schwarz_set_target(27, 1.0)
            c.schwarz_set_coarse(cells_per_dir ** 3)
            c.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_GDSW)
            _, i2, r2 = c.gmres(None, rtol=1e-8, max_it=400, restart=100, use_prec=True)
            c.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
            _, i1, r1 = c.gmres(None, rtol=1e-8, max_it=400, restart=100, use_prec=True)
            counts[cells_per_dir] = (i1, i2)
            assert r1 <= 1e-8 and r2 <= 1e-8
        print("one-level / GDSW iterations at H/h = 8:", counts)       # measured: (26, 26), (40, 29), (54, 30)
        assert counts[8][1] <= counts[4][1] + 6            # GDSW: levels off as the number of cells grows
        assert counts[8][0] >= counts[4][0] + 20           # one level: keeps growing
        assert counts[8][1] < counts[8][0]
    finally:
        c.close()


@pytest.mark.parametrize("dim,M,cells,kind", [(3, 8, 8, "gdsw"), (3, 12, 27, "gdsw"), (3, 8, 8, "rgdsw"), (3, 12, 27, "rgdsw"),
                                              (2, 16, 16, "gdsw"), (2, 16, 16, "rgdsw"), (3, 9, 12, "gdsw")])
def test_gdsw_rotations_in_the_null_space(fedd_lib, dim, M, cells, kind):
    """Option "gdsw_rotations" (FROSch with node lists and "Rotations" = true, steadyLinElas/parametersPrec.xml:6, 100):
    elasticity, every entity carries the translations AND the linearised rotations about its centre, the dependent ones
    dropped (vertices keep 3 functions, straight edges 5, faces 6 in 3D) -- against the oracle's
    CoarseGDSW(rotations=True): size and inverse of K0, the operator, the solve, and fewer iterations than with translations
    alone.  ((3, 9, 12): a 3 x 2 x 2 lattice whose planes do not all lie on mesh planes.)"""
    c = fedd_lib.Context(device=0)
    try:
        mu, nu = 2.0e6, 0.4
        lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
        m = fedd_lib.structured_mesh(dim, 1, M)
        c.mesh_set_dict(m)
        c.pattern_build(dim, fedd_lib.BLOCK_FULL)
        c.assemble(fedd_lib.FORM_LINELAS, [lam, mu])
        f = [0.0, 1.0, 0.0][:dim]
        c.assemble_rhs(f)
        c.dirichlet([2], [0.0] * dim)
        om = oracle_mesh(m)
        A_bc, rhs_bc, _, _, flags = fo.linelas_problem(om, mu, nu, f=tuple(f))
        is_dir = fo.dirichlet_rows(flags, (2,), dofs=dim)
        reduced = kind == "rgdsw"
        ck = fedd_lib.COARSE_RGDSW if reduced else fedd_lib.COARSE_GDSW
        tgt = 8 if dim == 3 else 9
        c.schwarz_set_target(tgt, 1.0)
        c.schwarz_set_coarse(cells)
        c.set_option("gdsw_tol", 1e-13)
        its = {}
        for rot in (0, 1):
            c.set_option("gdsw_rotations", rot)
            c.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=ck)
            g, Kinv = c.schwarz_coarse()
            co = fo.CoarseGDSW(A_bc, m["conn"], m["xyz"], is_dir, dim, cells_target=cells, reduced=reduced, rotations=bool(rot))
            nns = dim + ((3 if dim == 3 else 1) if rot else 0)
            assert co.nns == nns and Kinv.shape == (co.n0, co.n0)
            assert co.n0 == nns * (int(np.prod(np.where(co.g >= 2, co.g - 1, 1))) if reduced else int(np.prod(2 * co.g - 1)))
            np.testing.assert_allclose(Kinv, co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
            if rot and not reduced:
                # the functions the device dropped are the oracle's: unit rows of K0^-1 exactly where the oracle keeps none
                # although the entity has free interface dofs
                unit = np.abs(Kinv - np.eye(co.n0)).sum(axis=1) < 1e-14
                assert np.array_equal(unit, ~co.kept)
                if dim == 3 and M % int(co.g[0]) == 0 and co.g.min() >= 2:
                    per_entity = co.kept.reshape(-1, nns).sum(axis=1)
                    assert set(per_entity.tolist()) <= {0, 3, 5, 6}
            node_bin, nb, _ = fo.schwarz_bins(m["xyz"], tgt)
            ras = fo.RAS(A_bc, node_bin, nb, dofs=dim)
            r = np.random.default_rng(5).standard_normal(A_bc.shape[0])
            zc = c.schwarz_coarse_apply(r)
            np.testing.assert_allclose(zc, co.apply(r), rtol=0, atol=1e-10 * np.abs(zc).max(), err_msg="coarse level, rotations %d" % rot)
            z = c.schwarz_apply(r)
            zo = ras.apply(r) + co.apply(r)
            np.testing.assert_allclose(z, zo, rtol=0, atol=1e-10 * np.abs(zo).max(), err_msg="both levels, rotations %d" % rot)
            x, its[rot], rel = c.gmres(None, rtol=1e-12, max_it=400, restart=100, use_prec=True)
            xd = fo.direct_solve(A_bc, rhs_bc)
            np.testing.assert_allclose(x, xd, rtol=0, atol=1e-10 * np.abs(xd).max())
        print("iterations without / with rotations:", its)
        assert its[1] < its[0]
    finally:
        c.close()


@pytest.mark.parametrize("problem,kind", [("laplace", "gdsw"), ("laplace", "rgdsw"), ("elasticity", "gdsw"), ("elasticity", "rgdsw"),
                                          ("cylinder", "gdsw"), ("cylinder", "rgdsw")])
def test_extension_solves_sixteen_columns_at_a_time(fedd_lib, problem, kind):
    """option "gdsw_block": the harmonic extensions solved as stacked systems of sixteen columns (multi.hip: the matrix and the
    local inverses read once per sweep) against the same columns solved one by one -- the same K0^-1 and operator to 1e-10 when
    both are driven to 1e-13, the same outer iteration count at the default extension tolerance; 26 (Laplace) and 78 (elasticity)
    GDSW columns: several batches, the last one partly filled"""
    c = fedd_lib.Context(device=0)
    try:
        M = 12
        if problem == "cylinder":       # unstructured: every subdomain its own inverse (no slab shared through LDS), ragged boxes
            m = fedd_lib.read_mesh(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "DFG3DCylinder_1k.mesh"), 3)
        else:
            m = fedd_lib.structured_mesh(3, 1, M)
        c.mesh_set_dict(m)
        if problem == "cylinder":
            c.pattern_build(3, fedd_lib.BLOCK_DIAG)
            c.assemble(fedd_lib.FORM_LAPLACE_VEC)
            c.assemble_rhs([0.0, 1.0, 0.0])
            c.dirichlet([1, 2], [0.0, 0.0, 0.0])
        elif problem == "laplace":
            c.pattern_build(1, fedd_lib.BLOCK_SCALAR)
            c.assemble(fedd_lib.FORM_LAPLACE)
            c.assemble_rhs([1.0])
            c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        else:
            mu, nu = 2.0e6, 0.4
            c.pattern_build(3, fedd_lib.BLOCK_FULL)
            c.assemble(fedd_lib.FORM_LINELAS, [2.0 * mu * nu / (1.0 - 2.0 * nu), mu])
            c.assemble_rhs([0.0, 1.0, 0.0])
            c.dirichlet([2], [0.0, 0.0, 0.0])
        c.schwarz_set_target(8, 1.0)
        c.schwarz_set_coarse(27)
        ck = fedd_lib.COARSE_RGDSW if kind == "rgdsw" else fedd_lib.COARSE_GDSW
        rng = np.random.default_rng(11)
        r = rng.standard_normal(c.csr_sizes()[0])
        out = {}
        for block in (0, 1):
            c.set_option("gdsw_block", block)
            c.set_option("gdsw_tol", 1e-13)
            c.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=ck)
            Kinv = c.schwarz_coarse()[1]
            z = c.schwarz_apply(r)
            c.set_option("gdsw_tol", 1e-4)      # the default
            c.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=ck)
            x, its, rel = c.gmres(None, rtol=1e-8, max_it=400, restart=100, use_prec=True)
            out[block] = (Kinv, z, its, x)
        np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=1e-10 * np.abs(out[0][0]).max())
        np.testing.assert_allclose(out[1][1], out[0][1], rtol=0, atol=1e-10 * np.abs(out[0][1]).max())
        assert abs(out[1][2] - out[0][2]) <= 1, (out[0][2], out[1][2])
        np.testing.assert_allclose(out[1][3], out[0][3], rtol=0, atol=1e-6 * np.abs(out[0][3]).max())
    finally:
        c.close()
