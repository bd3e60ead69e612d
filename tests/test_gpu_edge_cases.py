"""Edge cases of the GPU path: degenerate right-hand sides, extreme restart lengths, the smallest
meshes, subdomain sizes on both sides of the kernels' size classes, 2D vector problems and P2 with
the coarse level, repeated setup on one context."""
import os

import numpy as np
import pytest

import fedd_oracle as fo
from test_gpu_parity import oracle_mesh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx(fedd_lib):
    c = fedd_lib.Context(device=0)
    yield c
    c.close()


def laplace(fedd_lib, ctx, dim, M):
    m = fedd_lib.structured_mesh(dim, 1, M)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.assemble_rhs([1.0])
    ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    A_bc, rhs_bc, _, _, flags = fo.laplace_problem(oracle_mesh(m))
    return m, A_bc, rhs_bc, flags


def test_zero_rhs_and_exact_initial_guess(fedd_lib, ctx):
    m, A_bc, rhs_bc, _ = laplace(fedd_lib, ctx, 3, 4)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    x, its, rel = ctx.gmres(np.zeros_like(rhs_bc), rtol=1e-10, max_it=50, restart=20, use_prec=True)
    assert its == 0 and rel == 0.0 and not x.any()            # b = 0: zero iterations, x = 0


@pytest.mark.parametrize("restart,max_it", [(1, 400), (2, 400), (1000, 60)])
def test_extreme_restart_lengths(fedd_lib, ctx, restart, max_it):
    m, A_bc, rhs_bc, _ = laplace(fedd_lib, ctx, 3, 5)
    ctx.schwarz_set_target(27, 1.0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    x, its, rel = ctx.gmres(None, rtol=1e-10, max_it=max_it, restart=restart, use_prec=True)
    assert rel <= 1e-10 and its <= max_it
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-7 * np.abs(xd).max())


@pytest.mark.parametrize("dim", [2, 3])
def test_one_cell_mesh_all_dirichlet(fedd_lib, ctx, dim):
    """M = 1: every node is on the boundary, the system is the identity; everything still runs."""
    m, A_bc, rhs_bc, _ = laplace(fedd_lib, ctx, dim, 1)
    ctx.schwarz_set_target(27, 1.0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
    x, its, rel = ctx.gmres(None, rtol=1e-12, max_it=10, restart=10, use_prec=True)
    assert its <= 1 and not x.any()


@pytest.mark.parametrize("target", [1, 2, 5, 40, 64])
def test_subdomain_size_classes(fedd_lib, ctx, target):
    """Box sizes from single nodes to 64 nodes: every size class of the local inversion kernels
    (16 T dofs, T = 2 ... 10) and both apply kernels produce the oracle's operator."""
    m, A_bc, rhs_bc, _ = laplace(fedd_lib, ctx, 3, 9)
    ctx.schwarz_set_target(target, 1.0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    node_bin, nb, _ = fo.schwarz_bins(m["xyz"], target)
    ras = fo.RAS(A_bc, node_bin, nb)
    info = ctx.schwarz_info()
    assert info["n_subdomains"] == nb and info["max_size"] == ras.max_size
    r = np.random.default_rng(target).standard_normal(A_bc.shape[0])
    zo = ras.apply(r)
    for kind in (0, 1):
        ctx.set_option("apply_kind", kind)
        np.testing.assert_allclose(ctx.schwarz_apply(r), zo, rtol=0, atol=1e-10 * np.abs(zo).max())
    ctx.set_option("apply_kind", 0)
    # the alternative local-inverse kernel (blocks of four pivots on the f64 matrix cores)
    ctx.set_option("inv_kind", 1)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    np.testing.assert_allclose(ctx.schwarz_apply(r), zo, rtol=0, atol=1e-10 * np.abs(zo).max())
    ctx.set_option("inv_kind", 0)


def test_2d_elasticity_two_level(fedd_lib, ctx):
    m = fedd_lib.structured_mesh(2, 1, 16)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(2, fedd_lib.BLOCK_FULL)
    mu, nu = 1.0, 0.3
    lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
    ctx.assemble(fedd_lib.FORM_LINELAS, [lam, mu])
    ctx.assemble_rhs([0.0, 1.0])
    ctx.dirichlet([2], [0.0, 0.0])
    A_bc, rhs_bc, _, _, flags = fo.linelas_problem(oracle_mesh(m), mu, nu, f=(0.0, 1.0))
    ctx.schwarz_set_target(0, 1.0)
    ctx.schwarz_set_coarse(16)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
    g, Kinv = ctx.schwarz_coarse()
    co = fo.CoarseQ1(A_bc, m["xyz"], np.repeat(np.isin(flags, (2,)), 2), 2, cells_target=16)
    np.testing.assert_allclose(Kinv, co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
    x, its, rel = ctx.gmres(None, rtol=1e-12, max_it=400, restart=200, use_prec=True)
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-9 * np.abs(xd).max())


def test_p2_laplace_two_level_on_unstructured_mesh(fedd_lib, ctx):
    """P2 nodes (vertices + edge midpoints) of the reference's cylinder mesh: the coarse space is
    evaluated at the dof-carrying nodes, whatever they are."""
    m1 = fedd_lib.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)
    m2 = fedd_lib.p2_of_p1(m1, volume_id=0)
    ctx.mesh_set_dict(m2)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.assemble_rhs([1.0])
    ctx.dirichlet([1, 2, 4], [0.0, 0.0, 0.0])
    om = oracle_mesh(m2)
    A_bc, rhs_bc, _, _, flags = fo.laplace_problem(om, bc_flags=(1, 2, 4))
    ctx.schwarz_set_target(1, 1.0)      # P2 neighbourhoods are large: one node per box
    ctx.schwarz_set_coarse(12)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
    g, Kinv = ctx.schwarz_coarse()
    co = fo.CoarseQ1(A_bc, m2["xyz"], np.isin(flags, (1, 2, 4)), 1, cells_target=12)
    np.testing.assert_array_equal(g, co.g)
    np.testing.assert_allclose(Kinv, co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
    x, its, rel = ctx.gmres(None, rtol=1e-12, max_it=600, restart=200, use_prec=True)
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-9 * np.abs(xd).max())


def test_repeated_setup_on_one_context(fedd_lib, ctx):
    """Mesh, pattern, preconditioner and solve repeated with different sizes on the same context:
    buffers grow and are reused, results do not depend on what ran before."""
    out = []
    for M in (6, 3, 6):
        m, A_bc, rhs_bc, _ = laplace(fedd_lib, ctx, 3, M)
        ctx.schwarz_set_target(27, 1.0)
        ctx.schwarz_set_coarse(8 if M == 6 else 1)
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)
        x, its, rel = ctx.gmres(None, rtol=1e-12, max_it=200, restart=100, use_prec=True)
        xd = fo.direct_solve(A_bc, rhs_bc)
        np.testing.assert_allclose(x, xd, rtol=0, atol=1e-9 * np.abs(xd).max())
        out.append((x, its))
    np.testing.assert_array_equal(out[0][0], out[2][0])        # bitwise the same solve
    assert out[0][1] == out[2][1]
    # 8 cells (27 lattice points) on the 4^3-node mesh with 8 free nodes: K0 cannot have full rank
    laplace(fedd_lib, ctx, 3, 3)
    ctx.schwarz_set_coarse(8)
    with pytest.raises(fedd_lib.FeddError, match="27 coarse dofs for 8 free dofs"):
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_Q1)


@pytest.mark.parametrize("two_level", [0, 1])
def test_delayed_and_plain_gram_schmidt_give_the_same_iterates(fedd_lib, ctx, two_level):
    """gmres_kind 0 (DCGS2, delayed second pass) and 1 (plain two-pass CGS2) are the same method in
    exact arithmetic: same iteration count, same residual history end point, same solution."""
    m, A_bc, rhs_bc, _ = laplace(fedd_lib, ctx, 3, 14)
    ctx.schwarz_set_target(27, 1.0)
    ctx.schwarz_set_coarse(27)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=two_level, coarse_kind=fedd_lib.COARSE_Q1 if two_level else 0)
    res = {}
    for kind in (0, 1):
        ctx.set_option("gmres_kind", kind)
        for rtol, restart in ((1e-8, 100), (1e-12, 7)):
            res[(kind, rtol)] = ctx.gmres(None, rtol=rtol, max_it=500, restart=restart, use_prec=True)
    ctx.set_option("gmres_kind", 0)
    xd = fo.direct_solve(A_bc, rhs_bc)
    for rtol in (1e-8, 1e-12):
        x0, its0, rel0 = res[(0, rtol)]
        x1, its1, rel1 = res[(1, rtol)]
        assert abs(its0 - its1) <= 1 and rel0 <= rtol and rel1 <= rtol
        np.testing.assert_allclose(x0, x1, rtol=0, atol=10 * rtol * np.abs(xd).max())
    np.testing.assert_allclose(res[(0, 1e-12)][0], xd, rtol=0, atol=1e-9 * np.abs(xd).max())
    # explicit residual of the delayed variant (the lagged recurrences must not drift from b - A x)
    x0 = res[(0, 1e-12)][0]
    assert np.linalg.norm(rhs_bc - A_bc @ x0) / np.linalg.norm(rhs_bc) < 1e-10


@pytest.mark.parametrize("dim,M", [(3, 13), (2, 50), (3, 1), (3, 2)])
def test_spmv_kernels_agree(fedd_lib, ctx, dim, M):
    """The three SpMV kernels (`spmv_kind` 0 = CSR-window, 1 = row-per-lane-group, 2 = CSR-stream) on matrices of
    less than one window, one window and many windows: against the oracle, and the two windowed kernels
    bit for bit (same products, same summation order)."""
    m, A_bc, rhs_bc, flags = laplace(fedd_lib, ctx, dim, M)
    x = np.random.default_rng(11).standard_normal(A_bc.shape[0])
    ctx.set_option("spmv_exact_public", 0)     # fedd_spmv = the solver's stream in this test
    yo = fo.spmv(A_bc, x)
    ys = {}
    try:
        for kind in (0, 1, 2):
            ctx.set_option("spmv_kind", kind)
            ys[kind] = ctx.spmv(x)
            np.testing.assert_allclose(ys[kind], yo, rtol=0, atol=1e-13 * np.abs(yo).max())
    finally:
        ctx.set_option("spmv_kind", 0)
    # The default kernel streams a solver-private compacted copy of the rows; fedd_csr_get still returns the
    # reference pattern and values.
    rowptr, col, val, _ = ctx.csr_get()
    n = rowptr.shape[0] - 1
    row_of = np.repeat(np.arange(n), np.diff(rowptr))
    rowmax = np.zeros(n)
    np.maximum.at(rowmax, row_of, np.abs(val))
    info = ctx.spmv_info()
    assert info["nnz_pattern"] == val.shape[0] == A_bc.nnz
    # default: entries below one ulp of their row's largest entry are left out (exact zeros and cancellation noise)
    assert info["nnz_streamed"] == np.count_nonzero(np.abs(val) > 2.0 ** -52 * rowmax[row_of]) < info["nnz_pattern"]
    try:
        # tolerance 0: exactly the entries that are 0.0 are left out, and y is bit for bit the y of the parity CSR
        # (same products, same summation order: CSR-stream, kind 2, reads the parity CSR)
        ctx.set_option("spmv_drop_tol", 0.0)
        y_exact = ctx.spmv(x)
        assert ctx.spmv_info()["nnz_streamed"] == np.count_nonzero(val)
        assert np.array_equal(y_exact, ys[2])
        ctx.set_option("spmv_compact", 0)
        assert np.array_equal(ctx.spmv(x), ys[2])
        assert ctx.spmv_info()["nnz_streamed"] == info["nnz_pattern"]
    finally:
        ctx.set_option("spmv_compact", 1)
        ctx.set_option("spmv_drop_tol", 2.0 ** -52)
    # what the default tolerance changes is below the rounding error of the row sums
    assert np.abs(ys[0] - ys[2]).max() <= 2.0 ** -50 * np.abs(A_bc).max() * np.abs(x).max()
    # the caller's own product (the default of fedd_spmv) is formed with the parity CSR, whatever the solver streams
    ctx.set_option("spmv_exact_public", 1)
    assert np.array_equal(ctx.spmv(x), ys[2])
    ctx.set_option("spmv_exact_public", 0)
    # the compacted copy follows the matrix when it changes
    ctx.matrix_scale(-1, -2.5)
    np.testing.assert_allclose(ctx.spmv(x), -2.5 * yo, rtol=0, atol=1e-12 * np.abs(yo).max())
    ctx.matrix_scale(-1, -0.4)
    ctx.set_option("spmv_exact_public", 1)


def test_box_lattice_refines_itself_when_a_subdomain_would_exceed_the_dense_solver(fedd_lib, ctx):
    """125-node boxes plus overlap are far beyond the 256 dofs the dense local solver takes: the setup refines
    the lattice (box edge x 0.85 per attempt) until every subdomain fits, and the solve is still exact."""
    m, A_bc, rhs_bc, flags = laplace(fedd_lib, ctx, 3, 14)
    ctx.schwarz_set_target(125, 1.0)
    try:
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
        info = ctx.schwarz_info()
        assert info["max_size"] <= 256 and info["n_subdomains"] > 27       # 3^3 boxes of 125 nodes were asked for
        x, its, rel = ctx.gmres(None, rtol=1e-12, max_it=300, restart=100, use_prec=True)
        xd = fo.direct_solve(A_bc, rhs_bc)
        np.testing.assert_allclose(x, xd, rtol=0, atol=1e-9 * np.abs(xd).max())
    finally:
        ctx.schwarz_set_target(0, 1.0)


@pytest.mark.parametrize("dim,M,dofs", [(3, 20, 1), (2, 70, 1), (3, 10, 3)])
def test_spmv_column_patterns_give_the_same_bits(fedd_lib, ctx, dim, M, dofs):
    """spmv_pattern: rows that repeat their column offsets share an offset list and the solver's stream carries values
    only.  Forced on (2) on these small matrices (by default only matrices beyond the Infinity Cache take it): a handful of
    patterns covers the structured mesh, and y is the y of the per-entry kernel bit for bit (same products, same order)."""
    m = fedd_lib.structured_mesh(dim, 1, M)
    ctx.mesh_set_dict(m)
    if dofs == 1:
        ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
        ctx.assemble(fedd_lib.FORM_LAPLACE)
        ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    else:
        ctx.pattern_build(dofs, fedd_lib.BLOCK_FULL)
        ctx.assemble(fedd_lib.FORM_LINELAS, [1.5, 1.0])
        ctx.dirichlet([2], np.zeros(dofs))
    nr = ctx.csr_sizes()[0]
    x = np.random.default_rng(7).standard_normal(nr)
    ctx.set_option("spmv_pattern", 0)
    ctx.set_option("spmv_exact_public", 0)     # fedd_spmv = the solver's stream in this test
    y0 = ctx.spmv(x)
    assert ctx.spmv_info()["column_patterns"] == 0
    try:
        ctx.set_option("spmv_pattern", 2)
        for nu in (0, 2, 3, 4, 5, 6, 7, 8):
            ctx.set_option("spmv_pat_nu", nu)
            y1 = ctx.spmv(x)
            info = ctx.spmv_info()
            assert np.array_equal(y1, y0), (nu, np.abs(y1 - y0).max())
            if dofs == 1:       # scalar stencil: every row finds a pattern, and there are few
                assert 1 <= info["column_patterns"] <= 64 and info["rows_with_explicit_columns"] == 0, info
        # the elasticity rows (3 x 15 entries) are longer than the 16 entries the pattern KERNEL unrolls: the dictionary holds
        # them (48 offsets per pattern, round 4) for the row classes' sake -- with classes the class kernel runs, without
        # them such a matrix goes back to the per-entry kernel (dictionary off)
        if dofs > 1:
            assert info["row_classes"] > 0 or info["column_patterns"] == 0, info
    finally:
        ctx.set_option("spmv_pattern", 1)
        ctx.set_option("spmv_exact_public", 1)
        ctx.set_option("spmv_pat_nu", 0)


def test_spmv_column_patterns_fall_back_on_an_unstructured_mesh(fedd_lib, ctx):
    m = fedd_lib.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.dirichlet([1, 2, 4], [0.0, 0.0, 0.0])
    nr = ctx.csr_sizes()[0]
    x = np.random.default_rng(8).standard_normal(nr)
    ctx.set_option("spmv_pattern", 0)
    y0 = ctx.spmv(x)
    try:
        ctx.set_option("spmv_pattern", 2)
        y1 = ctx.spmv(x)
        info = ctx.spmv_info()
        assert np.array_equal(y1, y0)
        # no two rows of an unstructured mesh share their offsets (beyond the Dirichlet rows' single entry): not used
        assert info["column_patterns"] == 0 and info["rows_with_explicit_columns"] == nr, info
    finally:
        ctx.set_option("spmv_pattern", 1)



@pytest.mark.parametrize("dim,M,dofs", [(3, 20, 1), (2, 70, 1), (3, 10, 3)])
def test_spmv_sixteen_bit_columns_give_the_same_bits(fedd_lib, ctx, dim, M, dofs):
    """option "spmv_col16" (default on): the per-entry window kernel reads its column indices as 16-bit offsets from a base per
    window of the stream -- the same entries in the same order, so y is bit for bit the y of the 32-bit stream"""
    m = fedd_lib.structured_mesh(dim, 1, M)
    ctx.mesh_set_dict(m)
    if dofs == 1:
        ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
        ctx.assemble(fedd_lib.FORM_LAPLACE)
        ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    else:
        ctx.pattern_build(dofs, fedd_lib.BLOCK_FULL)
        ctx.assemble(fedd_lib.FORM_LINELAS, [1.5, 1.0])
        ctx.dirichlet([2], np.zeros(dofs))
    x = np.random.default_rng(17).standard_normal(ctx.csr_sizes()[0])
    ctx.set_option("spmv_exact_public", 0)     # fedd_spmv = the solver's stream in this test
    try:
        ys = {}
        for on in (0, 1):
            ctx.set_option("spmv_col16", on)
            for nu in (0, 4, 6, 8):
                ctx.set_option("spmv_win_nu", nu)
                ys[(on, nu)] = ctx.spmv(x)
                assert ctx.spmv_info()["column_index_bytes"] == (2 if on else 4)
        for key, y in ys.items():
            assert np.array_equal(y, ys[(0, 0)]), key
    finally:
        ctx.set_option("spmv_col16", 1)
        ctx.set_option("spmv_win_nu", 0)
        ctx.set_option("spmv_exact_public", 1)


def test_spmv_sixteen_bit_columns_are_not_taken_where_a_window_spans_too_many_columns(fedd_lib, ctx):
    """a numbering that puts neighbours 40 000 rows apart: windows of the stream that span more than 65 535 columns keep their
    32-bit indices (decided from the data, window by window), the others take the 16-bit offsets, and the product is unchanged"""
    M = 42      # 43^3 = 79 507 nodes
    m = fedd_lib.structured_mesh(3, 1, M)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    x = np.random.default_rng(18).standard_normal(ctx.csr_sizes()[0])
    ctx.set_option("spmv_exact_public", 0)
    ctx.set_option("spmv_classes", 0)          # (this test is about the per-entry kernel's column indices)
    try:
        y = ctx.spmv(x)
        # the natural numbering of the cube couples nodes 43^2 apart: windows of 2048 entries span ~2 * 1849 + a few hundred columns
        info = ctx.spmv_info()
        assert info["column_index_bytes"] == 2 and info["entries_with_32bit_columns"] == 0
        # renumbered so that neighbours lie far apart: every second node sent to the far end
        n = m["xyz"].shape[0]
        perm = np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)])       # new id -> old id
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        m2 = dict(m)
        m2["xyz"] = m["xyz"][perm]
        for key in ("gid_rep", "flag_rep", "gid_uni", "flag_uni"):
            m2[key] = m[key][perm]
        m2["conn"] = inv[m["conn"]].astype(m["conn"].dtype)
        ctx.mesh_set_dict(m2)
        ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
        ctx.assemble(fedd_lib.FORM_LAPLACE)
        y2 = ctx.spmv(x[perm])
        info = ctx.spmv_info()
        assert info["column_index_bytes"] == 2 and 0 < info["entries_with_32bit_columns"] <= info["nnz_streamed"]
        np.testing.assert_allclose(y2, y[perm], rtol=0, atol=1e-12 * np.abs(y).max())
        ctx.set_option("spmv_col16", 0)
        assert np.array_equal(ctx.spmv(x[perm]), y2) and ctx.spmv_info()["column_index_bytes"] == 4
        ctx.set_option("spmv_col16", 1)
    finally:
        ctx.set_option("spmv_exact_public", 1)
        ctx.set_option("spmv_classes", 1)


@pytest.mark.parametrize("dim,M,dofs", [(3, 21, 1), (3, 16, 1), (2, 75, 1), (3, 13, 3), (2, 40, 2)])
def test_spmv_row_classes_give_the_same_bits(fedd_lib, ctx, dim, M, dofs):
    """spmv_classes (round 4): rows that repeat their column pattern AND their values bit for bit share a class, the SpMV reads a
    2-byte class id per row and the values from a table.  On a structured grid whose spacing is no power of two the assembled
    rows differ in their last bits by where the coordinates round -- several hundred classes --, with a power of two there is
    one class per pattern; either way y is the y of the pattern kernel and of the per-entry kernel BIT FOR BIT, a whole solve
    gives the same iterates (bitwise the same solution), and the rows that are in no class keep their stream entries."""
    m = fedd_lib.structured_mesh(dim, 1, M)
    ctx.mesh_set_dict(m)
    if dofs == 1:
        ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
        ctx.assemble(fedd_lib.FORM_LAPLACE)
        ctx.assemble_rhs([1.0])
        ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    else:       # elasticity, full node blocks: 3 x 15 (2 x 7) entries per row, a pattern per component
        ctx.pattern_build(dofs, fedd_lib.BLOCK_FULL)
        ctx.assemble(fedd_lib.FORM_LINELAS, [1.5e6, 1.0e6])
        ctx.assemble_rhs([0.0, 1.0, 0.0][:dofs])
        ctx.dirichlet([2], np.zeros(dofs))
    nr = ctx.csr_sizes()[0]
    x = np.random.default_rng(9).standard_normal(nr)
    ctx.set_option("spmv_exact_public", 0)     # fedd_spmv = the solver's stream in this test
    try:
        ctx.set_option("spmv_pattern", 0)
        y0 = ctx.spmv(x)
        ctx.set_option("spmv_pattern", 2)
        ctx.set_option("spmv_classes", 0)
        y1 = ctx.spmv(x)
        info1 = ctx.spmv_info()
        assert info1["row_classes"] == 0 and (info1["column_patterns"] >= 1 or dofs > 1)     # (long rows without classes: per-entry kernel)
        ctx.set_option("spmv_classes", 1)
        y2 = ctx.spmv(x)
        info2 = ctx.spmv_info()
        assert np.array_equal(y1, y0) and np.array_equal(y2, y0)
        assert 1 <= info2["row_classes"] <= 16384 and info2["rows_in_classes"] >= 0.9 * nr, info2
        # every stream entry is accounted for: in a class or outside
        rowptr, col, val, _ = ctx.csr_get()
        assert info2["nnz_streamed_outside_classes"] <= info2["nnz_streamed"]
        if M & (M - 1) == 0 and dofs == 1:      # h a power of two: the arithmetic is exact, one class per (pattern, Dirichlet or not)
            assert info2["row_classes"] <= 2 * info2["column_patterns"], info2
        # a solve on either stream: the same bits all the way (the shifted epilogue of the Newton basis included)
        ctx.schwarz_set_target((27 if dim == 3 else 16) if dofs == 1 else 8, 1.0)
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
        ctx.set_option("gmres_s", 16)
        sol = {}
        for cls in (0, 1):
            ctx.set_option("spmv_classes", cls)
            sol[cls] = ctx.gmres(None, rtol=1e-10 if dofs == 1 else 1e-6, max_it=400, restart=100, use_prec=True)
        assert sol[0][1] == sol[1][1] and np.array_equal(sol[0][0], sol[1][0])
    finally:
        ctx.set_option("spmv_pattern", 1)
        ctx.set_option("spmv_classes", 1)
        ctx.set_option("spmv_exact_public", 1)
        ctx.set_option("gmres_s", 0)


def test_spmv_row_classes_stay_off_where_rows_do_not_repeat(fedd_lib, ctx):
    m = fedd_lib.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.dirichlet([1, 2, 4], [0.0, 0.0, 0.0])
    nr = ctx.csr_sizes()[0]
    x = np.random.default_rng(8).standard_normal(nr)
    ctx.set_option("spmv_exact_public", 0)
    try:
        ctx.set_option("spmv_pattern", 0)
        y0 = ctx.spmv(x)
        ctx.set_option("spmv_pattern", 2)
        y1 = ctx.spmv(x)
        assert np.array_equal(y1, y0) and ctx.spmv_info()["row_classes"] == 0
    finally:
        ctx.set_option("spmv_pattern", 1)
        ctx.set_option("spmv_exact_public", 1)


def test_spmv_dictionary_is_kept_only_while_the_matrix_matches_it(fedd_lib, ctx):
    """option spmv_keep_dictionary: a reassembled matrix that matches the previous pattern dictionary and row classes bit for bit
    keeps them (one verifying pass instead of the build); a matrix with other values gets new ones.  The products are the same
    bits either way."""
    m = fedd_lib.structured_mesh(3, 1, 19)
    ctx.mesh_set_dict(m)
    x = np.random.default_rng(21).standard_normal(20 ** 3)
    ctx.set_option("spmv_exact_public", 0)
    ctx.set_option("spmv_pattern", 2)
    try:
        ys = {}
        for keep in (0, 1):
            ctx.set_option("spmv_keep_dictionary", keep)
            for rep in range(3):        # (keep: build, verify, verify)
                ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
                ctx.assemble(fedd_lib.FORM_LAPLACE)
                ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
                ys[(keep, rep)] = ctx.spmv(x)
                info = ctx.spmv_info()
                assert info["row_classes"] >= 1 and info["rows_in_classes"] >= 0.9 * x.shape[0]
            # other values (an exact scaling): the kept classes no longer match, new ones are built
            ctx.matrix_scale(-1, 2.0)
            ys[(keep, "scaled")] = ctx.spmv(x)
            assert ctx.spmv_info()["row_classes"] >= 1
        y0 = ys[(0, 0)]
        for key, y in ys.items():
            assert np.array_equal(y, 2.0 * y0 if key[1] == "scaled" else y0), key
    finally:
        ctx.set_option("spmv_keep_dictionary", 0)
        ctx.set_option("spmv_pattern", 1)
        ctx.set_option("spmv_exact_public", 1)
