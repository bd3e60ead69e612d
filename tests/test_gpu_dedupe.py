"""Shared local inverses (option schwarz_dedupe): subdomains whose local matrices agree are inverted once, and the
restricted apply then runs as batched products on the f64 matrix cores (apply_kind 4 forces that kernel on the small
meshes of these tests; by default it takes over from 4096 subdomains).  The operator must be the oracle's in every size
class of the kernel (row tiles x column steps), equal to the unshared path, and bitwise reproducible."""
import os

import numpy as np
import pytest

import fedd_oracle as fo
from test_gpu_parity import oracle_mesh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def ctx(fedd_lib):
    c = fedd_lib.Context(device=0)
    yield c
    c.close()


def problem(fedd_lib, ctx, kind, dim, M):
    m = fedd_lib.structured_mesh(dim, 1, M)
    ctx.mesh_set_dict(m)
    if kind == "laplace":
        ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
        ctx.assemble(fedd_lib.FORM_LAPLACE)
        ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        A_bc = fo.laplace_problem(oracle_mesh(m))[0]
        return m, A_bc, 1
    mu, nu = 1.0, 0.3
    lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
    ctx.pattern_build(dim, fedd_lib.BLOCK_FULL)
    ctx.assemble(fedd_lib.FORM_LINELAS, [lam, mu])
    ctx.dirichlet([2], np.zeros(dim))
    A_bc = fo.linelas_problem(oracle_mesh(m), mu, nu, f=[0.0, 1.0, 0.0][:dim], bc_flags=(2,))[0]
    return m, A_bc, dim


# (problem, dim, cells, nodes per box, overlap): owned rows / columns of the largest subdomain pick the kernel instance
CASES = [("laplace", 3, 14, 27, 1),     # 27 + 74: two row tiles, ten column steps per wave
         ("laplace", 3, 14, 8, 1),      # 8 + 56
         ("laplace", 3, 26, 12, 2),     # two layers of overlap: 27 + 220 columns, sixteen steps
         ("laplace", 3, 27, 64, 1),     # 64 + 144: four row tiles, sixteen steps
         ("laplace", 2, 60, 27, 1),     # 2D boxes
         ("linelas", 3, 16, 8, 1),      # 24 + 114
         ("linelas", 2, 48, 27, 1),     # 50-odd owned rows: four row tiles
         ("linelas", 2, 60, 36, 1)]     # 72 owned rows: six row tiles


@pytest.mark.parametrize("kind,dim,M,target,overlap", CASES)
def test_shared_inverses_give_the_oracle_operator(fedd_lib, ctx, kind, dim, M, target, overlap):
    m, A_bc, dofs = problem(fedd_lib, ctx, kind, dim, M)
    r = np.random.default_rng(M + target).standard_normal(A_bc.shape[0])
    ctx.schwarz_set_target(target, 1.0)
    ctx.set_option("schwarz_dedupe", 0)
    ctx.set_option("apply_kind", 0)
    ctx.schwarz_setup(overlap, fedd_lib.COMBINE_RESTRICTED)
    info0 = ctx.schwarz_info()
    assert info0["n_unique"] == info0["n_subdomains"]
    z0 = ctx.schwarz_apply(r)
    ctx.set_option("schwarz_dedupe", 1)
    ctx.set_option("apply_kind", 4)
    ctx.schwarz_setup(overlap, fedd_lib.COMBINE_RESTRICTED)
    info = ctx.schwarz_info()
    assert info["n_subdomains"] == info0["n_subdomains"] and info["max_size"] == info0["max_size"]
    assert info["max_size"] <= 256
    # a structured mesh repeats itself: few distinct local matrices, and only those are stored
    assert info["n_unique"] * 4 <= info["n_subdomains"], info
    assert info["inverse_bytes"] < info0["inverse_bytes"] / 3
    z1 = ctx.schwarz_apply(r)
    scale = np.abs(z0).max()
    np.testing.assert_allclose(z1, z0, rtol=0, atol=1e-11 * scale)
    assert np.array_equal(ctx.schwarz_apply(r), z1)            # fixed summation order
    # the batch-table kernel (k_apply_bt: what kind 4 runs when every box conforms to its representative) and the chunk-record
    # kernel (k_apply_mfma, kind 6) form the same products in the same order: the same bits
    ctx.set_option("apply_kind", 6)
    assert np.array_equal(ctx.schwarz_apply(r), z1)
    ctx.set_option("apply_kind", 4)
    # ... and the oracle's operator (when the lattice was not refined, whose rule the oracle shares only for Laplace)
    node_bin, nb, _ = fo.schwarz_bins(m["xyz"], target)
    if nb == info["n_subdomains"]:
        ras = fo.RAS(A_bc, node_bin, nb, dofs=dofs, overlap=overlap)
        assert ras.max_size == info["max_size"]
        zo = ras.apply(r)
        np.testing.assert_allclose(z1, zo, rtol=0, atol=1e-10 * np.abs(zo).max())
    # the flat kernel on the shared slabs (what small systems take by default) agrees too
    ctx.set_option("apply_kind", 0)
    np.testing.assert_allclose(ctx.schwarz_apply(r), z0, rtol=0, atol=1e-11 * scale)


def test_solve_with_shared_inverses_takes_the_same_iterations(fedd_lib, ctx):
    m, A_bc, _ = problem(fedd_lib, ctx, "laplace", 3, 20)
    ctx.assemble_rhs([1.0])
    ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    ctx.schwarz_set_target(27, 1.0)
    out = []
    for dedupe, kind in ((0, 0), (1, 4), (1, 0)):
        ctx.set_option("schwarz_dedupe", dedupe)
        ctx.set_option("apply_kind", kind)
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
        x, its, rel = ctx.gmres(None, rtol=1e-12, max_it=300, restart=100, use_prec=True)
        assert rel <= 1e-12
        out.append((x, its))
    assert abs(out[0][1] - out[1][1]) <= 1 and abs(out[0][1] - out[2][1]) <= 1
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=1e-10 * np.abs(out[0][0]).max())


def test_unstructured_mesh_shares_nothing_and_stays_on_the_streaming_kernel(fedd_lib, ctx):
    m = fedd_lib.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    ctx.dirichlet([1, 2, 4], [0.0, 0.0, 0.0])
    A_bc = fo.laplace_problem(oracle_mesh(m), bc_flags=(1, 2, 4))[0]
    ctx.schwarz_set_target(8, 1.0)
    ctx.set_option("schwarz_dedupe", 1)
    ctx.set_option("apply_kind", 4)          # asked for, but without sharing the matrix-core kernel is not taken
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    info = ctx.schwarz_info()
    assert info["n_unique"] * 4 > info["n_subdomains"]
    node_bin, nb, _ = fo.schwarz_bins(m["xyz"], 8)
    ras = fo.RAS(A_bc, node_bin, nb)
    r = np.random.default_rng(3).standard_normal(A_bc.shape[0])
    zo = ras.apply(r)
    np.testing.assert_allclose(ctx.schwarz_apply(r), zo, rtol=0, atol=1e-10 * np.abs(zo).max())


def test_matrix_change_invalidates_the_sharing(fedd_lib, ctx):
    """Scaling some rows of the matrix after a setup: the next setup fingerprints the new entries (nothing cached)."""
    m, A_bc, _ = problem(fedd_lib, ctx, "laplace", 3, 10)
    ctx.schwarz_set_target(27, 1.0)
    ctx.set_option("schwarz_dedupe", 1)
    ctx.set_option("apply_kind", 4)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    r = np.random.default_rng(5).standard_normal(A_bc.shape[0])
    z1 = ctx.schwarz_apply(r)
    ctx.matrix_scale(-1, 2.0)
    ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
    np.testing.assert_allclose(ctx.schwarz_apply(r), 0.5 * z1, rtol=0, atol=1e-12 * np.abs(z1).max())


def test_scaled_copies_of_a_local_matrix_do_not_share_an_inverse(fedd_lib, ctx):
    """A_j = 2 A_i must be two matrices (the fingerprint quantised every row against its own maximum and never
    hashed the scale).  A cube whose nodes are mapped per coordinate by g(t) = t (t <= 1/2), 2 t - 1/2 (t > 1/2): the cells of
    the octant beyond 1/2 are the cells of the first octant doubled, so 3D P1 Laplace entries there are exactly twice those of
    the first octant (K ~ h; powers of two: bit for bit).  One-node boxes + one layer of overlap = the 15-node stencil
    neighbourhood, the same graph in both octants: local matrices that differ by the factor 2 and by nothing else."""
    M = 10
    m = fedd_lib.structured_mesh(3, 1, M)
    g = lambda t: np.where(t <= 0.5, t, 2.0 * t - 0.5)
    m = dict(m)
    m["xyz"] = g(m["xyz"])
    ctx.mesh_set_dict(m)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    # (flags are those of the unit cube's surface, which the map keeps on the surface)
    ctx.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
    om = oracle_mesh(m)
    A_bc = fo.laplace_problem(om)[0]
    rowptr, col, val, gid = ctx.csr_get()
    d = val[rowptr[:-1] + np.array([np.searchsorted(col[rowptr[i]:rowptr[i + 1]], i) for i in range(rowptr.shape[0] - 1)])]
    x = m["xyz"]
    inner1 = np.flatnonzero(np.all((x > 0.15) & (x < 0.35), axis=1))
    inner2 = np.flatnonzero(np.all((x > 0.8) & (x < 1.2), axis=1))
    assert inner1.size and inner2.size
    assert np.allclose(d[inner2], 2.0 * d[inner1][0], rtol=1e-14, atol=0)     # the scenario is what it claims to be
    r = np.random.default_rng(11).standard_normal(A_bc.shape[0])
    node_bin, nb, _ = fo.schwarz_bins(m["xyz"], 1)
    zo = fo.RAS(A_bc, node_bin, nb).apply(r)
    out = {}
    for dedupe, kind in ((0, 0), (1, 0), (1, 4)):
        ctx.schwarz_set_target(1, 1.0)
        ctx.set_option("schwarz_dedupe", dedupe)
        ctx.set_option("apply_kind", kind)
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
        out[(dedupe, kind)] = ctx.schwarz_apply(r)
        info = ctx.schwarz_info()
        assert info["n_subdomains"] == nb
        np.testing.assert_allclose(out[(dedupe, kind)], zo, rtol=0, atol=1e-10 * np.abs(zo).max())
    assert info["n_unique"] < info["n_subdomains"]          # sharing still happens where the matrices ARE equal


@pytest.mark.parametrize("kind,dim,M,target", [("laplace", 3, 20, 27), ("laplace", 2, 60, 27), ("linelas", 3, 12, 8)])
def test_row_hash_fingerprints_find_the_classes_of_the_entry_fingerprints(fedd_lib, ctx, kind, dim, M, target):
    """schwarz_fp_kind 0 (one hash per matrix row, the default) against 1 (entry by entry inside the subdomain): the same
    number of distinct local matrices on the structured meshes, and the same operator."""
    m, A_bc, dofs = problem(fedd_lib, ctx, kind, dim, M)
    r = np.random.default_rng(M).standard_normal(A_bc.shape[0])
    ctx.schwarz_set_target(target, 1.0)
    ctx.set_option("schwarz_dedupe", 1)
    ctx.set_option("apply_kind", 4)
    out = {}
    for fp_kind in (1, 0):
        ctx.set_option("schwarz_fp_kind", fp_kind)
        ctx.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED)
        out[fp_kind] = (ctx.schwarz_info()["n_unique"], ctx.schwarz_apply(r))
    assert out[0][0] == out[1][0] and out[0][0] * 4 <= ctx.schwarz_info()["n_subdomains"]
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=1e-12 * np.abs(out[1][1]).max())
