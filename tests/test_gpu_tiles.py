"""Element-major tile assembly (assemble.hip k_assemble_tiles; the default for the P1 Laplace / vector-Laplace / elasticity forms,
option "asm_tiles") against the pair kernels it replaces and against the oracle: FE::assemblyLaplace / assemblyLaplaceVecField /
assemblyLinElasXDim (FE_def.hpp:604-734, 2739-3040) loop over elements; so does this kernel, once per tile an element touches."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import fedd_oracle as fo
from test_gpu_parity import assert_matrix_close, csr_global, oracle_mesh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def ctx(fedd_lib):
    c = fedd_lib.Context(device=0)
    yield c
    c.close()


def _values(ctx, fedd_lib, form, dofs, mode, params, tiles):
    ctx.set_option("asm_tiles", tiles)
    ctx.pattern_build(dofs, mode)
    ctx.assemble(form, params)
    return ctx.csr_get()[2].copy()


def _meshes(fedd_lib):
    yield "cube 3D M=9", fedd_lib.structured_mesh(3, 1, 9)
    yield "square 2D M=17", fedd_lib.structured_mesh(2, 1, 17)
    yield "cylinder 1k", fedd_lib.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)


def test_tiles_and_pairs_assemble_the_same_matrices(fedd_lib, ctx):
    """same pattern slots, values equal to rounding (Laplace: the same expressions in the same order, bit for bit), bitwise
    reproducible run to run"""
    for name, m in _meshes(fedd_lib):
        dim = m["dim"]
        ctx.mesh_set_dict(m)
        for form, dofs, mode, params in ((fedd_lib.FORM_LAPLACE, 1, fedd_lib.BLOCK_SCALAR, None),
                                         (fedd_lib.FORM_LAPLACE_VEC, dim, fedd_lib.BLOCK_DIAG, None),
                                         (fedd_lib.FORM_LINELAS, dim, fedd_lib.BLOCK_FULL, [1.5, 1.0])):
            v_pairs = _values(ctx, fedd_lib, form, dofs, mode, params, 0)
            v_tiles = _values(ctx, fedd_lib, form, dofs, mode, params, 1)
            v_again = _values(ctx, fedd_lib, form, dofs, mode, params, 1)
            assert np.array_equal(v_tiles, v_again), (name, form)
            assert np.abs(v_tiles - v_pairs).max() <= 4e-16 * np.abs(v_pairs).max(), (name, form)
            if form != fedd_lib.FORM_LINELAS:
                assert np.array_equal(v_tiles, v_pairs), (name, form)
    ctx.set_option("asm_tiles", 1)


@pytest.mark.parametrize("dim,M", [(3, 8), (2, 20)])
def test_tiles_against_the_oracle(fedd_lib, ctx, dim, M):
    m = fedd_lib.structured_mesh(dim, 1, M)
    om = oracle_mesh(m)
    ctx.mesh_set_dict(m)
    ctx.set_option("asm_tiles", 1)
    ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
    ctx.assemble(fedd_lib.FORM_LAPLACE)
    assert_matrix_close(csr_global(ctx, om.n_global)[0], fo.assembly_laplace(om))
    mu, nu = 2.0e6, 0.4
    lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
    ctx.pattern_build(dim, fedd_lib.BLOCK_FULL)
    ctx.assemble(fedd_lib.FORM_LINELAS, [lam, mu])
    assert_matrix_close(csr_global(ctx, dim * om.n_global)[0], fo.assembly_linelas(om, lam, mu))


def test_forms_and_elements_outside_the_tile_kernel_stay_on_the_pair_kernels(fedd_lib, ctx):
    """P2 elements and the mass / divergence forms: asm_tiles changes nothing"""
    m1 = fedd_lib.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)
    mv = fedd_lib.p2_of_p1(m1, volume_id=0)
    ctx.mesh_set_dict(mv)
    a0 = _values(ctx, fedd_lib, fedd_lib.FORM_LAPLACE, 1, fedd_lib.BLOCK_SCALAR, None, 0)
    a1 = _values(ctx, fedd_lib, fedd_lib.FORM_LAPLACE, 1, fedd_lib.BLOCK_SCALAR, None, 1)
    assert np.array_equal(a0, a1)
    ctx.mesh_set_dict(m1)
    b0 = _values(ctx, fedd_lib, fedd_lib.FORM_MASS, 1, fedd_lib.BLOCK_SCALAR, None, 0)
    b1 = _values(ctx, fedd_lib, fedd_lib.FORM_MASS, 1, fedd_lib.BLOCK_SCALAR, None, 1)
    assert np.array_equal(b0, b1)


def test_a_new_mesh_gets_new_tiles(fedd_lib, ctx):
    """the tile structures belong to a mesh: fedd_mesh_set drops them"""
    for M in (6, 11, 6):
        m = fedd_lib.structured_mesh(3, 1, M)
        ctx.mesh_set_dict(m)
        ctx.pattern_build(1, fedd_lib.BLOCK_SCALAR)
        ctx.assemble(fedd_lib.FORM_LAPLACE)
        rowptr, col, val, gid = ctx.csr_get()
        n = rowptr.shape[0] - 1
        A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
        assert np.abs(A @ np.ones(n)).max() <= 1e-13 * np.abs(val).max()       # Laplace rows sum to zero
        assert abs(A - A.T).max() <= 1e-15 * np.abs(val).max()


def test_device_built_tiles_equal_host_built_tiles(fedd_lib, ctx):
    """round 4: the tile structures are built by device kernels (k_tb_*), the round-3 host builder stays as option
    "asm_tiles_host" 1: both give the same matrices bit for bit (the entries of a slot are added in adjacency order either way),
    on the structured grids, on the graded cylinder mesh (whose cells are split), scalar and block forms, whatever the pattern
    the first assembly happened to see (the tiles are built once per mesh, under the first pattern)"""
    for name, m in _meshes(fedd_lib):
        dim = m["dim"]
        forms = ((fedd_lib.FORM_LAPLACE, 1, fedd_lib.BLOCK_SCALAR, None),
                 (fedd_lib.FORM_LAPLACE_VEC, dim, fedd_lib.BLOCK_DIAG, None),
                 (fedd_lib.FORM_LINELAS, dim, fedd_lib.BLOCK_FULL, [1.5, 1.0]))
        for first in range(3):       # which pattern the tile build runs under
            vals = {}
            for host in (1, 0):
                ctx.set_option("asm_tiles_host", host)
                ctx.mesh_set_dict(m)
                order = forms[first:] + forms[:first]
                for form, dofs, mode, params in order:
                    vals[(host, form)] = _values(ctx, fedd_lib, form, dofs, mode, params, 1)
                info = ctx.mesh_setup_info()
                assert info["tiles_state"] == 1 and info["n_tiles"] > 0 and info["tiles_ms"] > 0 and info["adjacency_ms"] > 0, (name, info)
            for form, _, _, _ in forms:
                assert np.array_equal(vals[(1, form)], vals[(0, form)]), (name, first, form)
    ctx.set_option("asm_tiles_host", 0)
