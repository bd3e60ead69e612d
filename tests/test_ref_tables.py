"""The product's reference-element tables (libfedd_hip.so) and the oracle's, both against the reference's own
numeric literals (tests/golden/ref_tables.json, written by tests/golden/make_ref_tables.py from FE_def.hpp and
MeshStructured_def.hpp in the build container).  Product and oracle were transcribed by the same hand; this pins
both to the source they follow.  No GPU needed."""
import json
import os

import numpy as np
import pytest

import fedd_oracle as fo

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "ref_tables.json")))
NEN = {(2, 1): 3, (2, 2): 6, (3, 1): 4, (3, 2): 10}
# degrees the hot path can ask for: determineDegree of P1 / P2 mass, stiffness, rhs (+ extra degree) and div blocks
DEGREES = {2: [1, 2, 3, 4, 5], 3: [1, 2, 3, 4, 5]}


def ref_rule(dim, degree):
    d = REF["degree_remap"].get(str(dim), {}).get(str(degree), degree)
    r = REF["quadrature"]["%d,%d" % (dim, d)]
    return np.array(r["points"]), np.array(r["weights"])


@pytest.mark.parametrize("dim", [2, 3])
def test_quadrature_rules_match_the_reference_literals(fedd_lib, dim):
    for deg in DEGREES[dim]:
        rp, rw = ref_rule(dim, deg)
        pp, pw = fedd_lib.fe_quadrature(dim, deg)
        np.testing.assert_allclose(pp, rp, rtol=0, atol=1e-15, err_msg="product points dim %d degree %d" % (dim, deg))
        np.testing.assert_allclose(pw, rw, rtol=0, atol=1e-16, err_msg="product weights dim %d degree %d" % (dim, deg))
        op, ow = fo.quadrature(dim, deg)
        np.testing.assert_allclose(np.asarray(op)[:, :dim], rp, rtol=0, atol=1e-15, err_msg="oracle points")
        np.testing.assert_allclose(ow, rw, rtol=0, atol=1e-16, err_msg="oracle weights")


@pytest.mark.parametrize("dim,fe", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_basis_values_and_gradients_match_the_reference_expressions(fedd_lib, dim, fe):
    ref = REF["basis"]["%d,P%d" % (dim, fe)]
    rpts = np.array(ref["points"]); rphi = np.array(ref["phi"]); rgrad = np.array(ref["grad"])
    nen = NEN[(dim, fe)]

    def ref_at(p):
        k = np.where(np.abs(rpts - p).max(axis=1) < 1e-15)[0]
        assert k.size, "point %r not in the reference table" % (p,)
        return rphi[k[0]], rgrad[k[0]]
    for deg in DEGREES[dim]:
        pts, _ = fedd_lib.fe_quadrature(dim, deg)
        phi, dphi = fedd_lib.fe_basis(dim, nen, deg)
        for q, p in enumerate(pts):
            a, g = ref_at(p)
            np.testing.assert_allclose(phi[q], a, rtol=0, atol=2e-16)
            np.testing.assert_allclose(dphi[q], g, rtol=0, atol=1e-15)
    # the oracle's basis at every tabulated point (quadrature points of all rules + generic interior points)
    np.testing.assert_allclose(fo.phi(dim, "P%d" % fe, rpts), rphi, rtol=0, atol=2e-16)
    np.testing.assert_allclose(fo.grad_phi(dim, "P%d" % fe, rpts), rgrad, rtol=0, atol=1e-15)


@pytest.mark.parametrize("dim", [2, 3])
def test_structured_cell_element_table_matches_the_reference(fedd_lib, dim):
    """one cell (N = 1, M = 1, unit size): element -> corner offsets, product generator and oracle"""
    want = np.array(REF["structured_cells"][str(dim)]["cell_elements"])
    m = fedd_lib.structured_mesh(dim, 1, 1)
    got = np.rint(m["xyz"][m["conn"]]).astype(int)          # [elements, nodes, dim] corner coordinates = offsets
    assert np.array_equal(got, want)
    om = fo.build_mesh_structured(dim, 1, 1)
    assert np.array_equal(np.rint(om.xyz[om.conn]).astype(int), want)
