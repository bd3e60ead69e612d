"""The C++ host facade (namespace FEDD: Domain / BCBuilder / Laplace / Problem::solve, mirroring
feddlib/problems/tests/laplace/main.cpp) driven end to end on the GPU through the C ABI and
checked against the oracle.  Reads like the reference's own laplace test: same XML files, same
call sequence; unlike it, this one asserts on the numbers."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import fedd_oracle as fo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XML = os.path.join(ROOT, "tests", "golden", "laplace_xml")


@pytest.fixture(scope="module")
def driver(fedd_lib):
    from feddlib_amd import build
    return build.build_driver(verbose=False)


def run_driver(driver, tmp_path, problem_xml, prec_xml, solver_xml):
    out = tmp_path / "sol.txt"
    r = subprocess.run([driver, "--problemfile=%s" % problem_xml, "--precfile=%s" % prec_xml,
                        "--solverfile=%s" % solver_xml, "--out=%s" % out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"iterations (\d+) relres (\S+)", r.stdout)
    assert m, r.stdout
    sol = np.loadtxt(out)
    x = np.zeros(int(sol[:, 0].max()) + 1)
    x[sol[:, 0].astype(int)] = sol[:, 1]
    return x, int(m.group(1)), float(m.group(2)), r.stdout


def test_reference_laplace_xml_files_2d(driver, tmp_path):
    """The reference's own parameter files, unchanged: 2D, P1, structured, H/h = 10, GMRES 1e-8."""
    x, its, rel, log = run_driver(driver, tmp_path, os.path.join(XML, "parametersProblem.xml"),
                                  os.path.join(XML, "parametersPrec.xml"), os.path.join(XML, "parametersSolver.xml"))
    m = fo.build_mesh_structured(2, 1, 10)
    A_bc, rhs_bc, _, _, _ = fo.laplace_problem(m)
    xd = fo.direct_solve(A_bc, rhs_bc)
    assert rel <= 1e-8 and 0 < its <= 100
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-6 * np.abs(xd).max())      # tolerance-limited (1e-8 residual)
    assert "Q1-lattice coarse space" in log        # TwoLevel=true in that file: honoured, and the substitution is said


def test_3d_tight_tolerance_matches_oracle(driver, tmp_path):
    prob = tmp_path / "p.xml"
    shutil.copy(os.path.join(XML, "parametersProblem.xml"), prob)
    txt = prob.read_text().replace('name="Dimension" type="int" value="2"', 'name="Dimension" type="int" value="3"') \
                          .replace('name="H/h" type="int" value="10"', 'name="H/h" type="int" value="9"')
    prob.write_text(txt)
    sol = tmp_path / "s.xml"
    sol.write_text(open(os.path.join(XML, "parametersSolver.xml")).read()
                   .replace('value="1e-8"', 'value="1e-13"').replace('"Maximum Iterations" type="int" value="100"',
                                                                     '"Maximum Iterations" type="int" value="400"'))
    prec = tmp_path / "c.xml"
    prec.write_text(open(os.path.join(XML, "parametersPrec.xml")).read()
                    .replace('name="Combine Values in Overlap" type="string" value="Averaging"',
                             'name="Combine Values in Overlap" type="string" value="Restricted"'))
    x, its, rel, log = run_driver(driver, tmp_path, prob, prec, sol)
    m = fo.build_mesh_structured(3, 1, 9)
    A_bc, rhs_bc, _, _, _ = fo.laplace_problem(m)
    xd = fo.direct_solve(A_bc, rhs_bc)
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-10 * np.abs(xd).max())
    assert rel <= 1e-13


LINELAS_XML = os.path.join(ROOT, "tests", "golden", "linelas_xml")


@pytest.fixture(scope="module")
def linelas_driver(fedd_lib):
    from feddlib_amd import build
    return build.build_driver(verbose=False, which="linelas")


def test_reference_steady_linelas_perf_xml_files(linelas_driver, tmp_path):
    """The reference's steadyLinElas_Perf parameter files, unchanged (3D, P1, H/h = 4, mu = 2e6,
    nu = 0.4, volume force 1, Dirichlet on flag 2, one-level FROSch, Block GMRES 1e-6): the driver's
    displacement against a direct solve of the oracle's system."""
    x, its, rel, log = run_driver(linelas_driver, tmp_path, os.path.join(LINELAS_XML, "parametersProblem.xml"),
                                  os.path.join(LINELAS_XML, "parametersPrec.xml"),
                                  os.path.join(LINELAS_XML, "parametersSolver.xml"))
    m = fo.build_mesh_structured(3, 1, 4)
    A_bc, rhs_bc, _, _, _ = fo.linelas_problem(m, 2.0e6, 0.4, f=(0.0, 1.0, 0.0), bc_flags=(2,))
    xd = fo.direct_solve(A_bc, rhs_bc)
    assert rel <= 1e-6 and 0 < its <= 100
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-4 * np.abs(xd).max())      # tolerance-limited (1e-6 residual)
    assert "Solve Problem" in log


def test_linelas_two_level_tight_tolerance(linelas_driver, tmp_path):
    prob = tmp_path / "p.xml"
    prob.write_text(open(os.path.join(LINELAS_XML, "parametersProblem.xml")).read()
                    .replace('name="H/h"							    	type="int"   	value="4"', 'name="H/h" type="int" value="8"'))
    assert 'name="H/h" type="int" value="8"' in prob.read_text()
    prec = tmp_path / "c.xml"
    prec.write_text(open(os.path.join(LINELAS_XML, "parametersPrec.xml")).read()
                    .replace('name="TwoLevel"                                          type="bool"     value="false"',
                             'name="TwoLevel" type="bool" value="true"'))
    assert 'name="TwoLevel" type="bool" value="true"' in prec.read_text()
    sol = tmp_path / "s.xml"
    sol.write_text(open(os.path.join(LINELAS_XML, "parametersSolver.xml")).read()
                   .replace('"Convergence Tolerance" type="double" value="1e-6"', '"Convergence Tolerance" type="double" value="1e-13"')
                   .replace('"Maximum Iterations" type="int" value="100"', '"Maximum Iterations" type="int" value="500"'))
    x, its, rel, log = run_driver(linelas_driver, tmp_path, prob, prec, sol)
    m = fo.build_mesh_structured(3, 1, 8)
    A_bc, rhs_bc, _, _, _ = fo.linelas_problem(m, 2.0e6, 0.4, f=(0.0, 1.0, 0.0), bc_flags=(2,))
    xd = fo.direct_solve(A_bc, rhs_bc)
    assert rel <= 1e-13
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-9 * np.abs(xd).max())
    assert "Q1-lattice coarse space" in log
