"""Regenerates the oracle-made fixtures in this directory:  python tests/golden/make_fixtures.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import fedd_oracle as fo  # noqa: E402


def save_csr(prefix, A):
    A = A.tocsr()
    A.sort_indices()
    return {prefix + "_indptr": A.indptr.astype(np.int64), prefix + "_indices": A.indices.astype(np.int64),
            prefix + "_data": A.data}


def main():
    # cfg 1: 2D P1 Laplace on square.mesh, Dirichlet on flags 1,2,3 (left-edge interior flag 4 stays natural)
    m = fo.read_mesh_file(os.path.join(HERE, "square.mesh"), 2, volume_id=10)
    A_bc, rhs_bc, A, rhs, flags = fo.laplace_problem(m, bc_flags=(1, 2, 3))
    x = fo.direct_solve(A_bc, rhs_bc)
    d = dict(conn=m.conn, xyz=m.xyz, flags=flags, rhs=rhs, rhs_bc=rhs_bc, x=x)
    d.update(save_csr("A", A)); d.update(save_csr("Abc", A_bc))
    np.savez_compressed(os.path.join(HERE, "laplace_square_mesh.npz"), **d)
    # N=2, M=2 3D mesh fixture: (rank, local id) -> gid, xyz, flag and all 48 tets per rank
    d = {}
    for r in range(8):
        mm = fo.build_mesh_structured(3, 2, 2, r)
        d["conn_%d" % r] = mm.conn; d["xyz_%d" % r] = mm.xyz; d["gid_rep_%d" % r] = mm.gid_rep
        d["gid_uni_%d" % r] = mm.gid_uni; d["flag_uni_%d" % r] = mm.flag_uni
    mg = fo.build_mesh_structured_global(3, 2, 2)
    A_bc, rhs_bc, A, rhs, flags = fo.laplace_problem(mg)
    d.update(save_csr("A", A)); d.update(save_csr("Abc", A_bc)); d["rhs"] = rhs; d["rhs_bc"] = rhs_bc
    d["x"] = fo.direct_solve(A_bc, rhs_bc)
    np.savez_compressed(os.path.join(HERE, "laplace_cube_N2M2.npz"), **d)
    # 3D elasticity, steadyLinElas_Perf parameters on a 3^3-cell cube
    mm = fo.build_mesh_structured(3, 1, 3)
    A_bc, rhs_bc, A, rhs, flags = fo.linelas_problem(mm, 2.0e6, 0.4)
    d = dict(rhs=rhs, rhs_bc=rhs_bc, x=fo.direct_solve(A_bc, rhs_bc))
    d.update(save_csr("A", A)); d.update(save_csr("Abc", A_bc))
    np.savez_compressed(os.path.join(HERE, "linelas_cube_M3.npz"), **d)


if __name__ == "__main__":
    main()
