"""CPU oracle for the FEDDLib assembly + Schwarz/GMRES hot path  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (feddlib_amd/, include/) never does.

PARITY STATUS
  * assembly half (mesh generator, quadrature, basis, element matrices, CSR fill,
    Dirichlet rows): a restatement of the reference C++ arithmetic, each function citing the
    reference file:line it follows (paths relative to /root/reference).  The reference ships
    NO golden numbers (every test passes on exit code only) and cannot be
    built here (Trilinos/Boost/METIS absent), so this half is pinned by analytic known-answer
    tests only (tests/test_oracle_known_answers.py): reference-tet stiffness/mass/rhs,
    quadrature exactness, partition of unity, Kuhn-cube 7-point stencil, rigid-body modes, and the
    whole chain generator -> assembly -> Dirichlet rows -> solve against the Fourier-series value of
    -Laplace u = 1 at the centre of the unit cube (second-order convergence).
    => "parity unpinned" in the sense of the task statement: no reference-produced vector exists.
  * solve half (GMRES + one-level overlapping Schwarz): the arithmetic lives in Trilinos
    (Belos / ShyLU_DDFROSch / Amesos2-KLU), an un-vendored and un-pinned dependency of the
    reference (cmake/TPLs/FindTPLTrilinos.cmake:64-65).  This file restates the *published*
    algorithms (right-preconditioned restarted GMRES, Saad 2003 Alg. 9.5; restricted additive
    Schwarz, Cai & Sarkis 1999) configured as the reference's call sites configure them
    (feddlib/problems/tests/laplace/parametersSolver.xml:5-15,
    feddlib/problems/tests/steadyLinElas_Perf/parametersPrec.xml:7-25).  "parity unpinned";
    pinned by mathematical invariants (direct-solve agreement, 1 subdomain => 1 iteration).

All heavy loops are numpy-vectorised over elements; `*_loops` variants restate the reference's
literal loop nest for small cases and are checked against the vectorised ones in the tests.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

EPS = np.finfo(np.float64).eps


# --------------------------------------------------------------------------------------
# a3: quadrature  (feddlib/core/FE/FE_def.hpp:6023-6727)
# --------------------------------------------------------------------------------------
def quadrature(dim: int, degree: int):
    """Reference-simplex rule. Returns (points[Q,dim], weights[Q]).

    2D: FE_def.hpp:6067-6239 (deg 3,4 -> 5; 6 -> 7).  3D tets: FE_def.hpp:6241-6460
    (deg 2 -> 3, 4 -> 5).  Degrees 7 (2D, 28 pt) and 6 (3D, Keast 24 pt) are not on the
    hot path and raise."""
    if dim == 2:
        if degree in (3, 4):
            degree = 5
        if degree == 1:  # :6076-6083
            return np.array([[1 / 3.0, 1 / 3.0]]), np.array([1 / 2.0])
        if degree == 2:  # :6085-6103
            a = 1 / 6.0
            return np.array([[0.5, 0.5], [0.0, 0.5], [0.5, 0.0]]), np.array([a, a, a])
        if degree == 5:  # :6105-6142
            a = 0.470142064105115
            b = 0.101286507323456
            P1 = 0.066197076394253
            P2 = 0.062969590272413
            pts = np.array([[1 / 3.0, 1 / 3.0], [a, a], [1 - 2.0 * a, a], [a, 1 - 2.0 * a],
                            [b, b], [1 - 2.0 * b, b], [b, 1 - 2.0 * b]])
            w = np.array([9 / 80.0, P1, P1, P1, P2, P2, P2])
            return pts, w
        raise NotImplementedError("2D quadrature degree %d not on the hot path" % degree)
    if dim == 3:
        if degree == 2:
            degree = 3
        if degree == 4:
            degree = 5
        if degree == 1:  # :6253-6260
            return np.array([[0.25, 0.25, 0.25]]), np.array([1 / 6.0])
        if degree == 3:  # :6262-6293
            a, b, c = 0.25, 1.0 / 6.0, 0.5
            pts = np.array([[a, a, a], [b, b, b], [b, b, c], [b, c, b], [c, b, b]])
            w = np.array([-2.0 / 15.0, 3.0 / 40.0, 3.0 / 40.0, 3.0 / 40.0, 3.0 / 40.0])
            return pts, w
        if degree == 5:  # :6366-6460
            s15 = math.sqrt(15.0)
            a = 0.25
            b1 = (7.0 + s15) / 34.0
            b2 = (7.0 - s15) / 34.0
            c1 = (13.0 - 3.0 * s15) / 34.0
            c2 = (13.0 + 3.0 * s15) / 34.0
            d = (5.0 - s15) / 20.0
            e = (5.0 + s15) / 20.0
            pts = np.array([[a, a, a],
                            [b1, b1, b1], [b1, b1, c1], [b1, c1, b1], [c1, b1, b1],
                            [b2, b2, b2], [b2, b2, c2], [b2, c2, b2], [c2, b2, b2],
                            [d, d, e], [d, e, d], [e, d, d], [d, e, e], [e, d, e], [e, e, d]])
            P1 = (2665.0 - 14.0 * s15) / 226800.0
            P2 = (2665.0 + 14.0 * s15) / 226800.0
            b = 5.0 / 567.0
            w = np.array([8.0 / 405.0, P1, P1, P1, P1, P2, P2, P2, P2, b, b, b, b, b, b])
            return pts, w
        raise NotImplementedError("3D quadrature degree %d not on the hot path" % degree)
    raise ValueError("dim must be 2 or 3")


# --------------------------------------------------------------------------------------
# a4: basis functions  (FE_def.hpp:4947-5087 phi, :5565-5713 gradPhi)
# --------------------------------------------------------------------------------------
def nodes_per_element(dim: int, fe: str) -> int:
    return {(2, "P1"): 3, (2, "P2"): 6, (3, "P1"): 4, (3, "P2"): 10}[(dim, fe)]


def phi(dim: int, fe: str, p: np.ndarray) -> np.ndarray:
    """phi[Q, nen] at points p[Q, dim]  (FE_def.hpp:5000-5087)."""
    p = np.atleast_2d(p)
    x = p[:, 0]
    y = p[:, 1]
    if dim == 2:
        if fe == "P1":  # :5000-5011
            return np.stack([1.0 - x - y, x, y], axis=1)
        if fe == "P2":  # :5013-5033
            l0 = 1.0 - x - y
            return np.stack([-l0 * (1 - 2.0 * l0), -x * (1 - 2 * x), -y * (1 - 2 * y),
                             4 * x * l0, 4 * x * y, 4 * y * l0], axis=1)
    if dim == 3:
        z = p[:, 2]
        if fe == "P1":  # :5039-5054
            return np.stack([1.0 - x - y - z, x, y, z], axis=1)
        if fe == "P2":  # :5055-5087  (edge order 4=(0,1) 5=(1,2) 6=(0,2) 7=(0,3) 8=(1,3) 9=(2,3))
            l0 = 1.0 - x - y - z
            return np.stack([l0 * (1 - 2 * x - 2 * y - 2 * z), x * (2 * x - 1), y * (2 * y - 1),
                             z * (2 * z - 1), 4 * x * l0, 4 * x * y, 4 * y * l0, 4 * z * l0,
                             4 * x * z, 4 * y * z], axis=1)
    raise NotImplementedError((dim, fe))


def grad_phi(dim: int, fe: str, p: np.ndarray) -> np.ndarray:
    """dphi[Q, nen, dim] at reference points p[Q, dim]  (FE_def.hpp:5580-5713)."""
    p = np.atleast_2d(p)
    Q = p.shape[0]
    x = p[:, 0]
    y = p[:, 1]
    o = np.ones(Q)
    zr = np.zeros(Q)
    if dim == 2:
        if fe == "P1":  # :5580-5594
            g = [[-o, -o], [o, zr], [zr, o]]
        elif fe == "P2":  # :5596-5622
            g = [[1.0 - 4.0 * (1 - x - y), 1.0 - 4.0 * (1 - x - y)],
                 [4.0 * x - 1, zr], [zr, 4.0 * y - 1],
                 [4 * (1.0 - 2 * x - y), -4 * x], [4.0 * y, 4.0 * x],
                 [-4.0 * y, 4 * (1.0 - x - 2 * y)]]
        else:
            raise NotImplementedError(fe)
    elif dim == 3:
        z = p[:, 2]
        if fe == "P1":  # :5637-5659
            g = [[-o, -o, -o], [o, zr, zr], [zr, o, zr], [zr, zr, o]]
        elif fe == "P2":  # :5661-5713
            s = -3.0 + 4.0 * x + 4.0 * y + 4.0 * z
            g = [[s, s, s],
                 [4.0 * x - 1, zr, zr], [zr, 4.0 * y - 1, zr], [zr, zr, 4.0 * z - 1],
                 [4.0 - 8.0 * x - 4.0 * y - 4.0 * z, -4.0 * x, -4.0 * x],
                 [4.0 * y, 4.0 * x, zr],
                 [-4.0 * y, 4.0 - 4.0 * x - 8.0 * y - 4.0 * z, -4.0 * y],
                 [-4.0 * z, -4.0 * z, 4.0 - 4.0 * x - 4.0 * y - 8.0 * z],
                 [4.0 * z, zr, 4.0 * x],
                 [zr, 4.0 * z, 4.0 * y]]
        else:
            raise NotImplementedError(fe)
    else:
        raise ValueError(dim)
    return np.stack([np.stack(gi, axis=1) for gi in g], axis=1)


# a5: FE_def.hpp:5431-5562
_DEG = {"P1": {"Std": 1, "Grad": 0}, "P2": {"Std": 2, "Grad": 1}}


def determine_degree(fe1: str, fe2: str, t1: str, t2: str, extra: int = 0) -> int:
    deg = _DEG[fe1][t1] + _DEG[fe2][t2] + extra   # :5507
    return 1 if deg == 0 else deg                 # :5508-5509


def determine_degree_single(fe: str, t: str) -> int:
    deg = _DEG[fe][t]                             # :5516-5543
    return 1 if deg == 0 else deg


def get_phi(dim, fe, deg):
    """(phi[Q,nen], w[Q])   FE_def.hpp:6730-6836"""
    pts, w = quadrature(dim, deg)
    return phi(dim, fe, pts), w


def get_dphi(dim, fe, deg):
    """(dphi[Q,nen,dim], w[Q])   FE_def.hpp:6846-6929"""
    pts, w = quadrature(dim, deg)
    return grad_phi(dim, fe, pts), w


# --------------------------------------------------------------------------------------
# meshes
# --------------------------------------------------------------------------------------
@dataclass
class Mesh:
    """Per-rank mesh in the reference's data model (Mesh_decl.hpp:132-169)."""
    dim: int
    fe: str
    conn: np.ndarray        # [E, nen] int32, LOCAL repeated ids
    xyz: np.ndarray         # [n_rep, dim] f64  (pointsRep_)
    gid_rep: np.ndarray     # [n_rep] int64     (mapRepeated_)
    flag_rep: np.ndarray    # [n_rep] int32     (bcFlagRep_)
    gid_uni: np.ndarray     # [n_uni] int64     (mapUnique_)
    flag_uni: np.ndarray    # [n_uni] int32     (bcFlagUni_)
    xyz_uni: np.ndarray     # [n_uni, dim]
    n_global: int           # global node count
    elem_flag: np.ndarray | None = None
    extra: dict = field(default_factory=dict)


# The six Kuhn tets of a cell as (dr,ds,dt) corner offsets, in the reference's exact order
# (MeshStructured_def.hpp:772-801).
KUHN_TETS = np.array([
    [[1, 0, 0], [0, 0, 0], [1, 0, 1], [1, 1, 1]],
    [[0, 0, 1], [0, 0, 0], [1, 0, 1], [1, 1, 1]],
    [[1, 0, 0], [0, 0, 0], [1, 1, 0], [1, 1, 1]],
    [[0, 0, 0], [0, 1, 0], [1, 1, 0], [1, 1, 1]],
    [[0, 0, 0], [0, 1, 0], [0, 1, 1], [1, 1, 1]],
    [[0, 0, 0], [0, 0, 1], [0, 1, 1], [1, 1, 1]],
], dtype=np.int64)
# The two triangles of a 2D cell (MeshStructured_def.hpp:431-446).
SQUARE_TRIS = np.array([
    [[1, 0], [0, 0], [1, 1]],
    [[0, 1], [0, 0], [1, 1]],
], dtype=np.int64)


def rank_offsets(rank: int, N: int, dim: int):
    """MeshStructured_def.hpp:712-722 (3D), :362-367 (2D)."""
    ox = rank % N
    oy = (rank % (N * N)) // N if (rank % (N * N)) >= N else 0
    oz = 0
    if dim == 3:
        oz = (rank % (N * N * N)) // (N * N) if (rank % (N * N * N)) >= N * N else 0
    return ox, oy, oz


def build_mesh_structured(dim: int, N: int, M: int, rank: int = 0, origin=None, size=None,
                          flags_option: int = 1, owner_of_gid=None) -> Mesh:
    """P1 structured square/cube block of rank `rank` out of N^dim.

    Restates MeshStructured::buildMesh2D P1 branch (MeshStructured_def.hpp:348-463) and
    buildMesh3D P1 branch (:703-806), followed by setStructuredMeshFlags(flags_option)
    (:2974-3203) exactly as Domain::buildMesh sequences them (Domain_def.hpp:201-265).

    Owner election: the reference uses two Tpetra imports with INSERT (Map_def.hpp:184-210),
    whose winner among sharing ranks is implementation defined.  Normative choice here (and in
    the product): the LOWEST rank that holds a node owns it; the unique list keeps the
    repeated order (Map_def.hpp:201-206).
    """
    origin = np.zeros(dim) if origin is None else np.asarray(origin, dtype=np.float64)
    size = np.ones(dim) if size is None else np.asarray(size, dtype=np.float64)
    length = size[0]
    # NB the reference uses `length` for h and H in every direction (:646-647, :311-312)
    h = length / (M * N)
    H = length / N
    P = N * (M + 1) - (N - 1)
    ox, oy, oz = rank_offsets(rank, N, dim)
    n1 = M + 1
    if dim == 3:
        t, s, r = np.meshgrid(np.arange(n1), np.arange(n1), np.arange(n1), indexing="ij")
        r = r.ravel(); s = s.ravel(); t = t.ravel()
        xyz = np.stack([r * h + ox * H, s * h + oy * H, t * h + oz * H], axis=1)
        snap = EPS                                                    # :643, :727-734
        gid = (r + s * P + t * P * P + ox * M + oy * P * M + oz * P * P * M).astype(np.int64)  # :736-737
    else:
        s, r = np.meshgrid(np.arange(n1), np.arange(n1), indexing="ij")
        r = r.ravel(); s = s.ravel()
        xyz = np.stack([r * h + ox * H, s * h + oy * H], axis=1)
        snap = 100 * EPS                                              # :371-374
        gid = (r + s * P + ox * M + oy * P * M).astype(np.int64)      # :375
    xyz = np.where((xyz < snap) & (xyz > -snap), 0.0, xyz)
    # NB: the reference never adds coorRec to the coordinates (only uses it in the flag tests)
    hi = origin + size
    flag = np.zeros(xyz.shape[0], dtype=np.int32)
    on_bnd = np.zeros(xyz.shape[0], dtype=bool)
    for d in range(dim):                                              # :739-744 / :376-380
        on_bnd |= (xyz[:, d] > hi[d] - snap) | (xyz[:, d] < origin[d] + snap)
    flag[on_bnd] = 1

    # elements (:757-803 / :417-455)
    if dim == 3:
        t, s, r = np.meshgrid(np.arange(M), np.arange(M), np.arange(M), indexing="ij")
        cells = np.stack([r.ravel(), s.ravel(), t.ravel()], axis=1)           # [C,3]
        corner = cells[:, None, None, :] + KUHN_TETS[None, :, :, :]            # [C,6,4,3]
        conn = corner[..., 0] + n1 * corner[..., 1] + n1 * n1 * corner[..., 2]
        conn = conn.reshape(-1, 4).astype(np.int32)
        elem_flag = np.zeros(conn.shape[0], dtype=np.int32)
    else:
        s, r = np.meshgrid(np.arange(M), np.arange(M), indexing="ij")
        cells = np.stack([r.ravel(), s.ravel()], axis=1)
        corner = cells[:, None, None, :] + SQUARE_TRIS[None, :, :, :]
        conn = (corner[..., 0] + n1 * corner[..., 1]).reshape(-1, 3).astype(np.int32)
        cx = xyz[conn, 0].sum(axis=1) / 3.0                                   # :434-441
        cy = xyz[conn, 1].sum(axis=1) / 3.0
        elem_flag = ((cx >= 0.3) & (cx <= 0.7) & (cy >= 0.6)).astype(np.int32)

    # unique map: lowest-rank owner, repeated order
    if owner_of_gid is None:
        owner = structured_owner(dim, N, M, gid)
    else:
        owner = owner_of_gid(gid)
    mine = owner == rank
    gid_uni = gid[mine]
    xyz_uni = xyz[mine].copy()
    flag_uni = flag[mine].copy()

    # setStructuredMeshFlags(flags_option)  (:2974-3203)
    if flags_option == 1:
        flag_uni = _structured_flags(dim, xyz_uni, flag_uni, origin, size)
        # the reference re-flags only the first |unique| repeated nodes in 3D (loop bound bug,
        # :3170); reproduced for fidelity although no hot-path consumer reads bcFlagRep_.
        nrep_flagged = xyz.shape[0] if dim == 2 else gid_uni.shape[0]
        flag[:nrep_flagged] = _structured_flags(dim, xyz[:nrep_flagged], flag[:nrep_flagged], origin, size)
    elif flags_option != 0:
        raise NotImplementedError("flags option %d" % flags_option)
    return Mesh(dim=dim, fe="P1", conn=conn, xyz=xyz, gid_rep=gid, flag_rep=flag, gid_uni=gid_uni,
                flag_uni=flag_uni, xyz_uni=xyz_uni, n_global=P ** dim, elem_flag=elem_flag,
                extra={"N": N, "M": M, "rank": rank, "P": P})


def structured_owner(dim: int, N: int, M: int, gid: np.ndarray) -> np.ndarray:
    """Lowest rank whose block contains global node `gid` (normative owner rule)."""
    P = N * (M + 1) - (N - 1)
    g = gid.astype(np.int64)
    gx = g % P
    gy = (g // P) % P
    gz = g // (P * P)
    # a lattice coordinate c in [0, N*M] lies in blocks floor((c-1)/M) (as its upper face) and
    # floor(c/M); the lowest block index is max(0, ceil(c/M) - 1).
    def lo(c):
        return np.maximum(0, (c + M - 1) // M - 1)
    owner = lo(gx) + N * lo(gy)
    if dim == 3:
        owner = owner + N * N * lo(gz)
    return owner


def _structured_flags(dim, pts, flag_in, origin, size):
    """setStructuredMeshFlags option 1 (MeshStructured_def.hpp:2984-3011 in 2D, :3136-3203 in 3D)."""
    tol = 1.0e-12                                              # :2976
    f = flag_in.copy()
    x = pts[:, 0]
    y = pts[:, 1]
    x0, y0 = origin[0], origin[1]
    if dim == 2:
        length, height = size[0], size[1]
        f[(x > x0 - tol) & (y < y0 + tol)] = 1                                   # :2986
        f[(x > x0 - tol) & (y > y0 + height - tol)] = 1                          # :2989
        f[(x > x0 + length - tol) & (y > y0 + tol) & (y < y0 + height - tol)] = 3  # :2992
        f[x < x0 + tol] = 2                                                      # :2995
        return f
    z = pts[:, 2]
    z0 = origin[2]
    length, width, height = size[0], size[1], size[2]
    f[x < x0 + tol] = 2                                                          # :3138
    inner = x > x0 + tol
    f[inner & (z < z0 + tol)] = 1                                                # bottom :3142
    f[inner & (z > z0 + height - tol)] = 1                                       # top    :3147
    f[inner & (y < y0 + tol)] = 1                                                # front  :3152
    f[inner & (y > y0 + width - tol)] = 1                                        # back   :3157
    f[(x > x0 + length - tol) & (y > y0 + tol) & (y < y0 + width - tol)
      & (z > z0 + tol) & (z < z0 + height - tol)] = 3                            # out    :3162
    return f


def build_mesh_structured_global(dim: int, N: int, M: int, **kw) -> Mesh:
    """The same global grid seen by ONE rank (N=1, M'=N*M): used as the ownership-independent
    comparison target ('the assembled global matrix is ownership independent')."""
    return build_mesh_structured(dim, 1, N * M, 0, **kw)


def read_mesh_file(path: str, dim: int, volume_id: int = 10) -> Mesh:
    """INRIA/medit .mesh reader (feddlib/core/Mesh/MeshFileReader.cpp:16-106,
    MeshFileReader.hpp:35-127; 1-based ids converted at MeshUnstructured_def.hpp:1181-1190).
    One rank: repeated = unique = identity ordering (MeshPartitioner_def.hpp:321-397).
    Only elements whose flag equals `volume_id` are kept when the file mixes flags
    (Domain_decl.hpp:122,187)."""
    with open(path) as fh:
        toks = fh.read().split()
    i = 0
    verts = None
    vflag = None
    elems = None
    eflag = None
    surf = None
    sflag = None
    key_elem = "Triangles" if dim == 2 else "Tetrahedra"
    key_surf = "Edges" if dim == 2 else "Triangles"
    nen = dim + 1
    while i < len(toks):
        t = toks[i]
        if t == "Vertices":
            n = int(toks[i + 1])
            arr = np.array(toks[i + 2:i + 2 + 4 * n], dtype=np.float64).reshape(n, 4)
            verts = arr[:, :dim].copy()           # vertex lines always carry 3 coords (:54)
            vflag = arr[:, 3].astype(np.int32)
            i += 2 + 4 * n
        elif t == key_elem:
            n = int(toks[i + 1])
            arr = np.array(toks[i + 2:i + 2 + (nen + 1) * n], dtype=np.int64).reshape(n, nen + 1)
            elems = (arr[:, :nen] - 1).astype(np.int32)
            eflag = arr[:, nen].astype(np.int32)
            i += 2 + (nen + 1) * n
        elif t == key_surf:
            n = int(toks[i + 1])
            arr = np.array(toks[i + 2:i + 2 + (dim + 1) * n], dtype=np.int64).reshape(n, dim + 1)
            surf = (arr[:, :dim] - 1).astype(np.int32)
            sflag = arr[:, dim].astype(np.int32)
            i += 2 + (dim + 1) * n
        elif t == "Edges" and dim == 3:
            n = int(toks[i + 1])
            i += 2 + 3 * n
        else:
            i += 1
    keep = eflag == volume_id
    if keep.any() and not keep.all():
        elems = elems[keep]
        eflag = eflag[keep]
    n = verts.shape[0]
    gid = np.arange(n, dtype=np.int64)
    return Mesh(dim=dim, fe="P1", conn=elems, xyz=verts, gid_rep=gid, flag_rep=vflag.copy(),
                gid_uni=gid.copy(), flag_uni=vflag.copy(), xyz_uni=verts.copy(), n_global=n,
                elem_flag=eflag, extra={"surf": surf, "surf_flag": sflag, "volume_id": volume_id})


# local edge order used to create P2 mid-edge nodes
# (MeshPartitioner_def.hpp:708-734) and the slot each edge takes in the 10-node tet
# (MeshUnstructured_def.hpp:755-772: (0,1)->4 (1,2)->5 (0,2)->6 (0,3)->7 (1,3)->8 (2,3)->9).
P2_EDGE_SLOTS_3D = {(0, 1): 4, (1, 2): 5, (0, 2): 6, (0, 3): 7, (1, 3): 8, (2, 3): 9}
P2_EDGE_SLOTS_2D = {(0, 1): 3, (1, 2): 4, (0, 2): 5}


def build_p2_of_p1(m: Mesh) -> Mesh:
    """P2 mesh from a P1 mesh, one rank (MeshUnstructured_def.hpp:129-410, EdgeElements.cpp:105-155):
    global edge id = rank in the lexicographically sorted unique (min,max) list; mid node
    gid = P1Offset + edge id, P1Offset = max P1 gid + 1 (:141,:372-374); coordinate = midpoint
    (:173-174).  Mid-node flag: smallest flag among the boundary entities containing the edge if
    any, else the volume default 10 -- simplified: min over the two end-node flags when both are
    boundary (<10) else 10.  (Flag derivation detail :177,:317-339 is only needed for BCs.)"""
    slots = P2_EDGE_SLOTS_3D if m.dim == 3 else P2_EDGE_SLOTS_2D
    pairs = list(slots.keys())
    g = m.gid_rep[m.conn]                                    # [E, nen] global vertex ids
    e_lo = np.stack([np.minimum(g[:, a], g[:, b]) for a, b in pairs], axis=1)
    e_hi = np.stack([np.maximum(g[:, a], g[:, b]) for a, b in pairs], axis=1)
    key = e_lo.astype(np.int64) * (m.n_global + 1) + e_hi
    uniq, inv = np.unique(key.ravel(), return_inverse=True)
    inv = inv.reshape(key.shape)
    n_p1 = m.xyz.shape[0]
    off = int(m.gid_rep.max()) + 1
    lo = (uniq // (m.n_global + 1)).astype(np.int64)
    hi = (uniq % (m.n_global + 1)).astype(np.int64)
    # one rank: local id == gid for P1 nodes
    mid = 0.5 * (m.xyz[lo] + m.xyz[hi])
    xyz = np.vstack([m.xyz, mid])
    nen2 = nodes_per_element(m.dim, "P2")
    conn = np.zeros((m.conn.shape[0], nen2), dtype=np.int32)
    conn[:, :m.dim + 1] = m.conn
    for k, pr in enumerate(pairs):
        conn[:, slots[pr]] = n_p1 + inv[:, k]
    # determineFlagP2 (MeshUnstructured_def.hpp:806-900): interior if an end node is interior, else
    # the lowest flag of the boundary entities (Edges in 2D, Triangles in 3D) that contain both end
    # nodes, interior (volume id) if there is none
    interior = int(m.extra.get("volume_id", 10))
    fl, fh = m.flag_rep[lo], m.flag_rep[hi]
    mflag = np.full(lo.shape[0], interior, dtype=np.int32)
    surf, sflag = m.extra.get("surf"), m.extra.get("surf_flag")
    if surf is not None and len(surf):
        pair_flag = {}
        d = surf.shape[1]
        for a in range(d):
            for b in range(a + 1, d):
                u = np.minimum(surf[:, a], surf[:, b]); v = np.maximum(surf[:, a], surf[:, b])
                for uu, vv, ff in zip(u.tolist(), v.tolist(), sflag.tolist()):
                    k = (uu, vv)
                    pair_flag[k] = min(pair_flag.get(k, ff), ff)
        both = (fl != interior) & (fh != interior)
        for i in np.nonzero(both)[0]:
            mflag[i] = pair_flag.get((int(lo[i]), int(hi[i])), interior)
    flag = np.concatenate([m.flag_rep, mflag])
    gid = np.concatenate([m.gid_rep, off + np.arange(uniq.shape[0], dtype=np.int64)])
    return Mesh(dim=m.dim, fe="P2", conn=conn, xyz=xyz, gid_rep=gid, flag_rep=flag, gid_uni=gid.copy(),
                flag_uni=flag.copy(), xyz_uni=xyz.copy(), n_global=int(gid.max()) + 1,
                elem_flag=m.elem_flag, extra=dict(m.extra))


# --------------------------------------------------------------------------------------
# a6-a8: affine map  (FE_def.hpp:5342-5357, SmallMatrix.hpp:306-357, FE_def.hpp:83-96)
# --------------------------------------------------------------------------------------
def build_transformation(m: Mesh) -> np.ndarray:
    """B[e, i, j] = x_{j+1}[i] - x_0[i]  (vertices only, also for P2)."""
    X = m.xyz[m.conn[:, :m.dim + 1]]                  # [E, dim+1, dim]
    return np.transpose(X[:, 1:, :] - X[:, :1, :], (0, 2, 1))


def det_small(B: np.ndarray) -> np.ndarray:
    """SmallMatrix::computeDet, Sarrus, same term order (SmallMatrix.hpp:338-357)."""
    if B.shape[-1] == 2:
        return B[:, 0, 0] * B[:, 1, 1] - B[:, 1, 0] * B[:, 0, 1]
    v = B
    return (v[:, 0, 0] * v[:, 1, 1] * v[:, 2, 2] + v[:, 0, 1] * v[:, 1, 2] * v[:, 2, 0]
            + v[:, 0, 2] * v[:, 1, 0] * v[:, 2, 1] - v[:, 2, 0] * v[:, 1, 1] * v[:, 0, 2]
            - v[:, 2, 1] * v[:, 1, 2] * v[:, 0, 0] - v[:, 2, 2] * v[:, 1, 0] * v[:, 0, 1])


def inv_small(B: np.ndarray):
    """SmallMatrix::computeInverse: adjugate / det, returns (Binv, det)  (SmallMatrix.hpp:306-335)."""
    det = det_small(B)
    v = B
    inv = np.empty_like(B)
    if B.shape[-1] == 2:
        inv[:, 0, 0] = v[:, 1, 1] / det
        inv[:, 0, 1] = (-v[:, 0, 1]) / det
        inv[:, 1, 0] = (-v[:, 1, 0]) / det
        inv[:, 1, 1] = v[:, 0, 0] / det
        return inv, det
    inv[:, 0, 0] = (v[:, 1, 1] * v[:, 2, 2] - v[:, 1, 2] * v[:, 2, 1]) / det
    inv[:, 0, 1] = (v[:, 0, 2] * v[:, 2, 1] - v[:, 0, 1] * v[:, 2, 2]) / det
    inv[:, 0, 2] = (v[:, 0, 1] * v[:, 1, 2] - v[:, 0, 2] * v[:, 1, 1]) / det
    inv[:, 1, 0] = (v[:, 1, 2] * v[:, 2, 0] - v[:, 1, 0] * v[:, 2, 2]) / det
    inv[:, 1, 1] = (v[:, 0, 0] * v[:, 2, 2] - v[:, 0, 2] * v[:, 2, 0]) / det
    inv[:, 1, 2] = (v[:, 0, 2] * v[:, 1, 0] - v[:, 0, 0] * v[:, 1, 2]) / det
    inv[:, 2, 0] = (v[:, 1, 0] * v[:, 2, 1] - v[:, 1, 1] * v[:, 2, 0]) / det
    inv[:, 2, 1] = (v[:, 0, 1] * v[:, 2, 0] - v[:, 0, 0] * v[:, 2, 1]) / det
    inv[:, 2, 2] = (v[:, 0, 0] * v[:, 1, 1] - v[:, 0, 1] * v[:, 1, 0]) / det
    return inv, det


def dphi_trans(m: Mesh, fe: str, deg: int):
    """(G[e,q,i,d1] = sum_d2 dPhi[q,i,d2] Binv[e,d2,d1], w[q], absdet[e])  -- applyBTinv :83-96."""
    dphi, w = get_dphi(m.dim, fe, deg)
    Binv, det = inv_small(build_transformation(m))
    G = np.einsum("qik,ekd->eqid", dphi, Binv)
    return G, w, np.abs(det)


# --------------------------------------------------------------------------------------
# a14: CSR fill with Tpetra semantics (Matrix_def.hpp:88-92,192-199)
# --------------------------------------------------------------------------------------
def fill_complete(rows, cols, vals, n_rows, n_cols=None) -> sp.csr_matrix:
    """insertGlobalValues + fillComplete: duplicate (row,col) summed, zeros kept structurally,
    columns sorted.  Global ids."""
    n_cols = n_rows if n_cols is None else n_cols
    A = sp.coo_matrix((np.asarray(vals, dtype=np.float64).ravel(),
                       (np.asarray(rows).ravel(), np.asarray(cols).ravel())),
                      shape=(n_rows, n_cols)).tocsr()     # sums duplicates, keeps explicit zeros
    A.sort_indices()
    return A


def _local_to_triplets(m: Mesh, K: np.ndarray, dofs: int = 1, diag_only: bool = False):
    """K[e,i,j] scalar local matrices -> global (row, col, val) triplets, node-wise dof ids
    dim*gid+d (Map_def.hpp:101-104)."""
    g = m.gid_rep[m.conn]
    nen = g.shape[1]
    R = np.broadcast_to(g[:, :, None], (g.shape[0], nen, nen))
    C = np.broadcast_to(g[:, None, :], (g.shape[0], nen, nen))
    if dofs == 1:
        return R.ravel(), C.ravel(), K.ravel()
    rows, cols, vals = [], [], []
    for d in range(dofs):
        rows.append((dofs * R + d).ravel())
        cols.append((dofs * C + d).ravel())
        vals.append(K.ravel())
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


# --------------------------------------------------------------------------------------
# a9-a12, a16, a17: element-wise assembly
# --------------------------------------------------------------------------------------
def local_laplace(m: Mesh, fe: str | None = None) -> np.ndarray:
    """K[e,i,j] = |detB| sum_q w_q sum_d G_qid G_qjd   (FE_def.hpp:637-656)."""
    fe = fe or m.fe
    deg = determine_degree(fe, fe, "Grad", "Grad")                    # :626
    G, w, absdet = dphi_trans(m, fe, deg)
    K = np.einsum("q,eqid,eqjd->eij", w, G, G)
    return K * absdet[:, None, None]


def assembly_laplace(m: Mesh, fe: str | None = None) -> sp.csr_matrix:
    """FE::assemblyLaplace (FE_def.hpp:604-667)."""
    K = local_laplace(m, fe)
    r, c, v = _local_to_triplets(m, K)
    return fill_complete(r, c, v, m.n_global)


def assembly_laplace_vecfield(m: Mesh, fe: str | None = None, set_zeros_eps: float = 0.0) -> sp.csr_matrix:
    """FE::assemblyLaplaceVecField (FE_def.hpp:670-734): scalar K on the dim diagonal blocks only.
    set_zeros_eps > 0: FE::doSetZeros(eps) (:74-79) -- an element value with |value| < eps is set to zero before it is
    inserted (:719-721)."""
    K = local_laplace(m, fe)
    if set_zeros_eps > 0.0:
        K = np.where(np.abs(K) < set_zeros_eps, 0.0, K)
    r, c, v = _local_to_triplets(m, K, dofs=m.dim)
    return fill_complete(r, c, v, m.dim * m.n_global)


def local_mass(m: Mesh, fe: str | None = None) -> np.ndarray:
    """M[e,i,j] = |detB| sum_q w_q phi_qi phi_qj  (FE_def.hpp:485-499)."""
    fe = fe or m.fe
    deg = determine_degree(fe, fe, "Std", "Std")                      # :472
    ph, w = get_phi(m.dim, fe, deg)
    absdet = np.abs(det_small(build_transformation(m)))
    Mloc = np.einsum("q,qi,qj->ij", w, ph, ph)
    return absdet[:, None, None] * Mloc[None]


def assembly_mass(m: Mesh, field_type: str = "Scalar", fe: str | None = None) -> sp.csr_matrix:
    """FE::assemblyMass (FE_def.hpp:454-524)."""
    K = local_mass(m, fe)
    if field_type == "Scalar":
        r, c, v = _local_to_triplets(m, K)
        return fill_complete(r, c, v, m.n_global)
    r, c, v = _local_to_triplets(m, K, dofs=m.dim)
    return fill_complete(r, c, v, m.dim * m.n_global)


def assembly_bd_stabilization(m: Mesh) -> sp.csr_matrix:
    """FE::assemblyBDStabilization (FE_def.hpp:2151-2220; P1 only, :2156): the mass matrix entry of every element pair minus
    |det B| * refElementSize * refElementScale, with (1/2, 1/9) in 2D and (1/6, 1/16) in 3D (:2183-2192); the operation
    order of :2204-2206 is kept (value *= absDetB; value -= refElementSize * absDetB * refElementScale)."""
    assert m.fe == "P1", "Only implemented for P1"
    dim = m.dim
    deg = determine_degree("P1", "P1", "Std", "Std")
    ph, w = get_phi(dim, "P1", deg)
    absdet = np.abs(det_small(build_transformation(m)))
    ref_size, ref_scale = (0.5, 1.0 / 9.0) if dim == 2 else (1.0 / 6.0, 1.0 / 16.0)
    nen = dim + 1
    base = np.zeros((nen, nen))
    for i in range(nen):
        for j in range(nen):
            v = 0.0
            for q in range(w.shape[0]):
                v += w[q] * ph[q][i] * ph[q][j]
            base[i, j] = v
    K = base[None, :, :] * absdet[:, None, None]
    K = K - (ref_size * absdet * ref_scale)[:, None, None]
    return fill_complete(*_local_to_triplets(m, K), m.n_global)


def assembly_rhs(m: Mesh, f_const, field_type: str = "Scalar", deg_func: int = 0,
                 fe: str | None = None) -> np.ndarray:
    """FE::assemblyRHS on the REPEATED vector (FE_def.hpp:4694-4766): f evaluated once (constant
    only, :4731-4736); deg = determineDegree(dim,FE,Std) + floor(lastParam + 1e-14) (:4717-4718)."""
    fe = fe or m.fe
    deg = determine_degree_single(fe, "Std") + int(deg_func + 1.0e-14)
    ph, w = get_phi(m.dim, fe, deg)
    absdet = np.abs(det_small(build_transformation(m)))
    base = ph.T @ w                                   # sum_q w_q phi_qi  (:4746-4747)
    f_const = np.atleast_1d(np.asarray(f_const, dtype=np.float64))
    n_rep = m.xyz.shape[0]
    if field_type == "Scalar":
        out = np.zeros(n_rep)
        contrib = (base[None, :] * (absdet * f_const[0])[:, None])
        np.add.at(out, m.conn.ravel(), contrib.ravel())
        return out
    dofs = m.dim
    out = np.zeros(n_rep * dofs)
    val = base[None, :] * absdet[:, None]
    for d in range(dofs):
        np.add.at(out, (dofs * m.conn + d).ravel(), (val * f_const[d]).ravel())
    return out


def export_add(values_rep: np.ndarray, gid_rep_dofs: np.ndarray, n_global_dofs: int) -> np.ndarray:
    """MultiVector::exportFromVector(..., "Add") seen globally: repeated -> global sum
    (MultiVector_def.hpp:297-330)."""
    out = np.zeros(n_global_dofs)
    np.add.at(out, gid_rep_dofs, values_rep)
    return out


def assembly_linelas(m: Mesh, lam: float, mu: float, fe: str | None = None) -> sp.csr_matrix:
    """FE::assemblyLinElasXDim (FE_def.hpp:2739-3040; epsilonTensor :4931-4944):
    v_ab = |detB| sum_q w_q ( 2 mu eps(phi_i e_a):eps(phi_j e_b) + lam tr eps_i tr eps_j )
         = |detB| sum_q w_q ( mu (delta_ab g_i.g_j + g_i[b] g_j[a]) + lam g_i[a] g_j[b] );
    row dim*gid_i + a, col dim*gid_j + b, every (a,b) inserted (also structural zeros)."""
    fe = fe or m.fe
    dim = m.dim
    deg = determine_degree(fe, fe, "Grad", "Grad")                    # :2757
    G, w, absdet = dphi_trans(m, fe, deg)
    gg = np.einsum("q,eqid,eqjd->eij", w, G, G)
    t1 = np.einsum("q,eqib,eqja->eiajb", w, G, G)                     # g_i[b] g_j[a]
    t2 = np.einsum("q,eqia,eqjb->eiajb", w, G, G)                     # g_i[a] g_j[b]
    V = mu * t1 + lam * t2
    for a in range(dim):
        V[:, :, a, :, a] += mu * gg
    V = V * absdet[:, None, None, None, None]
    g = m.gid_rep[m.conn]
    E, nen = g.shape
    R = dim * g[:, :, None, None, None] + np.arange(dim)[None, None, :, None, None]
    C = dim * g[:, None, None, :, None] + np.arange(dim)[None, None, None, None, :]
    R = np.broadcast_to(R, V.shape)
    C = np.broadcast_to(C, V.shape)
    return fill_complete(R.ravel(), C.ravel(), V.ravel(), dim * m.n_global)


def assembly_div_and_divt(mv: Mesh, mp: Mesh, set_zeros_eps: float = 0.0):
    """FE::assemblyDivAndDivT (FE_def.hpp:1932-2057): velocity mesh `mv` (FEType1), pressure
    mesh `mp` (FEType2) over the same elements.  B[row p_i, col dim*v_j+d] =
    |detB| sum_q w_q psi_qi dphi_qjd ; BT its transpose pattern/value.  (Unscaled; the -1 is
    applied by Stokes::assemble, Stokes_def.hpp:79-89.)"""
    dim = mv.dim
    deg = determine_degree(mv.fe, mp.fe, "Grad", "Std")               # :1962
    G, w, absdet = dphi_trans(mv, mv.fe, deg)
    psi, _ = get_phi(dim, mp.fe, deg)
    V = np.einsum("q,qi,eqjd->eijd", w, psi, G) * absdet[:, None, None, None]
    if set_zeros_eps > 0.0:        # doSetZeros: :2002-2004 (B) and :2032-2034 (B^T), the same element values
        V = np.where(np.abs(V) < set_zeros_eps, 0.0, V)
    gp = mp.gid_rep[mp.conn]
    gv = mv.gid_rep[mv.conn]
    R = np.broadcast_to(gp[:, :, None, None], V.shape)
    C = dim * gv[:, None, :, None] + np.arange(dim)[None, None, None, :]
    C = np.broadcast_to(C, V.shape)
    B = fill_complete(R.ravel(), C.ravel(), V.ravel(), mp.n_global, dim * mv.n_global)
    BT = fill_complete(C.ravel(), R.ravel(), V.ravel(), dim * mv.n_global, mp.n_global)
    return B, BT


def stokes_blocks(mv: Mesh, mp: Mesh, nu: float = 1.0):
    """Stokes::assemble (feddlib/problems/specific/Stokes_def.hpp:47-138): A = nu * vector Laplacian,
    B and B^T from assemblyDivAndDivT scaled by -1 (:79-89).  Returns (A, BT, B)."""
    A = assembly_laplace_vecfield(mv) * nu
    B, BT = assembly_div_and_divt(mv, mp)
    return A, BT * (-1.0), B * (-1.0)


def block_merge(A, BT, B, C=None) -> sp.csr_matrix:
    """BlockMatrix::merge + BlockMap::merge (BlockMatrix_def.hpp:119-148,212-287; BlockMap_def.hpp:55-80):
    monolithic [A B^T; B C], ids of block 1 shifted by (maxAllGlobalIndex + 1) of block 0; every stored
    entry (also structural zeros) is re-inserted."""
    def coo(Mx, ro, co):
        Mx = Mx.tocoo()
        return Mx.row + ro, Mx.col + co, Mx.data
    n0, n1 = A.shape[0], B.shape[0]
    parts = [coo(A, 0, 0), coo(BT, 0, n0), coo(B, n0, 0)]
    if C is not None:
        parts.append(coo(C, n0, n0))
    r = np.concatenate([p[0] for p in parts]); c = np.concatenate([p[1] for p in parts])
    v = np.concatenate([p[2] for p in parts])
    return fill_complete(r, c, v, n0 + n1)


# literal loop nests for small cases ----------------------------------------------------
def assembly_laplace_loops(m: Mesh) -> sp.csr_matrix:
    """Line-by-line restatement of FE::assemblyLaplace's loop nest (FE_def.hpp:637-662) for small
    meshes; checked against the vectorised version."""
    fe = m.fe
    dim = m.dim
    deg = determine_degree(fe, fe, "Grad", "Grad")
    dphi, w = get_dphi(dim, fe, deg)
    rows, cols, vals = [], [], []
    for T in range(m.conn.shape[0]):
        el = m.conn[T]
        B = np.zeros((dim, dim))
        for j in range(dim):                                          # :5349-5355
            for i in range(dim):
                B[i, j] = m.xyz[el[j + 1], i] - m.xyz[el[0], i]
        Binv, det = inv_small(B[None])
        Binv = Binv[0]
        absdet = abs(det[0])
        Q, nen = dphi.shape[0], dphi.shape[1]
        tr = np.zeros((Q, nen, dim))
        for q in range(Q):                                            # :83-96
            for i in range(nen):
                for d1 in range(dim):
                    for d2 in range(dim):
                        tr[q, i, d1] += dphi[q, i, d2] * Binv[d2, d1]
        for i in range(nen):
            for j in range(nen):
                v = 0.0
                for q in range(Q):
                    for d in range(dim):
                        v += w[q] * tr[q, i, d] * tr[q, j, d]          # :651
                v *= absdet                                            # :654
                rows.append(m.gid_rep[el[i]]); cols.append(m.gid_rep[el[j]]); vals.append(v)
    return fill_complete(rows, cols, vals, m.n_global)


# --------------------------------------------------------------------------------------
# a15: Dirichlet rows  (BCBuilder_def.hpp:589-707 setSystem, :93-170 setRHS)
# --------------------------------------------------------------------------------------
def dirichlet_rows(flags_global: np.ndarray, bc_flags, dofs: int = 1, comp_mask=None) -> np.ndarray:
    """Boolean mask over global dofs that get a Dirichlet row.  comp_mask[d] selects components
    (Dirichlet / Dirichlet_X / ..., BCBuilder_def.hpp:659-667)."""
    node = np.isin(flags_global, np.asarray(list(bc_flags)))
    if dofs == 1:
        return node
    comp_mask = np.ones(dofs, dtype=bool) if comp_mask is None else np.asarray(comp_mask, dtype=bool)
    return (node[:, None] & comp_mask[None, :]).ravel()


def set_dirichlet(A: sp.csr_matrix, rhs: np.ndarray, is_dir: np.ndarray, values) -> tuple:
    """setLocalRowOne on a diagonal block: row <- e_i^T with the pattern kept (zeros stay
    structural, BCBuilder_def.hpp:670-679); columns NOT eliminated; rhs <- g (:136-143)."""
    A = A.copy().tocsr()
    A.sort_indices()
    rhs = rhs.copy()
    rows = np.nonzero(is_dir)[0]
    indptr, indices, data = A.indptr, A.indices, A.data
    row_of = np.repeat(np.arange(A.shape[0]), np.diff(indptr))
    in_dir = is_dir[row_of]
    data[in_dir] = np.where(indices[in_dir] == row_of[in_dir], 1.0, 0.0)
    vals = np.broadcast_to(np.asarray(values, dtype=np.float64), rows.shape) if np.ndim(values) == 0 \
        else np.asarray(values, dtype=np.float64)[rows] if np.shape(values) == rhs.shape else np.asarray(values)
    rhs[rows] = vals
    return A, rhs


def set_dirichlet_offdiag(A: sp.csr_matrix, is_dir_rows: np.ndarray) -> sp.csr_matrix:
    """setLocalRowZero on an off-diagonal block (BCBuilder_def.hpp:687-707)."""
    A = A.copy().tocsr()
    row_of = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    A.data[is_dir_rows[row_of]] = 0.0
    return A


# --------------------------------------------------------------------------------------
# whole problems (drivers)
# --------------------------------------------------------------------------------------
def laplace_problem(meshes, bc_flags=(1, 2, 3), f: float = 1.0, bc_value: float = 0.0):
    """The `laplace` driver sequence (feddlib/problems/tests/laplace/main.cpp:194-208;
    Laplace_def.hpp:36-60; Problem_def.hpp:170-216,298-304) over a list of per-rank meshes.
    Returns (A_bc, rhs_bc, A_raw, rhs_raw, flags_global) in GLOBAL ids."""
    if isinstance(meshes, Mesh):
        meshes = [meshes]
    n = meshes[0].n_global
    rows, cols, vals = [], [], []
    rhs = np.zeros(n)
    flags = np.zeros(n, dtype=np.int32)
    for m in meshes:
        K = local_laplace(m)
        r, c, v = _local_to_triplets(m, K)
        rows.append(r); cols.append(c); vals.append(v)
        fr = assembly_rhs(m, [f], "Scalar", 0)
        rhs += export_add(fr, m.gid_rep, n)
        flags[m.gid_uni] = m.flag_uni
    A = fill_complete(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), n)
    is_dir = dirichlet_rows(flags, bc_flags)
    A_bc, rhs_bc = set_dirichlet(A, rhs, is_dir, bc_value)
    return A_bc, rhs_bc, A, rhs, flags


def linelas_problem(m: Mesh, mu: float, nu: float, f=(0.0, 1.0, 0.0), bc_flags=(2,)):
    """`steadyLinElas_Perf` driver sequence (LinElas_def.hpp:64-99; lambda, E at :76-77)."""
    dim = m.dim
    E = mu * 2.0 * (1.0 + nu)
    lam = nu * E / ((1.0 + nu) * (1.0 - 2.0 * nu))
    A = assembly_linelas(m, lam, mu)
    fr = assembly_rhs(m, f[:dim], "Vector", 0)
    gd = (dim * m.gid_rep[:, None] + np.arange(dim)[None, :]).ravel()
    rhs = export_add(fr, gd, dim * m.n_global)
    flags = np.zeros(m.n_global, dtype=np.int32)
    flags[m.gid_uni] = m.flag_uni
    is_dir = dirichlet_rows(flags, bc_flags, dofs=dim)
    A_bc, rhs_bc = set_dirichlet(A, rhs, is_dir, 0.0)
    return A_bc, rhs_bc, A, rhs, flags


# --------------------------------------------------------------------------------------
# a21: solver  (published algorithms; normative definitions shared with the product)
# --------------------------------------------------------------------------------------
def schwarz_bins(xyz: np.ndarray, target: int, scale: float = 1.0):
    """Normative node -> subdomain ('bin') map of the product's batched one-level Schwarz:
    a regular grid of boxes over the bounding box of the owned nodes with
    g_d = max(1, ceil(L_d / s - 1e-9)) made odd (+1 when even: the lattice has a centre box),
    s = scale * (V * target / n)^(1/dim) (V = product of the non-degenerate extents),
    bin = ix + gx*(iy + gy*iz), w_d = L_d / g_d, t = (x_d - lo_d) / w_d, i_d = floor(t) -- except for a
    node on a box boundary (|t - k| <= 1e-9, k = floor(t + 0.5)), which goes to the box on the centre
    side (k if 2k <= g_d - 1, else k - 1) -- clamped to [0, g_d - 1].  A point set that is symmetric about
    the centre of its bounding box thus gets a symmetric partition.  Empty bins are dropped, the rest
    renumbered in increasing bin id."""
    n, dim = xyz.shape
    lo = xyz.min(axis=0)
    L = xyz.max(axis=0) - lo
    Lpos = np.where(L > 0, L, 1.0)
    V = float(np.prod(Lpos))
    s = scale * (V * target / n) ** (1.0 / dim)
    g = np.maximum(1, np.ceil(Lpos / s - 1e-9).astype(np.int64))
    g = np.where(L > 0, g, 1)
    g = np.where(g % 2 == 0, g + 1, g)
    w = Lpos / g
    t = (xyz - lo) / w
    kb = np.floor(t + 0.5)
    idx = np.floor(t).astype(np.int64)
    on = np.abs(t - kb) <= 1e-9
    kbi = kb.astype(np.int64)
    idx = np.where(on, np.where(2 * kbi <= g - 1, kbi, kbi - 1), idx)
    idx = np.minimum(g - 1, idx)
    idx = np.maximum(idx, 0)
    b = idx[:, 0].copy()
    mul = g[0]
    for d in range(1, dim):
        b += mul * idx[:, d]
        mul *= g[d]
    uniq, inv = np.unique(b, return_inverse=True)
    return inv.astype(np.int64), uniq.shape[0], g


def rcb_bins(xyz_dof: np.ndarray, target: int):
    """Normative boxes of the product's large-subdomain Schwarz path (feddlib_amd/csrc/schwarz_big.hip): balanced
    recursive coordinate bisection of the points that carry the dofs -- a part with more than `target` points is
    split at its median (the first floor(count / 2) points in the order (coordinate, index) go left) along the
    longest edge of its bounding box (lowest axis on ties); parts are numbered left to right.  Returns
    (bin of every dof, number of bins)."""
    n, dim = xyz_dof.shape
    bins = np.zeros(n, dtype=np.int64)
    out = []

    def rec(idx):
        if idx.shape[0] <= target:
            out.append(idx)
            return
        ext = xyz_dof[idx].max(axis=0) - xyz_dof[idx].min(axis=0)
        ax = 0
        for d in range(1, dim):
            if ext[d] > ext[ax]:
                ax = d
        order = idx[np.lexsort((idx, xyz_dof[idx, ax]))]
        h = order.shape[0] // 2
        rec(order[:h])
        rec(order[h:])

    rec(np.arange(n))
    for k, idx in enumerate(out):
        bins[idx] = k
    return bins, len(out)


class RAS:
    """One-level overlapping additive Schwarz, many subdomains per rank (normative definition,
    DESIGN.md 'Schwarz'): subdomain i = bin_i plus `overlap` graph layers of the (Dirichlet-
    modified) matrix; A_i = principal submatrix; M^-1 r = sum_i P_i A_i^-1 R_i r with P_i =
    restricted (only bin-owned rows), averaging (divide by multiplicity) or full."""

    def __init__(self, A: sp.csr_matrix, node_bin: np.ndarray, nbins: int, dofs: int = 1,
                 overlap: int = 1, combine: str = "restricted"):
        A = A.tocsr()
        n = A.shape[0]
        self.n = n
        self.combine = combine
        dof_bin = np.repeat(node_bin, dofs)
        G = A.copy()
        G.data = np.ones_like(G.data)
        P0 = sp.csr_matrix((np.ones(n), (np.arange(n), dof_bin)), shape=(n, nbins))
        Pk = P0
        for _ in range(overlap):
            Pk = ((G @ Pk) + Pk)
            Pk.data[:] = 1.0
        Pk = Pk.tocsc()
        P0c = P0.tocsc()
        self.subs = []
        mult = np.zeros(n)
        for i in range(nbins):
            own = np.sort(P0c.indices[P0c.indptr[i]:P0c.indptr[i + 1]])
            allr = np.sort(Pk.indices[Pk.indptr[i]:Pk.indptr[i + 1]])
            ext = np.setdiff1d(allr, own, assume_unique=True)
            idx = np.concatenate([own, ext])
            Ai = A[idx][:, idx].toarray()
            self.subs.append((idx, own.shape[0], np.linalg.inv(Ai)))
            mult[idx] += 1.0
        self.mult = mult
        self.max_size = max(s[0].shape[0] for s in self.subs)

    def apply(self, r: np.ndarray) -> np.ndarray:
        z = np.zeros_like(r)
        for idx, n_own, Ainv in self.subs:
            if self.combine == "restricted":
                z[idx[:n_own]] += Ainv[:n_own] @ r[idx]
            else:
                z[idx] += Ainv @ r[idx]
        if self.combine == "averaging":
            z /= self.mult
        return z


def coarse_lattice(lo: np.ndarray, L: np.ndarray, cells_target: float):
    """Normative coarse lattice of the product's second level: g_d = max(1, floor(L_d / H + 0.5))
    cells per direction, H = (V / cells_target)^(1/dim), V = product of the non-degenerate extents
    of the GLOBAL bounding box [lo, lo + L]; degenerate directions get one cell."""
    dim = lo.shape[0]
    Lpos = np.where(L > 0, L, 1.0)
    H = (float(np.prod(Lpos)) / float(cells_target)) ** (1.0 / dim)
    g = np.maximum(1, np.floor(Lpos / H + 0.5).astype(np.int64))
    return np.where(L > 0, g, 1)


class CoarseQ1:
    """Second (coarse) level added to the one-level operator: M^-1 = M_RAS^-1 + Phi K0^-1 Phi^T.
    Normative definition (DESIGN.md 'two-level'): Phi = multilinear (Q1) hat functions of a regular
    lattice of (g_d + 1) points per direction over the global bounding box, evaluated at the node
    that carries the dof, one copy per dof component (coarse dof = dofs * lattice_node + k), rows of
    Dirichlet dofs zeroed; K0 = Phi^T A Phi, lattice dofs without support get a unit diagonal, the
    rest a relative diagonal shift of 1e-12.  This plays the role of FROSch's GDSWCoarseOperator
    (parametersPrec.xml:62-122, 'TwoLevel' = true); GDSW's interface-based space is defined for few
    large subdomains and is not what is built here (DESIGN.md explains the substitution)."""

    def __init__(self, A: sp.csr_matrix, xyz: np.ndarray, is_dir: np.ndarray, dofs: int = 1,
                 cells_target: float = 1000.0, lo=None, L=None):
        n_nodes, dim = xyz.shape
        lo = xyz.min(axis=0) if lo is None else np.asarray(lo, dtype=float)
        L = (xyz.max(axis=0) - lo) if L is None else np.asarray(L, dtype=float)
        g = coarse_lattice(lo, L, cells_target)
        Lpos = np.where(L > 0, L, 1.0)
        t = (xyz - lo) / Lpos * g
        i0 = np.clip(np.floor(t).astype(np.int64), 0, g - 1)
        f = t - i0
        stride = np.ones(dim, dtype=np.int64)
        for d in range(1, dim):
            stride[d] = stride[d - 1] * (g[d - 1] + 1)
        n_lat = int(np.prod(g + 1))
        rows, cols, vals = [], [], []
        free = (~np.asarray(is_dir, dtype=bool)).astype(float)
        for corner in range(1 << dim):
            w = np.ones(n_nodes)
            node = np.zeros(n_nodes, dtype=np.int64)
            for d in range(dim):
                bit = (corner >> d) & 1
                w = w * (f[:, d] if bit else 1.0 - f[:, d])
                node += stride[d] * (i0[:, d] + bit)
            for k in range(dofs):
                r = dofs * np.arange(n_nodes) + k
                rows.append(r)
                cols.append(dofs * node + k)
                vals.append(w * free[r])
        n = dofs * n_nodes
        self.g = g
        self.n0 = dofs * n_lat
        self.Phi = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                 shape=(n, self.n0))
        K0 = (self.Phi.T @ A.tocsr() @ self.Phi).toarray()
        d = np.diag(K0).copy()
        empty = ~(np.abs(K0).sum(axis=1) > 0)
        K0[np.diag_indices(self.n0)] = np.where(empty, 1.0, d * (1.0 + 1e-12))
        self.K0 = K0
        self.K0inv = np.linalg.inv(K0)

    def apply(self, r: np.ndarray) -> np.ndarray:
        return self.Phi @ (self.K0inv @ (self.Phi.T @ r))


class CoarseGDSW:
    """GDSW coarse level (FROSch GDSWCoarseOperator, parametersPrec.xml:13-23, 62-122) on a second, coarse
    decomposition -- normative definition of the product's FEDD_COARSE_GDSW (feddlib_amd/csrc/coarse.hip):
      * coarse subdomains = cells of the regular lattice `coarse_lattice(lo, L, cells_target)`; an element belongs to
        the cell of its centroid (mean of its first dim + 1 vertices; cell index = clamp(floor((c - lo) / L * g)));
      * a node whose incident elements lie in the cell index box [imin, imax] sits on the interface entity with the
        doubled-lattice coordinates e = imin + imax (e_d odd: between cells in direction d); all e_d even: interior;
      * Phi_Gamma[(node, a), (entity, a)] = 1 for free interface dofs (null space = constants per component);
      * Phi_I = -A_II^-1 A_IGamma Phi_Gamma on the free interior dofs (exact sparse solve here);
      * K0 = Phi^T A Phi over ALL (2g - 1)^dim entities x dofs (cell interiors and empty entities: unit diagonal, the
        rest a relative diagonal shift of 1e-12); the level adds Phi K0^-1 Phi^T.
    rotations = True (FROSch with "Use node lists" and "Rotations" = true, steadyLinElas/parametersPrec.xml:6, 100; vector
    problems with dofs = dim): every entity carries nns = dofs + (1 in 2D, 3 in 3D) functions, the translations and the
    linearised rotations about the entity's centre in the lattice, c_d = lo_d + (e_d + 1) H_d / 2, H_d = L_d / g_d:
    2D (-y, x); 3D about z (-y, x, 0), about x (0, -z, y), about y (z, 0, -x).  Functions that are linearly dependent on the
    entity's free interface dofs are dropped (FROSch: no rotations on vertices, two on straight edges): with G the Gram
    matrix of the entity's functions, a Cholesky sweep in the order of the functions keeps function k when the remainder
    r_k = G_kk - sum_{j kept} L_kj^2 exceeds 1e-8 * s_k * max_t G_tt (t over the translations; s_k = 1 for a translation,
    max_d H_d^2 for a rotation); dropped functions are zero columns (unit diagonal in K0)."""

    def __init__(self, A: sp.csr_matrix, conn: np.ndarray, xyz: np.ndarray, is_dir: np.ndarray, dofs: int = 1,
                 cells_target: float = 8.0, lo=None, L=None, reduced: bool = False, rotations: bool = False):
        """reduced = True: RGDSW, option 1 (FEDD_COARSE_RGDSW): coarse dofs only on the coarse nodes -- entities with
        an odd coordinate in every direction that has >= 2 cells --, numbered compactly ((e_d - 1) / 2 per such
        direction); an interface node of entity e carries 1 / |C(e)| for every coarse node of C(e) (odd e_d kept, even
        e_d of a direction with >= 2 cells moved to e_d - 1 or e_d + 1 inside the lattice)."""
        import itertools
        import scipy.sparse.linalg as spla
        n_nodes, dim = xyz.shape
        lo = xyz.min(axis=0) if lo is None else np.asarray(lo, dtype=float)
        L = (xyz.max(axis=0) - lo) if L is None else np.asarray(L, dtype=float)
        g = coarse_lattice(lo, L, cells_target)
        Lpos = np.where(L > 0, L, 1.0)
        cen = xyz[conn[:, :dim + 1]].sum(axis=1) / (dim + 1)
        ecell = np.clip(np.floor((cen - lo) / Lpos * g).astype(np.int64), 0, g - 1)        # [E, dim]
        imin = np.full((n_nodes, dim), np.iinfo(np.int64).max)
        imax = np.full((n_nodes, dim), -1)
        for k in range(conn.shape[1]):
            np.minimum.at(imin, conn[:, k], ecell)
            np.maximum.at(imax, conn[:, k], ecell)
        lone = imax[:, 0] < 0
        imin[lone] = 0
        imax[lone] = 0
        e = imin + imax                                            # entity coordinates
        m = 2 * g - 1
        stride = np.ones(dim, dtype=np.int64)
        for d in range(1, dim):
            stride[d] = stride[d - 1] * m[d - 1]
        ent = (e * stride).sum(axis=1)
        on_gamma = (e % 2 == 1).any(axis=1)
        n_ent = int(np.prod(m))
        n = dofs * n_nodes
        free = ~np.asarray(is_dir, dtype=bool)
        gamma_dof = np.repeat(on_gamma, dofs)
        comp = np.tile(np.arange(dofs), n_nodes)
        col = np.repeat(ent, dofs) * dofs + comp
        sel = gamma_dof & free
        rotations = bool(rotations) and dofs == dim and dim >= 2
        nns = dofs + ((3 if dim == 3 else 1) if rotations else 0)
        self.nns = nns
        Hd = Lpos / g

        def null_vec(k, dd):
            v = np.zeros(dofs)
            if k < dofs:
                v[k] = 1.0
            elif dim == 2:
                v[:] = (-dd[1], dd[0])
            else:
                v[:] = ((-dd[1], dd[0], 0.0), (0.0, -dd[2], dd[1]), (dd[2], 0.0, -dd[0]))[k - dofs]
            return v

        self.n0 = n_ent * nns
        if not reduced and not rotations:
            PhiG = sp.csr_matrix((np.ones(sel.sum()), (np.nonzero(sel)[0], col[sel])), shape=(n, self.n0))
        elif rotations:
            # (node, coarse id, entity coordinates of the function, weight) for every interface node
            multi = g >= 2
            cstride = np.ones(dim, dtype=np.int64)
            ncoarse = 1
            for d in range(dim):
                cstride[d] = ncoarse
                ncoarse *= (g[d] - 1) if multi[d] else 1
            if reduced:
                self.n0 = ncoarse * nns
            rows, cols, vals = [], [], []
            for node in np.nonzero(on_gamma)[0]:
                if not reduced:
                    targets = [(int(ent[node]), e[node])]
                else:
                    opts = []
                    for d in range(dim):
                        if not multi[d]:
                            opts.append([0])
                        elif e[node, d] % 2 == 1:
                            opts.append([e[node, d]])
                        else:
                            opts.append([v for v in (e[node, d] - 1, e[node, d] + 1) if 1 <= v <= 2 * g[d] - 3])
                    targets = [(int(sum(cstride[d] * ((v[d] - 1) // 2) for d in range(dim) if multi[d])), np.asarray(v))
                               for v in itertools.product(*opts)]
                for cid, ev in targets:
                    dd = xyz[node] - (lo + (np.asarray(ev) + 1) * Hd * 0.5)
                    for k in range(nns):
                        f = null_vec(k, dd) / len(targets)
                        for a in range(dofs):
                            if free[node * dofs + a] and f[a] != 0.0:
                                rows.append(node * dofs + a)
                                cols.append(cid * nns + k)
                                vals.append(f[a])
            PhiG = sp.csr_matrix((vals, (rows, cols)), shape=(n, self.n0))
            # independent functions per coarse id
            G = (PhiG.T @ PhiG).toarray() if self.n0 <= 4096 else None
            hmax2 = float((Hd[:dim] ** 2).max())
            keep = np.zeros(self.n0, dtype=bool)
            for E in range(self.n0 // nns):
                sl = slice(E * nns, (E + 1) * nns)
                Ge = G[sl, sl] if G is not None else (PhiG[:, sl].T @ PhiG[:, sl]).toarray()
                tmax = max(Ge[t, t] for t in range(dofs))
                Lc = np.zeros((nns, nns))
                kept = []
                for k in range(nns):
                    r = Ge[k, k]
                    for j in kept:
                        v = Ge[k, j] - sum(Lc[k, q] * Lc[j, q] for q in kept if q < j)
                        Lc[k, j] = v / Lc[j, j]
                        r -= Lc[k, j] ** 2
                    if tmax > 0.0 and r > 1e-8 * (1.0 if k < dofs else hmax2) * tmax:
                        kept.append(k)
                        Lc[k, k] = math.sqrt(r)
                        keep[E * nns + k] = True
            self.kept = keep
            PhiG = (PhiG @ sp.diags(keep.astype(float))).tocsr()
            PhiG.eliminate_zeros()
        else:
            multi = g >= 2
            cstride = np.ones(dim, dtype=np.int64)
            ncoarse = 1
            for d in range(dim):
                cstride[d] = ncoarse
                ncoarse *= (g[d] - 1) if multi[d] else 1
            self.n0 = ncoarse * dofs
            rows, cols, vals = [], [], []
            for node in np.nonzero(on_gamma)[0]:
                opts = []
                for d in range(dim):
                    if not multi[d]:
                        opts.append([0])
                    elif e[node, d] % 2 == 1:
                        opts.append([e[node, d]])
                    else:
                        opts.append([v for v in (e[node, d] - 1, e[node, d] + 1) if 1 <= v <= 2 * g[d] - 3])
                adj = list(itertools.product(*opts))
                for v in adj:
                    cid = sum(cstride[d] * ((v[d] - 1) // 2) for d in range(dim) if multi[d])
                    for a in range(dofs):
                        if free[node * dofs + a]:
                            rows.append(node * dofs + a)
                            cols.append(cid * dofs + a)
                            vals.append(1.0 / len(adj))
            PhiG = sp.csr_matrix((vals, (rows, cols)), shape=(n, self.n0))
        A = A.tocsr()
        I = np.nonzero(~gamma_dof & free)[0]
        Phi = PhiG.tolil()
        if I.shape[0]:
            AII = A[I][:, I].tocsc()
            rhs = -(A[I] @ PhiG).toarray()
            X = spla.splu(AII).solve(rhs)
            nz = np.abs(rhs).sum(axis=0) > 0
            Phi = (PhiG + sp.csr_matrix((X[:, nz].ravel(), (np.repeat(I, nz.sum()), np.tile(np.nonzero(nz)[0], I.shape[0]))),
                                        shape=(n, self.n0))).tocsr()
        self.g = g
        self.Phi = sp.csr_matrix(Phi)
        K0 = (self.Phi.T @ A @ self.Phi).toarray()
        d = np.diag(K0).copy()
        empty = ~(np.abs(K0).sum(axis=1) > 0)
        K0[np.diag_indices(self.n0)] = np.where(empty, 1.0, d * (1.0 + 1e-12))
        self.K0 = K0
        self.K0inv = np.linalg.inv(K0)
        self.n_interface_entities = int(np.unique(ent[on_gamma]).shape[0])

    def apply(self, r: np.ndarray) -> np.ndarray:
        return self.Phi @ (self.K0inv @ (self.Phi.T @ r))


def gmres_right(A, b, M=None, rtol=1e-8, max_it=100, restart=100, x0=None):
    """Right-preconditioned restarted GMRES, block size 1 (what Stratimikos/Belos 'Block GMRES'
    with an 'unspecified'-side Thyra preconditioner runs; parametersSolver.xml:5-15), classical
    Gram-Schmidt with DGKS-style re-orthogonalisation (always two passes here), Givens QR,
    convergence on the implicit residual ||r_k|| / ||r_0|| <= rtol, r_0 = b - A x0.
    Returns (x, iterations, relres_history)."""
    n = b.shape[0]
    x = np.zeros(n) if x0 is None else x0.copy()
    Mop = (lambda v: v) if M is None else M
    r = b - A @ x
    beta0 = np.linalg.norm(r)
    hist = [1.0]
    if beta0 == 0.0:
        return x, 0, hist
    its = 0
    while its < max_it:
        beta = np.linalg.norm(r)
        m = min(restart, max_it - its)
        V = np.zeros((m + 1, n))
        Z = np.zeros((m, n))
        H = np.zeros((m + 1, m))
        cs = np.zeros(m)
        sn = np.zeros(m)
        g = np.zeros(m + 1)
        g[0] = beta
        V[0] = r / beta
        k = 0
        done = False
        for j in range(m):
            Z[j] = Mop(V[j])
            w = A @ Z[j]
            h = V[:j + 1] @ w
            w = w - h @ V[:j + 1]
            h2 = V[:j + 1] @ w
            w = w - h2 @ V[:j + 1]
            h = h + h2
            hn = np.linalg.norm(w)
            H[:j + 1, j] = h
            H[j + 1, j] = hn
            for i in range(j):
                t = cs[i] * H[i, j] + sn[i] * H[i + 1, j]
                H[i + 1, j] = -sn[i] * H[i, j] + cs[i] * H[i + 1, j]
                H[i, j] = t
            d = math.hypot(H[j, j], H[j + 1, j])
            cs[j] = H[j, j] / d
            sn[j] = H[j + 1, j] / d
            H[j, j] = d
            H[j + 1, j] = 0.0
            g[j + 1] = -sn[j] * g[j]
            g[j] = cs[j] * g[j]
            its += 1
            k = j + 1
            hist.append(abs(g[j + 1]) / beta0)
            if hn != 0.0:
                V[j + 1] = w / hn
            if hist[-1] <= rtol or hn == 0.0:
                done = True
                break
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
        x = x + y @ Z[:k]
        r = b - A @ x
        if done or its >= max_it:
            break
    return x, its, hist


def solve_monolithic(A, b, prec=None, coarse_only=None, x_init=None, zero_initial_guess=True,
                     level_combination="Additive", rtol=1e-8, max_it=100, restart=100):
    """LinearSolver::solveMonolithic (feddlib/problems/Solver/LinearSolver_def.hpp:72-135), the part around Thyra::solve:
    "Zero Initial Guess" = true clears the solution vector (:76-78), otherwise the vector the problem holds is x_0;
    "Level Combination" = "Multiplicative" applies the preconditioner with "Only apply coarse" to the right-hand side, INTO
    the solution vector, before the solve (:98-104) -- the solve then starts from that vector with the preconditioner as it
    was built.  prec / coarse_only are callables r -> z.  Returns (x, iterations, relres history vs ||b - A x_0||)."""
    x = np.zeros(b.shape[0]) if (zero_initial_guess or x_init is None) else np.array(x_init, dtype=float, copy=True)
    if level_combination == "Multiplicative":
        x = coarse_only(b)
    return gmres_right(A, b, prec, rtol=rtol, max_it=max_it, restart=restart, x0=x)


def direct_solve(A: sp.csr_matrix, b: np.ndarray, refine: int = 2) -> np.ndarray:
    """Sparse LU + a few steps of iterative refinement (badly scaled systems, e.g. elasticity with
    unit Dirichlet rows next to 1e6-sized entries, otherwise keep O(1e-16) absolute noise at the
    Dirichlet dofs, which is above 1e-10 relative to max|x|)."""
    lu = spla.splu(A.tocsc())
    x = lu.solve(b)
    for _ in range(refine):
        x = x + lu.solve(b - A @ x)
    return x


def spmv(A: sp.csr_matrix, x: np.ndarray) -> np.ndarray:
    """Matrix::apply  (Matrix_def.hpp:245-254)."""
    return A @ x
