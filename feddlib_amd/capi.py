"""ctypes binding of include/fedd_hip.h -- the same C ABI a reference-side stub would bind
(INTEGRATION.md).  Host-side plumbing for tests and bench.py only: every numeric call lands in
the HIP library; if libfedd_hip.so is missing this module raises (no fallback path)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfedd_hip.so")

FORM_LAPLACE, FORM_LAPLACE_VEC, FORM_MASS, FORM_MASS_VEC, FORM_LINELAS, FORM_BDSTAB = range(6)
BLOCK_SCALAR, BLOCK_DIAG, BLOCK_FULL = range(3)
COMBINE_RESTRICTED, COMBINE_AVERAGING, COMBINE_FULL = range(3)
(T_SYMBOLIC, T_ASSEMBLE, T_RHS, T_DIRICHLET, T_SPMV, T_SCHWARZ_SETUP, T_SCHWARZ_APPLY, T_ORTHO, T_COARSE_SETUP,
 T_COARSE_APPLY, T_HALO, T_ALLREDUCE, T_SPMV_SETUP) = range(13)
TIMER_NAMES = ["symbolic", "assemble", "rhs", "dirichlet", "spmv", "schwarz_setup", "schwarz_apply", "ortho",
               "coarse_setup", "coarse_apply", "halo", "allreduce", "spmv_setup", "gs_dot", "gs_update", "gs_fused"]
COARSE_Q1 = 1
COARSE_GDSW = 2
COARSE_RGDSW = 3

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# name -> argtypes; every symbol declared in include/fedd_hip.h (tests check the two lists agree)
SIGNATURES = {
    "fedd_ctx_create": [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_int],
    "fedd_ctx_destroy": [C.c_void_p],
    "fedd_last_error": [],
    "fedd_sync": [C.c_void_p],
    "fedd_nccl_unique_id": [C.c_void_p],
    "fedd_mesh_structured_sizes": [C.c_int, _ip, _ip, C.c_int, C.c_int, _i64p, _i64p, _i64p, _i64p],
    "fedd_mesh_structured_build": [C.c_int, _ip, _ip, C.c_int, _f64p, _f64p, C.c_int, C.c_int, _i32p, _f64p,
                                   _i64p, _i32p, _i64p, _i32p],
    "fedd_mesh_structured_owner": [C.c_int, _ip, _ip, C.c_int64, _i64p, _i32p],
    "fedd_mesh_structured_row_ghosts": [C.c_int, _ip, _ip, C.c_int, C.c_int, _f64p, _f64p, C.c_int, _i64p, _i64p, _i32p],
    "fedd_mesh_read_sizes": [C.c_char_p, C.c_int, _i64p, _i64p, _i64p],
    "fedd_mesh_read": [C.c_char_p, C.c_int, _f64p, _i32p, _i32p, _i32p, _i32p, _i32p],
    "fedd_mesh_p2_sizes": [C.c_int, C.c_int64, _i32p, _i64p],
    "fedd_mesh_p2_build": [C.c_int, C.c_int64, C.c_int64, _i32p, _f64p, _i32p, C.c_int64, _i32p, _i32p, C.c_int,
                           _i32p, _f64p, _i32p],
    "fedd_mesh_partition": [C.c_int, C.c_int, C.c_int64, _i32p, C.c_int64, _f64p, C.c_int, _i32p],
    "fedd_mesh_partition_sizes": [C.c_int, C.c_int64, _i32p, C.c_int64, _i32p, C.c_int, C.c_int, C.c_int, _i64p, _i64p, _i64p, _i64p],
    "fedd_mesh_partition_extract": [C.c_int, C.c_int, C.c_int64, _i32p, C.c_int64, _f64p, _i32p, _i32p, C.c_int, C.c_int, C.c_int,
                                    _i32p, _f64p, _i64p, _i32p, _i32p, _i64p, _i32p, _i64p, _i32p, _i64p],
    "fedd_fe_quadrature": [C.c_int, C.c_int, _ip, _f64p, _f64p],
    "fedd_fe_basis": [C.c_int, C.c_int, C.c_int, _f64p, _f64p],
    "fedd_mesh_set": [C.c_void_p, C.c_int, C.c_int, C.c_int64, _i32p, C.c_int64, _f64p, _i64p, C.c_int64,
                      _i64p, _i32p],
    "fedd_mesh_set_rows": [C.c_void_p, C.c_int, C.c_int, C.c_int64, _i32p, C.c_int64, _f64p, _i64p, C.c_int64,
                           _i64p, _i32p, C.c_int64, _i64p, _i32p],
    "fedd_pattern_build": [C.c_void_p, C.c_int, C.c_int, _i64p],
    "fedd_assemble": [C.c_void_p, C.c_int, _f64p],
    "fedd_assemble_rhs": [C.c_void_p, C.c_int, _f64p, C.c_int],
    "fedd_dirichlet": [C.c_void_p, C.c_int, _i32p, _i32p, _f64p],
    "fedd_dirichlet_nodes": [C.c_void_p, C.c_int64, _i32p, _i32p, _f64p],
    "fedd_dirichlet_rows": [C.c_void_p, C.c_int64, _i32p, _f64p],
    "fedd_matrix_store": [C.c_void_p, C.c_int],
    "fedd_matrix_scale": [C.c_void_p, C.c_int, C.c_double],
    "fedd_assemble_div": [C.c_void_p, C.c_int64, C.c_int, C.c_int],
    "fedd_block_merge": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int],
    "fedd_matrix_sizes": [C.c_void_p, C.c_int, _i64p, _i64p, _i64p],
    "fedd_matrix_get": [C.c_void_p, C.c_int, _i64p, _i32p, _f64p],
    "fedd_csr_sizes": [C.c_void_p, _i64p, _i64p, _i64p],
    "fedd_csr_get": [C.c_void_p, _i64p, _i32p, _f64p, _i64p],
    "fedd_rhs_get": [C.c_void_p, _f64p],
    "fedd_rhs_set": [C.c_void_p, _f64p],
    "fedd_solution_get": [C.c_void_p, _f64p],
    "fedd_spmv": [C.c_void_p, _f64p, _f64p],
    "fedd_spmv_device": [C.c_void_p, C.c_int],
    "fedd_spmv_info": [C.c_void_p, _i64p, _i64p],
    "fedd_schwarz_setup": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int],
    "fedd_schwarz_set_target": [C.c_void_p, C.c_int, C.c_double],
    "fedd_schwarz_set_coarse": [C.c_void_p, C.c_double],
    "fedd_schwarz_coarse_sizes": [C.c_void_p, _i32p, _i64p],
    "fedd_schwarz_coarse_get": [C.c_void_p, _f64p],
    "fedd_schwarz_apply": [C.c_void_p, _f64p, _f64p],
    "fedd_schwarz_apply_device": [C.c_void_p, C.c_int],
    "fedd_schwarz_info": [C.c_void_p, _i64p, _i64p, _i64p],
    "fedd_schwarz_unique": [C.c_void_p, _i64p],
    "fedd_schwarz_sizes": [C.c_void_p, _i64p, _i64p],
    "fedd_schwarz_conforming": [C.c_void_p, _i64p],
    "fedd_spmv_patterns": [C.c_void_p, _i64p, _i64p],
    "fedd_spmv_classes": [C.c_void_p, _i64p, _i64p, _i64p],
    "fedd_spmv_col_bytes": [C.c_void_p, C.POINTER(C.c_int), _i64p],
    "fedd_gmres": [C.c_void_p, _f64p, _f64p, C.c_double, C.c_int, C.c_int, C.c_int, _ip, _f64p],
    "fedd_set_option": [C.c_void_p, C.c_char_p, C.c_double],
    "fedd_timing_enable": [C.c_void_p, C.c_int],
    "fedd_timing_reset": [C.c_void_p],
    "fedd_timing_get": [C.c_void_p, C.c_int, _f64p, _i64p],
    "fedd_timing_get_sampled": [C.c_void_p, C.c_int, _f64p, _i64p, _f64p],
    "fedd_gmres_info": [C.c_void_p, _ip, _ip, _ip, _ip],
    "fedd_gmres_fused_blocks": [C.c_void_p, C.POINTER(C.c_int)],
    "fedd_gmres_x0": [C.c_void_p, _f64p, _f64p, C.c_double, C.c_int, C.c_int, C.c_int, _ip, _f64p],
    "fedd_gmres_status": [C.c_void_p, _ip, _f64p],
    "fedd_mesh_setup_info": [C.c_void_p, _f64p, _f64p, _ip, _i64p],
    "fedd_schwarz_coarse_apply": [C.c_void_p, _f64p, _f64p],
    "fedd_read_bandwidth": [C.c_void_p, C.c_int64, C.c_int, _f64p],
    "fedd_rccl_selftest": [C.c_void_p, C.c_int, _f64p],
    "fedd_comm_selftest": [C.c_void_p, C.c_int, _f64p],
    "fedd_halo_plan_sizes": [C.c_void_p, _ip, _i64p, _i64p],
    "fedd_halo_plan_get": [C.c_void_p, _i32p, _i64p, _i32p, _i64p, _i32p],
    "fedd_halo_set_owners": [C.c_void_p, C.c_int64, _i64p, _i32p],
    "fedd_halo_requests_sizes": [C.c_void_p, _i64p],
    "fedd_halo_requests_get": [C.c_void_p, _i64p],
    "fedd_halo_requests_set": [C.c_void_p, _i64p, _i64p],
    "fedd_halo_exchange_setup": [C.c_void_p],
    "fedd_comm_set_host_callbacks": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
}

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _i32p, _i64p, _f64p, _i64p, _f64p, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _f64p, C.c_int)


class ThreadGroup:
    """In-process stand-in for a communicator: `world` threads of one process, one rank (and one
    fedd_ctx) each.  Lets the complete N > 1 path run with more ranks than a one-GPU box admits
    processes (the 2 x 2 x 2 decomposition: 8 ranks).  Functional tests only."""

    def __init__(self, world, timeout=120.0):
        import collections
        import threading
        self.world = world
        self.timeout = timeout
        self._barrier = threading.Barrier(world)
        self._slots = [None] * world
        self._cv = threading.Condition()
        self._box = collections.defaultdict(collections.deque)

    def allgather(self, rank, obj):
        self._slots[rank] = obj
        self._barrier.wait(self.timeout)
        out = list(self._slots)
        self._barrier.wait(self.timeout)
        return out

    def send(self, src, dst, arr):
        with self._cv:
            self._box[(src, dst)].append(np.array(arr, copy=True))
            self._cv.notify_all()

    def recv(self, src, dst):
        with self._cv:
            if not self._cv.wait_for(lambda: len(self._box[(src, dst)]) > 0, self.timeout):
                raise TimeoutError("ThreadGroup.recv %d <- %d" % (dst, src))
            return self._box[(src, dst)].popleft()

_lib = None


class FeddError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FeddError("%s is missing: build it with `python -m feddlib_amd.build` "
                            "(hipcc, gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.fedd_last_error.restype = C.c_char_p
        L.fedd_ctx_destroy.restype = None
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise FeddError(lib().fedd_last_error().decode())


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


def _ints(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _decomp(dim, N):
    return [int(N)] * dim if np.isscalar(N) else [int(x) for x in N]


def structured_mesh(dim, N, M, rank=0, origin=None, size=None, flags_option=1, ghosts=False):
    """Product-side structured generator (host code in the library).  N, M: ints or per-direction lists.
    ghosts: False / 0 = the reference's block, True / 1 = plus the elements that complete the owned rows,
    L >= 2 = L element layers around the owned nodes, which complete the rows of the ghost nodes within
    L - 1 layers (the dict then carries row_ghost_gid / row_ghost_flag, which Context.mesh_set_dict hands
    to fedd_mesh_set_rows)."""
    L = lib()
    dec, cel = _decomp(dim, N), _decomp(dim, M)
    ne, nr, nu, ng = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    _chk(L.fedd_mesh_structured_sizes(dim, _ints(dec), _ints(cel), rank, int(ghosts), C.byref(ne), C.byref(nr),
                                      C.byref(nu), C.byref(ng)))
    conn = np.zeros((ne.value, dim + 1), dtype=np.int32)
    xyz = np.zeros((nr.value, dim), dtype=np.float64)
    gid_rep = np.zeros(nr.value, dtype=np.int64)
    flag_rep = np.zeros(nr.value, dtype=np.int32)
    gid_uni = np.zeros(nu.value, dtype=np.int64)
    flag_uni = np.zeros(nu.value, dtype=np.int32)
    o = None if origin is None else np.ascontiguousarray(origin, dtype=np.float64)
    s = None if size is None else np.ascontiguousarray(size, dtype=np.float64)
    _chk(L.fedd_mesh_structured_build(dim, _ints(dec), _ints(cel), rank, _p(o, _f64p), _p(s, _f64p), flags_option,
                                      int(ghosts), _p(conn, _i32p), _p(xyz, _f64p), _p(gid_rep, _i64p),
                                      _p(flag_rep, _i32p), _p(gid_uni, _i64p), _p(flag_uni, _i32p)))
    out = dict(dim=dim, nen=dim + 1, conn=conn, xyz=xyz, gid_rep=gid_rep, flag_rep=flag_rep, gid_uni=gid_uni,
               flag_uni=flag_uni, n_global=ng.value, decomp=dec, cells=cel, rank=rank)
    if int(ghosts) >= 2:
        nrg = C.c_int64()
        _chk(L.fedd_mesh_structured_row_ghosts(dim, _ints(dec), _ints(cel), rank, int(ghosts), _p(o, _f64p), _p(s, _f64p),
                                               flags_option, C.byref(nrg), None, None))
        rg = np.zeros(nrg.value, dtype=np.int64)
        rf = np.zeros(nrg.value, dtype=np.int32)
        _chk(L.fedd_mesh_structured_row_ghosts(dim, _ints(dec), _ints(cel), rank, int(ghosts), _p(o, _f64p), _p(s, _f64p),
                                               flags_option, C.byref(nrg), _p(rg, _i64p), _p(rf, _i32p)))
        out["row_ghost_gid"], out["row_ghost_flag"] = rg, rf
    return out


def read_mesh(path, dim):
    """INRIA .mesh file -> one-rank P1 mesh dict (repeated = unique = identity numbering)."""
    L = lib()
    nv, ne, ns = C.c_int64(), C.c_int64(), C.c_int64()
    _chk(L.fedd_mesh_read_sizes(path.encode(), dim, C.byref(nv), C.byref(ne), C.byref(ns)))
    xyz = np.zeros((nv.value, dim)); vflag = np.zeros(nv.value, dtype=np.int32)
    conn = np.zeros((ne.value, dim + 1), dtype=np.int32); eflag = np.zeros(ne.value, dtype=np.int32)
    surf = np.zeros((ns.value, dim), dtype=np.int32); sflag = np.zeros(ns.value, dtype=np.int32)
    _chk(L.fedd_mesh_read(path.encode(), dim, _p(xyz, _f64p), _p(vflag, _i32p), _p(conn, _i32p), _p(eflag, _i32p),
                          _p(surf, _i32p), _p(sflag, _i32p)))
    gid = np.arange(nv.value, dtype=np.int64)
    return dict(dim=dim, nen=dim + 1, conn=conn, xyz=xyz, gid_rep=gid, flag_rep=vflag, gid_uni=gid.copy(),
                flag_uni=vflag.copy(), n_global=nv.value, elem_flag=eflag, surf=surf, surf_flag=sflag)


def p2_of_p1(m, volume_id=10):
    """P2 mesh dict from a one-rank P1 mesh dict (edge mid-points)."""
    L = lib()
    dim = m["dim"]
    conn = np.ascontiguousarray(m["conn"], dtype=np.int32)
    ned = C.c_int64()
    _chk(L.fedd_mesh_p2_sizes(dim, conn.shape[0], _p(conn, _i32p), C.byref(ned)))
    nv = m["xyz"].shape[0]
    nen2 = 10 if dim == 3 else 6
    conn2 = np.zeros((conn.shape[0], nen2), dtype=np.int32)
    xyz2 = np.zeros((nv + ned.value, dim)); flag2 = np.zeros(nv + ned.value, dtype=np.int32)
    xyz = np.ascontiguousarray(m["xyz"], dtype=np.float64); vf = np.ascontiguousarray(m["flag_rep"], dtype=np.int32)
    surf = np.ascontiguousarray(m.get("surf", np.zeros((0, dim), np.int32)), dtype=np.int32)
    sflag = np.ascontiguousarray(m.get("surf_flag", np.zeros(0, np.int32)), dtype=np.int32)
    _chk(L.fedd_mesh_p2_build(dim, nv, conn.shape[0], _p(conn, _i32p), _p(xyz, _f64p), _p(vf, _i32p), surf.shape[0],
                              _p(surf, _i32p), _p(sflag, _i32p), volume_id, _p(conn2, _i32p), _p(xyz2, _f64p),
                              _p(flag2, _i32p)))
    gid = np.arange(nv + ned.value, dtype=np.int64)
    return dict(dim=dim, nen=nen2, conn=conn2, xyz=xyz2, gid_rep=gid, flag_rep=flag2, gid_uni=gid.copy(),
                flag_uni=flag2.copy(), n_global=nv + ned.value, n_p1=nv, elem_flag=m.get("elem_flag"))


def partition_mesh(m, nparts):
    """element -> part of a one-rank mesh dict (read_mesh / p2_of_p1 / structured_mesh on one block)"""
    conn = np.ascontiguousarray(m["gid_rep"][m["conn"]], dtype=np.int32) if "gid_rep" in m else np.ascontiguousarray(m["conn"], np.int32)
    xyz = np.ascontiguousarray(m["xyz"], dtype=np.float64)
    part = np.zeros(conn.shape[0], dtype=np.int32)
    _chk(lib().fedd_mesh_partition(m["dim"], conn.shape[1], conn.shape[0], _p(conn, _i32p), xyz.shape[0], _p(xyz, _f64p),
                                   nparts, _p(part, _i32p)))
    return part


def partitioned_mesh(m, part, nparts, rank, ghosts=1):
    """rank's mesh dict (what structured_mesh(..., rank, ghosts=L) returns for the structured grid) cut out of the
    one-rank mesh dict m by the element partition `part`; carries owner_rep for Context.halo_set_owners"""
    L = lib()
    dim = m["dim"]
    conn = np.ascontiguousarray(m["conn"], dtype=np.int32)
    xyz = np.ascontiguousarray(m["xyz"], dtype=np.float64)
    flag = np.ascontiguousarray(m["flag_rep"], dtype=np.int32)
    part = np.ascontiguousarray(part, dtype=np.int32)
    nen = conn.shape[1]
    ne, nr, nu, ng = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    _chk(L.fedd_mesh_partition_sizes(nen, conn.shape[0], _p(conn, _i32p), xyz.shape[0], _p(part, _i32p), nparts, rank, int(ghosts),
                                     C.byref(ne), C.byref(nr), C.byref(nu), C.byref(ng)))
    out = dict(dim=dim, nen=nen, conn=np.zeros((ne.value, nen), np.int32), xyz=np.zeros((nr.value, dim)),
               gid_rep=np.zeros(nr.value, np.int64), flag_rep=np.zeros(nr.value, np.int32), owner_rep=np.zeros(nr.value, np.int32),
               gid_uni=np.zeros(nu.value, np.int64), flag_uni=np.zeros(nu.value, np.int32),
               row_ghost_gid=np.zeros(ng.value, np.int64), row_ghost_flag=np.zeros(ng.value, np.int32),
               elem_gid=np.zeros(ne.value, np.int64), n_global=xyz.shape[0], rank=rank)
    _chk(L.fedd_mesh_partition_extract(dim, nen, conn.shape[0], _p(conn, _i32p), xyz.shape[0], _p(xyz, _f64p), _p(flag, _i32p),
                                       _p(part, _i32p), nparts, rank, int(ghosts), _p(out["conn"], _i32p), _p(out["xyz"], _f64p),
                                       _p(out["gid_rep"], _i64p), _p(out["flag_rep"], _i32p), _p(out["owner_rep"], _i32p),
                                       _p(out["gid_uni"], _i64p), _p(out["flag_uni"], _i32p), _p(out["row_ghost_gid"], _i64p),
                                       _p(out["row_ghost_flag"], _i32p), _p(out["elem_gid"], _i64p)))
    if int(ghosts) < 2:
        del out["row_ghost_gid"], out["row_ghost_flag"]
    return out


def fe_quadrature(dim, degree):
    """(points [nq, dim], weights [nq]) of the product's quadrature rule"""
    nq = C.c_int()
    _chk(lib().fedd_fe_quadrature(dim, degree, C.byref(nq), None, None))
    pts = np.zeros((nq.value, dim)); w = np.zeros(nq.value)
    _chk(lib().fedd_fe_quadrature(dim, degree, C.byref(nq), _p(pts, _f64p), _p(w, _f64p)))
    return pts, w


def fe_basis(dim, nen, degree):
    """(phi [nq, nen], dphi [nq, nen, dim]) at the points of fe_quadrature(dim, degree)"""
    nq = fe_quadrature(dim, degree)[1].shape[0]
    phi = np.zeros((nq, nen)); dphi = np.zeros((nq, nen, dim))
    _chk(lib().fedd_fe_basis(dim, nen, degree, _p(phi, _f64p), _p(dphi, _f64p)))
    return phi, dphi


def structured_owner(dim, N, M, gids):
    dec, cel = _decomp(dim, N), _decomp(dim, M)
    g = np.ascontiguousarray(gids, dtype=np.int64)
    out = np.zeros(g.shape[0], dtype=np.int32)
    _chk(lib().fedd_mesh_structured_owner(dim, _ints(dec), _ints(cel), g.shape[0], _p(g, _i64p), _p(out, _i32p)))
    return out


def nccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _chk(lib().fedd_nccl_unique_id(buf))
    return buf.raw


class Context:
    """One per GPU / rank.  device < 0 gives a host-only context (numbering and halo planning only)."""

    def __init__(self, device=0, rank=0, nranks=1, nccl_id: bytes | None = None):
        self._L = lib()
        h = C.c_void_p()
        idbuf = C.create_string_buffer(nccl_id, 128) if nccl_id is not None else None
        _chk(self._L.fedd_ctx_create(C.byref(h), device, idbuf, rank, nranks))
        self._h = h
        self.rank, self.nranks = rank, nranks
        self.dofs = 1

    def close(self):
        if self._h:
            self._L.fedd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _chk(self._L.fedd_sync(self._h))

    def mesh_set(self, dim, conn, xyz, gid_rep, gid_uni, flag_uni, row_ghost_gid=None, row_ghost_flag=None):
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        gid_rep = np.ascontiguousarray(gid_rep, dtype=np.int64)
        gid_uni = np.ascontiguousarray(gid_uni, dtype=np.int64)
        flag_uni = None if flag_uni is None else np.ascontiguousarray(flag_uni, dtype=np.int32)
        if row_ghost_gid is not None and len(row_ghost_gid):
            rg = np.ascontiguousarray(row_ghost_gid, dtype=np.int64)
            rf = None if row_ghost_flag is None else np.ascontiguousarray(row_ghost_flag, dtype=np.int32)
            _chk(self._L.fedd_mesh_set_rows(self._h, dim, conn.shape[1], conn.shape[0], _p(conn, _i32p), xyz.shape[0],
                                            _p(xyz, _f64p), _p(gid_rep, _i64p), gid_uni.shape[0], _p(gid_uni, _i64p),
                                            _p(flag_uni, _i32p), rg.shape[0], _p(rg, _i64p), _p(rf, _i32p)))
        else:
            _chk(self._L.fedd_mesh_set(self._h, dim, conn.shape[1], conn.shape[0], _p(conn, _i32p), xyz.shape[0],
                                       _p(xyz, _f64p), _p(gid_rep, _i64p), gid_uni.shape[0], _p(gid_uni, _i64p),
                                       _p(flag_uni, _i32p)))
        self.n_own = gid_uni.shape[0]

    def mesh_set_dict(self, m):
        self.mesh_set(m["dim"], m["conn"], m["xyz"], m["gid_rep"], m["gid_uni"], m["flag_uni"],
                      m.get("row_ghost_gid"), m.get("row_ghost_flag"))

    def pattern_build(self, dofs=1, block_mode=BLOCK_SCALAR) -> int:
        nnz = C.c_int64()
        _chk(self._L.fedd_pattern_build(self._h, dofs, block_mode, C.byref(nnz)))
        self.dofs = dofs
        return nnz.value

    def assemble(self, form, params=None):
        p = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
        _chk(self._L.fedd_assemble(self._h, form, _p(p, _f64p)))

    def assemble_rhs(self, f_const, extra_degree=0):
        f = np.ascontiguousarray(np.atleast_1d(f_const), dtype=np.float64)
        _chk(self._L.fedd_assemble_rhs(self._h, self.dofs, _p(f, _f64p), extra_degree))

    def dirichlet(self, flags, values=None, comp_mask=None):
        fl = np.ascontiguousarray(flags, dtype=np.int32)
        n = fl.shape[0]
        v = np.zeros(n * self.dofs) if values is None else np.ascontiguousarray(values, dtype=np.float64).ravel()
        m = None if comp_mask is None else np.ascontiguousarray(comp_mask, dtype=np.int32).ravel()
        _chk(self._L.fedd_dirichlet(self._h, n, _p(fl, _i32p), _p(m, _i32p), _p(v, _f64p)))

    def dirichlet_nodes(self, nodes, values, comp_mask=None):
        nd = np.ascontiguousarray(nodes, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        m = None if comp_mask is None else np.ascontiguousarray(comp_mask, dtype=np.int32).ravel()
        _chk(self._L.fedd_dirichlet_nodes(self._h, nd.shape[0], _p(nd, _i32p), _p(m, _i32p), _p(v, _f64p)))

    def dirichlet_rows(self, rows, values):
        r = np.ascontiguousarray(rows, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=np.float64)
        _chk(self._L.fedd_dirichlet_rows(self._h, r.shape[0], _p(r, _i32p), _p(v, _f64p)))

    def matrix_store(self, slot):
        _chk(self._L.fedd_matrix_store(self._h, slot))

    def matrix_scale(self, slot, alpha):
        _chk(self._L.fedd_matrix_scale(self._h, slot, float(alpha)))

    def assemble_div(self, n_pressure_nodes, slot_b, slot_bt):
        _chk(self._L.fedd_assemble_div(self._h, n_pressure_nodes, slot_b, slot_bt))

    def block_merge(self, slot_a, slot_bt, slot_b, slot_c=-1):
        _chk(self._L.fedd_block_merge(self._h, slot_a, slot_bt, slot_b, slot_c))
        self.dofs = 1

    def matrix_get(self, slot):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(self._L.fedd_matrix_sizes(self._h, slot, C.byref(a), C.byref(b), C.byref(c)))
        rowptr = np.zeros(a.value + 1, dtype=np.int64)
        col = np.zeros(c.value, dtype=np.int32)
        val = np.zeros(c.value, dtype=np.float64)
        _chk(self._L.fedd_matrix_get(self._h, slot, _p(rowptr, _i64p), _p(col, _i32p), _p(val, _f64p)))
        import scipy.sparse as sp
        return sp.csr_matrix((val, col, rowptr), shape=(a.value, b.value))

    def csr_sizes(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(self._L.fedd_csr_sizes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def csr_get(self):
        nr, nc, nnz = self.csr_sizes()
        rowptr = np.zeros(nr + 1, dtype=np.int64)
        col = np.zeros(nnz, dtype=np.int32)
        val = np.zeros(nnz, dtype=np.float64)
        gid = np.zeros(nc, dtype=np.int64)
        _chk(self._L.fedd_csr_get(self._h, _p(rowptr, _i64p), _p(col, _i32p), _p(val, _f64p), _p(gid, _i64p)))
        return rowptr, col, val, gid

    def rhs_get(self):
        nr = self.csr_sizes()[0]
        out = np.zeros(nr)
        _chk(self._L.fedd_rhs_get(self._h, _p(out, _f64p)))
        return out

    def rhs_set(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        _chk(self._L.fedd_rhs_set(self._h, _p(b, _f64p)))

    def solution_get(self):
        nr = self.csr_sizes()[0]
        out = np.zeros(nr)
        _chk(self._L.fedd_solution_get(self._h, _p(out, _f64p)))
        return out

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        _chk(self._L.fedd_spmv(self._h, _p(x, _f64p), _p(y, _f64p)))
        return y

    def spmv_info(self):
        a, b = C.c_int64(), C.c_int64()
        _chk(self._L.fedd_spmv_info(self._h, C.byref(a), C.byref(b)))
        p, e = C.c_int64(), C.c_int64()
        _chk(self._L.fedd_spmv_patterns(self._h, C.byref(p), C.byref(e)))
        cb, wide = C.c_int(), C.c_int64()
        _chk(self._L.fedd_spmv_col_bytes(self._h, C.byref(cb), C.byref(wide)))
        nc, rc, rest = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(self._L.fedd_spmv_classes(self._h, C.byref(nc), C.byref(rc), C.byref(rest)))
        return dict(nnz_pattern=a.value, nnz_streamed=b.value, column_patterns=p.value, rows_with_explicit_columns=e.value,
                    column_index_bytes=cb.value, entries_with_32bit_columns=wide.value, row_classes=nc.value,
                    rows_in_classes=rc.value, nnz_streamed_outside_classes=rest.value)

    def spmv_device(self, reps):
        _chk(self._L.fedd_spmv_device(self._h, reps))

    def schwarz_set_target(self, target, scale=1.0):
        _chk(self._L.fedd_schwarz_set_target(self._h, target, scale))

    def schwarz_setup(self, overlap=1, combine=COMBINE_RESTRICTED, two_level=0, coarse_kind=0):
        _chk(self._L.fedd_schwarz_setup(self._h, overlap, combine, two_level, coarse_kind))

    def schwarz_set_coarse(self, cells_target):
        _chk(self._L.fedd_schwarz_set_coarse(self._h, float(cells_target)))

    def schwarz_coarse_sizes(self):
        g = np.zeros(3, dtype=np.int32)
        n0 = C.c_int64()
        _chk(self._L.fedd_schwarz_coarse_sizes(self._h, _p(g, _i32p), C.byref(n0)))
        return g, n0.value

    def schwarz_coarse(self):
        """(cells per direction, K0^-1) of the coarse level"""
        g = np.zeros(3, dtype=np.int32)
        n0 = C.c_int64()
        _chk(self._L.fedd_schwarz_coarse_sizes(self._h, _p(g, _i32p), C.byref(n0)))
        K = np.empty((n0.value, n0.value), dtype=np.float64)
        _chk(self._L.fedd_schwarz_coarse_get(self._h, _p(K, _f64p)))
        return g, K

    def schwarz_info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(self._L.fedd_schwarz_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        u = C.c_int64()
        _chk(self._L.fedd_schwarz_unique(self._h, C.byref(u)))
        out = dict(n_subdomains=a.value, max_size=b.value, inverse_bytes=c.value, n_unique=u.value)
        if a.value and b.value <= 256:      # (the large-subdomain path keeps no per-box lists)
            ss, so = C.c_int64(), C.c_int64()
            if self._L.fedd_schwarz_sizes(self._h, C.byref(ss), C.byref(so)) == 0:
                out.update(sum_sizes=ss.value, sum_owned=so.value)
            nc = C.c_int64()
            if self._L.fedd_schwarz_conforming(self._h, C.byref(nc)) == 0:
                out.update(n_conforming=nc.value)
        return out

    def schwarz_apply(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        _chk(self._L.fedd_schwarz_apply(self._h, _p(r, _f64p), _p(z, _f64p)))
        return z

    def schwarz_apply_device(self, reps):
        _chk(self._L.fedd_schwarz_apply_device(self._h, reps))

    def gmres(self, b=None, rtol=1e-8, max_it=100, restart=100, use_prec=True, want_x=True):
        bb = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
        nr = self.csr_sizes()[0]
        x = np.zeros(nr) if want_x else None
        its, rel = C.c_int(), C.c_double()
        _chk(self._L.fedd_gmres(self._h, _p(bb, _f64p), _p(x, _f64p), rtol, max_it, restart, int(use_prec),
                                C.byref(its), C.byref(rel)))
        return x, its.value, rel.value

    def gmres_x0(self, x0=None, b=None, rtol=1e-8, max_it=100, restart=100, use_prec=True):
        """the solve from an initial guess ("Zero Initial Guess" = false); x0 None: from the vector the device holds"""
        bb = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
        x = None if x0 is None else np.array(x0, dtype=np.float64, copy=True)
        its, rel = C.c_int(), C.c_double()
        _chk(self._L.fedd_gmres_x0(self._h, _p(bb, _f64p), _p(x, _f64p), rtol, max_it, restart, int(use_prec),
                                   C.byref(its), C.byref(rel)))
        return (x if x is not None else self.solution_get()), its.value, rel.value

    def mesh_setup_info(self):
        a, t, st, nt = C.c_double(), C.c_double(), C.c_int(), C.c_int64()
        _chk(self._L.fedd_mesh_setup_info(self._h, C.byref(a), C.byref(t), C.byref(st), C.byref(nt)))
        return {"adjacency_ms": a.value, "tiles_ms": t.value, "tiles_state": st.value, "n_tiles": nt.value}

    def gmres_status(self):
        fl, rr = C.c_int(), C.c_double()
        _chk(self._L.fedd_gmres_status(self._h, C.byref(fl), C.byref(rr)))
        return {"floor_reached": fl.value, "recurrence_relres": rr.value}

    def schwarz_coarse_apply(self, r=None):
        """z = Phi K0^-1 Phi^T r (FROSch's "Only apply coarse"); r None: the assembled rhs, z stays on the device as the
        solution vector (Level Combination = Multiplicative: then gmres_x0(None))"""
        if r is None:
            _chk(self._L.fedd_schwarz_coarse_apply(self._h, None, None))
            return None
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        _chk(self._L.fedd_schwarz_coarse_apply(self._h, _p(r, _f64p), _p(z, _f64p)))
        return z

    def set_option(self, key, value):
        _chk(self._L.fedd_set_option(self._h, key.encode(), float(value)))

    def timing_enable(self, on=True):
        _chk(self._L.fedd_timing_enable(self._h, int(on)))

    def timing_reset(self):
        _chk(self._L.fedd_timing_reset(self._h))

    def timing_get(self):
        out = {}
        for i, name in enumerate(TIMER_NAMES):
            ms, n = C.c_double(), C.c_int64()
            _chk(self._L.fedd_timing_get(self._h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def timing_get_sampled(self):
        """per class: (device ms, launches, algorithmic bytes) of the launches that were actually timed"""
        out = {}
        for i, name in enumerate(TIMER_NAMES):
            ms, n, b = C.c_double(), C.c_int64(), C.c_double()
            _chk(self._L.fedd_timing_get_sampled(self._h, i, C.byref(ms), C.byref(n), C.byref(b)))
            out[name] = (ms.value, n.value, b.value)
        return out

    def gmres_info(self):
        k, s, nb, nc, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _chk(self._L.fedd_gmres_info(self._h, C.byref(k), C.byref(s), C.byref(nb), C.byref(nc)))
        _chk(self._L.fedd_gmres_fused_blocks(self._h, C.byref(nf)))
        return {"kind": k.value, "s": s.value, "blocks": nb.value, "cut_blocks": nc.value, "fused_blocks": nf.value}

    def read_bandwidth(self, nbytes=2 << 30, reps=10):
        """GB/s of a read-only stream on this GPU (roofline calibration)."""
        g = C.c_double()
        _chk(self._L.fedd_read_bandwidth(self._h, int(nbytes), reps, C.byref(g)))
        return g.value

    # ---- halo plan ----
    def halo_set_owners(self, gid_rep, owner_rep):
        g = np.ascontiguousarray(gid_rep, dtype=np.int64)
        o = np.ascontiguousarray(owner_rep, dtype=np.int32)
        _chk(self._L.fedd_halo_set_owners(self._h, g.shape[0], _p(g, _i64p), _p(o, _i32p)))

    def halo_requests(self):
        cnt = np.zeros(self.nranks, dtype=np.int64)
        _chk(self._L.fedd_halo_requests_sizes(self._h, _p(cnt, _i64p)))
        g = np.zeros(int(cnt.sum()), dtype=np.int64)
        _chk(self._L.fedd_halo_requests_get(self._h, _p(g, _i64p)))
        return cnt, g

    def halo_requests_set(self, count_from_rank, gids):
        cnt = np.ascontiguousarray(count_from_rank, dtype=np.int64)
        g = np.ascontiguousarray(gids, dtype=np.int64)
        _chk(self._L.fedd_halo_requests_set(self._h, _p(cnt, _i64p), _p(g, _i64p)))

    def comm_set_torch_dist(self, dist):
        """Host-staged transport over an initialised torch.distributed group (gloo): functional
        tests of the N > 1 path where RCCL cannot run.  Also finalises the halo plan over it."""
        import torch

        def exchange(user, n_peers, peers, send_ptr, send_buf, recv_ptr, recv_buf, dofs):
            try:
                reqs, keep = [], []
                for k in range(n_peers):
                    s0, s1 = send_ptr[k] * dofs, send_ptr[k + 1] * dofs
                    r0, r1 = recv_ptr[k] * dofs, recv_ptr[k + 1] * dofs
                    if s1 > s0:
                        t = torch.from_numpy(np.ctypeslib.as_array(send_buf, shape=(send_ptr[n_peers] * dofs,))[s0:s1].copy())
                        keep.append(t)
                        reqs.append(dist.isend(t, int(peers[k])))
                    if r1 > r0:
                        t = torch.zeros(r1 - r0, dtype=torch.float64)
                        keep.append((t, r0, r1))
                        reqs.append(dist.irecv(t, int(peers[k])))
                for r in reqs:
                    r.wait()
                out = np.ctypeslib.as_array(recv_buf, shape=(max(1, recv_ptr[n_peers] * dofs),))
                for item in keep:
                    if isinstance(item, tuple):
                        out[item[1]:item[2]] = item[0].numpy()
                return 0
            except Exception as e:  # pragma: no cover
                print("exchange callback failed:", e, flush=True)
                return 1

        def allreduce(user, buf, n):
            try:
                a = np.ctypeslib.as_array(buf, shape=(n,))
                t = torch.from_numpy(a.copy())
                dist.all_reduce(t)
                a[:] = t.numpy()
                return 0
            except Exception as e:  # pragma: no cover
                print("allreduce callback failed:", e, flush=True)
                return 1

        self._cb = (EXCHANGE_FN(exchange), ALLREDUCE_FN(allreduce))     # keep alive
        _chk(self._L.fedd_comm_set_host_callbacks(self._h, C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None))
        # halo plan over the same group
        cnt, gids = self.halo_requests()
        world = self.nranks
        cnt_all = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(cnt_all, torch.from_numpy(cnt.copy()))
        from_me = np.array([int(cnt_all[p][self.rank]) for p in range(world)], dtype=np.int64)
        recv = [torch.zeros(int(n), dtype=torch.int64) for n in from_me]
        off = np.concatenate([[0], np.cumsum(cnt)])
        reqs = []
        for p in range(world):
            if p == self.rank:
                continue
            if cnt[p] > 0:
                reqs.append(dist.isend(torch.from_numpy(gids[off[p]:off[p + 1]].copy()), p))
            if from_me[p] > 0:
                reqs.append(dist.irecv(recv[p], p))
        for r in reqs:
            r.wait()
        self.halo_requests_set(from_me, np.concatenate([t.numpy() for t in recv]) if from_me.sum() else np.zeros(0, np.int64))

    def comm_set_thread_group(self, group):
        """Host-staged transport between the threads of a `ThreadGroup` (see there); finalises the halo
        plan over it.  Sums of the all-reduce are formed in rank order on every rank (same bits everywhere)."""
        rank = self.rank

        def exchange(user, n_peers, peers, send_ptr, send_buf, recv_ptr, recv_buf, dofs):
            try:
                if n_peers == 0:
                    return 0
                sb = np.ctypeslib.as_array(send_buf, shape=(max(1, send_ptr[n_peers] * dofs),))
                out = np.ctypeslib.as_array(recv_buf, shape=(max(1, recv_ptr[n_peers] * dofs),))
                for k in range(n_peers):
                    s0, s1 = send_ptr[k] * dofs, send_ptr[k + 1] * dofs
                    if s1 > s0:
                        group.send(rank, int(peers[k]), sb[s0:s1])
                for k in range(n_peers):
                    r0, r1 = recv_ptr[k] * dofs, recv_ptr[k + 1] * dofs
                    if r1 > r0:
                        out[r0:r1] = group.recv(int(peers[k]), rank)
                return 0
            except Exception as e:  # pragma: no cover
                print("exchange callback failed:", repr(e), flush=True)
                return 1

        def allreduce(user, buf, n):
            try:
                a = np.ctypeslib.as_array(buf, shape=(n,))
                parts = group.allgather(rank, a.copy())
                tot = parts[0].copy()
                for q in parts[1:]:
                    tot += q
                a[:] = tot
                return 0
            except Exception as e:  # pragma: no cover
                print("allreduce callback failed:", repr(e), flush=True)
                return 1

        self._cb = (EXCHANGE_FN(exchange), ALLREDUCE_FN(allreduce))     # keep alive
        _chk(self._L.fedd_comm_set_host_callbacks(self._h, C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None))
        cnt, gids = self.halo_requests()
        off = np.concatenate([[0], np.cumsum(cnt)])
        everyone = group.allgather(rank, (cnt.copy(), gids.copy(), off))
        from_me = np.array([int(everyone[p][0][rank]) for p in range(group.world)], dtype=np.int64)
        lists = [everyone[p][1][everyone[p][2][rank]:everyone[p][2][rank + 1]] for p in range(group.world)]
        self.halo_requests_set(from_me, np.concatenate(lists).astype(np.int64) if from_me.sum() else np.zeros(0, np.int64))

    def rccl_selftest(self, n=4096):
        e = C.c_double()
        _chk(self._L.fedd_rccl_selftest(self._h, n, C.byref(e)))
        return e.value

    def comm_selftest(self, n=4096):
        """the solver's collectives on this context's own communicator (every rank calls it); largest deviation"""
        e = C.c_double()
        _chk(self._L.fedd_comm_selftest(self._h, n, C.byref(e)))
        return e.value

    def halo_exchange_setup(self):
        _chk(self._L.fedd_halo_exchange_setup(self._h))

    def halo_plan(self):
        npeers, ns, nr = C.c_int(), C.c_int64(), C.c_int64()
        _chk(self._L.fedd_halo_plan_sizes(self._h, C.byref(npeers), C.byref(ns), C.byref(nr)))
        peers = np.zeros(npeers.value, dtype=np.int32)
        sp = np.zeros(npeers.value + 1, dtype=np.int64)
        rp = np.zeros(npeers.value + 1, dtype=np.int64)
        sl = np.zeros(ns.value, dtype=np.int32)
        rl = np.zeros(nr.value, dtype=np.int32)
        _chk(self._L.fedd_halo_plan_get(self._h, _p(peers, _i32p), _p(sp, _i64p), _p(sl, _i32p), _p(rp, _i64p),
                                        _p(rl, _i32p)))
        return dict(peers=peers, send_ptr=sp, send_lid=sl, recv_ptr=rp, recv_lid=rl)
