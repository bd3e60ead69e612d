"""The N > 1 path end to end on ONE GPU: 2 and 4 ranks (each its own process and fedd_ctx on
device 0), ghost-element meshes, halo plan, distributed assembly / SpMV / Schwarz / GMRES.  RCCL
refuses several ranks on one device, so the two communication steps go through the library's
host-staged transport hook over gloo; plan, pack/unpack kernels and solver flow are the production
ones.  The gathered solution must match a direct solve of the single-domain oracle system."""
import os
import socket

import numpy as np
import pytest

import scipy.sparse as sp

import fedd_oracle as fo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dec, M, ghosts, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from feddlib_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = capi.structured_mesh(3, dec, [M] * 3, rank, ghosts=ghosts)
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(m)
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [M] * 3, m["gid_rep"]))
        c.comm_set_torch_dist(dist)
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        # distributed SpMV against the global vector
        rng = np.random.default_rng(11)
        xg = rng.standard_normal(m["n_global"])
        y = c.spmv(xg[m["gid_uni"]])
        c.schwarz_set_target(27, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
        x, its, rel = c.gmres(None, rtol=1e-13, max_it=600, restart=100, use_prec=True)
        rowptr, col, val, gid = c.csr_get()
        rhs = c.rhs_get()
        # second level on top: K0^-1 (replicated), the coarse part of the apply, and a solve
        r = np.random.default_rng(12).standard_normal(m["n_global"])[m["gid_uni"]]
        z1 = c.schwarz_apply(r)
        c.schwarz_set_coarse(16)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)
        g, Kinv = c.schwarz_coarse()
        z2 = c.schwarz_apply(r)
        x2, its2, rel2 = c.gmres(None, rtol=1e-13, max_it=600, restart=100, use_prec=True)
        # the other combine modes run across ranks too (ghost contributions of a rank's subdomains are dropped)
        x_alt = {}
        for name, mode in (("averaging", capi.COMBINE_AVERAGING), ("full", capi.COMBINE_FULL)):
            c.schwarz_setup(1, mode)
            xa, ita, rela = c.gmres(None, rtol=1e-13, max_it=600, restart=100, use_prec=True)
            x_alt[name] = (xa, ita, rela)
        two = dict(g=g, Kinv=Kinv, zc=z2 - z1, x=x2, its=its2, rel=rel2, alt=x_alt)
        q.put((rank, m["gid_uni"], x, y, its, rel, rowptr, col, val, gid, rhs, two))
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dec,M,ghosts", [((1, 1, 2), 6, 1), ((1, 2, 2), 5, 1), ((1, 1, 2), 6, 2), ((1, 2, 2), 5, 2),
                                          ((1, 2, 2), 5, 4)])
def test_multirank_solve_on_one_gpu(fedd_lib, dec, M, ghosts):
    import scipy.sparse as sp
    import torch.multiprocessing as mp
    world = int(np.prod(dec))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dec, M, ghosts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = fedd_lib.structured_mesh(3, (1, 1, 1), [d * M for d in dec], 0)
    om = fo.Mesh(dim=3, fe="P1", conn=ref["conn"], xyz=ref["xyz"], gid_rep=ref["gid_rep"], flag_rep=ref["flag_rep"],
                 gid_uni=ref["gid_uni"], flag_uni=ref["flag_uni"], xyz_uni=None, n_global=ref["n_global"])
    A_bc, rhs_bc, _, _, flags = fo.laplace_problem(om)
    xd = fo.direct_solve(A_bc, rhs_bc)
    co = fo.CoarseQ1(A_bc, ref["xyz"], np.isin(flags, (1, 2, 3)), 1, cells_target=16)
    rc = np.random.default_rng(12).standard_normal(ref["n_global"])
    zc_ref = co.apply(rc)
    xx2 = np.zeros_like(xd)
    its2_all = set()
    xg = np.random.default_rng(11).standard_normal(ref["n_global"])
    yref = A_bc @ xg
    xx = np.zeros_like(xd)
    its_all = set()
    for rank, gu, x, y, its, rel, rowptr, col, val, gid, rhs, two in res:
        xx[gu] = x
        xx2[gu] = two["x"]
        its2_all.add(two["its"])
        assert two["rel"] <= 1e-13
        np.testing.assert_array_equal(two["g"], co.g)
        np.testing.assert_allclose(two["Kinv"], co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
        np.testing.assert_allclose(two["zc"], zc_ref[gu], rtol=0, atol=1e-10 * np.abs(zc_ref).max())
        its_all.add(its)
        assert rel <= 1e-13
        np.testing.assert_allclose(y, yref[gu], rtol=0, atol=1e-10 * np.abs(yref).max())
        # owned rows of the distributed matrix == the same rows of the global oracle matrix
        nr = rowptr.shape[0] - 1
        rows = np.repeat(np.arange(nr), np.diff(rowptr))
        Aloc = sp.csr_matrix((val, (gu[rows], gid[col])), shape=A_bc.shape)
        assert abs(Aloc[gu] - A_bc[gu]).max() <= 1e-10 * abs(A_bc).max()
        np.testing.assert_allclose(rhs, rhs_bc[gu], rtol=0, atol=1e-14)
    assert len(its_all) == 1                      # every rank took the same convergence decision
    np.testing.assert_allclose(xx, xd, rtol=0, atol=1e-10 * np.abs(xd).max())
    assert len(its2_all) == 1
    np.testing.assert_allclose(xx2, xd, rtol=0, atol=1e-10 * np.abs(xd).max())
    for name in ("averaging", "full"):
        xa = np.zeros_like(xd)
        its_a = set()
        for rank, gu, x, y, its, rel, rowptr, col, val, gid, rhs, two in res:
            xa[gu] = two["alt"][name][0]
            its_a.add(two["alt"][name][1])
            assert two["alt"][name][2] <= 1e-13
        assert len(its_a) == 1
        np.testing.assert_allclose(xa, xd, rtol=0, atol=1e-10 * np.abs(xd).max())


def _worker_elasticity(rank, world, port, dec, M, q, ghosts=4):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from feddlib_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = capi.structured_mesh(3, dec, [M] * 3, rank, ghosts=ghosts)
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(m)
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [M] * 3, m["gid_rep"]))
        c.comm_set_torch_dist(dist)
        mu, nu = 1.0, 0.3
        lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
        c.pattern_build(3, capi.BLOCK_FULL)
        c.assemble(capi.FORM_LINELAS, [lam, mu])
        c.assemble_rhs([0.0, 1.0, 0.0])
        c.dirichlet([2], [0.0, 0.0, 0.0])
        c.schwarz_set_target(8, 1.0)
        c.schwarz_set_coarse(8)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)
        g, Kinv = c.schwarz_coarse()
        x, its, rel = c.gmres(None, rtol=1e-13, max_it=800, restart=200, use_prec=True)
        q.put((rank, m["gid_uni"], x, its, rel, g, Kinv))
        c.close()
    finally:
        dist.destroy_process_group()


def test_multirank_elasticity_two_level(fedd_lib):
    """cfg 5 in miniature: 3 dofs per node, FULL blocks, ghost halo with 3 values per node, two ranks,
    coarse level replicated; solution and K0^-1 against the single-domain oracle."""
    import torch.multiprocessing as mp
    dec, M = (1, 1, 2), 4
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_elasticity, args=(r, world, port, dec, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = fedd_lib.structured_mesh(3, (1, 1, 1), [d * M for d in dec], 0)
    om = fo.Mesh(dim=3, fe="P1", conn=ref["conn"], xyz=ref["xyz"], gid_rep=ref["gid_rep"], flag_rep=ref["flag_rep"],
                 gid_uni=ref["gid_uni"], flag_uni=ref["flag_uni"], xyz_uni=None, n_global=ref["n_global"])
    A_bc, rhs_bc, _, _, flags = fo.linelas_problem(om, 1.0, 0.3)
    xd = fo.direct_solve(A_bc, rhs_bc)
    co = fo.CoarseQ1(A_bc, ref["xyz"], np.repeat(np.isin(flags, (2,)), 3), 3, cells_target=8)
    xx = np.zeros_like(xd)
    its_all = set()
    for rank, gu, x, its, rel, g, Kinv in res:
        gd = (3 * gu[:, None] + np.arange(3)[None, :]).ravel()
        xx[gd] = x
        its_all.add(its)
        assert rel <= 1e-13
        np.testing.assert_array_equal(g, co.g)
        np.testing.assert_allclose(Kinv, co.K0inv, rtol=0, atol=1e-10 * np.abs(co.K0inv).max())
    assert len(its_all) == 1
    np.testing.assert_allclose(xx, xd, rtol=0, atol=1e-9 * np.abs(xd).max())


def test_bench_contract_n2_rehearsal(fedd_lib):
    """bench.py's N > 1 code path (decomposition, ghost meshes, halo plan, barriers, max over ranks, one
    JSON line from rank 0) with two ranks on one GPU over gloo (`--rehearse-one-gpu`)."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "0", "--cells", "12", "--rehearse-one-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["dofs"] == 13 * 13 * 25 and d["config"]["relres"] <= 1e-8
    assert d["two_level_variant"]["gmres_iterations"] > 0


def test_bench_contract_n2_fixed_grid_rehearsal(fedd_lib):
    """The default mode of bench.py (one global grid split over the ranks, "scaling": "strong"), at a small
    grid: 24^3 cells as 1x1x2 blocks of 24x24x12."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "0", "--global-cells", "24", "--rehearse-one-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["self_check"]["halo_spmv_max_rel_err"] <= 1e-12 and d["self_check"]["true_relres"] <= 1e-7
    assert d["config"]["dofs"] == 25 ** 3 and d["config"]["relres"] <= 1e-8
    assert "1x1x2 blocks of 24x24x12 cells" in d["config"]["workload"]


def test_bench_contract_one_gpu(fedd_lib):
    """`python bench.py` end to end at a reduced grid: one JSON line with every contract key, the roofline and
    cpu_baseline objects, and the configs[1] extra."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--global-cells", "20"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["dofs"] == 21 ** 3
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"]
    assert d["roofline"]["traffic"] is None              # PMC passes are of the full grid only
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"]
    assert d["cfg2_one_gpu"]["gmres_iterations"] > 0 and "1030301 dofs" in d["cfg2_one_gpu"]["workload"]


def _thread_rank(capi, group, rank, dec, M, out, errs):
    try:
        world = group.world
        m = capi.structured_mesh(3, dec, [M] * 3, rank, ghosts=4)      # four ghost layers: what bench.py uses
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(m)
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [M] * 3, m["gid_rep"]))
        c.comm_set_thread_group(group)
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        xg = np.random.default_rng(11).standard_normal(m["n_global"])
        y = c.spmv(xg[m["gid_uni"]])
        c.schwarz_set_target(27, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
        x1, its1, rel1 = c.gmres(None, rtol=1e-13, max_it=600, restart=100, use_prec=True)
        c.schwarz_set_coarse(27)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)
        x2, its2, rel2 = c.gmres(None, rtol=1e-13, max_it=600, restart=100, use_prec=True)
        out[rank] = dict(gu=m["gid_uni"], y=y, x1=x1, its1=its1, rel1=rel1, x2=x2, its2=its2, rel2=rel2, rhs=c.rhs_get(),
                         plan=c.halo_plan())
        c.close()
    except Exception as e:      # pragma: no cover
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        try:
            group._barrier.abort()
        except Exception:
            pass


def test_eight_ranks_2x2x2_in_one_process(fedd_lib):
    """The reference's own decomposition of the headline (N = 2: 2 x 2 x 2 blocks, the 8-GPU case) with all eight
    ranks as threads of this process on one GPU (a one-GPU box admits 6 processes): nodes shared by 2, 4 and 8
    blocks, peers across faces, edges and the centre vertex.  Distributed SpMV, one- and two-level solves
    against the single-domain oracle system."""
    import threading
    dec, M = (2, 2, 2), 4
    group = fedd_lib.ThreadGroup(8)
    out, errs = [None] * 8, []
    th = [threading.Thread(target=_thread_rank, args=(fedd_lib, group, r, dec, M, out, errs)) for r in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, "\n".join(errs)
    assert all(o is not None for o in out)
    ref = fedd_lib.structured_mesh(3, (1, 1, 1), [2 * M] * 3, 0)
    om = fo.Mesh(dim=3, fe="P1", conn=ref["conn"], xyz=ref["xyz"], gid_rep=ref["gid_rep"], flag_rep=ref["flag_rep"],
                 gid_uni=ref["gid_uni"], flag_uni=ref["flag_uni"], xyz_uni=None, n_global=ref["n_global"])
    A_bc, rhs_bc, _, _, _ = fo.laplace_problem(om)
    xd = fo.direct_solve(A_bc, rhs_bc)
    yref = A_bc @ np.random.default_rng(11).standard_normal(ref["n_global"])
    x1, x2 = np.zeros_like(xd), np.zeros_like(xd)
    covered = np.zeros(ref["n_global"], dtype=int)
    for o in out:
        covered[o["gu"]] += 1
        x1[o["gu"]] = o["x1"]
        x2[o["gu"]] = o["x2"]
        np.testing.assert_allclose(o["y"], yref[o["gu"]], rtol=0, atol=1e-10 * np.abs(yref).max())
        np.testing.assert_allclose(o["rhs"], rhs_bc[o["gu"]], rtol=0, atol=1e-14)
        assert o["rel1"] <= 1e-13 and o["rel2"] <= 1e-13
    assert (covered == 1).all()                                   # every dof owned exactly once
    assert len({o["its1"] for o in out}) == 1 and len({o["its2"] for o in out}) == 1
    assert max(len(o["plan"]["peers"]) for o in out) == 7               # the block at the centre vertex talks to all 7 others
    np.testing.assert_allclose(x1, xd, rtol=0, atol=1e-10 * np.abs(xd).max())
    np.testing.assert_allclose(x2, xd, rtol=0, atol=1e-10 * np.abs(xd).max())


def test_row_ghosts_make_the_two_rank_schwarz_apply_equal_to_the_one_rank_apply(fedd_lib):
    """12^3 cells as 1x1x2 blocks with 8-node boxes: the lattice of boxes (one lattice over all ranks' nodes)
    has a box boundary between the node planes 6 and 7, i.e. exactly at the rank boundary, so both runs have
    the same subdomains.  With row ghosts (two ghost layers, fedd_mesh_set_rows) the local matrices of the
    subdomains at the rank boundary hold the true rows of the other rank's overlap nodes and M^-1 r is the
    one-rank result to rounding; with the single ghost layer those rows are identity rows and it is not."""
    import threading
    capi = fedd_lib
    G, dec, target = 12, (1, 1, 2), 8

    def setup(c):
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        c.schwarz_set_target(target, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)

    ref = capi.structured_mesh(3, 1, G)
    c0 = capi.Context(device=0)
    c0.mesh_set_dict(ref)
    setup(c0)
    r = np.random.default_rng(3).standard_normal(ref["n_global"])
    z_ref = c0.schwarz_apply(r)
    n_ref = c0.schwarz_info()["n_subdomains"]
    c0.close()
    diff = {}
    for ghosts in (1, 2):
        group = capi.ThreadGroup(2)
        out, errs = [None, None], []

        def rank_main(rank):
            try:
                m = capi.structured_mesh(3, dec, [G, G, G // 2], rank, ghosts=ghosts)
                c = capi.Context(device=0, rank=rank, nranks=2, nccl_id=None)
                c.mesh_set_dict(m)
                c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [G, G, G // 2], m["gid_rep"]))
                c.comm_set_thread_group(group)
                setup(c)
                out[rank] = (m["gid_uni"], c.schwarz_apply(r[m["gid_uni"]]), c.schwarz_info()["n_subdomains"])
                c.close()
            except Exception as e:      # pragma: no cover
                errs.append(repr(e))
                group._barrier.abort()

        th = [threading.Thread(target=rank_main, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        assert not errs, errs
        assert out[0][2] + out[1][2] == n_ref          # the same boxes
        z = np.zeros_like(z_ref)
        for gu, zz, _ in out:
            z[gu] = zz
        diff[ghosts] = np.abs(z - z_ref).max() / np.abs(z_ref).max()
    assert diff[2] <= 1e-13
    assert diff[1] > 1e-3


@pytest.mark.parametrize("G,dec,target,layers,problem", [(12, (2, 2, 2), 27, 4, "laplace"), (16, (1, 2, 2), 27, 4, "laplace"),
                                                         (12, (2, 2, 2), 8, 3, "laplace"), (30, (3, 2), 16, 5, "laplace"),
                                                         (24, (2, 2, 2), 64, 5, "laplace"),       # bench.py's boxes: 4 planes + 1

                                                         (8, (1, 2, 2), 0, 4, "linelas")])
def test_whole_boxes_make_the_preconditioner_independent_of_the_number_of_ranks(fedd_lib, G, dec, target, layers, problem):
    """With enough ghost layers (boxes of 3 node planes + 1 of overlap: 4 element layers; 2-plane boxes: 3) every
    rank builds every box that holds one of its nodes whole, rank boundaries or not, so the subdomains are those of
    a one-rank run: M^-1 r equal to rounding and the same GMRES iteration count, on decompositions whose rank
    boundaries cut through the boxes."""
    import threading
    capi = fedd_lib

    dofs = 3 if problem == "linelas" else 1

    def setup(c):
        if problem == "linelas":       # 3 dofs per node, FULL blocks, Dirichlet on the face x = 0 only, two levels
            c.pattern_build(3, capi.BLOCK_FULL)
            c.assemble(capi.FORM_LINELAS, [1.0 * 2 * 0.3 / (1 - 2 * 0.3), 1.0])
            c.assemble_rhs([0.0, 1.0, 0.0])
            c.dirichlet([2], [0.0, 0.0, 0.0])
            c.schwarz_set_target(target, 1.0)
            c.schwarz_set_coarse(8)
            c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)
            return
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        c.schwarz_set_target(target, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)

    dim = len(dec)
    ref = capi.structured_mesh(dim, 1, G)
    c0 = capi.Context(device=0)
    c0.mesh_set_dict(ref)
    setup(c0)
    r = np.random.default_rng(3).standard_normal(ref["n_global"] * dofs)
    z_ref = c0.schwarz_apply(r)
    x_ref, its_ref, _ = c0.gmres(None, rtol=1e-10, max_it=500, restart=100, use_prec=True)
    c0.close()
    world = int(np.prod(dec))
    cells = [G // d for d in dec]
    group = capi.ThreadGroup(world)
    out, errs = [None] * world, []

    def rank_main(rank):
        try:
            m = capi.structured_mesh(dim, dec, cells, rank, ghosts=layers)
            c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
            c.mesh_set_dict(m)
            c.halo_set_owners(m["gid_rep"], capi.structured_owner(dim, dec, cells, m["gid_rep"]))
            c.comm_set_thread_group(group)
            setup(c)
            gd = (m["gid_uni"][:, None] * dofs + np.arange(dofs)[None, :]).ravel()     # global dof ids, node-interleaved
            z = c.schwarz_apply(r[gd])
            x, its, _ = c.gmres(None, rtol=1e-10, max_it=500, restart=100, use_prec=True)
            out[rank] = (gd, z, x, its)
            c.close()
        except Exception as e:      # pragma: no cover
            errs.append(repr(e))
            group._barrier.abort()

    th = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    z, x = np.zeros_like(z_ref), np.zeros_like(x_ref)
    for gu, zz, xx, its in out:
        z[gu] = zz
        x[gu] = xx
        assert its == its_ref
    np.testing.assert_allclose(z, z_ref, rtol=0, atol=1e-13 * np.abs(z_ref).max())
    np.testing.assert_allclose(x, x_ref, rtol=0, atol=1e-9 * np.abs(x_ref).max())


def test_bench_contract_n4_fixed_grid_rehearsal(fedd_lib):
    """bench.py's 1x2x2 split of a fixed grid (four ranks on one GPU over gloo), with the communication self-check."""
    import json
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1",
           "--warmup", "0", "--global-cells", "32", "--rehearse-one-gpu", "--no-two-level"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["dofs"] == 33 ** 3
    assert "1x2x2 blocks of 32x16x16 cells" in d["config"]["workload"]
    assert d["self_check"]["halo_spmv_max_rel_err"] <= 1e-12 and d["self_check"]["true_relres"] <= 1e-7
    assert "halo" in d["phases_device_ms_per_step"] and "allreduce" in d["phases_device_ms_per_step"]


def _thread_rank_gdsw(capi, group, rank, dec, M, out, errs):
    try:
        world = group.world
        m = capi.structured_mesh(3, dec, [M] * 3, rank, ghosts=4)
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(m)
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [M] * 3, m["gid_rep"]))
        c.comm_set_thread_group(group)
        mu, nu = 2.0e6, 0.4
        lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
        c.pattern_build(3, capi.BLOCK_FULL)
        c.assemble(capi.FORM_LINELAS, [lam, mu])
        c.assemble_rhs([0.0, 1.0, 0.0])
        c.dirichlet([2], [0.0, 0.0, 0.0])
        c.schwarz_set_coarse(8)                 # coarse decomposition = the 2 x 2 x 2 rank blocks
        c.set_option("gdsw_tol", 1e-13)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_GDSW)
        g, Kinv = c.schwarz_coarse()
        x, its, rel = c.gmres(None, rtol=1e-12, max_it=600, restart=100, use_prec=True)
        # ... and with the rotations in the null space (their Gram matrices are summed over the ranks before the selection)
        c.set_option("gdsw_rotations", 1)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_GDSW)
        _, Kinv_rot = c.schwarz_coarse()
        x_rot, its_rot, rel_rot = c.gmres(None, rtol=1e-12, max_it=600, restart=100, use_prec=True)
        out[rank] = dict(gu=m["gid_uni"], g=g, Kinv=Kinv, x=x, its=its, rel=rel, Kinv_rot=Kinv_rot, x_rot=x_rot, its_rot=its_rot)
        c.close()
    except Exception as e:      # pragma: no cover
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        try:
            group._barrier.abort()
        except Exception:
            pass


def test_cfg5_miniature_gdsw_on_eight_ranks(fedd_lib):
    """cfg 5 in miniature with the coarse space the config names: 3D P1 linear elasticity (steadyLinElas_Perf
    parameters), 2 x 2 x 2 ranks (threads of this process), two-level Schwarz with the GDSW coarse level whose coarse
    decomposition is the rank decomposition (8 cells: 1 vertex + 6 edges + 12 faces = 19 interface components x 3
    translations).  K0^-1 (all-reduced Galerkin product, extensions solved by the distributed constrained GMRES) and
    the solution against the single-domain oracle; same coarse matrix and iteration count on every rank; and the
    same iteration count as the one-rank run of the same problem."""
    import threading
    dec, M = (2, 2, 2), 4
    group = fedd_lib.ThreadGroup(8)
    out, errs = [None] * 8, []
    th = [threading.Thread(target=_thread_rank_gdsw, args=(fedd_lib, group, r, dec, M, out, errs)) for r in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    assert not errs, "\n".join(errs)
    assert all(o is not None for o in out)
    ref = fedd_lib.structured_mesh(3, (1, 1, 1), [2 * M] * 3, 0)
    om = fo.Mesh(dim=3, fe="P1", conn=ref["conn"], xyz=ref["xyz"], gid_rep=ref["gid_rep"], flag_rep=ref["flag_rep"],
                 gid_uni=ref["gid_uni"], flag_uni=ref["flag_uni"], xyz_uni=None, n_global=ref["n_global"])
    A_bc, rhs_bc, _, _, flags = fo.linelas_problem(om, 2.0e6, 0.4)
    is_dir = fo.dirichlet_rows(flags, (2,), dofs=3)
    co = fo.CoarseGDSW(A_bc, ref["conn"], ref["xyz"], is_dir, 3, cells_target=8)
    assert co.n_interface_entities == 19 and co.n0 == 27 * 3
    xd = fo.direct_solve(A_bc, rhs_bc)
    x = np.zeros_like(xd)
    for o in out:
        np.testing.assert_array_equal(o["g"], [2, 2, 2])
        np.testing.assert_allclose(o["Kinv"], co.K0inv, rtol=0, atol=1e-8 * np.abs(co.K0inv).max())
        d = (3 * o["gu"][:, None] + np.arange(3)[None, :]).ravel()
        x[d] = o["x"]
        assert o["rel"] <= 1e-12
    assert len({o["its"] for o in out}) == 1
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-9 * np.abs(xd).max())
    co_rot = fo.CoarseGDSW(A_bc, ref["conn"], ref["xyz"], is_dir, 3, cells_target=8, rotations=True)
    assert co_rot.n0 == 27 * 6 and int(co_rot.kept.sum()) == 1 * 3 + 6 * 5 + 12 * 6     # the vertex, 6 straight edges, 12 faces
    x_rot = np.zeros_like(xd)
    for o in out:
        np.testing.assert_allclose(o["Kinv_rot"], co_rot.K0inv, rtol=0, atol=1e-8 * np.abs(co_rot.K0inv).max())
        x_rot[(3 * o["gu"][:, None] + np.arange(3)[None, :]).ravel()] = o["x_rot"]
    assert len({o["its_rot"] for o in out}) == 1 and out[0]["its_rot"] < out[0]["its"]
    np.testing.assert_allclose(x_rot, xd, rtol=0, atol=1e-9 * np.abs(xd).max())
    # one rank, same problem, same coarse decomposition: the same preconditioner (whole boxes + row ghosts make the first
    # level independent of the number of ranks; the coarse level is defined on the global lattice)
    c = fedd_lib.Context(device=0)
    try:
        c.mesh_set_dict(ref)
        mu, nu = 2.0e6, 0.4
        c.pattern_build(3, fedd_lib.BLOCK_FULL)
        c.assemble(fedd_lib.FORM_LINELAS, [2.0 * mu * nu / (1.0 - 2.0 * nu), mu])
        c.assemble_rhs([0.0, 1.0, 0.0])
        c.dirichlet([2], [0.0, 0.0, 0.0])
        c.schwarz_set_coarse(8)
        c.set_option("gdsw_tol", 1e-13)
        c.schwarz_setup(1, fedd_lib.COMBINE_RESTRICTED, two_level=1, coarse_kind=fedd_lib.COARSE_GDSW)
        _, its1, _ = c.gmres(None, rtol=1e-12, max_it=600, restart=100, use_prec=True)
    finally:
        c.close()
    assert abs(out[0]["its"] - its1) <= 1


def _thread_rank_unstructured(capi, group, rank, m, part, layers, out, errs):
    try:
        world = group.world
        pm = capi.partitioned_mesh(m, part, world, rank, ghosts=layers)
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(pm)
        c.halo_set_owners(pm["gid_rep"], pm["owner_rep"])
        c.comm_set_thread_group(group)
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 4], [0.0, 0.0, 0.0])
        rowptr, col, val, gid = c.csr_get()
        xg = np.random.default_rng(13).standard_normal(m["xyz"].shape[0])
        y = c.spmv(xg[pm["gid_uni"]])
        c.schwarz_set_target(8, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
        x, its, rel = c.gmres(None, rtol=1e-13, max_it=800, restart=200, use_prec=True)
        out[rank] = dict(gu=pm["gid_uni"], x=x, its=its, rel=rel, y=y, rhs=c.rhs_get(), rowptr=rowptr, col=col, val=val, gid=gid)
        c.close()
    except Exception as e:      # pragma: no cover
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        try:
            group._barrier.abort()
        except Exception:
            pass


@pytest.mark.parametrize("world,layers", [(2, 2), (4, 2), (4, 1)])
def test_partitioned_unstructured_mesh_on_several_ranks(fedd_lib, world, layers):
    """The reference's DFG3DCylinder_1k.mesh split by the element partitioner (mesh_partition.cpp: what
    MeshPartitioner::readAndPartitionMesh + METIS do in the reference) over 2 and 4 ranks (threads, host-staged
    transport): distributed assembly of the owned rows from the ghost-element layers, halo exchange, SpMV and the
    Schwarz-preconditioned solve against the one-rank oracle of the whole mesh."""
    import threading
    GOLD = os.path.join(ROOT, "tests", "golden")
    m = fedd_lib.read_mesh(os.path.join(GOLD, "DFG3DCylinder_1k.mesh"), 3)
    part = fedd_lib.partition_mesh(m, world)
    group = fedd_lib.ThreadGroup(world)
    out, errs = [None] * world, []
    th = [threading.Thread(target=_thread_rank_unstructured, args=(fedd_lib, group, r, m, part, layers, out, errs)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, "\n".join(errs)
    om = fo.Mesh(dim=3, fe="P1", conn=m["conn"], xyz=m["xyz"], gid_rep=m["gid_rep"], flag_rep=m["flag_rep"],
                 gid_uni=m["gid_uni"], flag_uni=m["flag_uni"], xyz_uni=None, n_global=m["n_global"])
    A_bc, rhs_bc, _, _, _ = fo.laplace_problem(om, bc_flags=(1, 2, 4))
    xd = fo.direct_solve(A_bc, rhs_bc)
    yref = A_bc @ np.random.default_rng(13).standard_normal(m["xyz"].shape[0])
    x = np.zeros_like(xd)
    covered = np.zeros(xd.shape[0], dtype=int)
    for o in out:
        covered[o["gu"]] += 1
        x[o["gu"]] = o["x"]
        n_own = o["gu"].shape[0]
        # owned rows of the distributed matrix = the same rows of the one-rank matrix
        Aloc = sp.csr_matrix((o["val"], o["gid"][o["col"]], o["rowptr"]), shape=(n_own, xd.shape[0]))
        assert abs(Aloc - A_bc[o["gu"]]).max() <= 1e-12 * abs(A_bc).max()
        np.testing.assert_allclose(o["rhs"], rhs_bc[o["gu"]], rtol=0, atol=1e-14 * np.abs(rhs_bc).max() + 1e-300)
        np.testing.assert_allclose(o["y"], yref[o["gu"]], rtol=0, atol=1e-11 * np.abs(yref).max())
        assert o["rel"] <= 1e-13
    assert (covered == 1).all()
    assert len({o["its"] for o in out}) == 1
    np.testing.assert_allclose(x, xd, rtol=0, atol=1e-10 * np.abs(xd).max())


def test_rccl_call_shapes_on_one_rank(fedd_lib):
    """What can run of the RCCL path on a one-GPU box: a one-rank communicator on the library's stream, the grouped
    ncclSend / ncclRecv of the halo import, the in-place f64 all-reduce of the Gram-Schmidt reductions and the int64
    all-gather of the plan setup, against the expected values."""
    c = fedd_lib.Context(device=0)
    try:
        assert c.rccl_selftest(100000) == 0.0
    finally:
        c.close()


def _thread_rank_overlap(capi, group, rank, dec, M, out, errs):
    try:
        world = group.world
        m = capi.structured_mesh(3, dec, [M] * 3, rank, ghosts=4)
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(m)
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [M] * 3, m["gid_rep"]))
        c.comm_set_thread_group(group)
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        rg = np.random.default_rng(5).standard_normal(m["n_global"])
        res = {}
        for dedupe, kind in ((1, 4), (1, 0), (0, 0), (1, 40)):   # matrix-core kernel, flat kernel on shared slabs, flat kernel; the
            # last: the matrix-core kernel with the SpMV on row classes (spmv_pattern 2 forces the dictionary on these small
            # matrices) -- with halo_overlap the ghost import of the SpMV travels on the second stream as well and the rows
            # that read ghost columns are redone when it has arrived (k_spmv_rows)
            c.set_option("spmv_pattern", 2 if kind == 40 else 1)
            c.set_option("spmv_classes_cover", 50 if kind == 40 else 90)     # (a fifth of these small blocks' rows read ghost columns)
            for ov in (0, 1):
                c.set_option("schwarz_dedupe", dedupe)
                c.set_option("apply_kind", kind % 10 if kind == 40 else kind)
                c.set_option("halo_overlap", ov)
                c.schwarz_set_target(27, 1.0)
                c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
                z = c.schwarz_apply(rg[m["gid_uni"]])
                x, its, rel = c.gmres(None, rtol=1e-12, max_it=400, restart=100, use_prec=True)
                res[(dedupe, kind, ov)] = (z, x, its)
                if kind == 40:
                    res[("spmv", ov)] = (c.spmv_info(), c.spmv(rg[m["gid_uni"]]))
        out[rank] = dict(gu=m["gid_uni"], res=res)
        c.close()
    except Exception as e:      # pragma: no cover
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        try:
            group._barrier.abort()
        except Exception:
            pass


def test_interior_first_order_with_the_ghost_import_on_a_second_stream(fedd_lib):
    """Option halo_overlap: the subdomains without ghost dofs are applied while the ghost entries of r are imported on a
    second stream, the others afterwards.  Every entry of z is still produced by one subdomain with the same arithmetic:
    the operator is the same bit for bit, whichever apply kernel runs, and so are the GMRES iterates."""
    import threading
    dec, M = (1, 2, 2), 8
    group = fedd_lib.ThreadGroup(4)
    out, errs = [None] * 4, []
    th = [threading.Thread(target=_thread_rank_overlap, args=(fedd_lib, group, r, dec, M, out, errs)) for r in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, "\n".join(errs)
    for o in out:
        assert o is not None
        for dedupe, kind in ((1, 4), (1, 0), (0, 0), (1, 40)):
            z0, x0, its0 = o["res"][(dedupe, kind, 0)]
            z1, x1, its1 = o["res"][(dedupe, kind, 1)]
            assert np.array_equal(z0, z1), (dedupe, kind)
            assert its0 == its1 and np.array_equal(x0, x1), (dedupe, kind)
        # the class SpMV ran (every rank's rows repeat) and the solve on it is the solve on the per-entry stream, bit for bit
        assert o["res"][("spmv", 1)][0]["row_classes"] >= 1
        assert o["res"][(1, 40, 1)][2] == o["res"][(1, 4, 1)][2] and np.array_equal(o["res"][(1, 40, 1)][1], o["res"][(1, 4, 1)][1])
        # ... and the three kernels agree with each other to rounding
        za, zb = o["res"][(1, 4, 1)][0], o["res"][(0, 0, 1)][0]
        np.testing.assert_allclose(za, zb, rtol=0, atol=1e-11 * np.abs(zb).max())


def _thread_rank_patterns(capi, group, rank, dec, M, out, errs):
    try:
        world = group.world
        m = capi.structured_mesh(3, dec, [M] * 3, rank, ghosts=4)
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(m)
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [M] * 3, m["gid_rep"]))
        c.comm_set_thread_group(group)
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        xg = np.random.default_rng(21).standard_normal(m["n_global"])
        c.set_option("spmv_exact_public", 0)     # fedd_spmv = the solver's stream in this test
        c.set_option("spmv_pattern", 0)
        y0 = c.spmv(xg[m["gid_uni"]])
        c.set_option("spmv_pattern", 2)
        y1 = c.spmv(xg[m["gid_uni"]])
        out[rank] = dict(y0=y0, y1=y1, info=c.spmv_info())
        c.close()
    except Exception as e:      # pragma: no cover
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        try:
            group._barrier.abort()
        except Exception:
            pass


def test_spmv_column_patterns_with_ghost_columns(fedd_lib):
    """The column patterns of the SpMV stream on several ranks: the offsets of rows at a rank boundary reach into the
    ghost part of the column space (other offsets, more patterns, or explicit columns) -- the product stays the
    per-entry kernel's bit for bit."""
    import threading
    dec, M = (1, 2, 2), 10
    group = fedd_lib.ThreadGroup(4)
    out, errs = [None] * 4, []
    th = [threading.Thread(target=_thread_rank_patterns, args=(fedd_lib, group, r, dec, M, out, errs)) for r in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, "\n".join(errs)
    used = 0
    for o in out:
        assert o is not None and np.array_equal(o["y0"], o["y1"])
        used += o["info"]["column_patterns"] > 0
    assert used >= 1        # (at least one rank's rows repeat enough for the dictionary to be used)



def _thread_rank_at_size(capi, group, rank, dec, cells, problem, target, ghosts, out, errs):
    """one rank of a headline configuration in its own decomposition, at full size: assemble, solve to the XML's tolerance,
    return the rank's pieces of ||b - A x||^2 and ||b||^2 (b - A x formed with fedd_spmv = the parity CSR + a halo import)"""
    try:
        world = group.world
        m = capi.structured_mesh(3, dec, cells, rank, ghosts=ghosts)
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(m)
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, cells, m["gid_rep"]))
        c.comm_set_thread_group(group)
        n_own = m["gid_uni"].shape[0]
        del m
        if problem == "laplace":
            nnz = c.pattern_build(1, capi.BLOCK_SCALAR)
            c.assemble(capi.FORM_LAPLACE)
            c.assemble_rhs([1.0])
            c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
            c.schwarz_set_target(target, 1.0)
            c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
            rtol = 1e-8
        else:
            mu, nu = 2.0e6, 0.4
            lam = 2.0 * mu * nu / (1.0 - 2.0 * nu)
            nnz = c.pattern_build(3, capi.BLOCK_FULL)
            c.assemble(capi.FORM_LINELAS, [lam, mu])
            c.assemble_rhs([0.0, 1.0, 0.0])
            c.dirichlet([2], [0.0, 0.0, 0.0])
            c.schwarz_set_target(target, 1.0)
            c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=1, coarse_kind=capi.COARSE_Q1)
            rtol = 1e-6
        x, its, rel = c.gmres(None, rtol=rtol, max_it=2000, restart=100, use_prec=True)
        b = c.rhs_get()
        r = b - c.spmv(x)
        out[rank] = dict(n_own=n_own, nnz=nnz, its=its, rel=rel, rr=float(r @ r), bb=float(b @ b), xmax=float(np.abs(x).max()),
                         peers=len(c.halo_plan()["peers"]))
        c.close()
    except Exception as e:      # pragma: no cover
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        try:
            group._barrier.abort()
        except Exception:
            pass


@pytest.mark.parametrize("problem,M,target,ghosts,n_nodes,its_lo,its_hi",
                         [("laplace", 107, 64, 5, 215 ** 3, 140, 150),      # BASELINE cfg 3 as the reference decomposes it
                          ("linelas", 94, 8, 3, 189 ** 3, 20, 400)])         # BASELINE cfg 5 (Q1 coarse space)
def test_headline_configs_in_their_own_decomposition_at_size(fedd_lib, problem, M, target, ghosts, n_nodes, its_lo, its_hi):
    """cfg 3 (214^3 cells) and cfg 5 (188^3 cells, 20 253 807 dofs) as 2 x 2 x 2 blocks, the eight ranks as threads on one
    GPU (host-staged transport; a one-GPU box admits no eight processes): every dof owned once, the same iteration count
    on every rank, and the gathered TRUE residual of the distributed solve below the XML's tolerance."""
    import threading
    dec = (2, 2, 2)
    group = fedd_lib.ThreadGroup(8, timeout=900.0)
    out, errs = [None] * 8, []
    th = [threading.Thread(target=_thread_rank_at_size, args=(fedd_lib, group, r, dec, [M] * 3, problem, target, ghosts, out, errs))
          for r in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=1100)
    assert not errs, "\n".join(errs)
    assert all(o is not None for o in out)
    assert sum(o["n_own"] for o in out) == n_nodes
    assert len({o["its"] for o in out}) == 1 and its_lo <= out[0]["its"] <= its_hi, [o["its"] for o in out]
    rtol = 1e-8 if problem == "laplace" else 1e-6
    true_rel = (sum(o["rr"] for o in out) / sum(o["bb"] for o in out)) ** 0.5
    assert true_rel <= rtol and all(o["rel"] <= rtol for o in out), (true_rel, [o["rel"] for o in out])
    assert max(o["peers"] for o in out) == 7
    if problem == "laplace":
        assert sum(o["nnz"] for o in out) == 147968803                     # the reference's nnz of cfg 3
        assert abs(max(o["xmax"] for o in out) - 0.05621) < 1e-4           # centre value of -lap u = 1 on the unit cube
    else:
        assert sum(o["nnz"] for o in out) == 903725973                     # ... and of cfg 5


def _thread_rank_cut(capi, group, rank, dec, M, out, errs):
    try:
        world = group.world
        m = capi.structured_mesh(3, dec, [M] * 3, rank, ghosts=4)
        c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
        c.mesh_set_dict(m)
        c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [M] * 3, m["gid_rep"]))
        c.comm_set_thread_group(group)
        selftest = c.comm_selftest(257)          # the all-reduce leg over the host-staged transport (no RCCL communicator here)
        c.pattern_build(1, capi.BLOCK_SCALAR)
        c.assemble(capi.FORM_LAPLACE)
        c.assemble_rhs([1.0])
        c.dirichlet([1, 2, 3], [0.0, 0.0, 0.0])
        c.schwarz_set_target(27, 1.0)
        c.schwarz_setup(1, capi.COMBINE_RESTRICTED)
        res = {}
        # fused: three sweeps per block (k_blockfuse) and the solution update folded into the last update of every restart
        # cycle (restart 20 = 8 + 8 + 4 columns, no preconditioner: several cycles), the reductions between them all-reduced
        for name, opts, rtol, restart, prec in (("cut", {"gmres_s": 8, "gmres_chol_tol": 0.5}, 1e-10, 50, True),
                                                ("claims", {"gmres_s": 16, "gmres_tol_blocks": 0}, 1e-13, 50, True),
                                                ("spec", {"gmres_s": 8, "gmres_spec": 2}, 1e-10, 50, True),
                                                ("fused", {"gmres_s": 16, "gmres_fuse": 1}, 1e-10, 20, False)):
            for k, v in opts.items():
                c.set_option(k, v)
            x, its, rel = c.gmres(None, rtol=rtol, max_it=600, restart=restart, use_prec=prec)
            res[name] = dict(x=x, its=its, rel=rel, info=c.gmres_info(), status=c.gmres_status())
            c.set_option("gmres_chol_tol", 1e-13)
            c.set_option("gmres_tol_blocks", 1)
            c.set_option("gmres_spec", 0)
            c.set_option("gmres_s", 0)
            c.set_option("gmres_fuse", -1)
        out[rank] = dict(gu=m["gid_uni"], res=res, selftest=selftest)
        c.close()
    except Exception as e:      # pragma: no cover
        import traceback
        errs.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
        try:
            group._barrier.abort()
        except Exception:
            pass


def test_cut_blocks_and_failed_claims_keep_the_ranks_in_step(fedd_lib):
    """the s-step solver's host decisions (block cuts, convergence claims that the true residual does not confirm,
    restarts, operator applications issued ahead of a block's outcome) are taken by every rank from all-reduced numbers.  Four
    ranks, three solves that force those paths -- an absurd Cholesky threshold (every block cut), 16-vector blocks at 1e-13
    (claims fail, restarts follow), speculative applications -- must finish (matched collectives), agree on every count, and
    end at the one-rank solution."""
    import threading
    dec, M = (1, 2, 2), 6
    group = fedd_lib.ThreadGroup(4)
    out, errs = [None] * 4, []
    th = [threading.Thread(target=_thread_rank_cut, args=(fedd_lib, group, r, dec, M, out, errs)) for r in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, "\n".join(errs)
    assert all(o is not None for o in out)
    ref = fedd_lib.structured_mesh(3, (1, 1, 1), [M, 2 * M, 2 * M], 0)
    om = fo.Mesh(dim=3, fe="P1", conn=ref["conn"], xyz=ref["xyz"], gid_rep=ref["gid_rep"], flag_rep=ref["flag_rep"],
                 gid_uni=ref["gid_uni"], flag_uni=ref["flag_uni"], xyz_uni=None, n_global=ref["n_global"])
    A_bc, rhs_bc, _, _, _ = fo.laplace_problem(om)
    xd = fo.direct_solve(A_bc, rhs_bc)
    assert max(o["selftest"] for o in out) <= 1e-9
    for name, tol in (("cut", 1e-7), ("claims", 1e-10), ("spec", 1e-7), ("fused", 1e-7)):
        assert len({o["res"][name]["its"] for o in out}) == 1, name
        assert len({o["res"][name]["rel"] for o in out}) == 1, name
        assert len({(o["res"][name]["info"]["blocks"], o["res"][name]["info"]["cut_blocks"]) for o in out}) == 1, name
        x = np.zeros_like(xd)
        for o in out:
            x[o["gu"]] = o["res"][name]["x"]
        tr = np.linalg.norm(rhs_bc - A_bc @ x) / np.linalg.norm(rhs_bc)
        assert abs(out[0]["res"][name]["rel"] - tr) <= 0.05 * tr + 1e-16, (name, out[0]["res"][name]["rel"], tr)
        np.testing.assert_allclose(x, xd, rtol=0, atol=tol * np.abs(xd).max())
    assert out[0]["res"]["cut"]["info"]["cut_blocks"] >= 1
    fi = out[0]["res"]["fused"]
    assert fi["info"]["fused_blocks"] == fi["info"]["blocks"] >= 4 and fi["its"] > 40, fi["info"]
