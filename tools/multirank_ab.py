"""2 / 4 ranks on one GPU (gloo transport): iteration counts with and without ghost dofs in the subdomains."""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def worker(rank, world, port, dec, M, q):
    import torch.distributed as dist
    from feddlib_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = capi.structured_mesh(3, dec, [M] * 3, rank, ghosts=True)
    c = capi.Context(device=0, rank=rank, nranks=world, nccl_id=None)
    c.mesh_set_dict(m)
    c.halo_set_owners(m["gid_rep"], capi.structured_owner(3, dec, [M] * 3, m["gid_rep"]))
    c.comm_set_torch_dist(dist)
    c.pattern_build(1, capi.BLOCK_SCALAR); c.assemble(capi.FORM_LAPLACE); c.assemble_rhs([1.0]); c.dirichlet([1, 2, 3], [0., 0., 0.])
    out = {}
    for go in (1, 0):
        c.set_option("ghost_overlap", go)
        for two in (0, 1):
            c.schwarz_setup(1, capi.COMBINE_RESTRICTED, two_level=two, coarse_kind=capi.COARSE_Q1 if two else 0)
            x, its, rel = c.gmres(None, rtol=1e-8, max_it=500, restart=100, use_prec=True)
            out[(go, two)] = its
    if rank == 0: q.put(out)
    c.close(); dist.destroy_process_group()

if __name__ == "__main__":
    import torch.multiprocessing as mp
    for dec, M in (((1, 1, 2), 24), ((1, 2, 2), 20)):
        world = int(np.prod(dec)); ctx = mp.get_context("spawn"); q = ctx.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ps = [ctx.Process(target=worker, args=(r, world, port, dec, M, q)) for r in range(world)]
        [p.start() for p in ps]; res = q.get(timeout=300); [p.join() for p in ps]
        print(dec, M, "iterations {(ghost_overlap, two_level): its}:", res, flush=True)
