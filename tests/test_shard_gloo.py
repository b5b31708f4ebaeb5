"""CPU test of the multi-GPU path's host logic: world_size-2 `gloo` processes
partition ncrms, scatter the strided shards from rank 0, advect their block and
gather f/flux back.  On the GPU box the same code runs with backend "nccl"
(RCCL) and the HIP kernel; here the per-rank compute is the CPU oracle (this is
a test, so it may call oracle/)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ncrms, nx, nz, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import codesign_kernels_amd as M
        from oracle import oracle as O
        names = ("adz", "f", "u", "w", "rho", "rhow", "flux")
        sh = M.shapes(ncrms, nx, nz)
        if rank == 0:
            inp = O.make_inputs(ncrms, nx, nz, seed=42, dist=1)
            full = {k: torch.from_numpy(np.ascontiguousarray(inp[k].T)) for k in names}
            arg = full
        else:
            arg = {k: sh[k][:-1] for k in names}
        mine = M.scatter_inputs(arg, ncrms, src=0)
        s0, n = M.partition(ncrms, world, rank)
        # every rank's shard equals the generator evaluated for its block
        loc = O.make_inputs(n, nx, nz, seed=42, dist=1, ncrms_global=ncrms, sl0=s0)
        for k in names:
            assert np.array_equal(mine[k].numpy().T, loc[k]), k
        f, flux = O.advect(loc)
        out = {"f": torch.from_numpy(np.ascontiguousarray(f.T)),
               "flux": torch.from_numpy(np.ascontiguousarray(flux.T))}
        tgt = {"f": full["f"].clone(), "flux": full["flux"].clone()} if rank == 0 else None
        M.gather_outputs(out, tgt, ncrms, dst=0)
        if rank == 0:
            f_ref, flux_ref = O.advect(inp)
            ok = (np.array_equal(tgt["f"].numpy().T, f_ref)
                  and np.array_equal(tgt["flux"].numpy().T, flux_ref))
            q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    """a port the OS hands out (bind to 0), instead of one derived from the pid"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("ncrms", [10, 7])
def test_scatter_advect_gather_world2(ncrms):
    from oracle import oracle as O
    O.build_lib()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ncrms, 8, 6, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_partition_covers_everything():
    import codesign_kernels_amd as M
    for ncrms in (1, 7, 64, 65536, 524288):
        for world in (1, 2, 3, 4, 8):
            blocks = [M.partition(ncrms, world, r) for r in range(world)]
            assert blocks[0][0] == 0
            for (a, n), (b, _) in zip(blocks, blocks[1:]):
                assert a + n == b
            assert blocks[-1][0] + blocks[-1][1] == ncrms
