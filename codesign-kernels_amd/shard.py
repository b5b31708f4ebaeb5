"""ncrms sharding across the GPUs of one node.

No statement of the reference routine couples different CRM instances
(reference :505-637 index every array with the same `sl`), so the path shards
along ncrms with NO data-path collective: each rank advects its contiguous
block of instances.  The only communication is the optional scatter of inputs
from / gather of outputs to a root rank, done here with grouped point-to-point
torch.distributed ops (backend "nccl" = RCCL over xGMI on GPU tensors, "gloo"
on CPU tensors in the tests).  Because ncrms is the FASTEST axis, a shard is a
strided slab of every array; it is packed contiguous before sending.
"""
import torch
import torch.distributed as dist


def partition(ncrms, world_size, rank):
    """Contiguous block of rank: (sl0, nloc); remainders go to the low ranks."""
    base, rem = divmod(ncrms, world_size)
    nloc = base + (1 if rank < rem else 0)
    sl0 = rank * base + min(rank, rem)
    return sl0, nloc


def _pack(full, sl0, nloc):
    if full.is_cuda:
        from .capi import pack_shard
        return pack_shard(full, sl0, nloc)
    return full[..., sl0:sl0 + nloc].contiguous()


def _unpack(full, shard, sl0):
    if full.is_cuda:
        from .capi import unpack_shard
        unpack_shard(full, shard, sl0)
    else:
        full[..., sl0:sl0 + shard.shape[-1]] = shard


def scatter_inputs(full, ncrms, src=0, group=None, device=None, dtype=torch.float64):
    """Root holds `full` = dict name -> tensor (reversed-axes layout, last axis
    ncrms); every rank returns its shard dict (last axis nloc).  Non-root
    ranks pass a dict name -> shape-without-last-axis."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sl0, nloc = partition(ncrms, world, rank)
    out, ops, keep = {}, [], []
    names = sorted(full.keys())
    if rank == src:
        for name in names:
            for r in range(world):
                s0, n = partition(ncrms, world, r)
                piece = _pack(full[name], s0, n)
                if r == src:
                    out[name] = piece
                else:
                    keep.append(piece)
                    ops.append(dist.P2POp(dist.isend, piece, r, group))
    else:
        for name in names:
            lead = tuple(full[name])
            out[name] = torch.empty(lead + (nloc,), dtype=dtype, device=device)
            ops.append(dist.P2POp(dist.irecv, out[name], src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def gather_outputs(shards, full, ncrms, dst=0, group=None):
    """Inverse of scatter_inputs for the outputs (f, flux): root's `full`
    tensors receive every rank's block."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ops, recv = [], []
    names = sorted(shards.keys())
    if rank == dst:
        for name in names:
            for r in range(world):
                s0, n = partition(ncrms, world, r)
                if r == dst:
                    _unpack(full[name], shards[name], s0)
                else:
                    buf = torch.empty(full[name].shape[:-1] + (n,), dtype=full[name].dtype,
                                      device=full[name].device)
                    recv.append((name, s0, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, r, group))
    else:
        for name in names:
            ops.append(dist.P2POp(dist.isend, shards[name].contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for name, s0, buf in recv:
        _unpack(full[name], buf, s0)
