"""Multi-GPU sharding of a shooting sweep: one process per GPU, segments partitioned, defect all-gathered.

Segment i needs only nodes i and i+1 (src/multiShoot_CRTBP_indirect.jl:71-86, src/multiShoot_CRTBP_direct.jl:77-105),
so a sweep shards into contiguous blocks of segments with a one-node halo on the input side and NO data-path
exchange during propagation.  The only collective is one all-gather of the per-rank defect slabs (RCCL over
xGMI when the process group's backend is "nccl"; gloo on CPU in the tests) so that every rank holds the full
defect vector for the convergence test / line-search decision (indirect.jl:240,331).  STM / Jacobian blocks stay
sharded (their consumer is block-structured).

The `sweep` callable is the propagation itself: the product passes a HIP-backed function
(`hip_indirect_defect` below).  There is no CPU fallback here; the gloo tests inject their own callable.
"""
import numpy as np


def partition(n_items, world, rank):
    """Contiguous block partition; the first (n_items % world) ranks get one extra item.  Returns (start, count)."""
    base, rem = divmod(int(n_items), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def local_nodes(nodes, t, world, rank):
    """Slice [ndim x n_nodes] node array and time grid for this rank's block of segments, including the
    one-node halo: segments [start, start+count) need nodes [start, start+count]."""
    n_seg = nodes.shape[1] - 1
    start, count = partition(n_seg, world, rank)
    return nodes[:, start:start + count + 1], t[start:start + count + 1], start, count


def all_gather_defect(local_defect, n_seg_total, world, rank, group=None, comm=None):
    """All-gather ragged per-rank defect slabs [ndim x count_r] into the full [ndim x n_seg_total] on every rank.

    Slabs are padded to the largest block so that ONE collective moves everything (it is latency-bound: 12 x 4096 doubles
    per rank at BASELINE configs[1]).  comm = a hotpath.Comm: the library's own RCCL all-gather (lto_comm_allgather_dev,
    device to device on torch's current stream); otherwise torch.distributed's all_gather_into_tensor on `group` (RCCL
    when its backend is "nccl", gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    ndim = local_defect.shape[0]
    cmax = partition(n_seg_total, world, 0)[1]
    pad = torch.zeros(ndim, cmax, dtype=local_defect.dtype, device=local_defect.device)
    pad[:, :local_defect.shape[1]] = local_defect
    out = torch.empty(world, ndim, cmax, dtype=local_defect.dtype, device=local_defect.device)
    if comm is not None:
        from .hotpath import current_stream_ptr
        comm.allgather(pad, out, ndim * cmax, stream=current_stream_ptr())
    elif world > 1:
        # concatenation form [world*ndim, cmax] <- [ndim, cmax]: accepted by both RCCL and gloo
        dist.all_gather_into_tensor(out.view(world * ndim, cmax), pad.contiguous(), group=group)
    else:
        out[0] = pad
    full = torch.empty(ndim, n_seg_total, dtype=local_defect.dtype, device=local_defect.device)
    for r in range(world):
        s, c = partition(n_seg_total, world, r)
        full[:, s:s + c] = out[r, :, :c]
    return full


def sharded_defect(sweep, nodes, t, world, rank, group=None, device=None, comm=None):
    """Run `sweep(local_nodes [ndim x (count+1)], local_t [count+1]) -> defect [ndim x count]` (numpy in/out or
    torch in/out) on this rank's block and all-gather.  Returns the full defect as a torch tensor."""
    import torch
    n_seg = nodes.shape[1] - 1
    ln, lt, start, count = local_nodes(nodes, t, world, rank)
    if count > 0:
        d = sweep(ln, lt)
    else:
        d = np.zeros((nodes.shape[0], 0))
    if not isinstance(d, torch.Tensor):
        d = torch.from_numpy(np.ascontiguousarray(d))
    if device is not None:
        d = d.to(device)
    return all_gather_defect(d, n_seg, world, rank, group, comm=comm)


def partition_batch(n_batch, world, rank):
    """Homotopy levels / line-search trial points shard by whole trajectories (BASELINE configs[3]:
    256 levels -> 32 per GPU)."""
    return partition(n_batch, world, rank)


def hip_indirect_defect(ctx, params, integ):
    """The product's sweep callable: host-pointer indirect defectCalc on this rank's GPU."""
    from . import hotpath

    def sweep(ln, lt):
        d, _ = hotpath.indirect_defectCalc(np.asfortranarray(ln), np.ascontiguousarray(lt), params, integ, ctx=ctx)
        return d
    return sweep
