"""Multi-GPU sharding of a batch of independent alignments.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  Problems are independent
([REF roman/align/submap_align.py:93-200] carries no state across iterations), so each rank aligns a
contiguous shard of the flattened pair list and ONE all_gather of fixed-size result records
collects the inlier sets and poses.  No other collective.
"""
import numpy as np

from .batch import AlignmentBatch, pack_records, run_batch, unpack_records


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced partition: the first n%world ranks get one extra item."""
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def align_sharded(registration, batch: AlignmentBatch, group=None, compute=None, device=None):
    """Align `batch` across the ranks of `group` (default: WORLD); every rank returns the full result
    (assoc list, T, status) in problem order.

    compute(registration, sub_batch) -> runtime.BatchResult defaults to the HIP path (run_batch).
    `device`: torch device for the gathered tensors (cuda:<local rank> under RCCL, cpu under gloo).
    """
    import torch
    import torch.distributed as dist
    compute = compute or run_batch
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = len(batch)
    kmax = batch.kmax()
    lo, hi = shard_bounds(B, rank, world)
    res = compute(registration, batch.subset(lo, hi))
    ints, poses = pack_records(res, kmax)
    if world == 1:
        return unpack_records(ints, poses, registration.dim)
    # pad every shard to the largest shard so all_gather sees equal shapes
    per = (B + world - 1) // world
    ints_p = np.full((per, ints.shape[1]), -1, dtype=np.int32); ints_p[:ints.shape[0]] = ints
    poses_p = np.full((per, 16), np.nan, dtype=np.float64); poses_p[:poses.shape[0]] = poses
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    ti = torch.from_numpy(ints_p).to(dev); tp = torch.from_numpy(poses_p).to(dev)
    gi = torch.empty((world * per, ti.shape[1]), dtype=ti.dtype, device=dev)    # concatenated along dim 0
    gp = torch.empty((world * per, tp.shape[1]), dtype=tp.dtype, device=dev)
    dist.all_gather_into_tensor(gi, ti, group=group)
    dist.all_gather_into_tensor(gp, tp, group=group)
    gi = gi.cpu().numpy().reshape(world, per, -1); gp = gp.cpu().numpy().reshape(world, per, -1)
    rows_i, rows_p = [], []
    for r in range(world):
        rlo, rhi = shard_bounds(B, r, world)
        rows_i.append(gi[r, :rhi - rlo]); rows_p.append(gp[r, :rhi - rlo])
    return unpack_records(np.concatenate(rows_i, axis=0), np.concatenate(rows_p, axis=0), registration.dim)
