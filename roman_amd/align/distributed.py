"""Multi-GPU sharding of a batch of independent alignments.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  Problems are independent
([REF roman/align/submap_align.py:93-200] carries no state across iterations), so every rank aligns its share of the
flattened pair list and ONE all_gather of fixed-size result records collects the inlier sets and poses.  No other
collective (one byte record per problem: int32 part and pose travel together).  The share is cost-balanced: problems are dealt longest-first (work estimate = the square of the number of
associations to score, n1*n2 for all-to-all) to the rank with the least work so far — the deal is a pure function of the
batch, identical on every rank.  On GPUs the records never leave the device before the gather.
"""
import numpy as np

from .batch import AlignmentBatch, pack_records, record_bytes, records_from_bytes, run_batch, unpack_records


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced partition: the first n%world ranks get one extra item."""
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def problem_costs(batch: AlignmentBatch):
    """Associations each problem scores (the O(A^2) build and the solver both grow with it)."""
    if batch.assoc_off is not None:
        a = np.diff(batch.assoc_off).astype(np.int64)
        full = batch.n1.astype(np.int64) * batch.n2.astype(np.int64)
        return np.where(a > 0, a, full)                       # an empty list means all-to-all
    return batch.n1.astype(np.int64) * batch.n2.astype(np.int64)


def problem_work(batch: AlignmentBatch):
    """Work estimate of each problem for the deal: the SQUARE of its association count.  The affinity build tests pairs of
    live associations and the matrix it leaves — the solver's stream — holds a fraction of them: both grow with A^2, not with A
    (at equal sizes, the all-pairs grid of BASELINE config 4, the two deals coincide; on submaps of 50 ... 300 objects a deal
    on A leaves the rank that drew the large pairs with up to 1.5x the mean A^2)."""
    a = problem_costs(batch).astype(np.int64)
    return a * a


def deal_by_cost(costs, world_size):
    """Longest-processing-time deal -> list of ascending index arrays, one per rank.  Deterministic: ties in cost keep
    problem order, ties in load go to the lowest rank."""
    costs = np.asarray(costs, dtype=np.int64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    count = np.zeros(world_size, dtype=np.int64)
    shards = [[] for _ in range(world_size)]
    for i in order:
        # lightest rank; among equally light ones the one holding fewer problems, then the lowest rank
        r = int(np.lexsort((np.arange(world_size), count, load))[0])
        shards[r].append(int(i)); load[r] += max(int(costs[i]), 1); count[r] += 1
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def take(batch: AlignmentBatch, idx) -> AlignmentBatch:
    """The problems `idx` of a batch over the same feature pool."""
    idx = np.asarray(idx, dtype=np.int64)
    a, ao = None, None
    if batch.assoc is not None:
        lens = np.diff(batch.assoc_off)[idx]
        ao = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        a = np.concatenate([batch.assoc[batch.assoc_off[i]:batch.assoc_off[i + 1]] for i in idx], axis=0) if len(idx) else np.zeros((0, 2), np.int32)
    return AlignmentBatch(batch.feats, batch.off1[idx], batch.n1[idx], batch.off2[idx], batch.n2[idx], a, ao,
                          None if batch.pair_index is None else batch.pair_index[idx])


def upload_pool(batch: AlignmentBatch, dev):
    """The feature pool of a batch as a device tensor.  A caller that aligns several batches over ONE pool (the chunks of an
    all-pairs grid) uploads it once and hands it to every align_sharded call (`pool=`); nothing is cached behind the
    caller's back: a pool refilled in place on the host is uploaded again by the next call that is not given a tensor."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(batch.feats, dtype=np.float64)).to(dev)


def _same_device(a, b):
    """torch devices compared by type and index (an index-less device means the current one)."""
    import torch
    norm = lambda d: (d.type, d.index if d.index is not None else (torch.cuda.current_device() if d.type == "cuda" else 0))
    return norm(torch.device(a)) == norm(torch.device(b))


def _device_records(registration, sub: AlignmentBatch, kmax, per, dev, pool=None, chunk=None, in_flight=3):
    """Align `sub` with device-resident inputs and outputs; -> (ints (per, 2+2*kmax) int32, poses (per,16) f64) torch
    tensors on `dev`, rows beyond len(sub) padded with -1 / NaN.  The problems go out as calls of `chunk` problems with
    `in_flight` of them on the device at once (pipeline.issue_chunked: a rank's share of a large grid in ONE call would last
    as long as its slowest problem); problems the speculatively sized workspace skipped (ROMAN_ST_WORKSPACE) are issued
    again — those only; what is still skipped after the attempts, or came back ROMAN_ST_INTERNAL (no further attempts then),
    keeps the flag in its record (align_sharded raises AFTER the gather, on every rank alike — an exception on one rank in
    front of a collective would leave the others waiting in it).  The feature pool is uploaded once per call (or taken from `pool`)."""
    import torch
    from .pipeline import issue_chunked
    ctx = registration._context()
    B = len(sub)
    ints = torch.full((per, 2 + 2 * kmax), -1, dtype=torch.int32, device=dev)
    poses = torch.full((per, 16), float("nan"), dtype=torch.float64, device=dev)
    if B == 0:
        return ints, poses
    P = registration._abi_params()
    if pool is not None and (tuple(pool.shape) != tuple(sub.feats.shape) or str(pool.dtype) != "torch.float64" or not _same_device(pool.device, dev)):
        raise ValueError(f"pool must be the batch's feature matrix {sub.feats.shape} as a float64 tensor on {dev}")
    feats = pool if pool is not None else upload_pool(sub, dev)
    a_out = torch.full((B, kmax, 2), -1, dtype=torch.int32, device=dev)
    n_out = torch.zeros(B, dtype=torch.int32, device=dev)
    T_out = torch.zeros((B, 16), dtype=torch.float64, device=dev)
    st_out = torch.zeros(B, dtype=torch.int32, device=dev)
    assoc = None if sub.assoc is None else torch.from_numpy(np.ascontiguousarray(sub.assoc, dtype=np.int32)).to(dev)
    if torch.device(dev).type == "cuda":
        torch.cuda.current_stream(dev).synchronize()           # inputs are in place before the library's streams read them
    issue_chunked(ctx, P, feats, sub, kmax, a_out, n_out, T_out, st_out, None, assoc, chunk, in_flight)
    ints[:B, 0] = n_out; ints[:B, 1] = st_out
    valid = torch.arange(kmax, device=dev)[None, :] < n_out[:, None]
    ints[:B, 2:] = torch.where(valid[:, :, None], a_out, torch.full_like(a_out, -1)).reshape(B, -1)
    poses[:B] = T_out
    return ints, poses


def align_sharded(registration, batch: AlignmentBatch, group=None, compute=None, device=None, pool=None, chunk=None, in_flight=3):
    """Align `batch` across the ranks of `group` (default: WORLD); every rank returns the full result
    (assoc list, T, status) in problem order.  A rank's share goes to its GPU as calls of `chunk` problems (default:
    pipeline.default_chunk — one call up to 512 problems, two up to 4096, calls of 2048 beyond) with `in_flight` of them on
    the device at once.

    compute(registration, sub_batch) -> runtime.BatchResult: a CPU double for tests; by default the HIP path runs with
    device-resident records (`device`: torch device of this rank, default cuda:<current device>; `pool`: the batch's
    feature matrix already resident there, see upload_pool()).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = len(batch)
    kmax = batch.kmax()
    shards = deal_by_cost(problem_work(batch), world)
    mine = shards[rank]
    per = max(1, max(len(s) for s in shards))                  # every rank gathers equal shapes
    sub = take(batch, mine)
    if compute is None and world == 1 and not torch.cuda.is_available():
        compute = run_batch                                    # no torch device memory to hold the records: the host-pointer entry
    on_device = compute is None
    if on_device:
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        ti, tp = _device_records(registration, sub, kmax, per, dev, pool=pool, chunk=chunk, in_flight=in_flight)
    else:
        res = compute(registration, sub)
        ints, poses = pack_records(res, kmax)
        ints_p = np.full((per, ints.shape[1]), -1, dtype=np.int32); ints_p[:ints.shape[0]] = ints
        poses_p = np.full((per, 16), np.nan, dtype=np.float64); poses_p[:poses.shape[0]] = poses
        dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                                 if dist.is_initialized() and dist.get_backend(group) == "nccl" else torch.device("cpu"))
        ti = torch.from_numpy(ints_p).to(dev); tp = torch.from_numpy(poses_p).to(dev)
    # ONE fixed-size byte record per problem — [count, status, index pairs] int32 | pose f64 (batch.record_bytes) — and ONE
    # all_gather_into_tensor; the shards are scattered back into problem order with one indexed store
    ib = 8 * (1 + kmax)
    rec = torch.empty((per, record_bytes(kmax)), dtype=torch.uint8, device=ti.device)
    rec[:, :ib].view(torch.int32).copy_(ti); rec[:, ib:].view(torch.float64).copy_(tp)
    if world > 1:
        g = torch.empty((world * per, rec.shape[1]), dtype=torch.uint8, device=rec.device)     # concatenated along dim 0
        dist.all_gather_into_tensor(g, rec, group=group)
    else:
        g = rec
    gi, gp = records_from_bytes(g.cpu().numpy(), kmax)
    src = np.concatenate([r * per + np.arange(len(shards[r]), dtype=np.int64) for r in range(world)]) if B else np.zeros(0, np.int64)
    dst = np.concatenate(shards) if B else np.zeros(0, np.int64)
    out_i = np.full((B, gi.shape[1]), -1, dtype=np.int32); out_p = np.full((B, 16), np.nan)
    out_i[dst] = gi[src]; out_p[dst] = gp[src]
    assoc, T, status = unpack_records(out_i, out_p, registration.dim)
    check_records(status)
    return assoc, T, status


def check_records(status):
    """Raise for results that are NOT results: a problem the library gave up on (ROMAN_ST_INTERNAL: a bounded wait at a grid
    barrier of the whole-device solver expired — the record holds no associations and a NaN pose) or one that found no
    workspace after the retries (ROMAN_ST_WORKSPACE).  Called on the GATHERED records, i.e. on every rank alike."""
    from .. import _abi
    status = np.asarray(status)
    bad = np.nonzero((status & (_abi.ROMAN_ST_INTERNAL | _abi.ROMAN_ST_WORKSPACE)) != 0)[0]
    if len(bad):
        n_int = int(np.count_nonzero(status[bad] & _abi.ROMAN_ST_INTERNAL))
        raise _abi.RomanHipError(f"{len(bad)} problem(s) without a result ({n_int} ROMAN_ST_INTERNAL, {len(bad) - n_int} ROMAN_ST_WORKSPACE "
                                 f"after the retries): problems {bad[:8].tolist()}")
