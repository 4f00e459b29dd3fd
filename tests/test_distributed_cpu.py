"""N>1 path on CPU: world_size-2 `gloo` run of roman_amd.align.distributed.align_sharded with the
oracle as the compute function (the HIP path needs a GPU).  Checks that sharding + one all_gather
of fixed-size records reproduces the serial result in problem order."""
import os
import socket

import numpy as np
import pytest

from conftest import registration_for
from roman_amd import synth
from roman_amd.align import batch as rb
from roman_amd.align.distributed import align_sharded, deal_by_cost, problem_costs, problem_work, shard_bounds, take
from roman_amd.runtime import BatchResult, stats_dtype


def oracle_compute(registration, sub):
    """BatchResult of a sub-batch computed by the CPU oracle (test double for run_batch)."""
    from oracle import oracle as orc
    P = registration._abi_params()
    d = registration.dim
    assoc, Ts, status = [], [], []
    for b in range(len(sub)):
        D1 = sub.feats[sub.off1[b]:sub.off1[b] + sub.n1[b]]; D2 = sub.feats[sub.off2[b]:sub.off2[b] + sub.n2[b]]
        if len(D1) == 0 or len(D2) == 0:
            assoc.append(np.zeros((0, 2), np.int32)); Ts.append(np.full((d + 1, d + 1), np.nan)); status.append(3); continue
        a = orc.register(P, D1, D2)["assoc"]
        assoc.append(a)
        if len(a) >= d:
            Ts.append(orc.t_align(D1[a[:, 0], :d], D2[a[:, 1], :d], d)); status.append(0)
        else:
            Ts.append(np.full((d + 1, d + 1), np.nan)); status.append(2)
    return BatchResult(assoc, np.array(Ts).reshape(-1, d + 1, d + 1), np.array(status, np.int32), np.zeros(len(sub), stats_dtype()))


def make_batch(reg):
    subs, _ = synth.make_submap_grid(4, n=24, d=0, seed0=40)
    subs[3] = []                                             # an empty submap: ragged + sentinel path
    return rb.batch_from_submap_grid(reg, subs[:2], subs[2:])      # 2x2 = 4 problems... plus ragged sizes


def make_grid_batch(reg):
    """A small all-pairs grid (config 4's shape: S x S submaps of two robots over one shared pool) with ragged sizes."""
    subs, _ = synth.make_submap_grid(6, n=20, d=0, seed0=43)
    subs[1] = subs[1][:11]; subs[4] = subs[4][:15]
    return rb.batch_from_submap_grid(reg, subs[:3], subs[3:])      # 3 x 3 = 9 problems dealt to 2 ranks


def _worker(rank, world, port, q, which="pairs"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reg = registration_for("gravity")
        batch = make_batch(reg) if which == "pairs" else make_grid_batch(reg)
        assoc, T, status = align_sharded(reg, batch, compute=oracle_compute)
        q.put((rank, [a.tolist() for a in assoc], T.tolist(), status.tolist()))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_partition():
    for n in (0, 1, 5, 7, 64, 4096):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_cost_balanced_deal_is_a_partition_and_deterministic():
    rng = np.random.default_rng(3)
    for n, w in [(0, 2), (1, 4), (7, 2), (64, 8), (4096, 8), (4096, 3)]:
        costs = rng.integers(1, 40000, size=n)
        shards = deal_by_cost(costs, w)
        assert len(shards) == w and sorted(np.concatenate(shards).tolist() if n else []) == list(range(n))
        assert all(np.all(np.diff(s) > 0) for s in shards if len(s) > 1)          # ascending inside a shard
        again = deal_by_cost(costs.copy(), w)
        assert all(np.array_equal(a, b) for a, b in zip(shards, again))
        if n >= 8 * w:
            loads = np.array([costs[s].sum() for s in shards], dtype=np.float64)
            assert loads.max() <= 1.05 * loads.mean() + costs.max()                # longest-first keeps the ranks level
    # equal costs (the all-pairs grid of equal-sized submaps): every rank gets the same number of problems
    shards = deal_by_cost(np.full(4096, 40000), 8)
    assert [len(s) for s in shards] == [512] * 8


def test_deal_on_a_heterogeneous_grid_balances_the_quadratic_work():
    """Round-4 review, item 13: the deal must hold on a grid of UNEQUAL submaps (n, m in [50, 300]: association counts from
    2 500 to 90 000, work — pair tests, matrix entries — growing with their square).  Dealt on the work estimate A^2 every one
    of 8 ranks stays within 3 % of the mean work; the same problems dealt on A itself (the round-4 deal) leave the ranks level
    in A but far apart in A^2 — the quantity the GPU time follows."""
    rng = np.random.default_rng(12)
    S = 24
    sizes0, sizes1 = rng.integers(50, 301, size=S), rng.integers(50, 301, size=S)
    n1 = np.repeat(sizes0, S).astype(np.int32); n2 = np.tile(sizes1, S).astype(np.int32)
    off = np.zeros(S * S, dtype=np.int64)
    batch = rb.AlignmentBatch(np.zeros((1, 3)), off, n1, off, n2)
    A = problem_costs(batch); W = problem_work(batch)
    assert np.array_equal(A, n1.astype(np.int64) * n2) and np.array_equal(W, A * A)
    shards = deal_by_cost(W, 8)
    assert sorted(np.concatenate(shards).tolist()) == list(range(S * S))
    loads = np.array([W[s].sum() for s in shards], dtype=np.float64)
    assert loads.max() <= 1.03 * loads.mean() and loads.min() >= 0.97 * loads.mean()
    linear = deal_by_cost(A, 8)
    lin_loads = np.array([W[s].sum() for s in linear], dtype=np.float64)
    assert np.array([A[s].sum() for s in linear]).max() <= 1.03 * A.sum() / 8        # level in A ...
    assert lin_loads.max() / lin_loads.mean() > loads.max() / loads.mean()               # ... but less so in A^2 than the deal on A^2
    # every rank computes the same deal from the same batch (no communication): a second evaluation is identical
    assert all(np.array_equal(a, b) for a, b in zip(shards, deal_by_cost(problem_work(batch), 8)))


def test_the_c_abis_deal_is_the_python_deal():
    """roman_deal_problems (include/roman_hip.h: the deal for a C / C++ caller that shards with its own collective) is a pure
    host function — callable without a GPU — and hands every rank exactly the share align_sharded computes, on all-to-all
    batches of unequal submaps and on explicit association lists (an empty list counting as all-to-all)."""
    import ctypes as C
    from roman_amd import _abi
    lib = _abi.load_library()
    rng = np.random.default_rng(21)
    for trial in range(4):
        B = int(rng.integers(1, 400)); world = int(rng.choice([1, 2, 3, 8]))
        n1 = rng.integers(0, 301, size=B).astype(np.int32); n2 = rng.integers(0, 301, size=B).astype(np.int32)
        assoc_off = None
        if trial % 2:
            lens = rng.integers(0, 500, size=B); lens[rng.random(B) < 0.2] = 0
            assoc_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        batch = rb.AlignmentBatch(np.zeros((1, 3)), np.zeros(B, np.int64), n1, np.zeros(B, np.int64), n2,
                                  None if assoc_off is None else np.zeros((int(assoc_off[-1]), 2), np.int32), assoc_off)
        want = deal_by_cost(problem_work(batch), world)
        for rank in range(world):
            idx = np.full(B, -1, dtype=np.int32); n = C.c_int32(0)
            rc = lib.roman_deal_problems(B, n1.ctypes.data_as(C.c_void_p), n2.ctypes.data_as(C.c_void_p),
                                         None if assoc_off is None else assoc_off.ctypes.data_as(C.c_void_p), world, rank,
                                         idx.ctypes.data_as(C.c_void_p), C.byref(n))
            assert rc == 0 and np.array_equal(idx[:n.value], want[rank]), (trial, rank)
    assert lib.roman_deal_problems(4, None, None, None, 2, 0, None, None) != 0          # bad arguments are refused, not dereferenced


def test_take_keeps_problem_semantics():
    reg = registration_for("clipper+prune", cosine_min=0.5)
    pairs = [(p.map1, p.map2) for p in (synth.make_pair(12 + k, 10, 16, 300 + k) for k in range(5))]
    b = rb.batch_from_pairs(reg, pairs)
    assert b.assoc is not None
    sub = take(b, [4, 1, 2])
    assert sub.feats is b.feats and len(sub) == 3 and sub.assoc_off[-1] == len(sub.assoc)
    for k, i in enumerate([4, 1, 2]):
        assert np.array_equal(sub.assoc[sub.assoc_off[k]:sub.assoc_off[k + 1]], b.assoc[b.assoc_off[i]:b.assoc_off[i + 1]])
        assert (sub.off1[k], sub.n1[k], sub.off2[k], sub.n2[k]) == (b.off1[i], b.n1[i], b.off2[i], b.n2[i])
    assert np.array_equal(problem_costs(b), np.diff(b.assoc_off))


def test_record_roundtrip():
    res = BatchResult([np.array([[1, 2], [3, 4]], np.int32), np.zeros((0, 2), np.int32)],
                      np.stack([np.arange(16.0).reshape(4, 4), np.full((4, 4), np.nan)]), np.array([0, 2], np.int32), None)
    ints, poses = rb.pack_records(res, kmax=5)
    assert ints.shape == (2, 12) and poses.shape == (2, 16)
    assoc, T, status = rb.unpack_records(ints, poses, 3)
    assert assoc[0].tolist() == [[1, 2], [3, 4]] and assoc[1].shape == (0, 2)
    assert np.array_equal(T[0], res.T[0]) and np.all(np.isnan(T[1])) and status.tolist() == [0, 2]


def test_batch_layout_shares_the_feature_pool():
    reg = registration_for("gravity")
    subs, _ = synth.make_submap_grid(5, n=10, d=0, seed0=41)
    b = rb.batch_from_submap_grid(reg, subs[:2], subs[2:], mask=np.array([[1, 0, 1], [1, 1, 0]], bool))
    assert len(b) == 4 and b.feats.shape == (50, 3)
    assert b.pair_index.tolist() == [[0, 0], [0, 2], [1, 0], [1, 1]]
    assert b.off1.tolist() == [0, 0, 10, 10] and b.off2.tolist() == [20, 40, 20, 30]
    sub = b.subset(1, 3)
    assert len(sub) == 2 and sub.feats is b.feats and sub.off2.tolist() == [40, 20]
    pairs = [(subs[0], subs[2]), (subs[1], [])]
    bp = rb.batch_from_pairs(reg, pairs)
    assert bp.n1.tolist() == [10, 10] and bp.n2.tolist() == [10, 0] and bp.assoc is None


@pytest.mark.parametrize("which", ["pairs", "grid"])
def test_world_size_2_gloo_matches_serial(which):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, which)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reg = registration_for("gravity")
    batch = make_batch(reg) if which == "pairs" else make_grid_batch(reg)
    serial = oracle_compute(reg, batch)
    for rank, assoc, T, status in outs:
        assert len(assoc) == len(batch)
        for b in range(len(batch)):
            assert assoc[b] == serial.assoc[b].tolist()
            assert np.allclose(np.array(T[b]), serial.T[b], equal_nan=True)
        assert status == serial.status.tolist()
    assert any(len(a) >= 3 for a in outs[0][1])


def test_records_without_a_result_raise_on_every_rank_alike():
    """check_records() runs on the GATHERED statuses (identical on every rank, so every rank raises or none does): a problem the
    library gave up on (ROMAN_ST_INTERNAL) or that found no workspace after the retries (ROMAN_ST_WORKSPACE) is an error, the
    reference's own per-pair conditions (insufficient associations, empty map, iteration limit, ties) are not."""
    from roman_amd import RomanHipError, _abi
    from roman_amd.align.distributed import check_records
    ok = np.array([0, _abi.ROMAN_ST_INSUFFICIENT, _abi.ROMAN_ST_EMPTY_MAP | _abi.ROMAN_ST_INSUFFICIENT, _abi.ROMAN_ST_MAXITER,
                   _abi.ROMAN_ST_TIE_FALLBACK, _abi.ROMAN_ST_ASSOC_TRUNCATED], dtype=np.int32)
    check_records(ok)
    for bad in (_abi.ROMAN_ST_INTERNAL, _abi.ROMAN_ST_WORKSPACE, _abi.ROMAN_ST_INTERNAL | _abi.ROMAN_ST_INSUFFICIENT):
        st = ok.copy(); st[3] = bad
        with pytest.raises(RomanHipError, match="without a result") as ei:
            check_records(st)
        assert "[3]" in str(ei.value)
