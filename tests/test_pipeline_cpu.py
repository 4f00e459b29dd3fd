"""roman_amd.align.pipeline's chunking / re-issue logic on a box without a GPU: the real functions over torch CPU tensors and a
stand-in context that computes with the oracle through the same raw pointers (tests/_stub_context.py).
(1) issue_chunked: calls of `chunk` problems at the requested depth, every problem's row written once; problems the library
    skipped for workspace are issued again — those only, in runs —; ROMAN_ST_INTERNAL records from a batch that ran in team mode
    are issued once more with teams off and the setting restored; the first call of an unknown parameter block is waited for.
(2) align_sharded at world size 2 on gloo with the DEVICE-record path (chunked _device_records, `device=cpu`): every rank gets
    the serial result in problem order."""
import os
import socket

import numpy as np
import pytest

from conftest import registration_for
from roman_amd import _abi, synth
from roman_amd.align import batch as rb
from roman_amd.align.pipeline import issue_chunked
from _stub_context import OracleContext
from test_distributed_cpu import oracle_compute


def _outputs(B, kmax):
    import torch
    return (torch.full((B, kmax, 2), -9, dtype=torch.int32), torch.full((B,), -9, dtype=torch.int32),
            torch.full((B, 16), 7.0, dtype=torch.float64), torch.full((B,), -9, dtype=torch.int32))


def _grid(reg, S=4, n=18):
    subs, _ = synth.make_submap_grid(2 * S, n=n, d=0, seed0=47)
    subs[1] = subs[1][:9]; subs[S + 2] = subs[S + 2][:13]
    return rb.batch_from_submap_grid(reg, subs[:S], subs[S:])


def test_issue_chunked_calls_rows_reissue_and_team_fallback():
    import torch
    reg = registration_for("gravity")
    batch = _grid(reg)                                              # 16 problems over one pool
    B, kmax = len(batch), batch.kmax()
    pool = torch.from_numpy(batch.feats.copy())
    want = oracle_compute(reg, batch)
    key = lambda b: (int(batch.off1[b]), int(batch.off2[b]))
    # problems 3, 4 and 9 are skipped once (3 and 4 form a run), 12 twice; problem 6 comes back INTERNAL while teams are on
    stub = OracleContext(batch.feats.shape[0], skip_times={key(3): 1, key(4): 1, key(9): 1, key(12): 2}, internal_with_teams={key(6)})
    a, n, T, st = _outputs(B, kmax)
    status = issue_chunked(stub, reg._abi_params(), pool, batch, kmax, a, n, T, st, None, None, chunk=5, in_flight=3)
    assert (status & (_abi.ROMAN_ST_WORKSPACE | _abi.ROMAN_ST_INTERNAL) == 0).all()
    for b in range(B):
        assert np.array_equal(a[b, :int(n[b])].numpy(), want.assoc[b]), b
        assert status[b] == want.status[b]
        if status[b] == 0:
            assert np.allclose(T[b].numpy().reshape(4, 4), want.T[b])
    sizes = [c[1] for c in stub.calls]
    # first call alone and waited for (unknown parameter block), then the rest in chunks of 5, then the re-issues: the run {3, 4},
    # {9}, {12} and the INTERNAL problem 6 (teams off by then), then {12} once more
    assert sizes[:4] == [5, 5, 5, 1] and stub.calls[0][2] == 0 and stub.calls[1][2] == 1     # one sync between call 0 and call 1
    assert sorted(sizes[4:]) == [1, 1, 1, 1, 2]
    assert all(c[0] == 3 for c in stub.calls) and stub.pipeline == 1 and stub.wide_teams == -1   # depth while issuing; both settings restored
    # a second batch of the same parameter block: no waited-for first call any more
    stub2_calls = len(stub.calls)
    issue_chunked(stub, reg._abi_params(), pool, batch, kmax, a, n, T, st, None, None, chunk=8, in_flight=2)
    # (problem 6 is INTERNAL again while teams are on: one more call for it)
    assert [c[1] for c in stub.calls[stub2_calls:]] == [8, 8, 1] and stub.calls[stub2_calls][2] == stub.calls[stub2_calls + 1][2]
    # a problem that stays ROMAN_ST_INTERNAL with teams off is final: its flag is handed back, nothing loops
    stub3 = OracleContext(batch.feats.shape[0], internal_with_teams=set())
    stub3.align_real = stub3.align_batch_dev

    def always_internal(P, fp, F, o1, n1, o2, n2, km, ap, np_, Tp, sp, *rest, **kw):
        stub3.align_real(P, fp, F, o1, n1, o2, n2, km, ap, np_, Tp, sp, *rest, **kw)
        from _stub_context import _view
        s = _view(sp, (len(n1),), np.int32)
        for b in range(len(n1)):
            if (int(o1[b]), int(o2[b])) == key(2):
                s[b] = _abi.ROMAN_ST_INTERNAL
    stub3.align_batch_dev = always_internal
    status = issue_chunked(stub3, reg._abi_params(), pool, batch, kmax, a, n, T, st, None, None, chunk=16, in_flight=1)
    assert status[2] & _abi.ROMAN_ST_INTERNAL and (np.delete(status, 2) & _abi.ROMAN_ST_INTERNAL == 0).all()
    assert len(stub3.calls) <= 4


def test_issue_chunked_slices_explicit_association_lists():
    import torch
    reg = registration_for("clipper+prune", cosine_min=0.5)
    pairs = [(p.map1, p.map2) for p in (synth.make_pair(12 + k, 10 + (k % 3), 16, 300 + k) for k in range(7))]
    batch = rb.batch_from_pairs(reg, pairs)
    assert batch.assoc is not None
    B, kmax = len(batch), batch.kmax()
    pool = torch.from_numpy(batch.feats.copy()); assoc = torch.from_numpy(batch.assoc.copy())
    stub = OracleContext(batch.feats.shape[0])
    a, n, T, st = _outputs(B, kmax)
    issue_chunked(stub, reg._abi_params(), pool, batch, kmax, a, n, T, st, None, assoc, chunk=3, in_flight=2)
    want = oracle_compute(reg, batch)                               # (all-to-all there; here: explicit pruned lists)
    from oracle import oracle as orc
    for b in range(B):
        A = batch.assoc[batch.assoc_off[b]:batch.assoc_off[b + 1]]
        o = orc.register(reg._abi_params(), batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]], batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]], A if len(A) else None)
        assert np.array_equal(a[b, :int(n[b])].numpy(), o["assoc"]), b
    assert want is not None


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from roman_amd.align.distributed import align_sharded
        reg = registration_for("gravity")
        batch = _grid(reg)
        key = lambda b: (int(batch.off1[b]), int(batch.off2[b]))
        stub = OracleContext(batch.feats.shape[0], skip_times={key(5): 1, key(10): 1})
        reg.set_context(stub)
        assoc, T, status = align_sharded(reg, batch, device=torch.device("cpu"), chunk=3, in_flight=3)
        q.put((rank, [a.tolist() for a in assoc], np.nan_to_num(T, nan=-1.0).tolist(), status.tolist(), [c[1] for c in stub.calls]))
    finally:
        dist.destroy_process_group()


def test_chunked_align_sharded_world_size_2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reg = registration_for("gravity")
    batch = _grid(reg)
    serial = oracle_compute(reg, batch)
    for rank, assoc, T, status, calls in outs:
        assert len(assoc) == len(batch) == 16
        for b in range(16):
            assert assoc[b] == serial.assoc[b].tolist(), (rank, b)
            assert np.allclose(np.array(T[b]), np.nan_to_num(serial.T[b], nan=-1.0))
        assert status == serial.status.tolist()
        assert max(calls) <= 3 and len(calls) >= 3                  # a rank's share in calls of at most 3 problems (+ the re-issues)
    assert sum(sum(o[4]) for o in outs) >= 16 + 2                   # every problem once, the two skipped ones twice


def _stream_worker(rank, world, port, q, in_flight):
    """The loop bench.py times at N > 1 (BASELINE config 4, strong scaling), on CPU tensors: the grid's pairs dealt by
    deal_by_cost, a rank's share as calls of `chunk` problems through the package's AlignStream WITH the process group — every
    call's fixed-size byte records collected on every rank by ONE all_gather_into_tensor."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from roman_amd.align.distributed import deal_by_cost, problem_work
        from roman_amd.align.pipeline import AlignStream
        reg = registration_for("gravity")
        batch = _grid(reg)
        shards = deal_by_cost(problem_work(batch), world)
        mine = shards[rank]
        chunk = 3
        kmax = batch.kmax()
        stub = OracleContext(batch.feats.shape[0])
        pool = torch.from_numpy(batch.feats.copy())
        seen = []                                                   # (tag, ints, poses) of every collection, in order

        def on_collect(k, tag):
            seen.append((tag, S.gathered_ints.clone().numpy(), S.gathered_T.clone().numpy()))
        S = AlignStream(reg, stub, torch.device("cpu"), rows=chunk, kmax=kmax, in_flight=in_flight, use_group=True, on_collect=on_collect)
        ncalls = max(-(-len(s) // chunk) for s in shards)            # every rank issues the same number of calls (a collective per call)
        for c in range(ncalls):
            ix = mine[c * chunk:(c + 1) * chunk]
            S.submit(pool.data_ptr(), batch.feats.shape[1], batch.off1[ix], batch.n1[ix], batch.off2[ix], batch.n2[ix], tag=c)
        S.drain()
        q.put((rank, S.gathers, [(t, i.tolist(), np.nan_to_num(p, nan=-1.0).tolist()) for t, i, p in seen], [c[1] for c in stub.calls]))
        S.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("in_flight", [1, 3])
def test_align_stream_with_a_process_group_one_gather_per_call_world_size_2_gloo(in_flight):
    import torch.multiprocessing as mp
    from roman_amd.align.distributed import deal_by_cost, problem_work
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, q, in_flight)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reg = registration_for("gravity")
    batch = _grid(reg)
    serial = oracle_compute(reg, batch)
    shards = deal_by_cost(problem_work(batch), 2)
    chunk, kmax = 3, batch.kmax()
    ncalls = max(-(-len(s_) // chunk) for s_ in shards)
    for rank, gathers, seen, calls in outs:
        assert gathers == ncalls == len(seen)                       # ONE collective per call, nothing else
        assert [t for t, _, _ in seen] == list(range(ncalls))       # collected in issue order
        for c, ints, poses in seen:
            ints = np.array(ints); poses = np.array(poses)
            assert ints.shape == (2 * chunk, 2 + 2 * kmax) and poses.shape == (2 * chunk, 16)
            for r in range(2):                                      # rows [r * rows, (r + 1) * rows) hold rank r's records of call c
                ix = shards[r][c * chunk:(c + 1) * chunk]
                for j, b in enumerate(ix):
                    row = ints[r * chunk + j]
                    k = int(row[0])
                    assert row[1] == serial.status[b] and np.array_equal(row[2:2 + 2 * k].reshape(-1, 2), serial.assoc[b]), (rank, c, r, b)
                    if serial.status[b] == 0:
                        assert np.allclose(poses[r * chunk + j].reshape(4, 4), serial.T[b])
        assert all(n <= chunk for n in calls)
