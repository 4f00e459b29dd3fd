"""Alignments in flight: the chunked, pipelined call loop around `roman_align_batch_dev`.

The reference aligns submap pairs one after the other ([REF roman/align/submap_align.py:93-200]: `register()` then `T_align()`
per pair).  One device call per pair — or one call for ALL pairs — leaves the GPU idle behind the slowest problem of a call:
the solver launch lasts as long as its longest problem while the other compute units have nothing to do.  The calls of this
module keep `in_flight` batch calls on the device at once (`roman_ctx_set_pipeline`), so that the affinity build of call k + 1
fills the compute units call k's solver has already left; every call is a pure enqueue, one host thread feeds them all.

Three layers, each usable on its own:

* `CallLoop` — the choreography alone (which output set a call writes, when the records of the previous call are collected),
  independent of the device: `tests/test_bench_loop_cpu.py` drives it at world size 2 on gloo with a stub context.
* `AlignStream` — a stream of batch calls over device-resident feature pools with rotating output sets and, with a process
  group, the all_gather of call k - 1's records while call k computes (the only collective of the path).
  This is the loop `bench.py` times: the headline number is produced by `AlignStream.submit()` / `drain()`.
* `align_resident` — ONE large batch (e.g. the 4096 pairs of an all-pairs grid) over a pool that is already in HBM: calls of
  `chunk` problems, `in_flight` of them at once, every call writing its own rows of the outputs, problems the speculatively
  sized workspace skipped (ROMAN_ST_WORKSPACE) issued again — those only —, results returned on the host.  `run_batch` (host
  arrays in) gets the same behaviour from the library itself (`roman_ctx_set_host_batching`).

torch is used for device memory, streams and the process group only; nothing numerical happens here.
"""
import numpy as np

from .. import _abi

MAX_ATTEMPTS = 5          # as the library's host-pointer entry points (roman_hip.hip MAX_ATTEMPTS)


class CallLoop:
    """The choreography of the calls of a step, independent of the device (tests/test_bench_loop_cpu.py drives it at world
    size 2 on gloo with a stub context): call number c writes output set c % nset; with batches in flight
    (`pipeline` >= 2) the records of call c - 1 are gathered while call c computes — `join(skip_latest=True)` makes the
    gathering stream wait for every call but the latest —, `drain()` gathers the last call's; at depth 1 a call is
    gathered right behind itself.  `launch(ci, k)`: enqueue call `ci` of the step into output set `k`; `join(skip_latest)`;
    `gather(k)`: collect output set k on every rank (None: single process, nothing to gather)."""

    def __init__(self, n_calls, nset, pipeline, launch, join, gather=None):
        assert nset >= max(pipeline, 2), "an output set is rewritten only after the call pipeline + 1 calls back has been gathered"
        self.n_calls, self.nset, self.pipeline = n_calls, nset, pipeline
        self.launch, self.join, self.gather = launch, join, gather
        self.call_no = 0

    def one_call(self, ci):
        k = self.call_no % self.nset
        self.call_no += 1
        self.launch(ci, k)
        if self.gather is not None:
            if self.pipeline >= 2:
                if self.call_no > 1:
                    self.join(True)                              # the gathering stream waits for the OLDER calls only
                    self.gather((k - 1) % self.nset)
            else:
                self.gather(k)
        return k

    def step(self):
        for ci in range(self.n_calls):
            self.one_call(ci)

    def drain(self):                                             # results of the last call
        if self.pipeline >= 2:
            self.join(False)
            if self.gather is not None and self.call_no > 0:
                self.gather((self.call_no - 1) % self.nset)

    def reset(self):
        self.call_no = 0


class OutputSet:
    """Device-resident outputs of one batch call of up to `rows` problems (torch tensors; what roman_align_batch_dev writes)."""

    def __init__(self, rows, kmax, device):
        import torch
        self.rows, self.kmax = int(rows), int(kmax)
        self.assoc = torch.zeros((rows, kmax, 2), dtype=torch.int32, device=device)
        self.n = torch.zeros(rows, dtype=torch.int32, device=device)
        self.T = torch.zeros((rows, 16), dtype=torch.float64, device=device)
        self.status = torch.zeros(rows, dtype=torch.int32, device=device)
        self.stats = torch.zeros(rows * _abi.STATS_NBYTES, dtype=torch.uint8, device=device)

    def host(self, count=None):
        """-> (assoc (count,kmax,2), n, T (count,16), status, stats structured array) on the host (synchronises the copy)."""
        from ..runtime import stats_dtype
        c = self.rows if count is None else int(count)
        st = np.frombuffer(self.stats.cpu().numpy().tobytes(), dtype=stats_dtype())[:c]
        return self.assoc.cpu().numpy()[:c], self.n.cpu().numpy()[:c], self.T.cpu().numpy()[:c], self.status.cpu().numpy()[:c], st


class _HostStream:
    """Stand-in for a torch.cuda.Stream when the tensors live on the CPU (tests over a stand-in context: every call is synchronous)."""
    cuda_stream = None

    def wait_event(self, ev):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _HostEvent:
    def record(self, stream=None):
        pass


class AlignStream:
    """A stream of batch calls with `in_flight` of them on the device at once.

        S = AlignStream(registration, ctx, device, rows=256, kmax=200, in_flight=3)
        for batch in batches:                    # pure enqueues: the host never waits here
            k = S.submit(pool.data_ptr(), F, off1, n1, off2, n2)
        S.drain()                                # every call issued so far has been ordered / collected
        torch.cuda.synchronize(device)           # results of the last `nset` calls are in S.sets[k]

    Call number c writes output set c % nset (nset = in_flight, one more with a process group); a set is rewritten
    `nset` calls later, so a caller that needs every call's results reads (or gathers) a set before then — `on_collect(k, tag)`
    is called once per call, in issue order, when the stream `collect_stream` has been made to wait for that call (with a
    process group the library's default collector all-gathers the set's records there first: `gathered_ints`,
    `gathered_T` hold them for every rank, rows [r * rows, (r + 1) * rows) from rank r).
    `stream`: the torch stream the context was created on (the library starts every call behind the work queued there)."""

    def __init__(self, registration, ctx, device, rows, kmax, in_flight=3, stream=None, group=None, use_group=False, on_collect=None):
        import torch
        self.reg, self.ctx, self.dev = registration, ctx, device
        self.P = registration._abi_params()
        self.rows, self.kmax, self.in_flight = int(rows), int(kmax), int(in_flight)
        self.dist_on = bool(use_group)
        self.group = group
        self.on_host = torch.device(device).type == "cpu"          # CPU tensors + a stand-in context (tests): no streams, synchronous calls
        self.stream = stream if stream is not None else (_HostStream() if self.on_host else torch.cuda.current_stream(device))
        # one output set per call in flight — and one more with a collector stream: the collection of call c - 1 runs on its own
        # stream while call c computes, and the set call c + in_flight - 1 rewrites must not be the one it is still reading
        self.collecting = self.dist_on or on_collect is not None
        self.nset = max(self.in_flight, 2) + (1 if self.collecting else 0)
        self.sets = [OutputSet(self.rows, self.kmax, device) for _ in range(self.nset)]
        self.tags = [None] * self.nset
        self.on_collect = on_collect
        self.ev_done = [None] * self.nset                          # behind the collection that read output set k
        if self.collecting:
            self.cstream = _HostStream() if self.on_host else torch.cuda.Stream(device)   # the collections' own stream: never between two batch calls
        if self.dist_on:
            import torch.distributed as dist
            self.world = dist.get_world_size(group)
            # ONE fixed-size byte record per problem (batch.record_bytes: [count, status, index pairs] int32 | pose f64) and ONE
            # all_gather_into_tensor per call; gathered_ints / gathered_T are typed VIEWS of the gathered block
            from .batch import record_bytes
            rb_, ib_ = record_bytes(self.kmax), 8 * (1 + self.kmax)
            self.rec = torch.empty((self.rows, rb_), dtype=torch.uint8, device=device)
            self.rec_i = self.rec[:, :ib_].view(torch.int32)
            self.rec_T = self.rec[:, ib_:].view(torch.float64)
            self.gathered = torch.empty((self.world * self.rows, rb_), dtype=torch.uint8, device=device)
            self.gathered_ints = self.gathered[:, :ib_].view(torch.int32)
            self.gathered_T = self.gathered[:, ib_:].view(torch.float64)
            self.gathers = 0                                           # collectives issued (tests: one per call)
        self._pending = None                                        # arguments of the call being launched
        self._depth_before = getattr(ctx, "pipeline_depth", 1)
        self.loop = CallLoop(1, self.nset, self.in_flight, self._launch,
                             (lambda skip: ctx.join(skip_latest=skip, stream=self.cstream.cuda_stream)) if self.collecting else (lambda skip: ctx.join(skip_latest=skip)),
                             self._collect if self.collecting else None)
        ctx.set_pipeline(self.in_flight)

    # -- CallLoop callbacks
    def _launch(self, ci, k):
        pool_ptr, F, o1, a1, o2, a2, assoc_ptr, assoc_off, u0_ptr, tag = self._pending
        if self.ev_done[k] is not None:                             # output set k is rewritten: behind the collection that read it
            self.stream.wait_event(self.ev_done[k])
        O = self.sets[k]
        self.tags[k] = tag
        self.ctx.align_batch_dev(self.P, pool_ptr, F, o1, a1, o2, a2, self.kmax, O.assoc.data_ptr(), O.n.data_ptr(), O.T.data_ptr(),
                                 O.status.data_ptr(), O.stats.data_ptr(), assoc_ptr=assoc_ptr, assoc_off=assoc_off, u0_ptr=u0_ptr)

    def _collect(self, k):
        import torch
        O = self.sets[k]
        if self.in_flight < 2:                                      # one call at a time, on the context's stream: the collector waits for what is queued there
            self.ctx.join(skip_latest=False, stream=self.cstream.cuda_stream)
        with (self.cstream if self.on_host else torch.cuda.stream(self.cstream)):
            if self.dist_on:
                import torch.distributed as dist
                self.rec_i[:, 0] = O.n; self.rec_i[:, 1] = O.status; self.rec_i[:, 2:] = O.assoc.view(self.rows, -1)
                self.rec_T.copy_(O.T)
                dist.all_gather_into_tensor(self.gathered, self.rec, group=self.group)
                self.gathers += 1
            if self.on_collect is not None:
                self.on_collect(k, self.tags[k])
            self.ev_done[k] = _HostEvent() if self.on_host else torch.cuda.Event(); self.ev_done[k].record(self.cstream)

    # -- the caller's side
    def submit(self, pool_ptr, F, off1, n1, off2, n2, assoc_ptr=None, assoc_off=None, u0_ptr=None, tag=None):
        """Enqueue one batch call (at most `rows` problems) over the feature pool at device address `pool_ptr`; -> the output
        set it writes.  A pure enqueue."""
        if len(n1) > self.rows:
            raise ValueError(f"{len(n1)} problems in a call of an AlignStream built for {self.rows}")
        if getattr(self.ctx, "pipeline_depth", self.in_flight) != self.in_flight:     # a synchronous entry point in between put it back
            self.ctx.set_pipeline(self.in_flight)
        self._pending = (pool_ptr, F, off1, n1, off2, n2, assoc_ptr, assoc_off, u0_ptr, tag)
        return self.loop.one_call(0)

    def drain(self):
        """Order / collect everything issued so far (the host does not wait: synchronise the device or the collector stream to
        read results)."""
        self.loop.drain()

    def reset(self):
        self.loop.reset()

    def close(self):
        self.ctx.set_pipeline(self._depth_before)


def default_chunk(B):
    """Problems per call for a ONE-SHOT batch of B problems: one call up to 512 problems, two calls up to 4096, calls of 2048
    beyond.  Measured on config 4 (bench.py `caller.one_shot`, one MI355X): 4096 pairs as 2 x 2048 take 37.7 ms, as 8 x 512 or
    16 x 256 41-42 ms; a rank's 512-pair share as ONE call 5.4-7.4 ms, as 4 x 128 6.2-8.0 ms — every call pays its launches
    and its own solver tail, two calls in flight already overlap one build with one tail."""
    B = int(B)
    if B <= 512:
        return max(B, 1)
    return min(2048, ((-(-B // 2)) + 63) & ~63)


def issue_chunked(ctx, P, pool, batch, kmax, a_out, n_out, T_out, st_out, stats_out=None, assoc_dev=None, chunk=None, in_flight=3):
    """The problems of `batch` over the device-resident `pool` as calls of `chunk` problems with `in_flight` of them on the device
    at once; problem b writes row b of the output tensors (torch, on the pool's device).  Problems a call skipped for workspace
    (ROMAN_ST_WORKSPACE: the pools are sized before the live counts are known) are issued again — those only, in runs of
    consecutive problems — up to MAX_ATTEMPTS times; with no sizing history for this parameter block (as far as this module has
    seen) the first call is waited for, so that the calls queued behind it size their pools from what it needed.
    -> the status array on the host (the caller decides what ROMAN_ST_INTERNAL / a still skipped problem mean).
    Synchronises the context; its pipeline depth and team mode are restored on return."""
    B = len(batch)
    F = int(pool.shape[1])
    chunk = default_chunk(B) if chunk is None else max(1, int(chunk))

    def issue(lo, hi):
        ctx.align_batch_dev(P, pool.data_ptr(), F, batch.off1[lo:hi], batch.n1[lo:hi], batch.off2[lo:hi], batch.n2[lo:hi], kmax,
                            a_out[lo:].data_ptr(), n_out[lo:].data_ptr(), T_out[lo:].data_ptr(), st_out[lo:].data_ptr(),
                            None if stats_out is None else stats_out[lo * _abi.STATS_NBYTES:].data_ptr(),
                            assoc_ptr=None if assoc_dev is None else assoc_dev.data_ptr(),
                            assoc_off=None if batch.assoc_off is None else batch.assoc_off[lo:hi + 1])

    status = np.zeros(0, dtype=np.int32)
    teams_off = False
    depth_before = getattr(ctx, "pipeline_depth", 1)              # restored on return (an AlignStream on the same context keeps its depth)
    teams_before = getattr(ctx, "wide_teams", -1)                 # a team mode the caller chose survives the teams-off retry
    ctx.set_pipeline(max(1, int(in_flight)))
    try:
        lo = 0
        # the library keeps ONE sizing history — that of the latest parameter block — and is asked for it (roman_ctx_has_history);
        # a context without the query (a stand-in in the CPU tests) is treated as having none
        has = getattr(ctx, "has_history", None)
        if B and not (has(P, F) if has is not None else False):
            issue(0, min(B, chunk)); ctx.sync(); lo = min(B, chunk)
        for l in range(lo, B, chunk):
            issue(l, min(B, l + chunk))
        again, teams_off = _abi.ROMAN_ST_WORKSPACE, False
        for attempt in range(1, MAX_ATTEMPTS + 1):
            ctx.sync()
            status = st_out.cpu().numpy()[:B]
            again = _abi.ROMAN_ST_WORKSPACE
            if (status & _abi.ROMAN_ST_INTERNAL).any() and not teams_off and teams_before != 0:
                # a team of the whole-device solver that could not hold its problem leaves ROMAN_ST_INTERNAL like an expired wait
                # does: those problems once more with the whole device per problem; a second ROMAN_ST_INTERNAL is final — but the
                # problems still flagged ROMAN_ST_WORKSPACE keep being issued again while attempts remain
                teams_off = True; ctx.set_wide_teams(0); again |= _abi.ROMAN_ST_INTERNAL
            skipped = np.nonzero((status & again) != 0)[0]
            if not len(skipped) or attempt == MAX_ATTEMPTS:
                break
            b = 0
            while b < len(skipped):                                 # runs of consecutive skipped problems, a chunk at most
                e = b + 1
                while e < len(skipped) and skipped[e] == skipped[e - 1] + 1 and e - b < chunk:
                    e += 1
                issue(int(skipped[b]), int(skipped[e - 1]) + 1)
                b = e
    finally:
        ctx.set_pipeline(depth_before)
        if teams_off:
            ctx.set_wide_teams(teams_before)
    return status.copy()


def align_resident(registration, batch, pool, chunk=None, in_flight=3, device=None, ctx=None, stats=True):
    """ONE batch of any size over a feature pool that is already in HBM (`pool`: the batch's (n_objects, F) float64 matrix as a
    torch tensor on `device`) -> runtime.BatchResult on the host.  On a real context this is ONE library call
    (roman_align_batch_resident: calls of `chunk` problems, `in_flight` at once, skipped problems issued again, one read-back);
    a stand-in context without that entry gets issue_chunked() + the read-back of the torch outputs.  ROMAN_ST_INTERNAL / a
    problem still skipped after MAX_ATTEMPTS raise RomanHipError (the result so far in `.result` where the library copied it)."""
    import torch
    from ..runtime import BatchResult, stats_dtype
    ctx = ctx or registration._context()
    dev = device if device is not None else pool.device
    P = registration._abi_params()
    B, kmax = len(batch), batch.kmax()
    if tuple(pool.shape) != tuple(batch.feats.shape) or pool.dtype != torch.float64:
        raise ValueError(f"pool must be the batch's feature matrix {batch.feats.shape} as a float64 tensor")
    if hasattr(ctx, "align_batch_resident"):
        # the library's own entry for resident inputs and host results (roman_align_batch_resident): it chunks, pipelines and
        # re-issues as issue_chunked() does, and the whole result comes back with ONE copy through pinned memory
        assoc_dev = None if batch.assoc is None else torch.from_numpy(np.ascontiguousarray(batch.assoc, dtype=np.int32)).to(dev)
        torch.cuda.current_stream(dev).synchronize()           # the pool (and the list) are in place before the library's streams read them
        before = getattr(ctx, "host_batching", (2048, 3))
        ctx.set_host_batching(default_chunk(B) if chunk is None else max(1, int(chunk)), max(1, int(in_flight)))
        try:
            return ctx.align_batch_resident(P, pool.data_ptr(), int(pool.shape[1]), batch.off1, batch.n1, batch.off2, batch.n2, kmax,
                                            assoc_ptr=None if assoc_dev is None else assoc_dev.data_ptr(), assoc_off=batch.assoc_off)
        finally:
            ctx.set_host_batching(*before)
    a_out = torch.full((max(B, 1), kmax, 2), -1, dtype=torch.int32, device=dev)
    n_out = torch.zeros(max(B, 1), dtype=torch.int32, device=dev)
    T_out = torch.zeros((max(B, 1), 16), dtype=torch.float64, device=dev)
    st_out = torch.zeros(max(B, 1), dtype=torch.int32, device=dev)
    stats_out = torch.zeros(max(B, 1) * _abi.STATS_NBYTES, dtype=torch.uint8, device=dev) if stats else None
    assoc_dev = None if batch.assoc is None else torch.from_numpy(np.ascontiguousarray(batch.assoc, dtype=np.int32)).to(dev)
    torch.cuda.current_stream(dev).synchronize()               # inputs and cleared outputs are in place before the library's streams touch them
    status = issue_chunked(ctx, P, pool, batch, kmax, a_out, n_out, T_out, st_out, stats_out, assoc_dev, chunk, in_flight)
    s = registration.dim + 1
    n_h = n_out.cpu().numpy()[:B]; a_h = a_out.cpu().numpy()[:B]
    T_h = T_out.cpu().numpy()[:B, :s * s].reshape(B, s, s).copy()
    st_h = (np.frombuffer(stats_out.cpu().numpy().tobytes(), dtype=stats_dtype())[:B].copy() if stats_out is not None
            else np.zeros(B, dtype=stats_dtype()))
    res = BatchResult([a_h[b, :n_h[b]].copy() for b in range(B)], T_h, status, st_h)
    bad = np.nonzero((status & (_abi.ROMAN_ST_INTERNAL | _abi.ROMAN_ST_WORKSPACE)) != 0)[0]
    if len(bad):
        err = _abi.RomanHipError(f"{len(bad)} problem(s) without a result (ROMAN_ST_INTERNAL or still ROMAN_ST_WORKSPACE after "
                                 f"{MAX_ATTEMPTS} attempts): problems {bad[:8].tolist()}")
        err.result = res
        raise err
    return res
