"""bench.py's N>1 machinery on the CPU box.

(1) `bench.CallLoop` — the launch / join / gather choreography of a timed step — at world size 2 on `gloo` with a STUB
    context that models the library's asynchrony: a launched call is "in flight" (its output set holds garbage) until a
    join completes it; `join(skip_latest=True)` completes every call but the latest (roman_ctx_join's contract).  The
    gather is a real all_gather of fixed-size records.  Checked on both ranks, at every pipeline depth: a set is never
    gathered while its call is in flight, never rewritten before it was gathered, every call's records are gathered exactly
    once (the last one by drain()), and what arrives is what every rank produced for that call.
(2) `python bench.py --gpus 2` started as a PLAIN process no longer exits with an error: it launches its own ranks
    through torch.distributed.run (checked through the command line it builds; no GPU is needed for that)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def _records(rank, step, ci, rows):
    """What rank `rank` 'computes' for call `ci` of step `step`: a deterministic (rows, 6) int32 record block."""
    base = 1000003 * (rank + 1) + 7919 * step + 104729 * ci
    return (base + np.arange(rows * 6, dtype=np.int64).reshape(rows, 6) * (rank + 3)).astype(np.int32)


class StubContext:
    """Stand-in for runtime.Context + the output sets: launches complete only at a join."""

    def __init__(self, rank, nset, rows):
        self.rank, self.rows = rank, rows
        self.sets = [np.full((rows, 6), -1, np.int32) for _ in range(nset)]
        self.state = ["free"] * nset            # free | flight | done (complete, not yet gathered)
        self.pending = []                       # (set, payload) of the calls in flight, oldest first
        self.tag = [None] * nset                # (step, ci) of the call a set holds
        self.step = 0
        self.errors = []

    def launch(self, ci, k):
        if self.state[k] != "free":
            self.errors.append(f"set {k} rewritten by call {ci} while {self.state[k]} (holds {self.tag[k]})")
        self.state[k] = "flight"; self.tag[k] = (self.step, ci)
        self.sets[k][:] = -7                                         # garbage until the call completes
        self.pending.append((k, _records(self.rank, self.step, ci, self.rows)))

    def join(self, skip_latest):
        keep = self.pending[-1:] if (skip_latest and self.pending) else []
        for k, payload in self.pending[:len(self.pending) - len(keep)]:
            self.sets[k][:] = payload; self.state[k] = "done"
        self.pending = keep


def _worker(rank, world, port, q, pipeline, n_calls, steps):
    import torch
    import torch.distributed as dist
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = 5
        nset = max(pipeline, 2) + 1                                  # (bench.py with a process group: one set more than calls in flight)
        ctx = StubContext(rank, nset, rows)
        gathered = []                                                # (step, ci) in gather order
        problems = []

        def gather(k):
            if ctx.state[k] != "done":
                problems.append(f"gather of set {k} while {ctx.state[k]}")
            rec = torch.from_numpy(ctx.sets[k].copy())
            out = torch.empty((world * rows, 6), dtype=torch.int32)
            dist.all_gather_into_tensor(out, rec)
            step, ci = ctx.tag[k]
            for r in range(world):
                if not np.array_equal(out[r * rows:(r + 1) * rows].numpy(), _records(r, step, ci, rows)):
                    problems.append(f"call {(step, ci)}: rank {r}'s records arrived wrong")
            gathered.append((step, ci)); ctx.state[k] = "free"

        def launch(ci, k):
            ctx.launch(ci, k)
            if pipeline == 1:                                        # depth 1: the call runs on the gathering stream itself
                ctx.join(False)

        loop = bench.CallLoop(n_calls, nset, pipeline, launch, ctx.join, gather)
        for s in range(steps):
            ctx.step = s
            loop.step()
        loop.drain()
        want = [(s, ci) for s in range(steps) for ci in range(n_calls)]
        if gathered != want:
            problems.append(f"gather order {gathered} != launch order {want}")
        if ctx.pending or any(st != "free" for st in ctx.state):
            problems.append(f"left over: pending {len(ctx.pending)}, states {ctx.state}")
        q.put((rank, problems + ctx.errors))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pipeline,n_calls", [(1, 1), (2, 1), (3, 1), (3, 2), (2, 8)])
def test_call_loop_gathers_every_call_once_and_only_when_complete(pipeline, n_calls):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q, pipeline, n_calls, 4)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, problems in outs:
        assert not problems, (rank, problems)


def test_call_loop_refuses_too_few_output_sets():
    import bench
    with pytest.raises(AssertionError):
        bench.CallLoop(1, 2, 3, lambda ci, k: None, lambda skip: None, lambda k: None)


def test_plain_start_with_gpus_n_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 2 --steps 3` with no WORLD_SIZE in the environment: main() hands the same arguments to
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1) and exits with ITS return code — not with an error."""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd; seen["env"] = env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False); monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_help_renders(capsys, monkeypatch):
    """`python bench.py --help` prints the options (a bare per-cent sign in a help string makes argparse raise)."""
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--help"])
    with pytest.raises(SystemExit) as ei:
        bench.parse()
    assert ei.value.code == 0 and "--pipeline" in capsys.readouterr().out

