#!/usr/bin/env python3
"""bench.py — submap-pair alignments/sec (+ p50 single-pair latency) of the roman.align hot path on
MI355X, at BASELINE.json's shape n=m=200 objects, d=512 (`method='semanticgrav'`).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Workloads (`--workload`, default `pairs` at every N, so that the per-N values of a scaling sweep are one workload):
  pairs  BASELINE config 3: a batch of 256 independent synthetic submap pairs per GPU per step (config 2 is the
         same shape at B=1 and is what `p50_latency_ms` is measured on).  With N>1 every rank aligns its own 256
         pairs: weak scaling.
  grid   BASELINE config 4: the all-pairs grid of 64 x 64 submaps (4096 alignments) of two robots.  The 128 submaps
         are packed ONCE into one feature pool that every rank holds; the 4096 pairs are dealt round-robin to the
         ranks (512 per rank at N=8) and aligned in calls of at most 512 pairs: strong scaling, a step = the whole
         grid.
The `pairs` run is followed by a `grid` leg (`--grid-steps`, `--grid-warmup`) whose result is reported in the same
line as `grid_config4` — config 4's strong-scaling number next to the weak one, at every N.
A "step" is one pass of the hot path (score -> solve -> select -> pose; roman_align_batch_dev calls) over the
workload.  Inputs are resident in HBM before the timed region (`value_incl_h2d` is the same loop with the upload of
every batch inside it).  There is no data-path collective; one RCCL all_gather of the fixed-size result records
(inlier sets + poses) per call collects the results on every rank.

The printed JSON line also carries (rank 0, N=1; `--no-extras` leaves the side legs out)
  roofline             the dominant kernel (k_solve_up): algorithmic bytes per launch / its hipEvent
                       time against the 8 TB/s HBM3E peak, the counter-based traffic fraction next to it, the other
                       kernels' own bounds (`kernels`) and the whole step against §8(d)'s ideal time (`step`);
  cpu_baseline         the CPU oracle (a restatement of the absent clipperpy, kind "port") on this box's host cores;
  decision_sensitivity what the headline becomes under every reading of the two formulas that live only in the absent
                       clipperpy sources (ROMAN_SINGLE_* x ROMAN_GRAV_*), and under random instead of all-ones starts;
  large_live           the large-live-set path (method 'gravity', every association live: k_solve_wide) with its own roofline;
  mid_live             64 pairs of method 'gravity' at n = m = 100 in ONE call (k_solve_wide in team mode), roofline + oracle check;
  demo_scale           the scale the reference's demo configuration runs at (method 'roman', n, m in [20, 40], d = 768): 4096 DISTINCT
                       pairs per call (the 64 x 64 grid of 128 distinct submaps), six calls in flight and one at a time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
F64_PEAK_TFLOPS = 78.6         # MI355X f64 vector (= matrix) peak, vendor figure quoted in SURVEY.md §8(d)
F64_MFMA_MEASURED_TFLOPS = 67.0   # v_mfma_f64_16x16x4 issue ceiling measured on this part (tools/ubench/mfma_rate.hip, round 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="auto", choices=["auto", "pairs", "grid"], help="auto = pairs")
    ap.add_argument("--no-grid", action="store_true", help="skip the config-4 grid leg")
    ap.add_argument("--grid-steps", type=int, default=6)
    ap.add_argument("--grid-warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="pairs workload: submap pairs per GPU per step (config 3: 256)")
    ap.add_argument("--grid", type=int, default=64, help="grid workload: submaps per robot (config 4: 64 -> 4096 alignments)")
    ap.add_argument("--chunk", type=int, default=2048, help="grid workload: pairs per roman_align_batch_dev call (2048: +5 %% over 512 on one GPU — launches and the solver tail amortise; capped at a rank's share)")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--m", type=int, default=200)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--method", default="semanticgrav")
    ap.add_argument("--cpu-sample", type=int, default=4, help="pairs timed on the CPU oracle in upstream-like mode (rank 0, N=1 only); 0 = skip the CPU legs")
    ap.add_argument("--check-pairs", type=int, default=256, help="problems of the first call compared with the oracle (pruned mode)")
    ap.add_argument("--latency-reps", type=int, default=30)
    ap.add_argument("--no-profile", action="store_true", help="do not bracket stages with hipEvents")
    ap.add_argument("--no-extras", action="store_true", help="main measurement only: no grid / sensitivity / large-live / demo-scale / sustained / h2d legs")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not re-run the main measurement under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for roofline.traffic (the committed pass is quoted instead)")
    ap.add_argument("--pipeline", type=int, default=3, choices=[1, 2, 3, 4, 5, 6],
                    help="batches in flight per GPU (roman_ctx_set_pipeline): the straggler tail of one call's solver overlaps the "
                         "next calls' affinity builds; results are complete at the closing device-wide synchronise")
    return ap.parse_args()


from roman_amd.align.pipeline import CallLoop  # noqa: E402,F401  (the choreography lives in the package; tests/test_bench_loop_cpu.py drives it)


def self_launch(args_list, n):
    """`python bench.py --gpus N` started as a plain process (no WORLD_SIZE): run the same command line under
    torch.distributed.run, one rank per GPU of this node, and hand its exit code back.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(args_list)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def live_traffic(kernel="k_solve_up<8", timeout_s=150):
    """HBM bytes per launch of the dominant kernel, MEASURED on this box with this build: bench.py's main measurement (one call in flight,
    3 timed steps, nothing else) re-run under rocprofv3 twice — `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate passes with
    `--kernel-trace` only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes — and the counter averaged over the kernel's dispatches of
    the largest grid (the B = 256 launches).  -> (bytes = (2 FETCH_SIZE + WRITE_SIZE) KB x 1024: the guide's gfx950 correction for wide
    coalesced reads, dict of the raw numbers) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    raw = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="roman_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", td, "-o", "b", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "3", "--warmup", "2", "--pipeline", "1", "--latency-reps", "-1", "--cpu-sample", "0", "--no-extras", "--no-grid", "--check-pairs", "0",
                   "--no-live-traffic"]
            env = dict(os.environ); env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-200:]}"
            per = {}
            gmax = 0
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    if kernel not in row.get("Kernel_Name", "") or row.get("Counter_Name") != ctr:
                        continue
                    g = int(row.get("Grid_Size", 0) or 0)
                    gmax = max(gmax, g)
                    per.setdefault((g, row["Dispatch_Id"]), 0.0)
                    per[(g, row["Dispatch_Id"])] += float(row["Counter_Value"])
            vals = [v for (g, _), v in per.items() if g == gmax]
            if not vals:
                return None, f"no {kernel} dispatch in the {ctr} pass"
            raw[ctr + "_KB_per_launch"] = sum(vals) / len(vals); raw[ctr + "_launches"] = len(vals)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {ctr} pass exceeded {timeout_s} s"
        except Exception as e:                                   # a reported extra: never lose the line over it
            return None, f"{ctr} pass: {e!r}"
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return (2.0 * raw["FETCH_SIZE_KB_per_launch"] + raw["WRITE_SIZE_KB_per_launch"]) * 1024.0, raw


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    import types
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("ROMAN_BENCH_FORCE_DIST"):
        # three internal streams + torch's + the gathers' + RCCL's own share the runtime's 4 hardware queues by default, and a wait
        # queued on one of them holds back whatever another stream put behind it: 8 queues (measured at world size 1 with the
        # process group forced on: 104 -> 112 k alignments/s; without a process group 4 queues are as good or better)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    import torch.distributed as dist
    from roman_amd import _abi, synth
    from roman_amd.align import SubmapAlignParams
    from roman_amd.align.batch import batch_from_pairs, batch_from_submap_grid
    from roman_amd.runtime import Context, stats_dtype

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:        # a plain `python bench.py --gpus N`: launch the ranks ourselves
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`, "
                 f"or plain `python bench.py --gpus {args.gpus}`)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ROMAN_BENCH_FORCE_DIST: run the process-group / all_gather path at world size 1 too (tools/scale_preflight.sh)
    dist_on = world > 1 or (os.environ.get("ROMAN_BENCH_FORCE_DIST") and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # auto: one GPU -> BASELINE config 3 (256 pairs per step); N > 1 -> BASELINE config 4, the 4096-pair grid dealt over the ranks
    # (STRONG scaling: north_star's ">= 6x at 8 GPUs vs 1" is about this grid), the per-rank pairs loop as the side key `weak_pairs`
    workload = args.workload if args.workload != "auto" else ("pairs" if world == 1 else "grid")
    extras = not args.no_extras

    sp = SubmapAlignParams(method=args.method, semantics_dim=args.d) if args.d > 0 else SubmapAlignParams(method=args.method)
    reg = sp.get_object_registration()
    P = reg._abi_params()
    F = P.feature_dim() if P.invariant == _abi.ROMAN_INV_ROMAN else reg.dim

    stream = torch.cuda.Stream(dev)                             # an explicit stream shared by torch (RCCL) and the library:
    torch.cuda.set_stream(stream)                              # the legacy default stream would not order against it
    ctx = Context(local_rank, stream=stream.cuda_stream)       # library launches on / behind torch's current stream
    reg.set_context(ctx)

    def out_sets(CB, kmax, n):
        return types.SimpleNamespace(
            assoc=[torch.zeros((CB, kmax, 2), dtype=torch.int32, device=dev) for _ in range(n)],
            n=[torch.zeros(CB, dtype=torch.int32, device=dev) for _ in range(n)],
            T=[torch.zeros((CB, 16), dtype=torch.float64, device=dev) for _ in range(n)],
            status=[torch.zeros(CB, dtype=torch.int32, device=dev) for _ in range(n)],
            stats=[torch.zeros(CB * _abi.STATS_NBYTES, dtype=torch.uint8, device=dev) for _ in range(n)])

    def measure(workload, steps, warmup, profile, h2d=False):
        """Build one workload, keep it resident in HBM, run `warmup` untimed and `steps` timed steps (barrier +
        device synchronise on both sides, MAX over ranks).  h2d: the feature pool of every call is uploaded inside the
        timed region from pinned host memory on a copy stream (two device buffers: the upload of step k+1 overlaps the
        compute of step k)."""
        truth = None
        if workload == "pairs":
            B = args.batch
            pairs = [synth.make_pair(args.n, args.m, args.d, 3000 + rank * B + k) for k in range(B)]
            batch = batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
            truth = [p.inliers for p in pairs]
            mine = np.arange(B)                                    # every rank owns all problems of its own batch
            total_per_step = world * B
            chunk = B
            scaling = "weak"
            wl_text = (f"config 3: batch of {B} submap pairs per GPU, n={args.n} m={args.m} d={args.d}, method={args.method} "
                       f"(xyz + {args.d}-d descriptors + gravity prior); p50 latency measured on config 2 (single pair)")
        else:
            S = args.grid
            subs, _poses = synth.make_submap_grid(2 * S, n=args.n, d=args.d, seed0=4000)
            batch = batch_from_submap_grid(reg, subs[:S], subs[S:])   # S*S problems over ONE pool of 2S submaps
            from roman_amd.align.distributed import deal_by_cost, problem_work
            mine = deal_by_cost(problem_work(batch), world)[rank]   # the package's deal (longest first on A^2; equal sizes: round-robin)
            total_per_step = S * S
            chunk = min(args.chunk, max(len(x) for x in deal_by_cost(problem_work(batch), world)))   # the same on every rank (the gathered records are fixed-size)
            scaling = "strong"
            wl_text = (f"config 4: all-pairs grid of {S} x {S} submaps ({S * S} alignments), n={args.n} d={args.d}, method={args.method}; "
                       f"{2 * S} submaps packed once and replicated, pairs dealt by deal_by_cost to {world} rank(s), {chunk} pairs per call, all_gather of the records inside the timed region")
        kmax = batch.kmax()
        feats = torch.from_numpy(batch.feats).to(dev)
        calls = [mine[i:i + chunk] for i in range(0, len(mine), chunk)]          # problem indices of every call of a step
        meta = [(batch.off1[ix], batch.n1[ix], batch.off2[ix], batch.n2[ix]) for ix in calls]
        CB = chunk                                                 # rows of the output sets / gathered records

        # The timed loop is the PACKAGE's: roman_amd.align.pipeline.AlignStream — `pipeline` batch calls in flight, call c writes
        # output set c % nset, and with a process group the records of call c - 1 are all-gathered on the stream's own collector
        # stream while call c computes.  What this function adds is the workload and the clock.
        from roman_amd.align.pipeline import AlignStream
        S = AlignStream(reg, ctx, dev, rows=CB, kmax=kmax, in_flight=args.pipeline, stream=stream, use_group=bool(dist_on))
        NSET = S.nset
        O = types.SimpleNamespace(assoc=[x.assoc for x in S.sets], n=[x.n for x in S.sets], T=[x.T for x in S.sets],
                                  status=[x.status for x in S.sets], stats=[x.stats for x in S.sets])

        # h2d: pinned host copy of the pool, two device buffers, a copy stream
        if h2d:
            host = torch.from_numpy(batch.feats).pin_memory()
            dbuf = [torch.empty_like(feats) for _ in range(2)]
            cstream = torch.cuda.Stream(dev)
            ev_up = [torch.cuda.Event() for _ in range(2)]
            ev_free = [torch.cuda.Event() for _ in range(2)]
            up_no = [0]

        fptr_now = [feats.data_ptr()]

        class _Loop:                                               # one step = the calls of the workload, in order
            @staticmethod
            def step():
                for ci in range(len(calls)):
                    o1, a1, o2, a2 = meta[ci]
                    S.submit(fptr_now[0], F, o1, a1, o2, a2)
            drain = staticmethod(S.drain)
            reset = staticmethod(S.reset)
        loop = _Loop

        def step():
            fptr = feats.data_ptr()
            if h2d:                                                # this step's pool arrives over PCIe: buffer j, behind the last reader of j
                j = up_no[0] % 2
                if up_no[0] >= 2:
                    cstream.wait_event(ev_free[j])
                with torch.cuda.stream(cstream):
                    dbuf[j].copy_(host, non_blocking=True)
                    ev_up[j].record(cstream)
                stream.wait_event(ev_up[j])
                fptr = dbuf[j].data_ptr()
            fptr_now[0] = fptr
            loop.step()
            if h2d:
                ctx.join(skip_latest=False)                        # the calls of this step are ordered on `stream` before the buffer is reused
                ev_free[up_no[0] % 2].record(stream)
                up_no[0] += 1

        drain = loop.drain

        def fence():
            torch.cuda.synchronize(dev)
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize(dev)

        ctx.set_pipeline(args.pipeline)
        for _ in range(warmup):
            step()
        drain(); fence()
        loop.reset()
        skipped0 = ctx.skipped(wait=True)
        if profile:
            ctx.profile_enable(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain(); fence()
        dt = time.perf_counter() - t0
        dt_own = dt
        prof = ctx.profile_get() if profile else None
        if profile:
            ctx.profile_enable(False)
        skipped = ctx.skipped(wait=True) - skipped0                 # problems the timed steps reported ROMAN_ST_WORKSPACE for (must be 0)
        ctx.set_pipeline(1)                                         # the latency probe and the checks below are single calls
        if dist_on:
            tt = torch.tensor([dt, float(skipped)], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0].item()); skipped = int(tt[1].item())

        rank_ms = [dt_own / max(steps, 1) * 1e3]
        if dist_on:                                                 # every rank's own clock for the same region (imbalance of the deal)
            tr = torch.zeros(world, dtype=torch.float64, device=dev); tr[rank] = rank_ms[0]
            dist.all_reduce(tr, op=dist.ReduceOp.SUM)
            rank_ms = [float(x) for x in tr.cpu().tolist()]
        return types.SimpleNamespace(workload=workload, batch=batch, truth=truth, calls=calls, meta=meta, feats=feats, kmax=kmax, CB=CB, O=O,
                                     total_per_step=total_per_step, scaling=scaling, wl_text=wl_text, dt=dt, prof=prof, steps=steps, warmup=warmup,
                                     skipped=skipped, rank_ms=rank_ms, gathers=getattr(S, "gathers", 0))

    M = measure(workload, args.steps, args.warmup, not args.no_profile)
    batch, truth, calls, meta, feats, kmax, CB, O = M.batch, M.truth, M.calls, M.meta, M.feats, M.kmax, M.CB, M.O
    total_per_step, scaling, wl_text, dt, prof = M.total_per_step, M.scaling, M.wl_text, M.dt, M.prof

    # a longer timed region of the same loop (the contract's K steps can be 46 ms of GPU time): >= 0.5 s
    SUS = None
    if extras and world == 1:
        per = dt / max(args.steps, 1)
        n_long = int(min(max(np.ceil(0.6 / max(per, 1e-6)), args.steps), 2000))
        SUS = measure(workload, n_long, 2, False)
    # BASELINE config 4 as a second leg — the 4096-pair grid (dealt over the ranks at N > 1: strong scaling)
    G = None
    if workload == "pairs" and not args.no_grid and (extras or world > 1):
        G = measure("grid", max(args.grid_steps, 10) if world > 1 else args.grid_steps, max(args.grid_warmup, 3) if world > 1 else args.grid_warmup, False)
    # N > 1 with the grid as the headline: the weak per-rank loop (every rank its own 256 pairs) as a side key
    WK = None
    if workload == "grid" and world > 1 and args.workload == "auto":
        WK = measure("pairs", 20, 5, False)
    # upload inside the timed region
    H2D = None
    if extras and world == 1:
        H2D = measure(workload, max(6, min(args.steps, 12)), 2, False, h2d=True)

    # ---- the first call once more, alone: its results are what the checks below look at, its kernels what
    #      `isolated` times (one untimed launch first: the first launch from this thread at depth 1 pays one-time costs)
    o1, a1, o2, a2 = meta[0]
    C0 = len(calls[0])

    def call0():
        ctx.align_batch_dev(P, feats.data_ptr(), F, o1, a1, o2, a2, kmax,
                            O.assoc[0].data_ptr(), O.n[0].data_ptr(), O.T[0].data_ptr(), O.status[0].data_ptr(), O.stats[0].data_ptr())
    call0(); torch.cuda.synchronize(dev)
    iso_launch_ms = []
    iso = None
    if prof is not None and rank == 0:
        runs = []
        for _ in range(5):
            ctx.profile_enable(True); ctx.profile_reset()
            call0(); torch.cuda.synchronize(dev)
            g = ctx.profile_get(); ctx.profile_enable(False)
            iso_launch_ms.append(g["solve"][0])
            runs.append(g)
        # the MEAN of the five single launches (stage -> (ms summed, launches summed)): what rocprofv3's average duration is compared with
        iso = {k: (sum(r[k][0] for r in runs), sum(r[k][1] for r in runs)) for k in runs[0]}
    st = np.frombuffer(O.stats[0].cpu().numpy().tobytes(), dtype=stats_dtype())[:C0]
    n_sel = O.n[0].cpu().numpy()[:C0]; stat_h = O.status[0].cpu().numpy()[:C0]; a_h = O.assoc[0].cpu().numpy()[:C0]
    ok_frac = float(np.mean(stat_h == 0))
    rec = None
    if truth is not None:
        rec = []
        for b in range(C0):
            got = set(map(tuple, a_h[b, :n_sel[b]].tolist())); tr = set(map(tuple, truth[calls[0][b]].tolist()))
            rec.append(len(got & tr) / max(len(tr), 1))

    # ---- pose stage alone: the reference times only register() ([REF roman/align/submap_align.py:155-157]); T_align
    #      of the same correspondences through the stand-alone pose entry (host pointers, copies included: an upper bound)
    pose_ms = None
    if rank == 0:
        feats_h = batch.feats
        p1 = np.concatenate([feats_h[o1[b] + a_h[b, :n_sel[b], 0], :3] for b in range(C0)]) if C0 else np.zeros((0, 3))
        p2 = np.concatenate([feats_h[o2[b] + a_h[b, :n_sel[b], 1], :3] for b in range(C0)]) if C0 else np.zeros((0, 3))
        off = np.concatenate([[0], np.cumsum(n_sel)]).astype(np.int64)
        ctx.pose_batch(3, p1, p2, off)
        tp = []
        for _ in range(5):
            t1 = time.perf_counter(); ctx.pose_batch(3, p1, p2, off); tp.append(time.perf_counter() - t1)
        pose_ms = float(np.median(tp) * 1e3)

    # ---- p50 single-pair latency (config 2: B=1) ----------------------------------------------------
    p50 = None
    p50_dev = None
    p50_same = None
    lat_break = None
    if rank == 0 and args.latency_reps >= 0:          # --latency-reps -1: batched launches only (counter passes)
        lat = []
        for r in range(args.latency_reps + 3):
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            ctx.align_batch_dev(P, feats.data_ptr(), F, o1[:1], a1[:1], o2[:1], a2[:1], kmax,
                                O.assoc[1].data_ptr(), O.n[1].data_ptr(), O.T[1].data_ptr(), O.status[1].data_ptr(), O.stats[1].data_ptr())
            torch.cuda.synchronize(dev)
            if r >= 3:
                lat.append(time.perf_counter() - t1)
        p50_dev = float(np.median(lat) * 1e3)
        # the same single pair with the RESULT ON THE HOST (associations, count, pose, status, statistics: what the reference's caller
        # holds after register() + T_align(), [REF roman/align/submap_align.py:155-166]): roman_align_batch_resident — one enqueue,
        # one read-back of the 1.8 KB record through pinned memory, one wait
        lat_h = []
        for r in range(args.latency_reps + 3):
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            r1 = ctx.align_batch_resident(P, feats.data_ptr(), F, o1[:1], a1[:1], o2[:1], a2[:1], kmax)
            if r >= 3:
                lat_h.append(time.perf_counter() - t1)
        p50 = float(np.median(lat_h) * 1e3)
        p50_same = bool(np.array_equal(r1.assoc[0], a_h[0, :n_sel[0]]))
        # where a single pair's time goes: host enqueue time of the call, and the stages' device time (hipEvents)
        enq = []; stg = []
        for r in range(5):
            ctx.profile_enable(True); ctx.profile_reset()
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            ctx.align_batch_dev(P, feats.data_ptr(), F, o1[:1], a1[:1], o2[:1], a2[:1], kmax,
                                O.assoc[1].data_ptr(), O.n[1].data_ptr(), O.T[1].data_ptr(), O.status[1].data_ptr(), O.stats[1].data_ptr())
            enq.append(time.perf_counter() - t1)
            torch.cuda.synchronize(dev)
            g = ctx.profile_get(); ctx.profile_enable(False)
            stg.append({k: v[0] for k, v in g.items()})
        lat_break = {"host_enqueue_ms": float(np.median(enq) * 1e3),
                     "stage_ms": {k: float(np.median([x[k] for x in stg])) for k in stg[0]},
                     "note": "B=1, stage timers on (they add event records to the call)"}
    world_seen = dist.get_world_size() if dist_on else 1
    if dist_on:                                                 # the last collective: everything below is rank 0's own (CPU baseline, side legs)
        dist.barrier()
        dist.destroy_process_group()
        dist_on = False
    if rank != 0:
        return

    value = total_per_step * args.steps / dt
    ms_step = dt / args.steps * 1e3
    out = {
        "metric": "submap-pair alignments/sec + p50 latency at n=m=200 objects, d=512",
        "value": value, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64 (fixed-point SpMV sums)", "data": "synthetic",
        "dtype_note": "every gate and every stored value is IEEE f64; the stream solver's matrix-vector sums are 64-bit FIXED POINT "
                      "(each term rint(v*x*2^s), s = 48 - exponent(max x), added as integers: order-free, absolute error 2^-48 * max x "
                      "per term against 2^-53 relative for a double sum; u agrees with the oracle's double sums to <= 1e-9, selected sets "
                      "identical).  Line-search passes sum (v + d) * x in ONE accumulator (weight v + d, scale lowered by exponent(1 + d) + 1), "
                      "a split pass (M x and C x apart) precedes every d update: n_pass counts both, as SURVEY 8(d) does, and the oracle states the "
                      "same order (oracle_set_pass_mode); the large-live-set solver sums plain doubles in a fixed order and keeps both sums of every pass",
        "config": {"workload": wl_text, "alignments_per_step": total_per_step, "pairs_per_call": CB, "calls_per_step_per_gpu": len(calls),
                   "n": args.n, "m": args.m, "d": args.d, "method": args.method,
                   "sharding": f"pairs x{world}, all_gather of records" if world > 1 else "single GPU",
                   "batches_in_flight": args.pipeline},
        "timed_region_s": dt,
        "skipped_in_timed_steps": M.skipped,
        "ranks": {"rccl_ranks_seen": world_seen, "ms_per_step_by_rank": M.rank_ms, "all_gathers_per_call": (M.gathers / max((args.steps + args.warmup) * len(calls), 1)) if M.gathers else 0},
        "weak_pairs": None if WK is None else {"value": WK.total_per_step * WK.steps / WK.dt, "unit": "alignments/s", "scaling": "weak", "steps": WK.steps,
                                               "ms_per_step": WK.dt / WK.steps * 1e3, "workload": "config 3: every rank its own 256 pairs per step"},
        "p50_latency_ms": p50, "p50_device_ms": p50_dev,
        "p50_note": "single pair (config 2 shape, B=1), inputs resident: p50_latency_ms = roman_align_batch_resident, result ON THE HOST "
                    f"(same associations as the batched call: {p50_same}); p50_device_ms = enqueue + device sync, outputs left in HBM",
        "sustained": None if SUS is None else {
            "value": SUS.total_per_step * SUS.steps / SUS.dt, "steps": SUS.steps, "timed_region_s": SUS.dt, "ms_per_step": SUS.dt / SUS.steps * 1e3,
            "skipped_in_timed_steps": SUS.skipped, "note": "the same loop timed over >= 0.5 s (the contract's K steps above can be tens of milliseconds)"},
        "value_incl_h2d": None if H2D is None else {
            "value": H2D.total_per_step * H2D.steps / H2D.dt, "unit": "alignments/s", "steps": H2D.steps, "ms_per_step": H2D.dt / H2D.steps * 1e3,
            "bytes_uploaded_per_step": int(batch.feats.nbytes),
            "note": "every step's feature pool (2 maps x 200 objects x 515 doubles per pair: independent pairs share nothing) uploaded INSIDE the "
                    "timed region from pinned memory on a copy stream, double-buffered against the compute; PCIe-bound. The all-pairs grid "
                    "uploads each submap once: see grid_config4.incl_h2d_estimate"},
        "grid_config4": None if G is None else {
            "value": G.total_per_step * G.steps / G.dt, "unit": "alignments/s", "scaling": "strong", "steps": G.steps, "warmup": G.warmup,
            "ms_per_step": G.dt / G.steps * 1e3, "timed_region_s": G.dt, "workload": G.wl_text, "calls_per_step_per_gpu": len(G.calls),
            "skipped_in_timed_steps": G.skipped,
            "status_ok_frac": float(np.mean(np.concatenate([x.cpu().numpy() for x in G.O.status]) == 0)),     # the last calls' output sets
            "incl_h2d_estimate": {"pool_bytes": int(G.batch.feats.nbytes),
                                  "note": "the grid's 128 submaps are uploaded once per grid (105 MB for 4096 alignments: < 2 ms at PCIe rates against the step time)"},
            "note": "second leg of this run, same timing rules (barrier + device synchronise, MAX over ranks)"},
        "latency_breakdown": lat_break,
        "alignments_per_s_batch1": (1e3 / p50) if p50 else None, "alignments_per_s_batch1_device_only": (1e3 / p50_dev) if p50_dev else None,
        "register_only": {"note": "the reference times register() alone; the pose is fused into the solver kernel's tail here, so the split is "
                                  "measured as the stand-alone pose entry on the same correspondences (host-pointer call, copies included: an upper bound)",
                          "t_align_ms_per_call": pose_ms, "pairs_per_call": C0,
                          "register_only_ms_per_step_lower_bound": (ms_step - pose_ms * len(calls)) if pose_ms is not None else None},
        "result_check": {"status_ok_frac": ok_frac, "planted_inlier_recall_mean": (float(np.mean(rec)) if rec else None),
                         "mean_live": float(st["n_live"].mean()), "mean_nnz_upper": float(st["nnz_upper"].mean()), "mean_passes": float(st["n_pass"].mean()),
                         "max_passes": int(st["n_pass"].max())},
    }
    # ---- roofline ---------------------------------------------------------------------------------------
    if prof is not None:
        out["stage_ms_per_call"] = {k: v[0] / max(v[1], 1) for k, v in prof.items()}
        dom = max(prof, key=lambda k: prof[k][0])
        solve_ms, solve_n = prof["solve"]
        Lb = st["n_live"].astype(np.float64); nnz = st["nnz_upper"].astype(np.float64); npass = st["n_pass"].astype(np.float64)
        alg_bytes = float(np.sum(npass * (12.0 * nnz + 24.0 * Lb)))
        if solve_n > 0 and solve_ms > 0:
            avg_s = solve_ms / solve_n * 1e-3
            ach = alg_bytes / avg_s / 1e9
            traffic, tsrc = None, None                          # HBM bytes per launch: from a committed rocprofv3 PMC pass of THIS workload
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                    pj = json.load(fh)
                if pj.get("kernel") == "k_solve_up" and workload == "pairs" and C0 == 256 and (args.n, args.m, args.d) == (200, 200, 512):
                    traffic = pj["traffic_bytes_per_launch"]; tsrc = pj.get("source", "profiles/pmc_traffic.json")
            except (OSError, ValueError, KeyError):
                pass
            out["roofline"] = {"kernel": "k_solve_up", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": (f"NOT measured in this run: committed rocprofv3 PMC pass ({tsrc}), 2 x FETCH_SIZE + WRITE_SIZE per launch" if traffic else None),
                               "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": solve_ms / solve_n,
                               "launches_timed": int(solve_n), "timing": "hipEvents on the stream the kernel runs on, with the other batches in flight",
                               "dominant_stage_by_time": dom}
            if iso is not None and iso["solve"][1] > 0:         # launch duration with no other batch in flight
                iso_ms = iso["solve"][0] / iso["solve"][1]
                iso_stage = {k: v[0] / max(v[1], 1) for k, v in iso.items()}
                out["roofline"]["isolated"] = {"avg_launch_ms": iso_ms, "per_launch_ms": iso_launch_ms,
                                               "note": "MEAN of 5 single launches after one untimed launch.  ALGORITHMIC credit (SURVEY 8(d): a full matrix stream per pass) over the "
                                                       "launch duration: it can reach and pass 1.0 of the HBM peak because the narrow passes re-read two slices out of L2 and the wide "
                                                       "passes of 256 problems (197 MB per pass) are served by the 256 MB infinity cache at up to 9.7 TB/s; the bytes the memory side "
                                                       "really moved are traffic_frac",
                                               "achieved": alg_bytes / (iso_ms * 1e-3) / 1e9,
                                               "frac": alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "stage_ms_per_call": iso_stage}
                if traffic:                                     # what the memory side really moved in that time (the algorithmic figure charges a full
                    out["roofline"]["traffic_frac"] = traffic / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS    # matrix stream per pass; narrow passes re-read two slices from L2)
                # the other kernels against THEIR bounds, from the isolated stage times (a stage = a few kernels; the named one dominates it)
                tests = float(np.sum(Lb * (Lb - 1.0) / 2.0))
                cosflops = 2.0 * float(np.sum(a1[:C0].astype(np.float64) * a2[:C0])) * args.d
                out["roofline"]["kernels"] = {
                    "k_count (stage 'count' = pair tests + mirror + sort + lists)": {
                        "bound": "f64 VALU", "pair_tests": tests, "flops_at_30_per_test": 30.0 * tests, "stage_ms": iso_stage["count"],
                        "achieved_TFLOPs": 30.0 * tests / (iso_stage["count"] * 1e-3) / 1e12, "peak_TFLOPs": F64_PEAK_TFLOPS,
                        "frac": 30.0 * tests / (iso_stage["count"] * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                        "note": "frac of the whole stage (k_count + k_lists + scans) on 30 flop per LIVE pair: algorithmic credit — since round 6 k_count tests every pair as 15-bit bins and only ~6 % in f64 (profiles/r06: k_count 0.36 of the stage's 0.55 ms)"},
                    "k_cos (stage 'single' = cosine MFMA + tables + single scores + live list)": {
                        "bound": "MFMA f64", "flops": cosflops, "stage_ms": iso_stage["single"],
                        "achieved_TFLOPs": cosflops / (iso_stage["single"] * 1e-3) / 1e12, "peak_TFLOPs": F64_PEAK_TFLOPS,
                        "measured_mfma_ceiling_TFLOPs": F64_MFMA_MEASURED_TFLOPS,
                        "frac": cosflops / (iso_stage["single"] * 1e-3) / 1e12 / F64_PEAK_TFLOPS,
                        "note": "frac of the whole stage on the f64 flops of all n1 x n2 products: algorithmic credit — since round 6 the products are screened on the bf16 matrix core and only the ~5 % that can pass the gate are contracted in f64 (k_cos_sel 0.23 of the stage's 0.34 ms; the dense k_cos_deal took 0.30)"}}
                # the whole step against SURVEY.md §8(d)'s ideal time t* = W/Pi + (B_b + B_s)/beta (pair tests among LIVE associations)
                Wb = 30.0 * tests + cosflops
                Bb = float(np.sum(12.0 * nnz)) + 8.0 * float(np.sum(a1[:C0].astype(np.float64) + a2[:C0])) * F
                t_star_ms = (Wb / (F64_PEAK_TFLOPS * 1e12) + (Bb + alg_bytes) / (HBM_PEAK_GBS * 1e9)) * 1e3
                out["roofline"]["step"] = {"t_star_ms": t_star_ms, "ms_per_step": ms_step, "frac": t_star_ms / ms_step,
                                           "build_flops": Wb, "build_bytes": Bb, "solver_bytes": alg_bytes,
                                           "note": "SURVEY.md §8(d): t* = W_b / 78.6 TF + (B_b + B_s) / 8 TB/s for one call of the batch; frac = t* / measured ms per step"}

    # roofline.traffic LIVE: the PMC passes of this box and this build (two short re-runs of the main measurement under rocprofv3)
    if "roofline" in out and extras and world == 1 and not args.no_live_traffic and workload == "pairs" and C0 == 256 and (args.n, args.m, args.d) == (200, 200, 512):
        tb, raw = live_traffic()
        r_ = out["roofline"]
        if tb is not None:
            r_["traffic"] = tb; r_["traffic_measured_live"] = True; r_["traffic_raw"] = raw
            r_["traffic_source"] = "MEASURED in this run: bench.py re-run under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 2 x FETCH_SIZE + WRITE_SIZE per k_solve_up launch"
            iso_ms_ = (r_.get("isolated") or {}).get("avg_launch_ms")
            if iso_ms_:
                r_["traffic_frac"] = tb / (iso_ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS
        else:
            r_["traffic_measured_live"] = False; r_["traffic_live_error"] = str(raw)[:300]
    from_oracle = args.cpu_sample > 0                           # rank 0 at every N (the other ranks have left; nothing of it is inside a timed region)
    orc = None
    if from_oracle or (extras and world == 1):
        from oracle import oracle as orc
    # ---- CPU baseline: the oracle on this box's host cores, bounded samples ----------------------------
    if from_oracle:
        fh_ = batch.feats

        def mats(b):
            return fh_[o1[b]:o1[b] + a1[b]], fh_[o2[b]:o2[b] + a2[b]]

        def run(b, faithful):
            D1, D2 = mats(b)
            r = orc.register(P, D1, D2, faithful=faithful)
            if len(r["assoc"]) >= 3:
                orc.t_align(D1[r["assoc"][:, 0], :3], D2[r["assoc"][:, 1], :3])
            return r["assoc"]
        nthr = orc.num_threads()
        S = min(args.cpu_sample, C0)
        tf0 = time.perf_counter()
        for b in range(S):
            run(b, True)
        tf = time.perf_counter() - tf0
        NC = min(args.check_pairs, C0)                          # every problem of the call, pruned mode: also the result check
        tp0 = time.perf_counter(); same = 0
        for b in range(NC):
            same += int(np.array_equal(run(b, False), a_h[b, :n_sel[b]]))
        tp = time.perf_counter() - tp0
        orc.set_threads(1)
        S1 = min(2, C0)
        t10 = time.perf_counter()
        for b in range(S1):
            run(b, False)
        t1 = time.perf_counter() - t10
        orc.set_threads(nthr)
        # the same pruned oracle with ONE thread per pair and all pairs of the call side by side: what a CPU caller with
        # many pairs and many cores would do (the reference's loop is serial; this is the box's throughput ceiling)
        pp = None
        try:
            tq0 = time.perf_counter()
            many = orc.register_many(P, fh_, o1[:NC], a1[:NC], o2[:NC], a2[:NC], kmax, faithful=False)
            for b in range(NC):
                if len(many[b]) >= 3:
                    D1, D2 = mats(b)
                    orc.t_align(D1[many[b][:, 0], :3], D2[many[b][:, 1], :3])
            tq = time.perf_counter() - tq0
            pp = {"value": NC / tq, "sample": f"{NC} pairs, pruned mode, one OpenMP thread per pair, {nthr} threads side by side, + numpy T_align",
                  "identical_to_gpu": bool(all(np.array_equal(many[b], a_h[b, :n_sel[b]]) for b in range(NC)))}
        except Exception as e:                                  # a reported extra: never lose the line over it
            pp = {"error": repr(e)}
        out["result_check"]["oracle_identical"] = f"{same}/{NC}"
        out["result_check"]["oracle_mode"] = "carried passes (findDenseClique as published), stated-order arithmetic; the GPU suite also checks plain libm arithmetic"
        out["cpu_baseline"] = {"value": S / tf, "unit": "alignments/s", "cores": nthr, "kind": "port",
                               "sample": f"{S} of the {C0} pairs of one call; oracle/clipper_oracle.c (C, OpenMP; a restatement, not the upstream binary) in "
                                         f"upstream-like mode: all A(A-1)/2 association pairs scored, + numpy T_align",
                               "value_pruned": NC / tp, "pruned_sample": f"{NC} pairs, same oracle skipping associations whose single score is 0 (identical results), {nthr} threads",
                               "value_pruned_1thread": S1 / t1, "one_thread_sample": f"{S1} pairs, pruned mode, 1 thread",
                               "pruned_pair_parallel": pp,
                               "identical_to_gpu": bool(same == NC), "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
        out["speedup_vs_cpu_baseline"] = {"vs_pruned": value / (NC / tp),
                                          "vs_pruned_pair_parallel": (value / pp["value"]) if pp and "value" in pp else None,
                                          "note": "GPU and CPU both skip associations whose single score is 0 here (like for like); a reported baseline, "
                                                  "not a target: the roofline fraction says how good the kernels are"}

    if extras and world == 1:
        try:
            caller_legs(out, args, reg, ctx, dev, stream, G, orc, from_oracle)
        except Exception as e:                                  # side legs never cost the headline line
            out["caller_legs_error"] = repr(e)
        try:
            side_legs(out, args, ctx, dev, G, orc, from_oracle)
        except Exception as e:                                  # side legs never cost the headline line
            out["side_legs_error"] = repr(e)
    emit(out)


def caller_legs(out, args, reg, ctx, dev, stream, G, orc, with_cpu):
    """What a CALLER of the package gets, outside the steady state of the timed loop (round-4 review, items 2, 3, 6):
      one_shot          ONE config-4 grid (4096 pairs) through roman_amd.align.pipeline.align_resident — calls of `chunk` pairs,
                        three in flight, skipped problems issued again — wall time from the first enqueue to the results on the host;
                        and the same grid through run_batch (host arrays in: roman_align_batch's own chunking, upload included);
      cold_call         the same call as the FIRST call of a fresh context (no sizing history, pools not allocated);
      varying           the timed loop with every call a DIFFERENT 256-pair subset of the grid (16 distinct batches);
      scale_projection  the 8-GPU strong-scaling number of config 4 projected on this one GPU: rank r's share of the deal (512 pairs),
                        one shot, for every r, against the one-shot time of the whole grid;
      dropin            the reference's serial loop — register() then T_align() per pair ([REF roman/align/submap_align.py:155-166]) —
                        through the package's stepwise path (host objects in, ndarray out), beside the same loop on the CPU oracle;
      config1 / p50     BASELINE config 1 (n = m = 30, xyz only) and config 2 (seed 2000) as single calls with the result read back."""
    import torch
    from roman_amd import _abi, synth
    from roman_amd.align import SubmapAlignParams
    from roman_amd.align import batch as rb
    from roman_amd.align.pipeline import AlignStream, align_resident
    from roman_amd.align.distributed import take
    from roman_amd.runtime import Context
    legs = {}
    if G is not None:
        gb, pool = G.batch, G.feats
        B = len(gb)
        steady = G.total_per_step * G.steps / G.dt
        # ---- one shot -------------------------------------------------------------------------------------------
        rows = []
        ref = None
        for chunk in (256, 512, 1024, 2048):
            align_resident(reg, gb, pool, chunk=chunk, in_flight=3, ctx=ctx)          # (history and pools of this shape in place)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                res = align_resident(reg, gb, pool, chunk=chunk, in_flight=3, ctx=ctx)
                ts.append(time.perf_counter() - t0)
            if ref is None:
                ref = res
            rows.append({"chunk": chunk, "in_flight": 3, "ms": float(np.median(ts) * 1e3), "alignments_per_s": B / float(np.median(ts)),
                         "vs_steady_state": B / float(np.median(ts)) / steady,
                         "identical_to_chunk_256": bool(all(np.array_equal(a, b) for a, b in zip(res.assoc, ref.assoc)) and np.array_equal(res.T, ref.T, equal_nan=True))})
        th = []
        for _ in range(2):
            t0 = time.perf_counter(); resh = rb.run_batch(reg, gb); th.append(time.perf_counter() - t0)
        legs["one_shot"] = {"workload": f"config 4: ONE grid of {B} pairs, inputs resident, wall time from the first enqueue to the results on the host "
                                        "(roman_amd.align.pipeline.align_resident: device outputs of every call read back at the end)",
                            "rows": rows, "steady_state_alignments_per_s": steady,
                            "host_arrays_in": {"ms": float(min(th) * 1e3), "alignments_per_s": B / min(th), "pool_bytes": int(gb.feats.nbytes),
                                               "identical": bool(all(np.array_equal(a, b) for a, b in zip(resh.assoc, ref.assoc))),
                                               "note": "run_batch(): roman_align_batch chunks and pipelines the batch itself (2048 pairs per call, three in flight); upload of the pool and read-back included"}}
        # ---- cold call ------------------------------------------------------------------------------------------
        c2 = Context(dev.index if dev.index is not None else 0)
        try:
            reg.set_context(c2)
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            resc = align_resident(reg, gb, pool, chunk=512, in_flight=3, ctx=c2)
            tc = time.perf_counter() - t0
            t0 = time.perf_counter()
            align_resident(reg, gb, pool, chunk=512, in_flight=3, ctx=c2)
            tc2 = time.perf_counter() - t0
            sk_cold = c2.skipped(wait=True)
        finally:
            reg.set_context(ctx); c2.close()
        legs["cold_call"] = {"workload": f"the same {B}-pair grid as the FIRST call of a fresh context (no sizing history: the first 512-pair call is waited for; no pool allocated)",
                             "first_call_ms": tc * 1e3, "second_call_ms": tc2 * 1e3, "problems_skipped_and_reissued_in_both_calls": int(sk_cold),
                             "identical": bool(all(np.array_equal(a, b) for a, b in zip(resc.assoc, ref.assoc)))}
        # ---- varying inputs in the timed loop ----------------------------------------------------------------------
        rng = np.random.default_rng(77)
        perm = rng.permutation(B)
        NB = 16
        subsets = [np.sort(perm[k * 256:(k + 1) * 256]) for k in range(NB)]
        metas = [(gb.off1[ix], gb.n1[ix], gb.off2[ix], gb.n2[ix]) for ix in subsets]
        kmax = gb.kmax()
        Fg = gb.feats.shape[1]
        S = AlignStream(reg, ctx, dev, rows=256, kmax=kmax, in_flight=args.pipeline, stream=stream)
        try:
            def loop(which, steps):
                for k in range(steps):
                    o1, a1, o2, a2 = metas[which(k)]
                    S.submit(pool.data_ptr(), Fg, o1, a1, o2, a2)
                S.drain(); torch.cuda.synchronize(dev)
            res_rates = {}
            for name, which in (("same_batch_every_step", lambda k: 0), ("different_batch_every_step", lambda k: k % NB)):
                loop(which, 2 * NB)
                sk0 = ctx.skipped(wait=True)
                steps = 4 * NB
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                loop(which, steps)
                dtv = time.perf_counter() - t0
                res_rates[name] = {"alignments_per_s": 256 * steps / dtv, "ms_per_step": dtv / steps * 1e3, "steps": steps,
                                   "skipped_in_timed_steps": ctx.skipped(wait=True) - sk0}
        finally:
            S.close()
        legs["varying"] = {"workload": f"the timed loop (256 pairs per call, {args.pipeline} calls in flight) over the grid's pool: {NB} DISTINCT 256-pair subsets of the "
                                       f"{B} grid pairs, one per step, against one subset repeated",
                           **res_rates,
                           "ratio": res_rates["different_batch_every_step"]["alignments_per_s"] / res_rates["same_batch_every_step"]["alignments_per_s"]}
        # ---- 8-GPU projection of config 4 on one GPU -----------------------------------------------------------------
        WORLD = 8
        t_all = min(r["ms"] for r in rows)                                       # the whole grid, one shot, at its best call size (pipeline.default_chunk: 2 x 2048)
        proj = []
        for chunk in (512, 256):                                                 # a rank's 512 pairs as ONE call (pipeline.default_chunk) / as two calls in flight
            per_rank = []
            for r in range(WORLD):
                sub = take(gb, np.arange(r, B, WORLD))                         # bench.py's deal of the grid: round-robin
                align_resident(reg, sub, pool, chunk=chunk, in_flight=3, ctx=ctx, stats=False)
                ts = []
                for _ in range(2):
                    torch.cuda.synchronize(dev); t0 = time.perf_counter()
                    align_resident(reg, sub, pool, chunk=chunk, in_flight=3, ctx=ctx, stats=False)
                    ts.append(time.perf_counter() - t0)
                per_rank.append(min(ts) * 1e3)
            gather_ms = 0.05                                                  # one all_gather pair of 512 x (2 + 2 kmax) int32 + 512 x 16 f64 per rank (0.9 MB): tens of microseconds over xGMI
            proj.append({"pairs_per_rank": B // WORLD, "chunk": chunk, "rank_ms": per_rank, "assumed_gather_ms": gather_ms,
                         "projection": [t_all / (t + gather_ms) for t in per_rank], "min_projection": t_all / (max(per_rank) + gather_ms)})
        legs["scale_projection"] = {"what": f"T(one shot, {B} pairs, one GPU) / (T(one shot, rank r's {B // WORLD} pairs, one GPU) + gather) for the 8 ranks of the round-robin deal: the strong-scaling "
                                            "factor of ONE grid at 8 GPUs if every rank behaves like this GPU (no 8-GPU node is reachable from the build container; the gather is an assumed figure)",
                                    "one_gpu_one_shot_ms": t_all, "rows": proj,
                                    "note": "a rank's share lasts at least as long as its slowest problem (a problem runs on ONE compute unit: the grid's longest takes "
                                            f"{int(ref.stats['n_pass'].max())} passes): the steady-state loop of `grid_config4` hides that tail behind the next grid, a single grid cannot"}
    # ---- the reference's serial loop through the package's stepwise path ---------------------------------------------
    def dropin(name, method, n_lo, n_hi, d, seed0, K, kw):
        sp = SubmapAlignParams(method=method, **kw)
        r = sp.get_object_registration(); r.set_context(ctx)
        rng = np.random.default_rng(seed0)
        prs = [synth.make_pair(int(rng.integers(n_lo, n_hi + 1)), int(rng.integers(n_lo, n_hi + 1)), d, seed0 + k,
                               tilt_deg=1.0 if r._abi_params().gravity_guided else 0.0) for k in range(K)]
        t_reg, t_pose, assoc = [], [], []
        for rep in range(2):                                       # (the first round allocates and sizes)
            t_reg, t_pose, assoc = [], [], []
            for pr in prs:
                t0 = time.perf_counter(); a = r.register(pr.map1, pr.map2); t1 = time.perf_counter()
                try:
                    r.T_align(pr.map1, pr.map2, a)
                except Exception:
                    pass
                t2 = time.perf_counter()
                t_reg.append(t1 - t0); t_pose.append(t2 - t1); assoc.append(a)
        t_one = []; same_one = 0
        for rep in range(2):
            t_one = []; same_one = 0
            for pr, a in zip(prs, assoc):
                t0 = time.perf_counter(); r1 = r.register_and_align_batch([(pr.map1, pr.map2)]); t_one.append(time.perf_counter() - t0)
                same_one += int(np.array_equal(np.asarray(a).reshape(-1, 2), r1.assoc[0]))
        row = {"workload": name, "pairs": K, "register_ms_per_pair": float(np.median(t_reg) * 1e3), "t_align_ms_per_pair": float(np.median(t_pose) * 1e3),
               "ms_per_pair": float((np.median(t_reg) + np.median(t_pose)) * 1e3),
               "one_call_per_pair": {"ms_per_pair": float(np.median(t_one) * 1e3), "identical": f"{same_one}/{K}",
                                     "note": "the same serial loop with register() + T_align() of a pair replaced by ONE registration.register_and_align_batch([(map1, map2)]) "
                                             "(the batch entry with B = 1: one upload, one enqueue, one read-back; demo-size pairs take the one-kernel path k_small)"},
               "note": "registration.register(map1, map2) then registration.T_align(map1, map2, associations), pair after pair, as the reference's loop does: per pair the Python "
                       "feature packing, the upload, roman_score / roman_solve / the getters and roman_pose_batch"}
        if with_cpu:
            Pm = r._abi_params()
            tc = []; same = 0
            for pr, a in zip(prs, assoc):
                t0 = time.perf_counter()
                D1, D2 = r.pack(pr.map1), r.pack(pr.map2)
                o = orc.register(Pm, D1, D2, A=r._association_list(pr.map1, pr.map2))
                if len(o["assoc"]) >= 3:
                    orc.t_align(D1[o["assoc"][:, 0], :3], D2[o["assoc"][:, 1], :3])
                tc.append(time.perf_counter() - t0)
                same += int(np.array_equal(np.asarray(a).reshape(-1, 2), o["assoc"]))
            row["cpu_oracle_loop"] = {"ms_per_pair": float(np.median(tc) * 1e3), "cores": orc.num_threads(), "kind": "port", "identical": f"{same}/{K}",
                                      "sample": f"the same {K} pairs, same packing, oracle register() (associations whose single score is 0 skipped) + numpy T_align, all host threads inside a pair"}
        return row
    legs["dropin"] = [dropin(f"config 2 shape: n = m = {args.n}, d = {args.d}, method {args.method}", args.method, args.n, args.n, args.d, 2000, 6,
                             {"semantics_dim": args.d} if args.d > 0 else {}),
                      dropin("demo scale: n, m in [20, 40], d = 768, method 'roman' ([REF params/demo/submap_align.yaml])", "roman", 20, 40, 768, 5200, 24, {"semantics_dim": 768})]
    # ---- BASELINE config 1 and config 2 as single calls with the result on the host --------------------------------------
    def single(method, n, d, seed, kw, reps=20):
        r = SubmapAlignParams(method=method, **kw).get_object_registration(); r.set_context(ctx)
        pr = synth.make_pair(n, n, d, seed, tilt_deg=1.0 if r._abi_params().gravity_guided else 0.0)
        bt = rb.batch_from_pairs(r, [(pr.map1, pr.map2)])
        poolp = torch.from_numpy(bt.feats).to(dev)
        align_resident(r, bt, poolp, chunk=1, in_flight=1, ctx=ctx)
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            res = align_resident(r, bt, poolp, chunk=1, in_flight=1, ctx=ctx)
            ts.append(time.perf_counter() - t0)
        row = {"p50_ms": float(np.median(ts) * 1e3), "selected": int(len(res.assoc[0])), "passes": int(res.stats["n_pass"][0]), "live": int(res.stats["n_live"][0]),
               "note": "inputs resident, ONE pair per call, associations / pose / status / statistics read back to the host inside the timed region"}
        if with_cpu:
            D1, D2 = r.pack(pr.map1), r.pack(pr.map2)
            cpu = {}
            nthr = orc.num_threads()
            for thr in (1, nthr):
                orc.set_threads(thr)
                tt = []
                for _ in range(3):
                    t0 = time.perf_counter(); o = orc.register(r._abi_params(), D1, D2)
                    if len(o["assoc"]) >= 3:
                        orc.t_align(D1[o["assoc"][:, 0], :3], D2[o["assoc"][:, 1], :3])
                    tt.append(time.perf_counter() - t0)
                cpu[f"{thr}_threads_ms"] = float(np.median(tt) * 1e3)
            orc.set_threads(nthr)
            cpu["identical"] = bool(np.array_equal(o["assoc"], res.assoc[0])); cpu["kind"] = "port"
            row["cpu_oracle"] = cpu
        return row
    legs["config1"] = {"workload": "BASELINE config 1: 2 synthetic submaps, 30 objects each, xyz centroids only (method 'clipper'), seed 1000", **single("clipper", 30, 0, 1000, {})}
    legs["config2_p50"] = {"workload": f"BASELINE config 2: n = m = {args.n}, d = {args.d}, method {args.method}, seed 2000",
                           **single(args.method, args.n, args.d, 2000, {"semantics_dim": args.d} if args.d > 0 else {})}
    out["caller"] = legs


def side_legs(out, args, ctx, dev, G, orc, with_cpu):
    """grid check against the oracle, decision sensitivity, random starts, the large-live-set path, the demo scale."""
    import torch
    from roman_amd import _abi, synth
    from roman_amd.align import SubmapAlignParams
    from roman_amd.align import batch as rb
    from roman_amd.runtime import stats_dtype

    def sets(res):
        return [frozenset(map(tuple, a.tolist())) for a in res.assoc]

    def timed_batch(reg, batch, reps=3, u0=None):
        """host-pointer entry (copies included) — used for the side legs only; -> (BatchResult, best seconds)"""
        best, res = None, None
        for _ in range(reps):
            t0 = time.perf_counter(); res = rb.run_batch(reg, batch, u0=u0); t = time.perf_counter() - t0
            best = t if best is None or t < best else best
        return res, best

    # ---- config 4: EVERY one of the 4096 grid pairs against the oracle (pair-parallel, ~30 s on the GPU box's host cores) ----------
    if G is not None and with_cpu:
        reg = SubmapAlignParams(method=args.method, semantics_dim=args.d).get_object_registration(); reg.set_context(ctx)
        res = rb.run_batch(reg, G.batch)
        pick = np.arange(len(G.batch))                           # every pair of the grid (one oracle thread per pair)
        kmax = G.batch.kmax()
        t0 = time.perf_counter()
        many = orc.register_many(reg._abi_params(), G.batch.feats, G.batch.off1[pick], G.batch.n1[pick], G.batch.off2[pick], G.batch.n2[pick], kmax, faithful=False)
        tq = time.perf_counter() - t0
        same = sum(int(np.array_equal(many[k], res.assoc[b])) for k, b in enumerate(pick))
        out["grid_config4"]["oracle_check"] = {"pairs_compared": int(len(pick)), "identical": int(same), "max_passes_in_grid": int(res.stats["n_pass"].max()),
                                               "includes": "every pair of the 64 x 64 grid",
                                               "oracle_seconds": tq}

    # ---- decision sensitivity: the readings of the absent clipperpy formulas and the start vector (H1) -------
    NP = 32
    pairs = [synth.make_pair(args.n, args.m, args.d, 3000 + k) for k in range(NP)]
    base_sets = None
    rows = []
    names_s = {0: "BOTH", 1: "OFFDIAG", 2: "DIAG", 3: "DIAG_KEEP"}; names_g = {0: "COMBINED", 1: "SEPARATE", 2: "ZGATE"}
    for sm in (0, 1, 2, 3):
        for gm in (0, 1, 2):
            reg = SubmapAlignParams(method=args.method, semantics_dim=args.d).get_object_registration(); reg.set_context(ctx)
            Pm = reg._abi_params(); Pm.single_mode = sm; Pm.gravity_mode = gm
            npairs = NP if sm != 3 else 4                      # DIAG_KEEP: every association live (L = 40 000): the large-live-set path
            bt = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs[:npairs]])
            res, secs = timed_batch(reg, bt, reps=2 if sm == 3 else 3)
            s = sets(res)
            if base_sets is None:
                base_sets = s
            truth = [frozenset(map(tuple, p.inliers.tolist())) for p in pairs[:npairs]]
            rows.append({"single_mode": names_s[sm], "gravity_mode": names_g[gm], "pairs": npairs,
                         "mean_live": float(res.stats["n_live"].mean()), "mean_nnz_upper": float(res.stats["nnz_upper"].mean()),
                         "mean_passes": float(res.stats["n_pass"].mean()),
                         "alignments_per_s": npairs / secs,
                         "selected_sets_differing_from_default": int(sum(a != b for a, b in zip(s, base_sets[:npairs]))),
                         "planted_inlier_recall_mean": float(np.mean([len(a & t) / max(len(t), 1) for a, t in zip(s, truth)])),
                         "status_ok": int((res.status == 0).sum())})
    out["decision_sensitivity"] = {
        "note": f"{NP} config-3 pairs (seeds 3000..) per reading, ONE host-pointer call each (upload + readback included, so the rates are "
                "below the pipelined headline: compare them with the first row); DIAG_KEEP keeps associations whose single score is 0 "
                "(L = A = 40 000) and runs 4 pairs on the large-live-set solver", "rows": rows}
    # random starts (upstream's default) against the all-ones start (this tree's default)
    reg = SubmapAlignParams(method=args.method, semantics_dim=args.d).get_object_registration(); reg.set_context(ctx)
    bt = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    nA = int(args.n * args.m)
    K = 8
    eq = 0; rec_min = 1.0
    truth = [frozenset(map(tuple, p.inliers.tolist())) for p in pairs]
    for k in range(K):
        u0 = np.concatenate([np.random.default_rng(900000 + 1000 * k + b).random(nA) for b in range(NP)])
        s = sets(rb.run_batch(reg, bt, u0=u0))
        eq += sum(int(a == b) for a, b in zip(s, base_sets))
        rec_min = min(rec_min, min(len(a & t) / max(len(t), 1) for a, t in zip(s, truth)))
    out["u0_stability"] = {"pairs": NP, "random_starts_per_pair": K, "fraction_selecting_the_all_ones_set": eq / (K * NP),
                           "min_planted_inlier_recall_over_all_starts": rec_min,
                           "note": "upstream solve() draws u0 ~ U[0,1) from std::random_device; this tree starts from all ones (decision H1); "
                                   "tests/test_u0_stability.py checks GPU == oracle for every start"}

    # ---- the large-live-set path: method 'gravity' (no semantic gate), every association live -------------------------------
    ll = []
    for n in (100, 200):
        reg = SubmapAlignParams(method="gravity").get_object_registration(); reg.set_context(ctx)
        pr = synth.make_pair(n, n, 0, 7001, tilt_deg=1.0)
        bt = rb.batch_from_pairs(reg, [(pr.map1, pr.map2)])
        rb.run_batch(reg, bt)                                   # sizes the pools
        best = None
        for _ in range(3):
            ctx.profile_enable(True); ctx.profile_reset()
            t0 = time.perf_counter(); res = rb.run_batch(reg, bt); t = time.perf_counter() - t0
            pf = ctx.profile_get(); ctx.profile_enable(False)
            if best is None or t < best[0]:
                best = (t, pf, res)
        t, pf, res = best
        npass = int(res.stats["n_pass"][0]); nnz = int(res.stats["nnz_upper"][0]); Lv = int(res.stats["n_live"][0])
        alg = npass * (12.0 * nnz + 24.0 * Lv)
        solve_ms = pf["solve"][0]
        row = {"workload": f"method 'gravity', n = m = {n}: L = {Lv} live associations, {nnz} stored pairs, {npass} passes",
               "alignment_ms": t * 1e3, "stage_ms": {k: v[0] for k, v in pf.items()}, "solve_us_per_pass": solve_ms * 1e3 / max(npass, 1),
               "roofline": {"kernel": "k_solve_wide", "bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / (solve_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": alg / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "§8(d) bytes: N_pass * (12 nnz_upper + 24 L); the layout stores both triangles with 16-bit labels: 20 bytes per upper non-zero and pass really streamed"},
               "status": int(res.status[0]), "selected": int(len(res.assoc[0]))}
        if with_cpu and (n <= 100 or (os.cpu_count() or 1) >= 32):
            D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
            t0 = time.perf_counter(); o = orc.register(reg._abi_params(), D1, D2, faithful=False); to = time.perf_counter() - t0
            row["oracle_identical"] = bool(np.array_equal(o["assoc"], res.assoc[0]) and o["stats"].n_pass == npass)
            row["cpu_oracle_seconds"] = to
        ll.append(row)
    out["large_live"] = ll

    # ---- MANY mid-size live sets in one call: 64 pairs of method 'gravity' at n = m = 100 (L = 10 000 each) — the scale a
    #      method without a semantic gate runs at ([REF roman/params/submap_align_params.py:98-116]); k_solve_wide in TEAM mode ----
    NM = 64
    reg = SubmapAlignParams(method="gravity").get_object_registration(); reg.set_context(ctx)
    prs = [synth.make_pair(100, 100, 0, 7100 + k, tilt_deg=1.0) for k in range(NM)]
    bt = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in prs])
    Pm = reg._abi_params(); Fm = bt.feats.shape[1]; kmax = bt.kmax()
    featsm = torch.from_numpy(bt.feats).to(dev)
    Om = [torch.zeros((NM, kmax, 2), dtype=torch.int32, device=dev), torch.zeros(NM, dtype=torch.int32, device=dev),
          torch.zeros((NM, 16), dtype=torch.float64, device=dev), torch.zeros(NM, dtype=torch.int32, device=dev),
          torch.zeros(NM * _abi.STATS_NBYTES, dtype=torch.uint8, device=dev)]

    def mcall():
        ctx.align_batch_dev(Pm, featsm.data_ptr(), Fm, bt.off1, bt.n1, bt.off2, bt.n2, kmax, Om[0].data_ptr(), Om[1].data_ptr(), Om[2].data_ptr(), Om[3].data_ptr(), Om[4].data_ptr())
    torch.cuda.synchronize(dev)
    for _ in range(3):                                          # (the first calls size the pools: ROMAN_ST_WORKSPACE until the history knows the need)
        mcall(); torch.cuda.synchronize(dev)
    reps = 4
    t0 = time.perf_counter()
    for _ in range(reps):
        mcall()
    torch.cuda.synchronize(dev)
    tm = (time.perf_counter() - t0) / reps
    ctx.profile_enable(True); ctx.profile_reset(); mcall(); torch.cuda.synchronize(dev); pf = ctx.profile_get(); ctx.profile_enable(False)
    stm = np.frombuffer(Om[4].cpu().numpy().tobytes(), dtype=stats_dtype())[:NM]
    nm_ = Om[1].cpu().numpy(); am_ = Om[0].cpu().numpy(); sm_ = Om[3].cpu().numpy()
    algm = float(np.sum(stm["n_pass"].astype(np.float64) * (12.0 * stm["nnz_upper"].astype(np.float64) + 24.0 * stm["n_live"].astype(np.float64))))
    solve_ms = pf["solve"][0]
    mid = {"workload": f"{NM} submap pairs in ONE call, method 'gravity', n = m = 100: L = 10 000 live associations each",
           "value": NM / tm, "unit": "alignments/s", "ms_per_call": tm * 1e3, "stage_ms": {k: v[0] for k, v in pf.items()},
           "mean_nnz_upper": float(stm["nnz_upper"].mean()), "mean_passes": float(stm["n_pass"].mean()), "max_passes": int(stm["n_pass"].max()),
           "status_ok_frac": float(np.mean(sm_ == 0)),
           "roofline": {"kernel": "k_solve_wide (team mode: the workgroups of an XCD, or half of one, per problem)", "bound": "hbm", "algorithmic_bytes": algm,
                        "achieved": algm / (solve_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algm / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "§8(d) bytes summed over the 64 problems / the solve stage's hipEvent time of the call"}}
    if with_cpu:
        t0 = time.perf_counter()
        many = orc.register_many(Pm, bt.feats, bt.off1, bt.n1, bt.off2, bt.n2, kmax, faithful=False)
        tq = time.perf_counter() - t0
        mid["oracle_identical"] = f"{sum(int(np.array_equal(many[b], am_[b, :nm_[b]])) for b in range(NM))}/{NM}"
        mid["cpu_baseline"] = {"value": NM / tq, "unit": "alignments/s", "cores": orc.num_threads(), "kind": "port", "sample": f"the {NM} pairs, oracle, one OpenMP thread per pair"}
    out["mid_live"] = mid

    # ---- the scale the reference's demo runs at: method 'roman' (pca + volume + gravity + 768-d descriptors), submaps of 20..40 objects:
    #      the all-pairs grid of two robots' 64 + 64 DISTINCT submaps = 4096 distinct pairs in one call -----------------------------------
    reg = SubmapAlignParams(method="roman", semantics_dim=768).get_object_registration(); reg.set_context(ctx)
    rng = np.random.default_rng(5000)
    SD = 64
    ND = SD * SD
    dsubs, _ = synth.make_submap_grid(2 * SD, n=40, d=768, seed0=5000)
    dsizes = rng.integers(20, 41, size=2 * SD)
    dsubs = [sm[:int(k)] for sm, k in zip(dsubs, dsizes)]                       # (objects are in random order inside a submap: a prefix is a random subset)
    bt = rb.batch_from_submap_grid(reg, dsubs[:SD], dsubs[SD:])
    Pd = reg._abi_params(); Fd = Pd.feature_dim(); kmax = bt.kmax()
    featsd = torch.from_numpy(bt.feats).to(dev)
    Od = [torch.zeros((ND, kmax, 2), dtype=torch.int32, device=dev), torch.zeros(ND, dtype=torch.int32, device=dev),
          torch.zeros((ND, 16), dtype=torch.float64, device=dev), torch.zeros(ND, dtype=torch.int32, device=dev),
          torch.zeros(ND * _abi.STATS_NBYTES, dtype=torch.uint8, device=dev)]

    DEPTH = 6                                                   # calls in flight for this leg (a call lasts as long as its slowest problem: they pack better at 6)
    Od2 = [[torch.zeros_like(t) for t in Od] for _ in range(DEPTH - 1)]     # one output set per call in flight

    def dcall(k=0):
        o = Od if k % DEPTH == 0 else Od2[k % DEPTH - 1]
        ctx.align_batch_dev(Pd, featsd.data_ptr(), Fd, bt.off1, bt.n1, bt.off2, bt.n2, kmax, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr())
    torch.cuda.synchronize(dev)
    for _ in range(3):
        dcall()
    torch.cuda.synchronize(dev)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):                                       # one call at a time: a call lasts as long as its slowest problem
        dcall()
    torch.cuda.synchronize(dev)
    td1 = (time.perf_counter() - t0) / reps
    ctx.set_pipeline(DEPTH)                                     # like the headline, calls in flight: the tail of one call's kernel — a problem of
    for k in range(DEPTH):                                      # 200+ passes among thousands of 10 — overlaps the next calls
        dcall(k)
    ctx.sync(); torch.cuda.synchronize(dev)
    reps = 60
    t0 = time.perf_counter()
    for k in range(reps):
        dcall(k)
    ctx.sync(); torch.cuda.synchronize(dev)
    td = (time.perf_counter() - t0) / reps
    ctx.set_pipeline(1)
    ctx.profile_enable(True); ctx.profile_reset(); dcall(); torch.cuda.synchronize(dev); pf = ctx.profile_get(); ctx.profile_enable(False)
    std = np.frombuffer(Od[4].cpu().numpy().tobytes(), dtype=stats_dtype())[:ND]
    nd = Od[1].cpu().numpy(); ad = Od[0].cpu().numpy(); sd_ = Od[3].cpu().numpy()
    demo = {"workload": f"{ND} submap pairs (64 x 64 grid of 128 distinct submaps), method 'roman' (xyz + pca + volume + 768-d descriptors + gravity prior), n, m uniform in [20, 40] "
                        f"([REF params/demo/submap_align.yaml]: submap_max_size 40, DINOv2 768-d): A <= 1600 per pair",
            "value": ND / td, "unit": "alignments/s", "ms_per_call": td * 1e3, "batches_in_flight": DEPTH,
            "one_call_at_a_time": {"value": ND / td1, "ms_per_call": td1 * 1e3},
            "max_passes": int(std["n_pass"].max()),
            "stage_ms": {k: v[0] for k, v in pf.items()},
            "mean_live": float(std["n_live"].mean()), "mean_nnz_upper": float(std["nnz_upper"].mean()), "mean_passes": float(std["n_pass"].mean()),
            "status_ok_frac": float(np.mean((sd_ == 0) | (sd_ == _abi.ROMAN_ST_INSUFFICIENT))),
            "note": "one roman_align_batch_dev call of 4096 DISTINCT pairs: the 64 x 64 cross pairs of 128 distinct submaps (each packed once), inputs resident"}
    if with_cpu:
        NCd = 1024
        pickd = np.sort(np.random.default_rng(11).choice(ND, size=NCd, replace=False))
        t0 = time.perf_counter()
        many = orc.register_many(Pd, bt.feats, bt.off1[pickd], bt.n1[pickd], bt.off2[pickd], bt.n2[pickd], kmax, faithful=False)
        tq = time.perf_counter() - t0
        demo["cpu_baseline"] = {"value": NCd / tq, "unit": "alignments/s", "cores": orc.num_threads(), "kind": "port",
                                "sample": f"{NCd} random pairs of the grid, oracle, one OpenMP thread per pair",
                                "identical_to_gpu": int(sum(int(np.array_equal(many[k], ad[b, :nd[b]])) for k, b in enumerate(pickd))), "compared": NCd}
    out["demo_scale"] = demo


def _sig(x, n=6):
    """floats at n significant digits (the line is read by a parser, not a plotter)"""
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if np.isfinite(x) else None
    if isinstance(x, (np.floating,)):
        return _sig(float(x), n)
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return default
        d = d[k]
    return d


HEADLINE_MAX_BYTES = 4096


def headline(out):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline`, nothing else — under
    HEADLINE_MAX_BYTES (tests/test_bench_line.py).  Everything the side legs measured stays in the full record
    (`bench_extras.json`, see emit()).  Mirrors the reference's timing contract of one number per run
    ([REF roman/align/submap_align.py:155-157], [REF roman/align/results.py:139-144])."""
    r = out.get("roofline") or {}
    c = out.get("cpu_baseline") or {}
    cfg = out.get("config") or {}
    kern = {}
    for name, row in (r.get("kernels") or {}).items():
        kern[name.split(" ")[0]] = {"bound": row.get("bound"), "frac": row.get("frac"), "stage_ms": row.get("stage_ms")}
    line = {
        "metric": out.get("metric"), "value": out.get("value"), "unit": out.get("unit"), "n_gpus": out.get("n_gpus"),
        "steps": out.get("steps"), "warmup": out.get("warmup"), "ms_per_step": out.get("ms_per_step"),
        "higher_is_better": True, "scaling": out.get("scaling"), "vs_baseline": None,
        "dtype": out.get("dtype"), "data": out.get("data"), "timed_region_s": out.get("timed_region_s"),
        "config": {k: cfg.get(k) for k in ("workload", "alignments_per_step", "pairs_per_call", "calls_per_step_per_gpu", "n", "m", "d",
                                           "method", "sharding", "batches_in_flight") if k in cfg},
        "p50_latency_ms": out.get("p50_latency_ms"), "p50_device_ms": out.get("p50_device_ms"),
        "p50_note": out.get("p50_note"),
        "skipped_in_timed_steps": out.get("skipped_in_timed_steps"),
        "sustained_value": _get(out, "sustained", "value"),
        "grid_config4": None if not out.get("grid_config4") else {
            k: _get(out, "grid_config4", k) for k in ("value", "scaling", "steps", "ms_per_step", "timed_region_s")},
        "result_check": {k: _get(out, "result_check", k) for k in ("oracle_identical", "oracle_mode", "status_ok_frac", "planted_inlier_recall_mean",
                                                                   "mean_live", "mean_nnz_upper", "mean_passes", "max_passes")},
    }
    if out.get("weak_pairs"):
        line["weak_pairs"] = out["weak_pairs"]
    if out.get("ranks"):
        line["ranks"] = out["ranks"]
    if r:
        line["roofline"] = {
            "kernel": r.get("kernel"), "bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
            "frac": r.get("frac"), "traffic": r.get("traffic"), "traffic_frac": r.get("traffic_frac"), "traffic_measured_live": bool(r.get("traffic_measured_live")) if r.get("traffic") else None,
            "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"), "avg_launch_ms": r.get("avg_launch_ms"),
            "launches_timed": r.get("launches_timed"), "timing": "hipEvents, timed region, other batches in flight",
            "isolated": {"avg_launch_ms": _get(r, "isolated", "avg_launch_ms"), "frac": _get(r, "isolated", "frac"), "of": "mean of 5 single launches"},
            "step": {"t_star_ms": _get(r, "step", "t_star_ms"), "frac": _get(r, "step", "frac")},
            "kernels": kern}
    if c:
        line["cpu_baseline"] = {
            "value": c.get("value"), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"), "sample": c.get("sample"),
            "value_pruned": c.get("value_pruned"), "value_pruned_pair_parallel": _get(c, "pruned_pair_parallel", "value"),
            "identical_to_gpu": c.get("identical_to_gpu"), "cpu_model": c.get("cpu_model")}
        line["speedup_vs_cpu_pruned_pair_parallel"] = _get(out, "speedup_vs_cpu_baseline", "vs_pruned_pair_parallel")
    for k in ("caller_legs_error", "side_legs_error"):
        if k in out:
            line[k] = str(out[k])[:160]
    line["extras"] = out.get("extras_file")
    line = _sig(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= HEADLINE_MAX_BYTES:                         # never lose the number over a long note
        for k in ("p50_note", "weak_pairs", "ranks", "caller_legs_error", "side_legs_error", "grid_config4", "sustained_value"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < HEADLINE_MAX_BYTES:
                break
    if len(text) >= HEADLINE_MAX_BYTES:
        line["config"]["workload"] = str(line["config"].get("workload"))[:120]
        if "cpu_baseline" in line:
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample"))[:80]
        text = json.dumps(line, separators=(",", ":"))
    return text


def emit(out):
    """The full record -> bench_extras.json (next to this file; and gpurun_out/ when that directory exists, so that it comes
    back from a GPU box) and to stderr; the compact line -> stdout, LAST and alone."""
    names = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                fn = os.path.join(d, "bench_extras.json")
                with open(fn, "w") as fh:
                    json.dump(out, fh, indent=1, default=lambda o: o.item() if hasattr(o, "item") else str(o))
                names.append(os.path.relpath(fn, ROOT))
            except OSError:
                pass
    out["extras_file"] = names[0] if names else None
    sys.stderr.write("bench extras (full record): " + (", ".join(names) if names else "not written") + "\n")
    sys.stderr.flush()
    print(headline(out), flush=True)


if __name__ == "__main__":
    main()
