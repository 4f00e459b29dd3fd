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
With N>1 (or `--also-grid`) the `pairs` run is followed by a short `grid` leg (`--grid-steps`, default 3) whose
result is reported in the same line as `grid_config4` — config 4's strong-scaling number next to the weak one.
A "step" is one pass of the hot path (score -> solve -> select -> pose; roman_align_batch_dev calls) over the
workload.  Inputs are resident in HBM before the timed region.  There is no data-path collective; one RCCL
all_gather of the fixed-size result records (inlier sets + poses) per call collects the results on every rank.

The printed JSON line also carries
  roofline      — the dominant kernel (k_solve_up, HBM-bound SpMV passes): algorithmic bytes per launch
                  (sum_b N_pass,b * (12*nnz_upper,b + 24*L_b)) / its hipEvent time,
                  against the 8 TB/s HBM3E peak;
  cpu_baseline  — the CPU oracle (a restatement of the absent clipperpy, kind "port") timed on this
                  box's host cores on a bounded sample of the same workload, in three modes.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="auto", choices=["auto", "pairs", "grid"], help="auto = pairs")
    ap.add_argument("--also-grid", action="store_true", help="N=1: run the config-4 grid leg too (always run at N>1)")
    ap.add_argument("--grid-steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="pairs workload: submap pairs per GPU per step (config 3: 256)")
    ap.add_argument("--grid", type=int, default=64, help="grid workload: submaps per robot (config 4: 64 -> 4096 alignments)")
    ap.add_argument("--chunk", type=int, default=512, help="grid workload: pairs per roman_align_batch_dev call")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--m", type=int, default=200)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--method", default="semanticgrav")
    ap.add_argument("--cpu-sample", type=int, default=4, help="pairs timed on the CPU oracle in upstream-like mode (rank 0, N=1 only); 0 = skip the CPU legs")
    ap.add_argument("--check-pairs", type=int, default=256, help="problems of the first call compared with the oracle (pruned mode)")
    ap.add_argument("--latency-reps", type=int, default=30)
    ap.add_argument("--no-profile", action="store_true", help="do not bracket stages with hipEvents")
    ap.add_argument("--pipeline", type=int, default=3, choices=[1, 2, 3],
                    help="batches in flight per GPU (roman_ctx_set_pipeline): the straggler tail of one call's solver overlaps the "
                         "next calls' affinity builds (2: 105 k alignments/s, 3: 110 k); results are complete at the closing "
                         "device-wide synchronise")
    return ap.parse_args()


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
    import torch
    import torch.distributed as dist
    from roman_amd import _abi, synth
    from roman_amd.align import SubmapAlignParams
    from roman_amd.align.batch import batch_from_pairs, batch_from_submap_grid
    from roman_amd.runtime import Context, stats_dtype

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    workload = args.workload if args.workload != "auto" else "pairs"

    sp = SubmapAlignParams(method=args.method, semantics_dim=args.d) if args.d > 0 else SubmapAlignParams(method=args.method)
    reg = sp.get_object_registration()
    P = reg._abi_params()
    F = P.feature_dim() if P.invariant == _abi.ROMAN_INV_ROMAN else reg.dim

    stream = torch.cuda.Stream(dev)                             # an explicit stream shared by torch (RCCL) and the library:
    torch.cuda.set_stream(stream)                              # the legacy default stream would not order against it
    ctx = Context(local_rank, stream=stream.cuda_stream)       # library launches on / behind torch's current stream
    reg.set_context(ctx)

    import types

    def measure(workload, steps, warmup, profile):
        """Build one workload, keep it resident in HBM, run `warmup` untimed and `steps` timed steps (barrier +
        device synchronise on both sides, MAX over ranks).  Returns everything the later sections look at."""
        # ---- synthetic workload, packed once, resident in HBM --------------------------------
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
            mine = np.arange(rank, S * S, world)                     # dealt round-robin: every rank gets the same mix
            total_per_step = S * S
            chunk = min(args.chunk, -(-(S * S) // world))         # the same on every rank (the gathered records are fixed-size)
            scaling = "strong"
            wl_text = (f"config 4: all-pairs grid of {S} x {S} submaps ({S * S} alignments), n={args.n} d={args.d}, method={args.method}; "
                       f"{2 * S} submaps packed once and replicated, pairs dealt round-robin to {world} rank(s), {chunk} pairs per call")
        kmax = batch.kmax()
        feats = torch.from_numpy(batch.feats).to(dev)
        calls = [mine[i:i + chunk] for i in range(0, len(mine), chunk)]          # problem indices of every call of a step
        meta = [(batch.off1[ix], batch.n1[ix], batch.off2[ix], batch.n2[ix]) for ix in calls]
        CB = chunk                                                 # rows of the output sets / gathered records

        # one output set per call in flight: call k writes set k % NSET while older sets are gathered
        NSET = max(args.pipeline, 2)
        assoc_o = [torch.zeros((CB, kmax, 2), dtype=torch.int32, device=dev) for _ in range(NSET)]
        n_o = [torch.zeros(CB, dtype=torch.int32, device=dev) for _ in range(NSET)]
        T_o = [torch.zeros((CB, 16), dtype=torch.float64, device=dev) for _ in range(NSET)]
        status_o = [torch.zeros(CB, dtype=torch.int32, device=dev) for _ in range(NSET)]
        stats_o = [torch.zeros(CB * _abi.STATS_NBYTES, dtype=torch.uint8, device=dev) for _ in range(NSET)]

        if world > 1:
            rec_i = torch.empty((CB, 2 + 2 * kmax), dtype=torch.int32, device=dev)
            gat_i = torch.empty((world * CB, 2 + 2 * kmax), dtype=torch.int32, device=dev)
            gat_T = torch.empty((world * CB, 16), dtype=torch.float64, device=dev)

        def gather(k):                                             # collect inlier sets + poses of output set k on every rank
            rec_i[:, 0] = n_o[k]; rec_i[:, 1] = status_o[k]; rec_i[:, 2:] = assoc_o[k].view(CB, -1)
            dist.all_gather_into_tensor(gat_i, rec_i)
            dist.all_gather_into_tensor(gat_T, T_o[k])

        call_no = [0]

        def one_call(ci):
            k = call_no[0] % NSET
            call_no[0] += 1
            o1, a1, o2, a2 = meta[ci]
            ctx.align_batch_dev(P, feats.data_ptr(), F, o1, a1, o2, a2, kmax,
                                assoc_o[k].data_ptr(), n_o[k].data_ptr(), T_o[k].data_ptr(), status_o[k].data_ptr(), stats_o[k].data_ptr())
            if world > 1:
                if args.pipeline >= 2:
                    if call_no[0] > 1:
                        ctx.join(skip_latest=True)                 # torch's stream waits for the OLDER calls only
                        gather((k - 1) % NSET)
                else:
                    gather(k)
            return k

        def step():
            for ci in range(len(calls)):
                one_call(ci)

        def drain():                                               # results of the last call
            if args.pipeline >= 2:
                ctx.join(skip_latest=False)
                if world > 1 and call_no[0] > 0:
                    gather((call_no[0] - 1) % NSET)

        def fence():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        ctx.set_pipeline(args.pipeline)
        for _ in range(warmup):
            step()
        drain(); fence()
        call_no[0] = 0
        if profile:
            ctx.profile_enable(True); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain(); fence()
        dt = time.perf_counter() - t0
        prof = ctx.profile_get() if profile else None
        if profile:
            ctx.profile_enable(False)
        ctx.set_pipeline(1)                                         # the latency probe and the checks below are single calls
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())

        return types.SimpleNamespace(workload=workload, batch=batch, truth=truth, calls=calls, meta=meta, feats=feats, kmax=kmax, CB=CB,
                                     assoc_o=assoc_o, n_o=n_o, T_o=T_o, status_o=status_o, stats_o=stats_o, total_per_step=total_per_step,
                                     scaling=scaling, wl_text=wl_text, dt=dt, prof=prof, steps=steps)

    M = measure(workload, args.steps, args.warmup, not args.no_profile)
    batch, truth, calls, meta, feats, kmax, CB = M.batch, M.truth, M.calls, M.meta, M.feats, M.kmax, M.CB
    assoc_o, n_o, T_o, status_o, stats_o = M.assoc_o, M.n_o, M.T_o, M.status_o, M.stats_o
    total_per_step, scaling, wl_text, dt, prof = M.total_per_step, M.scaling, M.wl_text, M.dt, M.prof
    # N > 1 (or --also-grid): BASELINE config 4 as a second, short leg — the 4096-pair grid dealt over the ranks (strong
    # scaling) next to the weak-scaling `value` above; reported in `grid_config4`
    G = None
    if workload == "pairs" and (world > 1 or args.also_grid):
        G = measure("grid", args.grid_steps, 1, False)

    # ---- the first call once more, alone: its results are what the checks below look at, its kernels what
    #      `isolated` times (one untimed launch first: the first launch from this thread at depth 1 pays one-time costs)
    o1, a1, o2, a2 = meta[0]
    C0 = len(calls[0])

    def call0():
        ctx.align_batch_dev(P, feats.data_ptr(), F, o1, a1, o2, a2, kmax,
                            assoc_o[0].data_ptr(), n_o[0].data_ptr(), T_o[0].data_ptr(), status_o[0].data_ptr(), stats_o[0].data_ptr())
    call0(); torch.cuda.synchronize(dev)
    iso_launch_ms = []
    iso = None
    if prof is not None and rank == 0:
        for _ in range(5):
            ctx.profile_enable(True); ctx.profile_reset()
            call0(); torch.cuda.synchronize(dev)
            g = ctx.profile_get(); ctx.profile_enable(False)
            iso_launch_ms.append(g["solve"][0])
            iso = g if iso is None or g["solve"][0] < iso["solve"][0] else iso
    st = np.frombuffer(stats_o[0].cpu().numpy().tobytes(), dtype=stats_dtype())[:C0]
    n_sel = n_o[0].cpu().numpy()[:C0]; stat_h = status_o[0].cpu().numpy()[:C0]; a_h = assoc_o[0].cpu().numpy()[:C0]
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
    lat_break = None
    if rank == 0 and args.latency_reps >= 0:          # --latency-reps -1: batched launches only (counter passes)
        lat = []
        for r in range(args.latency_reps + 3):
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            ctx.align_batch_dev(P, feats.data_ptr(), F, o1[:1], a1[:1], o2[:1], a2[:1], kmax,
                                assoc_o[1].data_ptr(), n_o[1].data_ptr(), T_o[1].data_ptr(), status_o[1].data_ptr(), stats_o[1].data_ptr())
            torch.cuda.synchronize(dev)
            if r >= 3:
                lat.append(time.perf_counter() - t1)
        p50 = float(np.median(lat) * 1e3)
        # where a single pair's time goes: host enqueue time of the call, and the stages' device time (hipEvents)
        enq = []; stg = []
        for r in range(5):
            ctx.profile_enable(True); ctx.profile_reset()
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            ctx.align_batch_dev(P, feats.data_ptr(), F, o1[:1], a1[:1], o2[:1], a2[:1], kmax,
                                assoc_o[1].data_ptr(), n_o[1].data_ptr(), T_o[1].data_ptr(), status_o[1].data_ptr(), stats_o[1].data_ptr())
            enq.append(time.perf_counter() - t1)
            torch.cuda.synchronize(dev)
            g = ctx.profile_get(); ctx.profile_enable(False)
            stg.append({k: v[0] for k, v in g.items()})
        lat_break = {"host_enqueue_ms": float(np.median(enq) * 1e3),
                     "stage_ms": {k: float(np.median([x[k] for x in stg])) for k in stg[0]},
                     "note": "B=1, stage timers on (they add event records to the call)"}
    if world > 1:
        dist.barrier()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = total_per_step * args.steps / dt
    ms_step = dt / args.steps * 1e3
    out = {
        "metric": "submap-pair alignments/sec + p50 latency at n=m=200 objects, d=512",
        "value": value, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": wl_text, "alignments_per_step": total_per_step, "pairs_per_call": CB, "calls_per_step_per_gpu": len(calls),
                   "n": args.n, "m": args.m, "d": args.d, "method": args.method,
                   "sharding": f"pairs x{world}, all_gather of records" if world > 1 else "single GPU",
                   "batches_in_flight": args.pipeline},
        "p50_latency_ms": p50,
        "grid_config4": None if G is None else {
            "value": G.total_per_step * G.steps / G.dt, "unit": "alignments/s", "scaling": "strong", "steps": G.steps, "warmup": 1,
            "ms_per_step": G.dt / G.steps * 1e3, "workload": G.wl_text, "calls_per_step_per_gpu": len(G.calls),
            "status_ok_frac": float(np.mean(np.concatenate([x.cpu().numpy() for x in G.status_o]) == 0)),     # the last calls' output sets
            "note": "second leg of this run, same timing rules (barrier + device synchronise, MAX over ranks)"},
        "latency_breakdown": lat_break,
        "alignments_per_s_batch1": (1e3 / p50) if p50 else None,
        "register_only": {"note": "the reference times register() alone; the pose is fused into the solver kernel's tail here, so the split is "
                                  "measured as the stand-alone pose entry on the same correspondences (host-pointer call, copies included: an upper bound)",
                          "t_align_ms_per_call": pose_ms, "pairs_per_call": C0,
                          "register_only_ms_per_step_lower_bound": (ms_step - pose_ms * len(calls)) if pose_ms is not None else None},
        "result_check": {"status_ok_frac": ok_frac, "planted_inlier_recall_mean": (float(np.mean(rec)) if rec else None),
                         "mean_live": float(st["n_live"].mean()), "mean_nnz_upper": float(st["nnz_upper"].mean()), "mean_passes": float(st["n_pass"].mean()),
                         "max_passes": int(st["n_pass"].max())},
    }
    # ---- roofline of the dominant kernel ---------------------------------------------------------------
    if prof is not None:
        out["stage_ms_per_call"] = {k: v[0] / max(v[1], 1) for k, v in prof.items()}
        dom = max(prof, key=lambda k: prof[k][0])
        solve_ms, solve_n = prof["solve"]
        alg_bytes = float(np.sum(st["n_pass"].astype(np.float64) * (12.0 * st["nnz_upper"] + 24.0 * st["n_live"])))
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
                               "traffic_source": (f"NOT measured in this run: committed rocprofv3 PMC pass ({tsrc}), FETCH_SIZE + WRITE_SIZE per launch" if traffic else None),
                               "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": solve_ms / solve_n,
                               "launches_timed": int(solve_n), "timing": "hipEvents on the stream the kernel runs on, with the other batch in flight",
                               "dominant_stage_by_time": dom}
            if iso is not None and iso["solve"][1] > 0:         # launch duration with no other batch in flight
                iso_ms = iso["solve"][0] / iso["solve"][1]
                out["roofline"]["isolated"] = {"avg_launch_ms": iso_ms, "per_launch_ms": iso_launch_ms, "note": "minimum of 5 single launches after one untimed launch",
                                               "achieved": alg_bytes / (iso_ms * 1e-3) / 1e9,
                                               "frac": alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "stage_ms_per_call": {k: v[0] / max(v[1], 1) for k, v in iso.items()}}
    # ---- CPU baseline: the oracle on this box's host cores, bounded samples ----------------------------
    if world == 1 and args.cpu_sample > 0:
        from oracle import oracle as orc
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
        out["cpu_baseline"] = {"value": S / tf, "unit": "alignments/s", "cores": nthr, "kind": "port",
                               "sample": f"{S} of the {C0} pairs of one call; oracle/clipper_oracle.c (C, OpenMP; a restatement, not the upstream binary) in "
                                         f"upstream-like mode: all A(A-1)/2 association pairs scored, + numpy T_align",
                               "value_pruned": NC / tp, "pruned_sample": f"{NC} pairs, same oracle skipping associations whose single score is 0 (identical results), {nthr} threads",
                               "value_pruned_1thread": S1 / t1, "one_thread_sample": f"{S1} pairs, pruned mode, 1 thread",
                               "pruned_pair_parallel": pp,
                               "identical_to_gpu": bool(same == NC), "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
        out["speedup_vs_cpu_baseline"] = {"vs_upstream_like_all_pairs": value / (S / tf), "vs_pruned": value / (NC / tp),
                                          "vs_pruned_pair_parallel": (value / pp["value"]) if pp and "value" in pp else None,
                                          "note": "a reported baseline, not a target: the roofline fraction says how good the kernels are"}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
