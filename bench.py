#!/usr/bin/env python3
"""bench.py — submap-pair alignments/sec (+ p50 single-pair latency) of the roman.align hot path on
MI355X, at BASELINE.json's shape n=m=200 objects, d=512 (`method='semanticgrav'`).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path (score -> solve -> select -> pose, one roman_align_batch_dev
call) over one batch of B independent synthetic submap pairs per GPU (BASELINE config 3: B=256;
config 2 is the same shape at B=1 and is what `p50_latency_ms` is measured on).  Inputs are
resident in HBM before the timed region.  With N>1 every rank aligns its own B pairs (weak
scaling, no data-path collective) and one RCCL all_gather of the fixed-size result records
(inlier sets + poses) closes each step.

The printed JSON line also carries
  roofline      — the dominant kernel (k_solve_stream, HBM-bound SpMV passes): algorithmic bytes per launch
                  (sum_b N_pass,b * (12*nnz_upper,b + 24*L_b)) / its hipEvent time,
                  against the 8 TB/s HBM3E peak;
  cpu_baseline  — the CPU oracle (a restatement of the absent clipperpy, kind "port") timed on this
                  box's host cores on a bounded sample of the same workload.
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
    ap.add_argument("--batch", type=int, default=256, help="submap pairs per GPU per step (config 3: 256)")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--m", type=int, default=200)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--method", default="semanticgrav")
    ap.add_argument("--cpu-sample", type=int, default=4, help="pairs timed on the CPU oracle (rank 0, N=1 only); 0 = skip")
    ap.add_argument("--latency-reps", type=int, default=30)
    ap.add_argument("--no-profile", action="store_true", help="do not bracket stages with hipEvents")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2, 3],
                    help="batches in flight per GPU (roman_ctx_set_pipeline): 2 overlaps the straggler tail of one step's kernels "
                         "with the next step's affinity build; results are complete at the closing device-wide synchronise")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from roman_amd import _abi, synth
    from roman_amd.align import SubmapAlignParams
    from roman_amd.align.batch import batch_from_pairs
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

    B = args.batch
    sp = SubmapAlignParams(method=args.method, semantics_dim=args.d) if args.d > 0 else SubmapAlignParams(method=args.method)
    reg = sp.get_object_registration()
    P = reg._abi_params()
    F = P.feature_dim() if P.invariant == _abi.ROMAN_INV_ROMAN else reg.dim

    # ---- synthetic workload (seeds 3000+k), packed once, resident in HBM ----
    pairs = [synth.make_pair(args.n, args.m, args.d, 3000 + rank * B + k) for k in range(B)]
    batch = batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    kmax = batch.kmax()
    feats = torch.from_numpy(batch.feats).to(dev)
    # one output set per batch in flight: step k writes set k % NSET while older sets are gathered
    NSET = max(args.pipeline, 2)
    assoc_o = [torch.zeros((B, kmax, 2), dtype=torch.int32, device=dev) for _ in range(NSET)]
    n_o = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NSET)]
    T_o = [torch.zeros((B, 16), dtype=torch.float64, device=dev) for _ in range(NSET)]
    status_o = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(NSET)]
    stats_o = [torch.zeros(B * _abi.STATS_NBYTES, dtype=torch.uint8, device=dev) for _ in range(NSET)]
    assoc_out, n_out, T_out, status, stats = assoc_o[0], n_o[0], T_o[0], status_o[0], stats_o[0]
    stream = torch.cuda.Stream(dev)                             # an explicit stream shared by torch (RCCL) and the library:
    torch.cuda.set_stream(stream)                              # the legacy default stream would not order against it
    ctx = Context(local_rank, stream=stream.cuda_stream)       # library launches on / behind torch's current stream
    reg.set_context(ctx)

    if world > 1:
        rec_i = torch.empty((B, 2 + 2 * kmax), dtype=torch.int32, device=dev)
        gat_i = torch.empty((world * B, 2 + 2 * kmax), dtype=torch.int32, device=dev)
        gat_T = torch.empty((world * B, 16), dtype=torch.float64, device=dev)

    def gather(k):                                             # collect inlier sets + poses of output set k on every rank
        rec_i[:, 0] = n_o[k]; rec_i[:, 1] = status_o[k]; rec_i[:, 2:] = assoc_o[k].view(B, -1)
        dist.all_gather_into_tensor(gat_i, rec_i)
        dist.all_gather_into_tensor(gat_T, T_o[k])

    step_no = [0]

    def step():
        k = step_no[0] % NSET
        step_no[0] += 1
        ctx.align_batch_dev(P, feats.data_ptr(), F, batch.off1, batch.n1, batch.off2, batch.n2, kmax,
                            assoc_o[k].data_ptr(), n_o[k].data_ptr(), T_o[k].data_ptr(), status_o[k].data_ptr(), stats_o[k].data_ptr())
        if world > 1:
            if args.pipeline >= 2:
                if step_no[0] > 1:
                    ctx.join(skip_latest=True)                 # torch's stream waits for the OLDER batches only
                    gather((k - 1) % NSET)                     # (with 3 in flight this is conservative: it also waits for k-1)
            else:
                gather(k)

    def drain():                                               # results of the last batch
        if args.pipeline >= 2:
            ctx.join(skip_latest=False)
            if world > 1 and step_no[0] > 0:
                gather((step_no[0] - 1) % NSET)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ctx.set_pipeline(args.pipeline)
    for _ in range(args.warmup):
        step()
    drain(); fence()
    step_no[0] = 0
    if not args.no_profile:
        ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain(); fence()
    dt = time.perf_counter() - t0
    prof = ctx.profile_get() if not args.no_profile else None
    if not args.no_profile:
        ctx.profile_enable(False)
    ctx.set_pipeline(1)                                         # the latency probe and the checks below are single calls
    iso = None
    if prof is not None and args.pipeline >= 2 and rank == 0:   # the same kernels without a second batch beside them
        ctx.profile_enable(True); ctx.profile_reset()
        for _ in range(3):
            ctx.align_batch_dev(P, feats.data_ptr(), F, batch.off1, batch.n1, batch.off2, batch.n2, kmax,
                                assoc_o[0].data_ptr(), n_o[0].data_ptr(), T_o[0].data_ptr(), status_o[0].data_ptr(), stats_o[0].data_ptr())
        torch.cuda.synchronize(dev)
        iso = ctx.profile_get(); ctx.profile_enable(False)
    last = (args.steps - 1) % NSET
    assoc_out, n_out, T_out, status, stats = assoc_o[last], n_o[last], T_o[last], status_o[last], stats_o[last]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- sanity of what was timed: results are real (planted inliers recovered) -------------------
    st = np.frombuffer(stats.cpu().numpy().tobytes(), dtype=stats_dtype())
    n_sel = n_out.cpu().numpy(); stat_h = status.cpu().numpy(); a_h = assoc_out.cpu().numpy()
    rec = []
    for b in range(min(B, 16)):
        got = set(map(tuple, a_h[b, :n_sel[b]].tolist())); truth = set(map(tuple, pairs[b].inliers.tolist()))
        rec.append(len(got & truth) / max(len(truth), 1))
    ok_frac = float(np.mean(stat_h == 0))

    # ---- p50 single-pair latency (config 2: B=1) ----------------------------------------------------
    p50 = None
    if rank == 0 and args.latency_reps >= 0:          # --latency-reps -1: B=256 launches only (counter passes)
        b1 = batch.subset(0, 1)
        lat = []
        for r in range(args.latency_reps + 3):
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            ctx.align_batch_dev(P, feats.data_ptr(), F, b1.off1, b1.n1, b1.off2, b1.n2, kmax,
                                assoc_out.data_ptr(), n_out.data_ptr(), T_out.data_ptr(), status.data_ptr(), stats.data_ptr())
            torch.cuda.synchronize(dev)
            if r >= 3:
                lat.append(time.perf_counter() - t1)
        p50 = float(np.median(lat) * 1e3)
    if world > 1:
        dist.barrier()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = world * B * args.steps / dt
    out = {
        "metric": "submap-pair alignments/sec + p50 latency at n=m=200 objects, d=512",
        "value": value, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"config 3: batch of {B} submap pairs per GPU, n={args.n} m={args.m} d={args.d}, method={args.method} "
                               f"(xyz + {args.d}-d descriptors + gravity prior); p50 latency measured on config 2 (single pair)",
                   "pairs_per_gpu": B, "n": args.n, "m": args.m, "d": args.d, "method": args.method, "sharding": f"pairs x{world}, all_gather of records" if world > 1 else "single GPU",
                   "batches_in_flight": args.pipeline},
        "p50_latency_ms": p50,
        "alignments_per_s_batch1": (1e3 / p50) if p50 else None,
        "result_check": {"status_ok_frac": ok_frac, "planted_inlier_recall_mean": float(np.mean(rec)),
                         "mean_live": float(st["n_live"].mean()), "mean_nnz_upper": float(st["nnz_upper"].mean()), "mean_passes": float(st["n_pass"].mean())},
    }
    # ---- roofline of the dominant kernel ---------------------------------------------------------------
    if prof is not None:
        out["stage_ms_per_step"] = {k: v[0] / max(v[1], 1) for k, v in prof.items()}
        dom = max(prof, key=lambda k: prof[k][0])
        solve_ms, solve_n = prof["solve"]
        alg_bytes = float(np.sum(st["n_pass"].astype(np.float64) * (12.0 * st["nnz_upper"] + 24.0 * st["n_live"])))
        if solve_n > 0 and solve_ms > 0:
            avg_s = solve_ms / solve_n * 1e-3
            ach = alg_bytes / avg_s / 1e9
            traffic = None                                      # HBM bytes per launch from the committed PMC pass (same workload)
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                    pj = json.load(fh)
                if pj.get("kernel") == "k_solve_stream" and B == 256 and (args.n, args.m, args.d) == (200, 200, 512):
                    traffic = pj["traffic_bytes_per_launch"]
            except (OSError, ValueError, KeyError):
                pass
            out["roofline"] = {"kernel": "k_solve_stream", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json)",
                               "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": solve_ms / solve_n,
                               "dominant_stage_by_time": dom}
            if iso is not None and iso["solve"][1] > 0:         # launch duration with no other batch in flight (3 extra untimed steps)
                iso_ms = iso["solve"][0] / iso["solve"][1]
                out["roofline"]["isolated"] = {"avg_launch_ms": iso_ms, "achieved": alg_bytes / (iso_ms * 1e-3) / 1e9,
                                               "frac": alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "stage_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in iso.items()}}
    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----------------------------
    if world == 1 and args.cpu_sample > 0:
        from oracle import oracle as orc
        S = min(args.cpu_sample, B)
        packs = [(reg.pack(pairs[b].map1), reg.pack(pairs[b].map2)) for b in range(S)]
        tf0 = time.perf_counter(); k_or = []
        for D1, D2 in packs:
            r = orc.register(P, D1, D2, faithful=True)
            if len(r["assoc"]) >= 3:
                orc.t_align(D1[r["assoc"][:, 0], :3], D2[r["assoc"][:, 1], :3])
            k_or.append(r["assoc"])
        tf = time.perf_counter() - tf0
        tp0 = time.perf_counter()
        for D1, D2 in packs:
            r = orc.register(P, D1, D2, faithful=False)
            if len(r["assoc"]) >= 3:
                orc.t_align(D1[r["assoc"][:, 0], :3], D2[r["assoc"][:, 1], :3])
        tp = time.perf_counter() - tp0
        same = all(np.array_equal(k_or[b], a_h[b, :n_sel[b]]) for b in range(S))
        out["cpu_baseline"] = {"value": S / tf, "unit": "alignments/s", "cores": orc.num_threads(), "kind": "port",
                               "sample": f"{S} of the {B} pairs of this workload; oracle/clipper_oracle.c (C, OpenMP) in upstream-like mode: all A(A-1)/2 association pairs scored, + numpy T_align",
                               "value_pruned": S / tp, "pruned_note": "same oracle skipping associations whose single score is 0 (identical results)",
                               "identical_to_gpu": bool(same), "host_cpus": os.cpu_count()}
        out["speedup_vs_cpu_baseline"] = value / (S / tf)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
