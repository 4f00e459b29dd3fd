#!/usr/bin/env python3
"""Batch of mid-size live sets (method 'gravity': L = n * m) through k_solve_wide under each team setting
(ROMAN_WIDE_TEAMS = 0 whole device, 1 / 2 / 4 teams per XCD, unset = the library's own choice): stage times of one call.
usage (GPU box): python tools/gpu_mid_live.py [pairs=64] [n=100]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roman_amd import synth                                     # noqa: E402
from roman_amd.align import SubmapAlignParams                   # noqa: E402
from roman_amd.align import batch as rb                         # noqa: E402
from roman_amd.runtime import Context                           # noqa: E402

NP = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = Context(0)
reg = SubmapAlignParams(method="gravity").get_object_registration(); reg.set_context(ctx)
pairs = [synth.make_pair(N, N, 0, 7100 + k, tilt_deg=1.0) for k in range(NP)]
bt = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
ref = None
for teams in tuple(os.environ.get("MIDLIVE_TEAMS", "None,0,1,2,4").split(",")):
    teams = None if teams == "None" else teams
    if teams is None:
        os.environ.pop("ROMAN_WIDE_TEAMS", None)
    else:
        os.environ["ROMAN_WIDE_TEAMS"] = teams
    rb.run_batch(reg, bt); rb.run_batch(reg, bt)
    best = None
    for _ in range(3):
        ctx.profile_enable(True); ctx.profile_reset()
        t0 = time.perf_counter(); res = rb.run_batch(reg, bt); t = time.perf_counter() - t0
        pf = ctx.profile_get(); ctx.profile_enable(False)
        if best is None or pf["solve"][0] < best[1]["solve"][0]:
            best = (t, pf, res)
    t, pf, res = best
    if ref is None:
        ref = res
    same = sum(int(np.array_equal(a, b)) for a, b in zip(res.assoc, ref.assoc))
    npass = res.stats["n_pass"].astype(np.float64); nnz = res.stats["nnz_upper"].astype(np.float64); L = res.stats["n_live"].astype(np.float64)
    alg = float(np.sum(npass * (12 * nnz + 24 * L))); streamed = float(np.sum(npass * 20 * nnz))
    sm = pf["solve"][0]
    print(f"teams={teams}: call {t * 1e3:.1f} ms, stages " + ", ".join(f"{k} {v[0]:.2f}" for k, v in pf.items()) +
          f" | solve: {alg / sm / 1e6:.0f} GB/s on 8(d) bytes, {streamed / sm / 1e6:.0f} GB/s streamed, {sm * 1e3 / npass.sum():.2f} us per problem-pass"
          f" | status ok {int((res.status == 0).sum())}/{NP}, same as first setting {same}/{NP}, mean passes {npass.mean():.1f}", flush=True)
ctx.close()
