#!/usr/bin/env python3
"""the large-live-set fallback, timed and compared with the oracle.
  (a) method 'gravity' (no semantic gate), n = m = 200: every one of the 40 000 associations is live (kind 1:
      symmetric SELL-64 layout, fallback solver, vectors in HBM);
  (b) one such problem inside a batch of 255 ordinary config-3 pairs (method 'gravity' needs d = 0, so the mix is
      255 small 'gravity' problems (n = m = 40: L = 1600, stream layout) + the large one).
Prints wall-clock of the calls and whether associations equal the oracle's."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roman_amd import _abi, synth
from roman_amd.align import SubmapAlignParams
from roman_amd.runtime import Context
from oracle import oracle as orc

T0 = time.time()
def say(*a): print(f"[{time.time()-T0:7.2f}s]", *a, flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
check = (sys.argv[2] != "nocheck") if len(sys.argv) > 2 else True
reg = SubmapAlignParams(method="gravity").get_object_registration()
ctx = Context(0); reg.set_context(ctx)
big = synth.make_pair(n, n, 0, 7001, tilt_deg=1.0)
say("warm-up (small problem)")
reg.register_and_align_batch([(synth.make_pair(20, 20, 0, 1).map1, synth.make_pair(20, 20, 0, 1).map2)])
for rep in range(3):
    ctx.profile_enable(True); ctx.profile_reset()
    t = time.perf_counter()
    res = reg.register_and_align_batch([(big.map1, big.map2)])
    dt = time.perf_counter() - t
    prof = ctx.profile_get(); ctx.profile_enable(False)
    npass = int(res.stats['n_pass'][0]); nnz = int(res.stats['nnz_upper'][0]); Lv = int(res.stats['n_live'][0])
    solve_ms = prof["solve"][0]
    alg = npass * (12.0 * nnz + 24.0 * Lv)
    say(f"(a) gravity n=m={n}: {dt*1e3:.1f} ms  status={res.status[0]} L={Lv} nnz_upper={nnz} passes={npass} selected={len(res.assoc[0])}  "
        f"stages ms: " + " ".join(f"{k}={v[0]:.2f}" for k, v in prof.items()) +
        f" | solve {solve_ms*1e3/max(npass,1):.1f} us/pass, algorithmic {alg/1e9:.2f} GB -> {alg/max(solve_ms,1e-9)/1e6:.0f} GB/s = {alg/max(solve_ms,1e-9)/1e6/8000:.3f} of 8 TB/s")
truth = set(map(tuple, big.inliers.tolist()))
say("planted inliers recovered:", len(truth & set(map(tuple, res.assoc[0].tolist()))), "of", len(truth))
if check:
    D1, D2 = reg.pack(big.map1), reg.pack(big.map2)
    t = time.perf_counter()
    o = orc.register(reg._abi_params(), D1, D2, faithful=False)
    say(f"oracle (pruned mode, all host threads): {time.perf_counter()-t:.1f} s; associations identical: {np.array_equal(res.assoc[0], o['assoc'])}; "
        f"nnz_upper {o['stats'].nnz_upper} passes {o['stats'].n_pass}")
small = [synth.make_pair(40, 40, 0, 7100 + k, tilt_deg=1.0) for k in range(255)]
pairs = [(p.map1, p.map2) for p in small]
for rep in range(2):
    t = time.perf_counter(); r1 = reg.register_and_align_batch(pairs); d1 = time.perf_counter() - t
say(f"(b) 255 small problems alone: {d1*1e3:.1f} ms")
for rep in range(2):
    t = time.perf_counter(); r2 = reg.register_and_align_batch(pairs + [(big.map1, big.map2)]); d2 = time.perf_counter() - t
say(f"(b) 255 small + the large one: {d2*1e3:.1f} ms; small results unchanged: "
    f"{all(np.array_equal(r1.assoc[b], r2.assoc[b]) for b in range(255))}; large unchanged: {np.array_equal(r2.assoc[255], res.assoc[0])}")
ctx.close()
