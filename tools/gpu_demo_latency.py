#!/usr/bin/env python3
"""Where the time of ONE demo-size pair goes (the reference's serial loop at its own scale: method 'roman', 20-40 objects, 768-d):
Python packing, the C call (roman_align_batch with host pointers: upload, enqueue, read-back), the device stages of that call
(hipEvents), and the same for the stepwise register() + T_align() pair.  usage (GPU box): python tools/gpu_demo_latency.py [pairs=24]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roman_amd import synth                                     # noqa: E402
from roman_amd.align import SubmapAlignParams                   # noqa: E402
from roman_amd.align import batch as rb                         # noqa: E402
from roman_amd.runtime import Context                           # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = Context(0)
reg = SubmapAlignParams(method="roman", semantics_dim=768).get_object_registration(); reg.set_context(ctx)
rng = np.random.default_rng(5200)
prs = [synth.make_pair(int(rng.integers(20, 41)), int(rng.integers(20, 41)), 768, 5200 + k, tilt_deg=1.0) for k in range(K)]


def med(f, reps=3):
    best = None
    for _ in range(reps):
        ts = []
        for pr in prs:
            t0 = time.perf_counter(); f(pr); ts.append(time.perf_counter() - t0)
        m = float(np.median(ts)) * 1e3
        best = m if best is None else min(best, m)
    return best


bts = [rb.batch_from_pairs(reg, [(p.map1, p.map2)]) for p in prs]
print(f"pack (batch_from_pairs)           : {med(lambda pr: rb.batch_from_pairs(reg, [(pr.map1, pr.map2)])):.3f} ms per pair")
it = iter(range(10 ** 9))
print(f"run_batch (C call, host pointers) : {med(lambda pr: rb.run_batch(reg, bts[next(it) % K])):.3f} ms per pair")
print(f"register_and_align_batch          : {med(lambda pr: reg.register_and_align_batch([(pr.map1, pr.map2)])):.3f} ms per pair")
print(f"register()                        : {med(lambda pr: reg.register(pr.map1, pr.map2)):.3f} ms per pair")
assoc = [reg.register(pr.map1, pr.map2) for pr in prs]
ia = iter(range(10 ** 9))


def pose(pr):
    a = assoc[next(ia) % K]
    try:
        reg.T_align(pr.map1, pr.map2, a)
    except Exception:
        pass


print(f"T_align()                         : {med(pose):.3f} ms per pair")
ctx.profile_enable(True); ctx.profile_reset()
for b in bts:
    rb.run_batch(reg, b)
pf = ctx.profile_get(); ctx.profile_enable(False)
print("device stages of run_batch (hipEvents, ms per call): " + ", ".join(f"{k} {v[0] / max(v[1], 1):.4f}" for k, v in pf.items()))
# the C call alone, arguments prepared (what a C caller pays)
import ctypes as C                                              # noqa: E402
from roman_amd import _abi                                      # noqa: E402
P = reg._abi_params()
b0 = bts[0]
for name in ("align_batch",):
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.align_batch(P, b0.feats, b0.off1, b0.n1, b0.off2, b0.n2, assoc=None, assoc_off=None, u0=None, kmax=b0.kmax())
    print(f"Context.{name} x 200 on one pair : {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per call")
ctx.close()
