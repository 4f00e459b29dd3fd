#!/usr/bin/env python3
"""The reference's demo scale (method 'roman', n, m in [20, 40], d = 768) as the all-pairs grid of 64 + 64 distinct submaps:
stage times of one device-pointer call of 4096 problems.  Run under different environment switches (ROMAN_COO=0, ...) for A/B.
usage (GPU box): python tools/gpu_demo_scale.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roman_amd import _abi, synth                               # noqa: E402
from roman_amd.align import SubmapAlignParams                   # noqa: E402
from roman_amd.align import batch as rb                         # noqa: E402
from roman_amd.runtime import Context, stats_dtype              # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = Context(0, stream=stream.cuda_stream)
reg = SubmapAlignParams(method="roman", semantics_dim=768).get_object_registration(); reg.set_context(ctx)
rng = np.random.default_rng(5000)
SD = 64; ND = SD * SD
subs, _ = synth.make_submap_grid(2 * SD, n=40, d=768, seed0=5000)
sizes = rng.integers(20, 41, size=2 * SD)
subs = [sm[:int(k)] for sm, k in zip(subs, sizes)]
bt = rb.batch_from_submap_grid(reg, subs[:SD], subs[SD:])
P = reg._abi_params(); F = P.feature_dim(); kmax = bt.kmax()
feats = torch.from_numpy(bt.feats).to(dev)
O = [torch.zeros((ND, kmax, 2), dtype=torch.int32, device=dev), torch.zeros(ND, dtype=torch.int32, device=dev), torch.zeros((ND, 16), dtype=torch.float64, device=dev),
     torch.zeros(ND, dtype=torch.int32, device=dev), torch.zeros(ND * _abi.STATS_NBYTES, dtype=torch.uint8, device=dev)]


def call():
    ctx.align_batch_dev(P, feats.data_ptr(), F, bt.off1, bt.n1, bt.off2, bt.n2, kmax, O[0].data_ptr(), O[1].data_ptr(), O[2].data_ptr(), O[3].data_ptr(), O[4].data_ptr())


torch.cuda.synchronize(dev)
for _ in range(3):
    call()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(10):
    call()
torch.cuda.synchronize(dev)
td = (time.perf_counter() - t0) / 10
tps = {}
for depth in (3, 4, 6):                                         # calls in flight (the outputs are the same every call)
    ctx.set_pipeline(depth)
    for _ in range(depth):
        call()
    ctx.sync(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(60):
        call()
    ctx.sync(); torch.cuda.synchronize(dev)
    tps[depth] = (time.perf_counter() - t0) / 60
tp = tps[3]
print("ms per call by calls in flight:", {k: round(v * 1e3, 3) for k, v in tps.items()})
ctx.set_pipeline(1)
ctx.profile_enable(True); ctx.profile_reset(); call(); torch.cuda.synchronize(dev); pf = ctx.profile_get(); ctx.profile_enable(False)
st = np.frombuffer(O[4].cpu().numpy().tobytes(), dtype=stats_dtype())[:ND]
print(f"ROMAN_COO={os.environ.get('ROMAN_COO')} ROMAN_SMALL_FUSED={os.environ.get('ROMAN_SMALL_FUSED')} ROMAN_SMALL_ONLY={os.environ.get('ROMAN_SMALL_ONLY')}: "
      f"one call at a time {ND / td / 1e6:.2f} M alignments/s ({td * 1e3:.3f} ms per call), three in flight {ND / tp / 1e6:.2f} M/s ({tp * 1e3:.3f} ms per call), stages " + ", ".join(f"{k} {v[0]:.3f}" for k, v in pf.items()) +
      f" | mean live {st['n_live'].mean():.1f}, nnz {st['nnz_upper'].mean():.1f}, passes {st['n_pass'].mean():.2f}, L<=128: {(st['n_live'] <= 128).mean():.3f}, nnz<=384: {(st['nnz_upper'] <= 384).mean():.3f}, "
      f"checksum {int(O[1].sum().item())} {int(O[0].sum().item())}")
ctx.close()
