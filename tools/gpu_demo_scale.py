#!/usr/bin/env python3
"""The scale the reference's demo configuration runs at ([REF params/demo/submap_align.yaml]: submap_max_size 40, method
'roman', 768-d descriptors): B problems with n, m uniform in [20, 40] in ONE roman_align_batch_dev call.  Prints the
rate, the stage times and (with the timing build, ROMAN_HIP_LIBRARY=.../variants/libT.so) the solver's phase counters."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from roman_amd import _abi, synth
from roman_amd.align import SubmapAlignParams
from roman_amd.align import batch as rb
from roman_amd.runtime import Context, stats_dtype

ND = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = Context(0, stream=stream.cuda_stream)
reg = SubmapAlignParams(method="roman", semantics_dim=768).get_object_registration(); reg.set_context(ctx)
rng = np.random.default_rng(5000)
sizes = rng.integers(20, 41, size=(256, 2))
base = [synth.make_pair(int(a), int(b), 768, 5000 + k, tilt_deg=1.0) for k, (a, b) in enumerate(sizes)]
b256 = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in base])
rep = max(1, ND // 256)
bt = rb.AlignmentBatch(b256.feats, np.tile(b256.off1, rep), np.tile(b256.n1, rep), np.tile(b256.off2, rep), np.tile(b256.n2, rep))
ND = len(bt)
P = reg._abi_params(); F = P.feature_dim(); kmax = bt.kmax()
feats = torch.from_numpy(bt.feats).to(dev)
O = [torch.zeros((ND, kmax, 2), dtype=torch.int32, device=dev), torch.zeros(ND, dtype=torch.int32, device=dev),
     torch.zeros((ND, 16), dtype=torch.float64, device=dev), torch.zeros(ND, dtype=torch.int32, device=dev),
     torch.zeros(ND * _abi.STATS_NBYTES, dtype=torch.uint8, device=dev)]
def call():
    ctx.align_batch_dev(P, feats.data_ptr(), F, bt.off1, bt.n1, bt.off2, bt.n2, kmax, O[0].data_ptr(), O[1].data_ptr(), O[2].data_ptr(), O[3].data_ptr(), O[4].data_ptr())
torch.cuda.synchronize(dev)
for _ in range(3): call()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(10): call()
torch.cuda.synchronize(dev)
td = (time.perf_counter() - t0) / 10
ctx.profile_enable(True); ctx.profile_reset(); call(); torch.cuda.synchronize(dev); pf = ctx.profile_get(); ctx.profile_enable(False)
st = np.frombuffer(O[4].cpu().numpy().tobytes(), dtype=stats_dtype())[:ND]
print(f"{ND} problems: {td*1e3:.3f} ms per call = {ND/td:.0f} alignments/s; stages ms: " + " ".join(f"{k}={v[0]:.3f}" for k, v in pf.items()))
print(f"mean live {st['n_live'].mean():.1f} (max {st['n_live'].max()}), mean nnz {st['nnz_upper'].mean():.1f}, mean passes {st['n_pass'].mean():.1f} (max {st['n_pass'].max()}), "
      f"tie fallbacks {(O[3].cpu().numpy() & _abi.ROMAN_ST_TIE_FALLBACK != 0).sum()}, insufficient {(O[3].cpu().numpy() & _abi.ROMAN_ST_INSUFFICIENT != 0).sum()}")
