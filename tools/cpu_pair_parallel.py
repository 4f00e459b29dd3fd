#!/usr/bin/env python3
"""CPU-only: the pruned oracle with one thread per pair on N config-3 pairs (bench.py's `pruned_pair_parallel` leg)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roman_amd import synth
from roman_amd.align import SubmapAlignParams
from roman_amd.align.batch import batch_from_pairs
from oracle import oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reg = SubmapAlignParams(method="semanticgrav", semantics_dim=512).get_object_registration()
P = reg._abi_params()
t = time.perf_counter()
pairs = [synth.make_pair(200, 200, 512, 3000 + k) for k in range(N)]
batch = batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
print(f"setup {time.perf_counter()-t:.1f} s; threads {orc.num_threads()} cpus {os.cpu_count()} OMP env:", {k: v for k, v in os.environ.items() if k.startswith("OMP") or k.startswith("GOMP")})
for rep in range(2):
    t = time.perf_counter()
    orc.register_many(P, batch.feats, batch.off1, batch.n1, batch.off2, batch.n2, batch.kmax())
    dt = time.perf_counter() - t
    print(f"pair-parallel: {N/dt:.1f} alignments/s ({dt:.2f} s for {N})")
orc.set_threads(1)
t = time.perf_counter()
orc.register(P, batch.feats[batch.off1[0]:batch.off1[0]+batch.n1[0]], batch.feats[batch.off2[0]:batch.off2[0]+batch.n2[0]], faithful=False)
print(f"one pair, one thread: {time.perf_counter()-t:.3f} s")
