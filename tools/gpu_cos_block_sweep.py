#!/usr/bin/env python3
"""k_cos_block (one wave per 16 x 16 cosine block) against k_cos_wave (one wave per problem) by batch size, demo-size maps
(20-40 objects, 768-d): the 'single' stage of one call (hipEvents), best of 5.  usage (GPU box): python tools/gpu_cos_block_sweep.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roman_amd import synth                                     # noqa: E402
from roman_amd.align import SubmapAlignParams                   # noqa: E402
from roman_amd.align import batch as rb                         # noqa: E402
from roman_amd.runtime import Context                           # noqa: E402

ctx = Context(0)
reg = SubmapAlignParams(method="roman", semantics_dim=768).get_object_registration(); reg.set_context(ctx)
rng = np.random.default_rng(5200)
prs = [synth.make_pair(int(rng.integers(20, 41)), int(rng.integers(20, 41)), 768, 5200 + k, tilt_deg=1.0) for k in range(1024)]
for B in (1, 16, 64, 128, 256, 512, 1024):
    bt = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in prs[:B]])
    row = []
    for setting in ("0", "1"):
        os.environ["ROMAN_COS_BLOCK"] = setting
        rb.run_batch(reg, bt)
        best = None
        for _ in range(5):
            ctx.profile_enable(True); ctx.profile_reset(); rb.run_batch(reg, bt); pf = ctx.profile_get(); ctx.profile_enable(False)
            s = pf["single"][0] / max(pf["single"][1], 1)
            best = s if best is None else min(best, s)
        row.append(best)
    print(f"B={B:5d}: single stage {row[0] * 1e3:7.1f} us per problem-wave kernel, {row[1] * 1e3:7.1f} us per block-wave kernel", flush=True)
ctx.close()
