#!/usr/bin/env python3
"""Probe of the workspace-overflow path of the batch entry (kind 2 problems): small stream-layout problem with a forced
tiny capacity, then a large fallback-layout problem on a fresh context."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roman_amd import synth
from roman_amd.align import SubmapAlignParams
from roman_amd.runtime import Context
T0 = time.time()
def say(*a): print(f"[{time.time()-T0:6.2f}s]", *a, flush=True)
which = sys.argv[1]
reg = SubmapAlignParams(method="clipper").get_object_registration()
ctx = Context(0); reg.set_context(ctx)
if which == "small":
    pr = synth.make_pair(30, 30, 0, 1000)
else:
    pr = synth.make_pair(150, 150, 0, 61)
say("calling batch on a fresh context", which)
res = reg.register_and_align_batch([(pr.map1, pr.map2)])
say("status", res.status, "k", len(res.assoc[0]), "n_pass", res.stats["n_pass"][0])
ctx.close()
