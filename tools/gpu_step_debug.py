#!/usr/bin/env python3
"""Step-by-step run of one small problem through the stepwise API with flushed progress lines (to locate a hang or a
wrong stage quickly): python -u tools/gpu_step_debug.py [method n m d seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from roman_amd import synth
from roman_amd.align import SubmapAlignParams
from roman_amd.runtime import Context
from oracle import oracle as orc


def say(*a):
    print(f"[{time.time() - T0:7.2f}s]", *a, flush=True)


T0 = time.time()
method = sys.argv[1] if len(sys.argv) > 1 else "clipper"
n, m, d, seed = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (30, 30, 0, 1000)
kw = {"semantics_dim": d} if d > 0 and method not in ("clipper", "gravity") else {}
reg = SubmapAlignParams(method=method, **kw).get_object_registration()
P = reg._abi_params()
pr = synth.make_pair(n, m, d, seed, tilt_deg=1.0 if P.gravity_guided else 0.0)
D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
say("oracle ...")
mat, A = orc.build_matrix(P, D1, D2)
sol = orc.solve(P, mat)
say("oracle done: nnz", mat.nnz, "passes", sol["stats"].n_pass, "nodes", len(sol["nodes"]))
ctx = Context(0)
say("context created")
ctx.score(P, D1, D2, None)
say("score done")
idx, sc = ctx.live()
say("live", len(idx), "oracle live", int((orc.single_scores(P, D1, D2) > 0).sum()))
rp, cc, vv, dd = ctx.upper_csr()
rp_o, c_o, v_o, d_o = mat.export()
say("csr: nnz", len(cc), "oracle", len(c_o), "pattern equal", np.array_equal(rp, rp_o) and np.array_equal(cc, c_o),
    "values equal", np.array_equal(vv, v_o), "diag equal", np.array_equal(dd, d_o))
ctx.solve(None)
say("solve done")
nodes, u, score, st = ctx.solution()
say("nodes equal", np.array_equal(nodes, sol["nodes"]), "passes", st.n_pass, "oracle", sol["stats"].n_pass, "score", score, sol["stats"].score,
    "max|du|", float(np.max(np.abs(u - sol["u"]))))
res = reg.register_and_align_batch([(pr.map1, pr.map2)]) if False else None
reg.set_context(ctx)
res = reg.register_and_align_batch([(pr.map1, pr.map2)])
say("batch: status", res.status[0], "assoc equal", np.array_equal(res.assoc[0], A[sol["nodes"]]), "n_pass", res.stats["n_pass"][0])
ctx.close()
say("closed")
