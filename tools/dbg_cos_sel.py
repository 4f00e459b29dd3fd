"""Scratch: where does a batch scored through k_cos_sel differ from the dense kernel?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import registration_for
from roman_amd import _abi, synth
from roman_amd.align import batch as rb
from roman_amd.runtime import Context

ctx = Context(0)
B, nlo, nhi, d, seed = 40, 30, 90, 33, 2
reg = registration_for("semanticgrav", semantics_dim=d); reg.set_context(ctx)
rng = np.random.default_rng(seed)
pairs = []
for k in range(B):
    n, m = int(rng.integers(nlo, nhi + 1)), int(rng.integers(nlo, nhi + 1))
    pr = synth.make_pair(n, m, d, 7000 + 10 * seed + k, tilt_deg=1.0)
    pairs.append((pr.map1, pr.map2))
batch = rb.batch_from_pairs(reg, pairs)
got = {}
for setting in ("0", "1", "0", "1"):
    os.environ["ROMAN_COS_SEL"] = setting
    r = rb.run_batch(reg, batch)
    got.setdefault(setting, []).append(r)
a, b = got["0"][0], got["1"][0]
print("repeat dense equal:", np.array_equal(got["0"][0].stats["score"], got["0"][1].stats["score"]), " repeat sel equal:", np.array_equal(got["1"][0].stats["score"], got["1"][1].stats["score"]))
bad = np.nonzero(a.stats["score"] != b.stats["score"])[0]
print("problems whose score differs:", bad, [(a.stats["score"][k], b.stats["score"][k]) for k in bad[:5]])
for f in ("n_live", "nnz_upper", "n_pass", "d_final"):
    print(f, np.array_equal(a.stats[f], b.stats[f]))
P = reg.abi_params() if hasattr(reg, "abi_params") else None
print("params:", None if P is None else (P.cosine_min, P.cosine_max, P.cos_feature_dim, P.ratio_feature_dim, P.point_dim))
P = reg._abi_params()
print("params:", P.cosine_min, P.cosine_max, P.cos_feature_dim, P.ratio_feature_dim, P.point_dim, "F", batch.feats.shape)
for k in list(bad[:4]) + [0]:
    D1 = np.ascontiguousarray(batch.feats[batch.off1[k]:batch.off1[k] + batch.n1[k]]); D2 = np.ascontiguousarray(batch.feats[batch.off2[k]:batch.off2[k] + batch.n2[k]])
    os.environ.pop("ROMAN_COS_SEL", None)
    ex = ctx.debug_cosine(P, D1, D2)
    os.environ["ROMAN_COS_SEL"] = "gated"
    ga = ctx.debug_cosine(P, D1, D2)
    same = ex.view(np.uint64) == ga.view(np.uint64)
    above = ex > P.cosine_min
    print("problem", k, "n", batch.n1[k], batch.n2[k], "entries", ex.size, "identical", same.sum(), "above cos_min", above.sum(), "above & differ", (above & ~same).sum(),
          "max |diff| of those", (np.abs(ex - ga)[above & ~same].max() if (above & ~same).any() else 0), "cand overflow?", (ex >= P.cosine_min - 2**-6).sum())
