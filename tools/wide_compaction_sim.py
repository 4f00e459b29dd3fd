#!/usr/bin/env python3
"""Column-compaction policies of k_solve_wide, simulated on the CPU oracle's support sets (no GPU needed).

The oracle records WHICH elements are positive in the vector fed to each pass (oracle_set_support_dump); positions are ranks by
degree as on the device; the matrix is both triangles, so a stream restricted to a column set S holds sum(deg[c], c in S) entries.
A policy decides per pass whether the vector fits the copy, whether to (re)compact, and what that costs (entries read + written).
The output is the number of entries streamed per problem under each policy, relative to no compaction at all.

usage: python tools/wide_compaction_sim.py [n=100] [seeds=4]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc                                # noqa: E402
from roman_amd import synth                                     # noqa: E402
from roman_amd.align import SubmapAlignParams                   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def support_sets(reg, pair):
    """-> (deg by position (descending), list of boolean support vectors by position, one per pass)"""
    D1, D2 = reg.pack(pair.map1), reg.pack(pair.map2)
    P = reg._abi_params()
    mat, _ = orc.build_matrix(P, D1, D2)
    n = mat.n
    rowptr, cols, vals, diag = mat.export()
    deg = np.diff(rowptr).astype(np.int64)                      # strict upper: add the column side
    deg += np.bincount(cols, minlength=n)
    W = (n + 63) // 64
    cap = 2048
    buf = np.zeros((cap, W), dtype=np.uint64)
    L = orc.lib()
    L.oracle_set_support_dump.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
    L.oracle_set_support_dump(buf.ctypes.data, W, cap)
    try:
        out = orc.solve(P, mat, trace=True)
    finally:
        L.oracle_set_support_dump(None, 0, 0)
    npass = int(out["stats"].n_pass)
    bits = np.unpackbits(buf[:npass].view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
    order = np.argsort(-deg, kind="stable")                     # position = rank by degree
    return deg[order], bits[:, order]


def simulate(deg, sup, window=4, thr=0.5, budget=6, early_thr=None, early_budget=0, min_gain=None):
    """The device's policy (kernels.hip.h, k_solve_wide `stream`): union of the supports over a window; at a window's end a copy
    is made when the union is at most thr x (columns of the copy | L); a vector outside the copy takes the full matrix for that
    pass.  early_thr / early_budget: a SEPARATE allowance of copies at a looser threshold while no copy exists yet.
    -> (entries streamed, entries moved by compactions, copies made)"""
    L = len(deg)
    full = int(deg.sum())
    have = False
    ccols = np.zeros(L, bool); centries = 0; ccount = 0
    acc = np.zeros(L, bool); wp = 0; wout = 0; ncomp = 0; nearly = 0
    streamed = 0; moved = 0
    for x in sup:
        acc = (x if wp == 0 else (acc | x))
        fits = have and not np.any(x & ~ccols)
        wp += 1
        if have and not fits:
            wout += 1
        doc = 0
        if wp >= window:
            cntA = int(acc.sum())
            sub = have and not np.any(acc & ~ccols)
            if ncomp < budget:
                if sub:
                    if cntA <= thr * ccount:
                        doc = 1
                elif (not have or wout >= 2) and cntA <= thr * L:
                    doc = 2
            if not doc and early_thr is not None and nearly < early_budget and (not have or wout >= 2) and not sub and cntA <= early_thr * L:
                doc = 3
            wp = 0; wout = 0
        if doc:
            src = centries if doc == 1 else full
            ccols = acc.copy(); ccount = int(acc.sum()); centries = int(deg[ccols].sum())
            moved += src + centries
            have = True; fits = True
            if doc == 3:
                nearly += 1
            else:
                ncomp += 1
        streamed += centries if fits else full
    return streamed, moved, ncomp + nearly


def main():
    reg = SubmapAlignParams(method="gravity").get_object_registration()
    probs = []
    for k in range(SEEDS):
        pr = synth.make_pair(N, N, 0, 7100 + k, tilt_deg=1.0)
        deg, sup = support_sets(reg, pr)
        probs.append((deg, sup))
        cnt = sup.sum(1)
        # how stable is the early plateau?  union / intersection of the supports of passes 3 ... first pass below L/4
        end = next((i for i in range(3, len(cnt)) if cnt[i] < len(deg) // 4), len(cnt))
        uni = np.any(sup[3:end], 0).sum(); inter = np.all(sup[3:end], 0).sum()
        print(f"seed {7100 + k}: L {len(deg)}, passes {len(sup)}, early plateau passes 3..{end}: support {cnt[3:end].min()}..{cnt[3:end].max()}, "
              f"union {uni}, intersection {inter}; entries under the union's columns {deg[np.any(sup[3:end], 0)].sum() / deg.sum():.3f} of all")
    policies = [("no compaction", dict(budget=0)),
                ("default 4 / 0.50 / 6", dict()),
                ("4 / 0.70 / 6", dict(thr=0.7)),
                ("8 / 0.70 / 8", dict(window=8, thr=0.7, budget=8)),
                ("default + 1 early copy at 0.70, window 4", dict(early_thr=0.7, early_budget=1)),
                ("default + 2 early copies at 0.70", dict(early_thr=0.7, early_budget=2)),
                ("default + 1 early copy at 0.80", dict(early_thr=0.8, early_budget=1)),
                ("default + 2 early copies at 0.80", dict(early_thr=0.8, early_budget=2)),
                ("window 8 + 1 early copy at 0.80", dict(window=8, early_thr=0.8, early_budget=1)),
                ("window 2 + 2 early copies at 0.80", dict(window=2, early_thr=0.8, early_budget=2)),
                ]
    for name, kw in policies:
        tot = []
        for deg, sup in probs:
            s, m, c = simulate(deg, sup, **kw)
            base = len(sup) * int(deg.sum())
            tot.append(((s + m) / base, s / base, m / base, c))
        t = np.array(tot)
        print(f"{name:48s}: streamed + moved {t[:, 0].mean():.3f} of the uncompacted run (stream {t[:, 1].mean():.3f}, copies {t[:, 2].mean():.3f}, {t[:, 3].mean():.1f} copies)")


if __name__ == "__main__":
    main()
