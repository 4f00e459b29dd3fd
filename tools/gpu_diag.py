#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle diagnostic (run on the MI355X box through gpurun).

Prints, for a ladder of configurations: device math bit-exactness, the MFMA cosine kernel, the
live association list, the sparsity pattern / values of M, the solver trajectory statistics, the
selected associations and the pose — each compared with the CPU oracle.  Not a test (tests/ holds
those); a tool for bring-up and for the numbers quoted in DESIGN.md.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from roman_amd import _abi, synth                      # noqa: E402
from roman_amd.align import SubmapAlignParams          # noqa: E402
from roman_amd.align.batch import batch_from_pairs     # noqa: E402
from roman_amd.runtime import Context, version         # noqa: E402
from oracle import oracle as orc                       # noqa: E402


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy(); ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2**63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2**63) - ib[ib < 0]
    return np.abs(ia - ib)


def section(t):
    print("\n" + "=" * 100 + "\n" + t + "\n" + "=" * 100, flush=True)


def math_checks(ctx):
    section("device math vs host (bit exactness of the pattern-shaping ops)")
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 1000, 500000), rng.uniform(0, 1e-3, 100000), 10.0 ** rng.uniform(-300, 300, 100000)])
    d = ulp_diff(ctx.debug_math(0, x), np.sqrt(x))
    print(f"sqrt : n={x.size} mismatches={np.count_nonzero(d)} max_ulp={d.max()}")
    y = rng.uniform(1e-3, 1e3, x.size)
    d = ulp_diff(ctx.debug_math(3, x, y), x / y)
    print(f"div  : mismatches={np.count_nonzero(d)} max_ulp={d.max()}")
    e = rng.uniform(-40, 0, 600000)
    d = ulp_diff(ctx.debug_math(1, e), np.exp(e))
    print(f"exp  : mismatches={np.count_nonzero(d)} max_ulp={d.max()}")
    c = rng.uniform(1e-12, 1, 600000)
    d = ulp_diff(ctx.debug_math(2, c), np.cbrt(c))
    print(f"cbrt : mismatches={np.count_nonzero(d)} max_ulp={d.max()}")
    d = ulp_diff(ctx.debug_math(4, c, np.full_like(c, 0.25)), np.power(c, 0.25))
    print(f"pow  : mismatches={np.count_nonzero(d)} max_ulp={d.max()}")


def cosine_checks(ctx):
    section("f64 MFMA cosine kernel vs numpy")
    rng = np.random.default_rng(1)
    for (n1, n2, d) in [(16, 16, 4), (37, 53, 70), (200, 200, 512), (5, 3, 1)]:
        P = _abi.RomanParams.default(); P.cos_feature_dim = d
        D1 = rng.standard_normal((n1, 3 + d)); D2 = rng.standard_normal((n2, 3 + d))
        D1[:, 3:] *= np.linspace(0.5, 2.0, d)            # asymmetric in k
        if n1 > 2:
            D1[2, 3:] = 0.0                              # zero-norm guard
        got = ctx.debug_cosine(P, D1, D2)
        a, b = D1[:, 3:], D2[:, 3:]
        na, nb = np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            ref = (a @ b.T) / np.outer(na, nb)
        ref[~np.isfinite(ref)] = 0.0
        print(f"n1={n1} n2={n2} d={d}: max|diff|={np.max(np.abs(got - ref)):.3e}")


def make_case(name):
    """-> (registration, pair, label)"""
    if name == "cfg1":
        reg = SubmapAlignParams(method="clipper").get_object_registration(); pr = synth.make_pair(30, 30, 0, 1000)
    elif name == "gravity40":
        reg = SubmapAlignParams(method="gravity").get_object_registration(); pr = synth.make_pair(40, 40, 0, 11)
    elif name == "semgrav60":
        reg = SubmapAlignParams(method="semanticgrav", semantics_dim=64).get_object_registration(); pr = synth.make_pair(60, 50, 64, 12)
    elif name == "roman50":
        reg = SubmapAlignParams(method="roman", semantics_dim=32).get_object_registration(); pr = synth.make_pair(50, 50, 32, 13)
    elif name == "sevg45":
        reg = SubmapAlignParams(method="sevg", semantics_dim=16, epsilon_shape=0.3).get_object_registration(); pr = synth.make_pair(45, 40, 16, 14)
    elif name == "cfg2":
        reg = SubmapAlignParams(method="semanticgrav", semantics_dim=512).get_object_registration(); pr = synth.make_pair(200, 200, 512, 2000)
    elif name == "prune60":
        reg = SubmapAlignParams(method="clipper+prune", cosine_min=0.5).get_object_registration(); pr = synth.make_pair(60, 60, 64, 15)
    elif name == "gravity100":
        reg = SubmapAlignParams(method="gravity").get_object_registration(); pr = synth.make_pair(100, 100, 0, 7)
    else:
        raise ValueError(name)
    return reg, pr


def compare_case(ctx, name):
    section(f"case {name}")
    reg, pr = make_case(name)
    reg.set_context(ctx)
    P = reg._abi_params()
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    A = reg._associations_to_score(pr.map1, pr.map2)
    t0 = time.time(); mat, Ao = orc.build_matrix(P, D1, D2, A); sol = orc.solve(P, mat); t_or = time.time() - t0
    st_o = sol["stats"]
    t0 = time.time(); ctx.score(P, D1, D2, A); t_sc = time.time() - t0
    # live list
    s_o = orc.single_scores(P, D1, D2, Ao)
    live_o = np.nonzero(s_o > 0)[0]
    idx, sc = ctx.live()
    same_live = idx.shape == live_o.shape and np.array_equal(idx, live_o)
    print(f"A={Ao.shape[0]} live: gpu={idx.size} oracle={live_o.size} identical={same_live}", end="")
    if same_live and idx.size:
        print(f" max|ds|={np.max(np.abs(sc - s_o[live_o])):.3e} max_ulp={ulp_diff(sc, s_o[live_o]).max()}")
    else:
        print()
    # matrix
    rp_o, c_o, v_o, d_o = mat.export()
    rp, cc, vv, dd = ctx.upper_csr()
    same_pat = np.array_equal(rp, rp_o) and np.array_equal(cc, c_o)
    print(f"nnz_upper: gpu={cc.size} oracle={c_o.size} pattern_identical={same_pat}", end="")
    if same_pat and cc.size:
        print(f" max_rel_dv={np.max(np.abs(vv - v_o) / np.abs(v_o)):.3e} max_ulp={ulp_diff(vv, v_o).max()} diag_maxdiff={np.max(np.abs(dd - d_o)):.3e}")
    else:
        print()
        if cc.size and c_o.size:
            so = set(zip(np.repeat(np.arange(len(rp_o) - 1), np.diff(rp_o)).tolist(), c_o.tolist()))
            sg = set(zip(np.repeat(np.arange(len(rp) - 1), np.diff(rp)).tolist(), cc.tolist()))
            print(f"   only_gpu={len(sg - so)} only_oracle={len(so - sg)} sample_gpu_only={list(sg - so)[:5]} sample_or_only={list(so - sg)[:5]}")
    # solve
    t0 = time.time(); ctx.solve(None); t_so = time.time() - t0
    nodes, u, score, st = ctx.solution()
    print(f"oracle: pass={st_o.n_pass} outer={st_o.outer_iters} inner={st_o.inner_iters} ls={st_o.ls_trials} F={st_o.score:.12f} d={st_o.d_final:.9f} k={len(sol['nodes'])}")
    print(f"gpu   : pass={st.n_pass} outer={st.outer_iters} inner={st.inner_iters} ls={st.ls_trials} F={st.score:.12f} d={st.d_final:.9f} k={len(nodes)} nnzU={st.nnz_upper} L={st.n_live}")
    print(f"nodes identical (ordered)={np.array_equal(nodes, sol['nodes'])} same set={set(nodes.tolist()) == set(sol['nodes'].tolist())} max|du|={np.max(np.abs(u - sol['u'])) if u.size else 0:.3e}")
    sel = ctx.selected_associations()
    print(f"selected assoc == A[nodes]: {np.array_equal(sel, Ao[nodes])}; inliers recovered {len(set(map(tuple, sel.tolist())) & set(map(tuple, pr.inliers.tolist())))}/{len(pr.inliers)}")
    # batch path + pose
    res = reg.register_and_align_batch([(pr.map1, pr.map2)])
    a_b = res.assoc[0]
    print(f"batch assoc identical to stepwise={np.array_equal(a_b, sel)} status={res.status[0]} ")
    if len(sel) >= reg.dim:
        p1 = np.array([pr.map1[i].center.reshape(-1)[:reg.dim] for i, _ in Ao[sol['nodes']]])
        p2 = np.array([pr.map2[j].center.reshape(-1)[:reg.dim] for _, j in Ao[sol['nodes']]])
        T_o = orc.t_align(p1, p2, reg.dim)
        T_s = reg.T_align(pr.map1, pr.map2, sel)
        print(f"pose: |T_batch-T_oracle|_F={np.linalg.norm(res.T[0] - T_o):.3e} |T_align-T_oracle|_F={np.linalg.norm(T_s - T_o):.3e} |T-T_gt^-1...|: trans_err={np.linalg.norm(res.T[0][:3, 3] - pr.T_gt[:3, 3]):.3f}")
    print(f"times: oracle build+solve {t_or*1e3:.1f} ms | gpu score {t_sc*1e3:.1f} ms solve {t_so*1e3:.1f} ms (first-call overheads included)")


def ragged(ctx):
    section("ragged batch: per-problem stats GPU vs oracle")
    reg = SubmapAlignParams(method="semanticgrav", semantics_dim=24).get_object_registration(); reg.set_context(ctx)
    sizes = [(30, 30), (12, 40), (40, 9), (3, 3), (0, 10), (10, 0), (1, 1), (25, 26), (2, 30)]
    pairs = []
    for k, (n, m) in enumerate(sizes):
        pr = synth.make_pair(max(n, 1), max(m, 1), 24, 100 + k)
        pairs.append((pr.map1[:n], pr.map2[:m]))
    res = reg.register_and_align_batch(pairs)
    P = reg._abi_params()
    for b, (m1, m2) in enumerate(pairs):
        if not len(m1) or not len(m2):
            continue
        D1, D2 = reg.pack(m1), reg.pack(m2)
        o = orc.register(P, D1, D2)
        so = o["stats"]; sg = res.stats[b]
        print(f"b={b} sizes={sizes[b]} oracle: L={so.n_live} nnzU={so.nnz_upper} pass={so.n_pass} outer={so.outer_iters} inner={so.inner_iters} ls={so.ls_trials} F={so.score!r} d={so.d_final!r} k={len(o['assoc'])}")
        print(f"      gpu   : L={sg['n_live']} nnzU={sg['nnz_upper']} pass={sg['n_pass']} outer={sg['outer_iters']} inner={sg['inner_iters']} ls={sg['ls_trials']} F={float(sg['score'])!r} d={float(sg['d_final'])!r} k={len(res.assoc[b])}")
        # single-problem stepwise for u
        ctx.score(P, D1, D2, None); ctx.solve(None)
        nodes, u, score, st = ctx.solution()
        live = np.nonzero(o['u'] > 0)[0]
        print(f"      stepwise gpu pass={st.n_pass}; u_or[live]={o['u'][live][:8]!r} u_gpu[live]={u[live][:8]!r}")


def timing(ctx):
    section("throughput probe: batch of cfg2-shaped pairs (semanticgrav, n=m=200, d=512)")
    reg = SubmapAlignParams(method="semanticgrav", semantics_dim=512).get_object_registration()
    reg.set_context(ctx)
    for B in (1, 16, 64):
        pairs = [(p.map1, p.map2) for p in (synth.make_pair(200, 200, 512, 3000 + k) for k in range(B))]
        batch = batch_from_pairs(reg, pairs)
        ctx.profile_enable(True)
        from roman_amd.align.batch import run_batch
        run_batch(reg, batch)                      # warm-up (allocations)
        ctx.profile_reset()
        t0 = time.time(); reps = 3
        for _ in range(reps):
            res = run_batch(reg, batch)
        dt = (time.time() - t0) / reps
        prof = ctx.profile_get()
        print(f"B={B}: {dt*1e3:.2f} ms/batch (host-pointer API incl. H2D) -> {B/dt:.1f} align/s ; stages(ms total over {reps} reps)=" +
              ", ".join(f"{k}:{v[0]:.2f}" for k, v in prof.items()) +
              f" ; passes mean={res.stats['n_pass'].mean():.1f} L mean={res.stats['n_live'].mean():.0f} nnzU mean={res.stats['nnz_upper'].mean():.0f} k mean={np.mean([len(a) for a in res.assoc]):.1f}")
        ctx.profile_enable(False)


def main():
    print(version())
    ctx = Context(0)
    which = sys.argv[1:] or ["math", "cos", "cfg1", "gravity40", "semgrav60", "roman50", "sevg45", "prune60", "cfg2", "gravity100", "timing"]
    for w in which:
        try:
            if w == "math":
                math_checks(ctx)
            elif w == "cos":
                cosine_checks(ctx)
            elif w == "ragged":
                ragged(ctx)
            elif w == "timing":
                timing(ctx)
            else:
                compare_case(ctx, w)
        except Exception as e:                         # keep going: one call should tell us everything
            import traceback
            print(f"!! {w} raised {type(e).__name__}: {e}")
            traceback.print_exc()
    ctx.close()


if __name__ == "__main__":
    main()
