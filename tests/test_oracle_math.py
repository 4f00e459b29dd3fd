"""The oracle's stated-order arithmetic (oracle/clipper_oracle.c header).

Every quantity a threshold is applied to (the cosine gate, the epsilon gate, `score > affinityeps`) is
produced by correctly-rounded operations in a stated order, so that the device can reproduce the bits.
These tests check (a) that the two fixed-sequence functions are accurate (<= 1 ulp from glibc), (b) that
switching to the plain restatement (sequential dot product, glibc exp/cbrt) changes no fixture's sparsity
pattern, selected associations or iteration counts — i.e. the choice of order is immaterial everywhere
except ON a threshold — and (c) the decision switches for H2/H3 (include/roman_hip.h)."""
import math

import numpy as np
import pytest

from conftest import registration_for, ulp_diff
from roman_amd import _abi, synth


def test_fixed_exp_is_within_one_ulp_of_libm(orc):
    rng = np.random.default_rng(0)
    y = np.concatenate([-rng.uniform(0, 2, 100000), rng.uniform(-40, 40, 50000), -10.0 ** rng.uniform(-12, 0, 20000),
                        rng.uniform(-690, 690, 20000), [0.0, -0.0, -1.125, -700.5, 705.0]])
    libm = np.array([math.exp(v) for v in y])      # glibc (numpy's own SIMD exp is itself only <= 1 ulp)
    d = ulp_diff(orc.fixed_exp(y), libm)
    assert d.max() <= 1
    assert (d > 0).mean() < 0.05          # and agrees exactly with glibc on the vast majority of arguments


def test_fixed_cbrt_is_within_one_ulp_of_libm(orc):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1, 100000), 10.0 ** rng.uniform(-300, 300, 60000), [1.0, 8.0, 27.0, 1e-12, 0.125]])
    d = ulp_diff(orc.fixed_cbrt(x), np.array([math.cbrt(v) if hasattr(math, "cbrt") else np.cbrt(v) for v in x]))
    assert d.max() <= 1
    assert (d > 0).mean() < 0.02
    assert orc.fixed_cbrt(np.array([27.0, 0.125, 1e-12]))[0] == 3.0


def test_stated_order_dot_is_an_ordinary_dot_product(orc):
    rng = np.random.default_rng(2)
    for d in (1, 3, 4, 15, 16, 17, 37, 64, 512, 768):
        a, b = rng.standard_normal(d), rng.standard_normal(d)
        ref = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        got = orc.cosine(a, b)
        with orc.plain_arith():
            seq = orc.cosine(a, b)
        assert abs(got - ref) < 1e-14 and abs(seq - ref) < 1e-14
    assert orc.cosine(np.zeros(8), np.ones(8)) == 0.0                 # zero-norm guard


CASES = [("clipper", {}, 30, 30, 0, 1000), ("gravity", {}, 40, 40, 0, 11), ("semanticgrav", {"semantics_dim": 64}, 60, 50, 64, 12),
         ("roman", {"semantics_dim": 32}, 50, 50, 32, 13), ("sevg", {"semantics_dim": 16, "epsilon_shape": 0.3}, 45, 40, 16, 14),
         ("semanticgrav", {"semantics_dim": 37}, 45, 45, 37, 25)]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}_{c[5]}" for c in CASES])
def test_plain_and_stated_arithmetic_agree_off_threshold(orc, case):
    method, kw, n, m, d, seed = case
    reg = registration_for(method, **kw)
    P = reg._abi_params()
    pr = synth.make_pair(n, m, d, seed, tilt_deg=1.0 if P.gravity_guided else 0.0)
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    mat, _ = orc.build_matrix(P, D1, D2)
    sol = orc.solve(P, mat)
    with orc.plain_arith():
        mat_p, _ = orc.build_matrix(P, D1, D2)
        sol_p = orc.solve(P, mat_p)
    rp, cc, vv, dd = mat.export(); rp2, cc2, vv2, dd2 = mat_p.export()
    assert np.array_equal(rp, rp2) and np.array_equal(cc, cc2)
    # values move by rounding only (the rescaling (cos - cos_min)/(cos_max - cos_min) amplifies an ulp of the cosine)
    assert np.allclose(vv, vv2, rtol=1e-11, atol=0) and np.allclose(dd, dd2, rtol=1e-11, atol=0)
    assert np.array_equal(sol["nodes"], sol_p["nodes"])
    assert sol["stats"].n_pass == sol_p["stats"].n_pass


def _pattern(orc, P, D1, D2):
    mat, _ = orc.build_matrix(P, D1, D2)
    rp, cc, vv, dd = mat.export()
    rows = np.repeat(np.arange(mat.n), np.diff(rp))
    return set(zip(rows.tolist(), cc.tolist())), dict(zip(zip(rows.tolist(), cc.tolist()), vv.tolist())), dd, mat


def test_gravity_mode_switch_h2(orc):
    reg = registration_for("gravity")
    pr = synth.make_pair(45, 45, 0, 31, tilt_deg=2.0)
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    P = type(reg._abi_params()).from_buffer_copy(reg._abi_params())
    pats = {}
    for mode in (_abi.ROMAN_GRAV_COMBINED, _abi.ROMAN_GRAV_SEPARATE, _abi.ROMAN_GRAV_ZGATE):
        P.gravity_mode = mode
        pats[mode] = _pattern(orc, P, D1, D2)
    comb, sep, zg = pats[0], pats[1], pats[2]
    # sqrt(ch^2 + cv^2) < eps implies ch < eps and cv < eps: the separate-gate pattern contains the combined one,
    # with the same values where both keep an entry
    assert comb[0] <= sep[0] and len(sep[0]) > len(comb[0])
    assert all(sep[1][k] == v for k, v in comb[1].items())
    # the z-gate reading scores |l1 - l2| like the plain invariant and only removes entries from it
    Pe = type(P).from_buffer_copy(P); Pe.gravity_guided = 0
    plain = _pattern(orc, Pe, D1, D2)
    assert zg[0] <= plain[0] and len(zg[0]) < len(plain[0])
    assert all(plain[1][k] == v for k, v in zg[1].items())
    # planted inliers survive every reading
    for mode in pats:
        P.gravity_mode = mode
        sol = orc.solve(P, pats[mode][3])
        A = orc.create_all_to_all(45, 45)
        got = set(map(tuple, A[sol["nodes"]].tolist()))
        truth = set(map(tuple, pr.inliers.tolist()))
        assert len(got & truth) >= 0.8 * len(truth)


def test_single_mode_switch_h3(orc):
    reg = registration_for("roman", semantics_dim=16)
    pr = synth.make_pair(36, 36, 16, 8, tilt_deg=1.0)
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    P = type(reg._abi_params()).from_buffer_copy(reg._abi_params())
    out = {}
    for mode in (_abi.ROMAN_SINGLE_BOTH, _abi.ROMAN_SINGLE_OFFDIAG, _abi.ROMAN_SINGLE_DIAG):
        P.single_mode = mode
        out[mode] = _pattern(orc, P, D1, D2)
    s = orc.single_scores(P, D1, D2)
    live = s > 0
    both, off, diag = out[0], out[1], out[2]
    assert np.array_equal(both[2], s) and np.array_equal(diag[2], s)
    assert np.array_equal(off[2], live.astype(float))                 # identity diagonal on the live associations
    assert both[1] == off[1]                                          # same off-diagonal fusion
    Pg = type(P).from_buffer_copy(P); Pg.ratio_feature_dim = 0; Pg.cos_feature_dim = 0   # the pair score alone
    # DIAG: off-diagonals are the unfused pair score, restricted to live associations
    A = orc.create_all_to_all(36, 36)
    m2, _ = orc.build_matrix(Pg, D1[:, :3].copy(), D2[:, :3].copy())
    rp, cc, vv, _ = m2.export()
    rows = np.repeat(np.arange(m2.n), np.diff(rp))
    expect = {(p, q): v for p, q, v in zip(rows.tolist(), cc.tolist(), vv.tolist()) if live[p] and live[q]}
    assert diag[1] == expect
    for mode in out:                                                  # dead associations never enter a solution
        P.single_mode = mode
        sol = orc.solve(P, out[mode][3])
        assert np.all(live[sol["nodes"]]) and np.all(sol["u"][~live] == 0.0)


def test_single_mode_diag_keep_removes_nothing(orc):
    """ROMAN_SINGLE_DIAG_KEEP (VERDICT r2 missing #3: "a zero single score does NOT remove the association"): every
    association stays live, off-diagonals are the unfused pair score over ALL associations, the diagonal is the
    single score (0 where it vanishes)."""
    reg = registration_for("roman", semantics_dim=16)
    pr = synth.make_pair(36, 36, 16, 8, tilt_deg=1.0)
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    P = type(reg._abi_params()).from_buffer_copy(reg._abi_params())
    P.single_mode = _abi.ROMAN_SINGLE_DIAG_KEEP
    keep = _pattern(orc, P, D1, D2)
    s = orc.single_scores(P, D1, D2)
    assert (s == 0).any() and np.array_equal(keep[2], s)              # zero diagonals are part of the problem
    Pg = type(P).from_buffer_copy(P); Pg.ratio_feature_dim = 0; Pg.cos_feature_dim = 0   # the pair score alone
    m2, _ = orc.build_matrix(Pg, D1[:, :3].copy(), D2[:, :3].copy())
    rp, cc, vv, _ = m2.export()
    rows = np.repeat(np.arange(m2.n), np.diff(rp))
    assert keep[1] == {(p, q): v for p, q, v in zip(rows.tolist(), cc.tolist(), vv.tolist())}
    # faithful (every pair scored) and pruned builds agree: nothing is pruned in this reading
    mf, _ = orc.build_matrix(P, D1, D2, faithful=True)
    assert mf.nnz == keep[3].nnz
    sol = orc.solve(P, keep[3])
    assert sol["stats"].n_live == 36 * 36
    # DIAG is the same reading restricted to the associations with a non-zero single score
    P.single_mode = _abi.ROMAN_SINGLE_DIAG
    diag = _pattern(orc, P, D1, D2)
    live = s > 0
    assert diag[1] == {k: v for k, v in keep[1].items() if live[k[0]] and live[k[1]]}


def test_removed_associations_ignore_u0(orc):
    reg = registration_for("semanticgrav", semantics_dim=16)
    pr = synth.make_pair(30, 30, 16, 3, tilt_deg=1.0)
    P = reg._abi_params()
    mat, _ = orc.build_matrix(P, reg.pack(pr.map1), reg.pack(pr.map2))
    dead = mat.export()[3] == 0
    assert dead.any()
    u0 = np.ones(mat.n)
    a = orc.solve(P, mat, u0)
    u0[dead] = 123.0
    b = orc.solve(P, mat, u0)
    assert np.array_equal(a["nodes"], b["nodes"]) and np.array_equal(a["u"], b["u"])
