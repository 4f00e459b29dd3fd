"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: bit-exact integer/index results (live list, sparsity pattern, selected nodes in
order, association indices); M values within a few ulp (exp/cbrt/pow are the only inexact ops);
pose within 1e-5 Frobenius (north_star) — in practice ~1e-15."""
import os

import numpy as np
import pytest

from conftest import registration_for, ulp_diff
from roman_amd import _abi, synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-5          # BASELINE.json north_star: "pose within 1e-5 Frobenius"

LADDER = [  # (id, method, kwargs, n, m, d, seed)
    ("cfg1", "clipper", {}, 30, 30, 0, 1000),
    ("clipper_ragged", "clipper", {}, 17, 41, 0, 21),
    ("clipper_2d", "clipper", {"dim": 2}, 25, 25, 0, 22),
    ("gravity40", "gravity", {}, 40, 40, 0, 11),
    ("pcavolgrav", "pcavolgrav", {"epsilon_shape": 0.2}, 40, 36, 0, 23),
    ("extentvolgrav", "extentvolgrav", {"epsilon_shape": 0.1}, 33, 40, 0, 24),
    ("semgrav60", "semanticgrav", {"semantics_dim": 64}, 60, 50, 64, 12),
    ("semgrav_oddd", "semanticgrav", {"semantics_dim": 37}, 45, 45, 37, 25),
    ("roman50", "roman", {"semantics_dim": 32}, 50, 50, 32, 13),
    ("sevg45", "sevg", {"semantics_dim": 16, "epsilon_shape": 0.3}, 45, 40, 16, 14),
    ("spv", "spv", {}, 30, 30, 8, 26),
    ("prune60", "clipper+prune", {"cosine_min": 0.5}, 60, 60, 64, 15),
    ("cfg2", "semanticgrav", {"semantics_dim": 512}, 200, 200, 512, 2000),
    ("roman_d768", "roman", {}, 40, 40, 768, 27),
    ("gravity100", "gravity", {}, 100, 100, 0, 7),
    # edge shapes of the sparse build: one mask word per row (no lower-triangle blocks), two to three words, and
    # rows with several hundred candidates (slices wider than one LDS image pass of the fill kernel)
    ("tiny_4x5", "clipper", {}, 4, 5, 0, 32),
    ("words2_9x10", "gravity", {}, 9, 10, 0, 33),
    ("words3_12x14", "clipper", {}, 12, 14, 0, 34),
    ("dense45", "clipper", {"epsilon": 1.5, "sigma": 0.8}, 45, 45, 0, 31),
    # maps of more than 256 objects: one table slice per wave in the pair tests (no second row), table rows loaded
    # without the register prefetch, and a live set beyond the streaming solver's size (SELL-64 fill and solver)
    ("maps300", "semanticgrav", {"semantics_dim": 32, "cosine_min": 0.6, "cosine_max": 0.8}, 300, 300, 32, 41),
    # the alternative readings of the two formulas pinned by decision (include/roman_hip.h ROMAN_GRAV_*, ROMAN_SINGLE_*)
    ("grav_separate", "gravity", {"_gravity_mode": 1}, 45, 45, 0, 31),
    ("grav_zgate", "gravity", {"_gravity_mode": 2}, 45, 45, 0, 31),
    ("semgrav_separate", "semanticgrav", {"semantics_dim": 64, "_gravity_mode": 1}, 60, 50, 64, 12),
    ("semgrav_zgate_200", "semanticgrav", {"semantics_dim": 128, "_gravity_mode": 2}, 200, 200, 128, 2001),
    ("roman_offdiag", "roman", {"semantics_dim": 32, "_single_mode": 1}, 50, 50, 32, 13),
    ("roman_diagonly", "roman", {"semantics_dim": 32, "_single_mode": 2}, 50, 50, 32, 13),
    ("sevg_offdiag_sep", "sevg", {"semantics_dim": 16, "epsilon_shape": 0.3, "_single_mode": 1, "_gravity_mode": 1}, 45, 40, 16, 14),
    # ROMAN_SINGLE_DIAG_KEEP: a zero single score removes nothing (every association live, zero diagonals): once inside the
    # stream layout (L = 2500), once beyond it (L = 8100: symmetric SELL-64 + the whole-device solver k_solve_wide)
    ("roman_diagkeep", "roman", {"semantics_dim": 32, "_single_mode": 3}, 50, 50, 32, 13),
    ("semgrav_diagkeep_90", "semanticgrav", {"semantics_dim": 64, "_single_mode": 3}, 90, 90, 64, 17),
]


def make(case):
    _, method, kw, n, m, d, seed = case
    reg = registration_for(method, **{k: v for k, v in kw.items() if not k.startswith("_")})
    reg._abi_params().gravity_mode = kw.get("_gravity_mode", 0)        # decision switches: not part of the reference's
    reg._abi_params().single_mode = kw.get("_single_mode", 0)          # parameter set, set on the ABI block directly
    pr = synth.make_pair(n, m, d, seed, tilt_deg=1.0 if reg._abi_params().gravity_guided else 0.0)
    if kw.get("dim") == 2:
        for o in pr.map1 + pr.map2:
            o.centroid = o.centroid[:2]; o.dim = 2
    return reg, pr


@pytest.mark.parametrize("lists", ["0", "1"], ids=["four_list_kernels", "k_lists"])
@pytest.mark.parametrize("case", LADDER, ids=[c[0] for c in LADDER])
def test_stagewise_parity(ctx, orc, case, lists, monkeypatch):
    # positions and kept-candidate lists of the stream layout: through the symmetric bit matrix (k_mirror, k_rowprefix, k_rowsort,
    # k_upper) or from its upper blocks in one kernel (k_lists, what batches of >= 32 problems take): the same matrix either way
    monkeypatch.setenv("ROMAN_LISTS", lists)
    reg, pr = make(case)
    reg.set_context(ctx)
    P = reg._abi_params()
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    A = reg._association_list(pr.map1, pr.map2)
    mat, Ao = orc.build_matrix(P, D1, D2, A)
    sol = orc.solve(P, mat)
    ctx.score(P, D1, D2, A)

    # (1) live association list: identical indices; single scores to rounding
    s_o = orc.single_scores(P, D1, D2, Ao)
    live_o = np.arange(len(s_o)) if P.single_mode == _abi.ROMAN_SINGLE_DIAG_KEEP else np.nonzero(s_o > 0)[0]
    idx, sc = ctx.live()
    assert np.array_equal(idx, live_o)
    assert np.array_equal(sc, s_o[live_o])          # bit-identical: stated-order dot / norms, exact ops otherwise

    # (2) affinity matrix: bit-identical sparsity pattern, values within a few ulp
    rp_o, c_o, v_o, d_o = mat.export()
    rp, cc, vv, dd = ctx.upper_csr()
    assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o)
    assert np.array_equal(vv, v_o)                  # bit-identical values: fixed-sequence exp / cbrt on both sides
    assert np.array_equal(dd, d_o)

    # (3) solver: same selected nodes in the same order, same trajectory
    ctx.solve(None)
    nodes, u, score, st = ctx.solution()
    so = sol["stats"]
    assert np.array_equal(nodes, sol["nodes"])
    assert np.max(np.abs(u - sol["u"])) < 1e-9 if u.size else True
    assert abs(score - so.score) < 1e-8 * max(1.0, abs(so.score))
    assert (st.n_live, st.nnz_upper, st.n_assoc_in) == (so.n_live, so.nnz_upper, so.n_assoc_in)
    assert (st.n_pass, st.outer_iters, st.inner_iters, st.ls_trials) == (so.n_pass, so.outer_iters, so.inner_iters, so.ls_trials)

    # (4) associations + pose through the fused batch entry
    sel = ctx.selected_associations()
    assert np.array_equal(sel, Ao[sol["nodes"]])
    res = reg.register_and_align_batch([(pr.map1, pr.map2)])
    assert np.array_equal(res.assoc[0], sel)
    d = reg.dim
    if len(sel) >= d:
        p1 = np.array([pr.map1[i].center.ravel()[:d] for i, _ in sel]); p2 = np.array([pr.map2[j].center.ravel()[:d] for _, j in sel])
        T_o = orc.t_align(p1, p2, d)
        assert res.status[0] == 0
        assert np.linalg.norm(res.T[0] - T_o) < POSE_TOL
        assert np.linalg.norm(reg.T_align(pr.map1, pr.map2, sel) - T_o) < POSE_TOL
    # the planted inliers come back
    if case[3] >= 20 and "epsilon" not in case[2]:              # (edge shapes are about parity, not about recall)
        got = set(map(tuple, sel.tolist())); truth = set(map(tuple, pr.inliers.tolist()))
        assert len(got & truth) >= 0.85 * len(truth)


@pytest.mark.parametrize("lists", ["0", "1"], ids=["k_rowsort", "k_lists"])
@pytest.mark.parametrize("name", ["cfg2", "roman50", "dense45", "tiny_4x5", "words3_12x14", "roman_diagkeep", "gravity40", "semgrav_zgate_200"])
def test_positions_without_a_sort_equal_the_sorted_ones(ctx, orc, name, lists, monkeypatch):
    """The stream layout's positions — rank by (degree descending, row ascending) — come from a histogram of the degrees, its scan
    and ranks inside the ranges of equal degree (place_keys, kernels.hip.h); the bitonic sort of the keys they stand for remains for
    live sets in which very many rows share a degree.  ROMAN_SORT_EQMAX sets where: default (512), 0 (always the sort), 2 (the sort
    as soon as three rows share a degree: both branches inside one process), 100000 (never the sort, whatever the degrees: dense45
    and the DIAG_KEEP reading have hundreds of rows per degree).  Every setting: the oracle's matrix, the oracle's nodes in the
    oracle's order, the same iterate bit for bit (the solver's sums are exact: the layout cannot show in them), the same pass counts."""
    case = next(c for c in LADDER if c[0] == name)
    reg, pr = make(case)
    reg.set_context(ctx)
    P = reg._abi_params()
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    A = reg._association_list(pr.map1, pr.map2)
    mat, Ao = orc.build_matrix(P, D1, D2, A)
    sol = orc.solve(P, mat)
    rp_o, c_o, v_o, d_o = mat.export()
    monkeypatch.setenv("ROMAN_LISTS", lists)
    first = None
    for eq in (None, "0", "2", "100000"):
        if eq is None:
            monkeypatch.delenv("ROMAN_SORT_EQMAX", raising=False)
        else:
            monkeypatch.setenv("ROMAN_SORT_EQMAX", eq)
        ctx.score(P, D1, D2, A)
        rp, cc, vv, dd = ctx.upper_csr()
        assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o) and np.array_equal(vv, v_o) and np.array_equal(dd, d_o), eq
        ctx.solve(None)
        nodes, u, score, st = ctx.solution()
        assert np.array_equal(nodes, sol["nodes"]), eq
        assert (st.n_pass, st.outer_iters, st.inner_iters, st.ls_trials) == (sol["stats"].n_pass, sol["stats"].outer_iters, sol["stats"].inner_iters, sol["stats"].ls_trials), eq
        res = reg.register_and_align_batch([(pr.map1, pr.map2)] * 3)
        got = (u.copy(), [a.copy() for a in res.assoc], res.T.copy())
        if first is None:
            first = got
        else:
            assert np.array_equal(got[0], first[0]), eq
            assert all(np.array_equal(a, b) for a, b in zip(got[1], first[1])) and np.array_equal(got[2], first[2], equal_nan=True), eq     # (fewer associations than dimensions: the NaN pose)


def test_device_arithmetic_is_bit_exact(ctx, orc):
    """The ops every threshold is applied to (+,-,*,/,sqrt,fma and the two fixed-sequence functions) give the
    host's bits on gfx950; the device's own libm stays within 2 ulp of glibc (it is only used for pow with
    non-integer weights)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(0, 1000, 200000), rng.uniform(0, 1e-3, 50000), 10.0 ** rng.uniform(-300, 300, 50000)])
    assert np.array_equal(ctx.debug_math(0, x), np.sqrt(x))
    y = rng.uniform(1e-3, 1e3, x.size)
    assert np.array_equal(ctx.debug_math(3, x, y), x / y)
    e = rng.uniform(-40, 0, 200000)
    assert ulp_diff(ctx.debug_math(1, e), np.exp(e)).max() <= 2
    c = rng.uniform(1e-12, 1, 200000)
    assert ulp_diff(ctx.debug_math(2, c), np.cbrt(c)).max() <= 2
    assert ulp_diff(ctx.debug_math(4, c, np.full_like(c, 0.25)), np.power(c, 0.25)).max() <= 2
    # fma, and the fixed sequences of exact operations that replace exp / cbrt on the scoring path
    a, b = rng.standard_normal(100000), rng.standard_normal(100000)
    from fractions import Fraction
    got = ctx.debug_math(7, a, b)
    for k in range(0, 100000, 997):
        assert got[k] == float(Fraction(a[k]) * Fraction(b[k]) + Fraction(a[k]))
    e2 = np.concatenate([-rng.uniform(0, 2, 200000), rng.uniform(-40, 40, 50000), [0.0, -1.125, -650.0]])
    assert np.array_equal(ctx.debug_math(5, e2), orc.fixed_exp(e2))
    c2 = np.concatenate([rng.uniform(0, 1, 200000), 10.0 ** rng.uniform(-30, 3, 50000), [1.0, 0.125, 1e-12]])
    assert np.array_equal(ctx.debug_math(6, c2), orc.fixed_cbrt(c2))


@pytest.mark.parametrize("n1,n2,d", [(16, 16, 4), (37, 53, 70), (200, 200, 512), (5, 3, 1), (1, 1, 9)])
def test_mfma_cosine_kernel(ctx, n1, n2, d):
    """v_mfma_f64_16x16x4 tile kernel against numpy on an asymmetric operand (catches row/col swaps)."""
    rng = np.random.default_rng(n1 * 1000 + n2 + d)
    P = _abi.RomanParams.default(); P.cos_feature_dim = d
    D1 = rng.standard_normal((n1, 3 + d)); D2 = rng.standard_normal((n2, 3 + d))
    D1[:, 3:] *= np.linspace(0.5, 2.0, d)
    if n1 > 2:
        D1[2, 3:] = 0.0
    got = ctx.debug_cosine(P, D1, D2)
    a, b = D1[:, 3:], D2[:, 3:]
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = (a @ b.T) / np.outer(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1))
    ref[~np.isfinite(ref)] = 0.0
    assert np.max(np.abs(got - ref)) < 1e-14


@pytest.mark.parametrize("n1,n2,d", [(16, 16, 4), (20, 24, 16), (37, 53, 70), (64, 64, 512), (9, 7, 768), (5, 3, 1),
                                     # the tiled kernel's cases: several balanced tiles, waves owning block rows (n1 >= n2 per tile) and block
                                     # columns (the transposed product), descriptor lengths with 0, 1, 2, 3 full stages of 16 and a ragged one
                                     (65, 17, 33), (200, 200, 48), (130, 40, 16), (40, 130, 31), (49, 49, 15), (1, 1, 16), (63, 200, 96),
                                     # maps of at most 48 objects: the one-wave-per-problem kernel (k_cos_wave), 1-3 blocks per dimension
                                     (40, 40, 768), (48, 48, 100), (33, 47, 70), (17, 48, 16), (32, 16, 37), (48, 1, 15)])
def test_cosine_bits_equal_the_oracles_stated_order(ctx, orc, monkeypatch, n1, n2, d):
    """The f64 matrix-core contraction accumulates in the order the oracle states (dot_fixed / norm_fixed in
    oracle/clipper_oracle.c): the cosine matrix is BIT-identical, so the cosine gate decides on the same value.  A call of one
    problem takes the one-wave-per-block kernel (k_cos_block) by itself; ROMAN_COS_BLOCK=0 gives it the tile / one-wave kernels."""
    rng = np.random.default_rng(n1 * 977 + n2 * 13 + d)
    P = _abi.RomanParams.default(); P.cos_feature_dim = d
    D1 = rng.standard_normal((n1, 3 + d)); D2 = rng.standard_normal((n2, 3 + d))
    ref = np.array([[orc.cosine(D1[i, 3:], D2[j, 3:]) for j in range(n2)] for i in range(n1)])
    for setting in (None, "0"):
        if setting is None:
            monkeypatch.delenv("ROMAN_COS_BLOCK", raising=False)
        else:
            monkeypatch.setenv("ROMAN_COS_BLOCK", setting)
        assert np.array_equal(ctx.debug_cosine(P, D1, D2), ref), setting


@pytest.mark.parametrize("n1,n2,d", [(40, 40, 768), (48, 48, 100), (33, 47, 70), (17, 48, 16), (32, 16, 37), (48, 1, 15), (1, 48, 160), (20, 37, 7), (16, 16, 128), (9, 7, 768),
                                     # larger maps in a call of one problem: k_cos_block as well (one wave per block) against the tile kernel
                                     (200, 200, 512), (137, 53, 70), (65, 17, 33), (49, 300, 16), (300, 130, 129)])
def test_demo_scale_cosine_kernels_give_the_oracles_bits(ctx, orc, monkeypatch, n1, n2, d):
    """Maps of at most 48 objects: one wave per problem (k_cos_wave, what a batch takes) and one wave per 16 x 16 block (k_cos_block,
    what a serial caller's single pair takes: ROMAN_COS_BLOCK forces either; for larger maps the switch chooses between k_cos_block
    and the tile kernel k_cos_tile) — 1 to 3 blocks per dimension, descriptor lengths
    below one chunk of 16, below and above the eight chunks of k_cos_block's load ring, ragged tails: both BIT-identical to the
    oracle's stated order."""
    rng = np.random.default_rng(n1 * 977 + n2 * 13 + d)
    P = _abi.RomanParams.default(); P.cos_feature_dim = d
    D1 = rng.standard_normal((n1, 3 + d)); D2 = rng.standard_normal((n2, 3 + d))
    if n1 > 2:
        D1[2, 3:] = 0.0                                  # a zero descriptor: cosine 0 by the reference's guard
    ref = np.array([[orc.cosine(D1[i, 3:], D2[j, 3:]) for j in range(n2)] for i in range(n1)])
    for setting in ("0", "1"):
        monkeypatch.setenv("ROMAN_COS_BLOCK", setting)
        assert np.array_equal(ctx.debug_cosine(P, D1, D2), ref), f"ROMAN_COS_BLOCK={setting}"


@pytest.mark.parametrize("n1,n2,d", [(200, 200, 512), (200, 200, 48), (200, 200, 7), (113, 97, 31), (300, 130, 33), (130, 300, 16), (224, 225, 17),
                                     (1, 1, 16), (16, 112, 64), (49, 49, 15), (64, 64, 512), (5, 3, 1), (111, 113, 80), (80, 96, 24), (81, 79, 40)])
def test_dealt_cosine_kernel_gives_the_oracles_bits(ctx, orc, monkeypatch, n1, n2, d):
    """k_cos_deal (tiles of up to 5 x 5 blocks whose blocks are dealt to the waves as equal runs; what a batch of mid-size maps
    takes) forced on for a single problem: one tile and several (exactly 5 and 6 blocks per dimension among them), runs that
    start in the middle of a block row, waves without a block, ragged descriptor lengths, norms handed over through LDS —
    BIT-identical to the oracle's stated order."""
    rng = np.random.default_rng(n1 * 977 + n2 * 13 + d)
    P = _abi.RomanParams.default(); P.cos_feature_dim = d
    D1 = rng.standard_normal((n1, 3 + d)); D2 = rng.standard_normal((n2, 3 + d))
    if n1 > 2:
        D1[2, 3:] = 0.0                                  # a zero descriptor: cosine 0 by the reference's guard
    monkeypatch.setenv("ROMAN_COS_DEAL", "1")
    got = ctx.debug_cosine(P, D1, D2)
    monkeypatch.setenv("ROMAN_COS_DEAL", "0")
    other = ctx.debug_cosine(P, D1, D2)
    ref = np.array([[orc.cosine(D1[i, 3:], D2[j, 3:]) for j in range(n2)] for i in range(n1)])
    assert np.array_equal(got, ref)
    assert np.array_equal(other, ref)


def test_large_single_cosine_product_takes_the_dealt_kernel_by_itself(ctx, monkeypatch):
    """A single product large enough for two workgroups per compute unit of k_cos_deal's tiles (the submap-descriptor gate of a
    long session: thousands of submaps a side) picks that kernel without the switch, its tiles spread over all XCDs (fewer than
    eight problems): equal to numpy within rounding, and bit-identical to k_cos_tile on the same operands."""
    rng = np.random.default_rng(77)
    A = rng.standard_normal((1900, 40)); Bm = rng.standard_normal((1830, 40))
    A[5] = 0.0
    monkeypatch.delenv("ROMAN_COS_DEAL", raising=False)
    got = ctx.cosine_matrix(A, Bm)
    monkeypatch.setenv("ROMAN_COS_DEAL", "0")
    other = ctx.cosine_matrix(A, Bm)
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = (A @ Bm.T) / np.outer(np.linalg.norm(A, axis=1), np.linalg.norm(Bm, axis=1))
    ref[~np.isfinite(ref)] = 0.0
    assert np.max(np.abs(got - ref)) < 1e-14
    assert np.array_equal(got, other)


def test_both_cosine_kernels_give_the_same_bits(ctx, tmp_path):
    """ROMAN_COS=0 selects the per-wave kernel k_cos (32x32 tile per wave, operands from global memory) that k_cos_tile
    replaced as the default: same contraction order per element, same bits.  The switch is read once per process, so the
    other kernel runs in a child process."""
    import subprocess, sys
    rng = np.random.default_rng(5)
    P = _abi.RomanParams.default(); P.cos_feature_dim = 70
    D1 = rng.standard_normal((137, 73)); D2 = rng.standard_normal((53, 73))
    here = ctx.debug_cosine(P, D1, D2)
    np.savez(tmp_path / "in.npz", D1=D1, D2=D2)
    code = ("import numpy as np, sys; from roman_amd import _abi; from roman_amd.runtime import Context\n"
            "z = np.load(sys.argv[1]); P = _abi.RomanParams.default(); P.cos_feature_dim = 70\n"
            "c = Context(0); np.save(sys.argv[2], c.debug_cosine(P, z['D1'], z['D2'])); c.close()\n")
    env = dict(os.environ, ROMAN_COS="0", PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + sys.path))
    subprocess.run([sys.executable, "-c", code, str(tmp_path / "in.npz"), str(tmp_path / "out.npy")], check=True, env=env, timeout=300)
    assert np.array_equal(here, np.load(tmp_path / "out.npy"))


def _nudge(x, k):
    for _ in range(abs(k)):
        x = np.nextafter(x, np.inf if k > 0 else -np.inf)
    return float(x)


def test_associations_on_the_cosine_threshold(ctx, orc):
    """Associations whose cosine sits exactly ON cos_min, one ulp below and one ulp above it: the live set is the
    oracle's in every case and flips where it must (`c > 0` after the rescaling, strict)."""
    rng = np.random.default_rng(3)
    n, d = 24, 96
    reg = registration_for("semanticgrav", semantics_dim=d); reg.set_context(ctx)
    pr = synth.make_pair(n, n, d, 77, tilt_deg=1.0)
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    cos = np.array([[orc.cosine(D1[i, 3:], D2[j, 3:]) for j in range(n)] for i in range(n)])
    flat = np.sort(cos.ravel())
    targets = [flat[len(flat) // 2], flat[int(0.9 * len(flat))], flat[int(0.97 * len(flat))]]   # three distinct on-threshold cases
    seen_flip = 0
    for t in targets:
        lives = []
        for k in (-1, 0, 1):
            P = type(reg._abi_params()).from_buffer_copy(reg._abi_params())
            P.cosine_min = _nudge(t, k); P.cosine_max = P.cosine_min + 0.2
            s_o = orc.single_scores(P, D1, D2)
            ctx.score(P, D1, D2, None)
            idx, sc = ctx.live()
            assert np.array_equal(idx, np.nonzero(s_o > 0)[0]) and np.array_equal(sc, s_o[idx])
            mat, _ = orc.build_matrix(P, D1, D2)
            rp_o, c_o, v_o, _ = mat.export()
            rp, cc, vv, _ = ctx.upper_csr()
            assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o) and np.array_equal(vv, v_o)
            lives.append(len(idx))
        on = int((cos == t).sum())
        assert lives[0] - lives[1] == on and lives[1] >= lives[2]     # cos == cos_min is dead, one ulp below cos_min it is live
        seen_flip += on
    assert seen_flip >= 3


@pytest.mark.parametrize("method", ["clipper", "gravity"])
def test_pairs_on_the_epsilon_threshold(ctx, orc, method):
    """Object pairs whose length difference is exactly epsilon, and epsilon moved by +-1 and +-2 ulp around it:
    the sparsity pattern is the oracle's in every case (`c < epsilon`, strict) and changes where it must."""
    reg = registration_for(method); reg.set_context(ctx)
    # map 1: points on the x axis 3 apart; map 2: points 2.4 apart -> |3 - fl(2.4)| is exactly one ulp above fl(0.6)
    xs1 = [0.0, 3.0, 6.0, 9.0, 12.5]; xs2 = [0.0, 2.4, 4.8, 7.2, 12.5]
    D1 = np.array([[x, 0.0, 0.0] for x in xs1]); D2 = np.array([[x, 0.0, 0.0] for x in xs2])
    c_exact = 3.0 - 2.4
    sizes = []
    for k in (-2, -1, 0, 1, 2):
        P = type(reg._abi_params()).from_buffer_copy(reg._abi_params())
        P.epsilon = _nudge(c_exact, k); P.mindist = 0.0
        mat, _ = orc.build_matrix(P, D1, D2)
        rp_o, c_o, v_o, _ = mat.export()
        ctx.score(P, D1, D2, None)
        rp, cc, vv, _ = ctx.upper_csr()
        assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o) and np.array_equal(vv, v_o)
        sizes.append(len(c_o))
    assert sizes[0] == sizes[1] == sizes[2] < sizes[3] == sizes[4]    # c == epsilon is rejected, epsilon one ulp larger accepts it


def test_entries_on_the_affinityeps_threshold(ctx, orc):
    """Single scores small enough that fused entries straddle affinityeps: kept and dropped entries are the
    oracle's (the gate is applied to values both sides compute with the same fixed operation sequences)."""
    n, d = 30, 48
    reg = registration_for("semanticgrav", semantics_dim=d); reg.set_context(ctx)
    pr = synth.make_pair(n, n, d, 78, tilt_deg=1.0)
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    P0 = reg._abi_params()
    mat, _ = orc.build_matrix(P0, D1, D2)
    vals = np.sort(mat.export()[2])
    assert vals.size > 50
    for t in (vals[3], vals[len(vals) // 3], vals[len(vals) // 2]):
        for k in (-1, 0, 1):
            P = type(P0).from_buffer_copy(P0); P.affinityeps = _nudge(t, k)
            m2, _ = orc.build_matrix(P, D1, D2)
            rp_o, c_o, v_o, _ = m2.export()
            ctx.score(P, D1, D2, None)
            rp, cc, vv, _ = ctx.upper_csr()
            assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o) and np.array_equal(vv, v_o)
            assert (v_o > P.affinityeps).all()


def test_explicit_u0_and_no_rescale(ctx, orc):
    reg = registration_for("gravity"); reg.set_context(ctx)
    pr = synth.make_pair(30, 28, 0, 31)
    P = reg._abi_params()
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    u0 = np.random.default_rng(5).uniform(0.1, 1.0, 30 * 28)
    for rescale in (1, 0):
        P.rescale_u0 = rescale
        mat, _ = orc.build_matrix(P, D1, D2)
        sol = orc.solve(P, mat, u0)
        ctx.score(P, D1, D2, None); ctx.solve(u0)
        nodes, u, _, st = ctx.solution()
        assert np.array_equal(nodes, sol["nodes"]) and np.max(np.abs(u - sol["u"])) < 1e-9
        assert st.n_pass == sol["stats"].n_pass


def test_fusion_variants_and_weights(ctx, orc):
    pr = synth.make_pair(30, 30, 16, 32)
    for fusion, wd, wr, wc in [(1, 1.0, 1.0, 1.0), (2, 1.0, 1.0, 1.0), (0, 2.0, 1.0, 3.0), (0, 0.5, 2.0, 1.0)]:
        reg = registration_for("roman", semantics_dim=16); reg.set_context(ctx)
        P = reg._abi_params()
        P.fusion_method = fusion; P.distance_weight, P.ratio_weight, P.cosine_weight = wd, wr, wc
        D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
        mat, Ao = orc.build_matrix(P, D1, D2)
        sol = orc.solve(P, mat)
        ctx.score(P, D1, D2, None)
        rp, cc, vv, dd = ctx.upper_csr(); rp_o, c_o, v_o, d_o = mat.export()
        assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o)
        assert np.allclose(vv, v_o, rtol=1e-12, atol=0) and np.allclose(dd, d_o, rtol=1e-12, atol=0)
        ctx.solve(None)
        assert np.array_equal(ctx.solution()[0], sol["nodes"])


@pytest.mark.parametrize("offset", [0.0, 1.0e3, 3.0e7])
def test_pair_tests_far_from_the_origin(ctx, orc, offset):
    """Both maps shifted far from the origin (the tables then hold differences of large coordinates): pattern and values
    still equal the oracle's on the same shifted inputs — every gate is a stated sequence of exactly rounded operations."""
    reg = registration_for("semanticgrav", semantics_dim=32); reg.set_context(ctx)
    P = reg._abi_params()
    pr = synth.make_pair(70, 64, 32, 515, tilt_deg=1.0)
    D1, D2 = reg.pack(pr.map1).copy(), reg.pack(pr.map2).copy()
    D1[:, :3] += offset; D2[:, :3] += offset * np.array([1.0, -1.0, 0.5])
    mat, _ = orc.build_matrix(P, D1, D2)
    ctx.score(P, D1, D2, None)
    rp_o, c_o, v_o, d_o = mat.export()
    rp, cc, vv, dd = ctx.upper_csr()
    assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o) and np.array_equal(vv, v_o) and np.array_equal(dd, d_o)
    assert len(c_o) > 0


PRE_CASES = ["cfg1", "clipper_2d", "gravity40", "semgrav60", "cfg2", "gravity100", "tiny_4x5", "words3_12x14", "dense45", "maps300",
             "grav_separate", "grav_zgate", "semgrav_zgate_200", "roman_diagkeep"]


@pytest.mark.parametrize("name", PRE_CASES)
def test_pair_tests_with_candidate_generation_give_the_plain_sweeps_matrix(ctx, orc, name, monkeypatch):
    """k_count with the integer prefilter in front of the exact gate (count_rows_pre: 16-bit bins of the two table entries, survivors
    compacted into an LDS queue, the exact f64 gate on the queue; ROMAN_COUNT_PRE=1, the default) against the plain sweep that
    tests every live pair (ROMAN_COUNT_PRE=0): the same upper CSR — pattern AND values — which is the oracle's.  Replaces
    clipper.score_pairwise_and_single_consistency [REF roman/align/roman_registration.py:95]."""
    case = next(c for c in LADDER if c[0] == name)
    reg, pr = make(case)
    reg.set_context(ctx)
    P = reg._abi_params()
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    A = reg._association_list(pr.map1, pr.map2)
    mat, _ = orc.build_matrix(P, D1, D2, A)
    rp_o, c_o, v_o, _ = mat.export()
    got = {}
    for pre in ("1", "0"):
        monkeypatch.setenv("ROMAN_COUNT_PRE", pre)
        ctx.score(P, D1, D2, A)
        got[pre] = ctx.upper_csr()
        assert np.array_equal(got[pre][0], rp_o) and np.array_equal(got[pre][1], c_o) and np.array_equal(got[pre][2], v_o), pre
    assert all(np.array_equal(a, b) for a, b in zip(got["0"], got["1"]))


@pytest.mark.parametrize("scale,epsilon,noise", [(1.0, 0.6, 0.1), (400.0, 0.6, 0.1), (1.0, 0.004, 0.0005), (1.0, 30.0, 0.1), (1.0e-3, 0.6, 0.0)])
def test_candidate_generation_at_the_ends_of_the_bin_range(ctx, orc, scale, epsilon, noise, monkeypatch):
    """The prefilter's bins are epsilon / 32 wide and the last of the 32768 takes everything beyond: maps 400 times as large (distances
    up to 12 km: everything beyond 614 m shares the last bin), an epsilon of 4 mm (everything beyond 4 m shares it), an epsilon of 30 m
    and maps shrunk to centimetres (every distance in the first bins: every pair is a candidate) — the matrix stays the oracle's and
    the plain sweep's in every case (a false positive costs an exact test, never a bit)."""
    reg = registration_for("semanticgrav", semantics_dim=32, epsilon=epsilon, sigma=max(epsilon / 1.5, 1e-3), mindist=0.0 if scale < 1 else 0.2)
    reg.set_context(ctx)
    P = reg._abi_params()
    pr = synth.make_pair(90, 80, 32, 616, tilt_deg=1.0, noise=noise)
    D1, D2 = reg.pack(pr.map1).copy(), reg.pack(pr.map2).copy()
    D1[:, :3] *= scale; D2[:, :3] *= scale
    mat, _ = orc.build_matrix(P, D1, D2)
    rp_o, c_o, v_o, _ = mat.export()
    for pre in ("1", "0"):
        monkeypatch.setenv("ROMAN_COUNT_PRE", pre)
        ctx.score(P, D1, D2, None)
        rp, cc, vv, _ = ctx.upper_csr()
        assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o) and np.array_equal(vv, v_o), pre


@pytest.mark.parametrize("name", ["cfg2", "semgrav60", "roman50", "tiny_4x5", "words3_12x14", "dense45", "semgrav_zgate_200", "sevg45", "prune60", "clipper_ragged"])
def test_whole_problem_pair_tests_hand_k_lists_the_degrees(ctx, orc, name, monkeypatch):
    """Batches of at least a problem per compute unit give k_count whole problems as work items; its exact gate then counts every live
    row's degree as the pairs pass (LDS atomics) and k_lists skips its degree sweep.  Forced here for single problems
    (ROMAN_COUNT_WHOLE=1, ROMAN_LISTS=1) against the row-block path: the same layout — matrix pattern and values, the iterate bit for
    bit, selection and pass counts —, and the oracle's."""
    case = next(c for c in LADDER if c[0] == name)
    reg, pr = make(case)
    reg.set_context(ctx)
    P = reg._abi_params()
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    A = reg._association_list(pr.map1, pr.map2)
    mat, Ao = orc.build_matrix(P, D1, D2, A)
    sol = orc.solve(P, mat)
    rp_o, c_o, v_o, d_o = mat.export()
    monkeypatch.setenv("ROMAN_LISTS", "1")
    got = {}
    for whole in ("1", "0"):
        monkeypatch.setenv("ROMAN_COUNT_WHOLE", whole)
        ctx.score(P, D1, D2, A)
        rp, cc, vv, dd = ctx.upper_csr()
        assert np.array_equal(rp, rp_o) and np.array_equal(cc, c_o) and np.array_equal(vv, v_o) and np.array_equal(dd, d_o), whole
        ctx.solve(None)
        nodes, u, score, st = ctx.solution()
        assert np.array_equal(nodes, sol["nodes"]) and st.n_pass == sol["stats"].n_pass, whole
        got[whole] = (u, score)
    assert np.array_equal(got["0"][0], got["1"][0]) and got["0"][1] == got["1"][1]
