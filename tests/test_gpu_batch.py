"""Batched entry point: ragged batches, shared feature pools, explicit lists, edge cases, and
size-independent properties at BASELINE.json's full sizes (n=m=200, d=512)."""
import numpy as np
import pytest

from conftest import registration_for
from roman_amd import _abi, synth
from roman_amd.align import batch as rb

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-5


def oracle_one(orc, reg, m1, m2):
    D1, D2 = reg.pack(m1), reg.pack(m2)
    A = reg._association_list(m1, m2) if len(m1) and len(m2) else None
    return orc.register(reg._abi_params(), D1, D2, A)


def test_ragged_batch_equals_individual_oracle_solves(ctx, orc):
    reg = registration_for("semanticgrav", semantics_dim=24); reg.set_context(ctx)
    sizes = [(30, 30), (12, 40), (40, 9), (3, 3), (0, 10), (10, 0), (1, 1), (25, 26), (2, 30)]
    pairs = []
    for k, (n, m) in enumerate(sizes):
        pr = synth.make_pair(max(n, 1), max(m, 1), 24, 100 + k)
        pairs.append((pr.map1[:n], pr.map2[:m]))
    res = reg.register_and_align_batch(pairs)
    for b, (m1, m2) in enumerate(pairs):
        if len(m1) == 0 or len(m2) == 0:
            assert res.status[b] & _abi.ROMAN_ST_EMPTY_MAP and res.assoc[b].shape == (0, 2) and np.all(np.isnan(res.T[b]))
            continue
        o = oracle_one(orc, reg, m1, m2)
        assert np.array_equal(res.assoc[b], o["assoc"]), b
        assert res.stats["n_live"][b] == o["stats"].n_live and res.stats["nnz_upper"][b] == o["stats"].nnz_upper
        # Trajectory statistics are compared when M has at least one edge.  Without any consistent pair
        # the iteration normalises a vector that is zero up to rounding (a discontinuity of the
        # published algorithm), so pass counts there depend on summation order; with no live
        # association the device skips the solver outright.
        if o["stats"].nnz_upper > 0:
            assert res.stats["n_pass"][b] == o["stats"].n_pass
        if len(o["assoc"]) >= 3:
            p1 = np.array([m1[i].center.ravel() for i, _ in o["assoc"]]); p2 = np.array([m2[j].center.ravel() for _, j in o["assoc"]])
            assert res.status[b] == 0 and np.linalg.norm(res.T[b] - orc.t_align(p1, p2)) < POSE_TOL
        else:
            assert res.status[b] & _abi.ROMAN_ST_INSUFFICIENT and np.all(np.isnan(res.T[b]))


def test_all_pairs_grid_shares_submaps(ctx, orc):
    reg = registration_for("gravity"); reg.set_context(ctx)
    subs, poses = synth.make_submap_grid(5, n=30, d=0, seed0=50)
    batch = rb.batch_from_submap_grid(reg, subs[:2], subs[2:])
    assert len(batch) == 6 and batch.feats.shape[0] == 150
    res = rb.run_batch(reg, batch)
    for b, (i, j) in enumerate(batch.pair_index):
        o = oracle_one(orc, reg, subs[i], subs[2 + j])
        assert np.array_equal(res.assoc[b], o["assoc"])
        if res.status[b] == 0:                       # pose agrees with the ground-truth relative pose
            T_gt = np.linalg.inv(poses[i]) @ poses[2 + j]
            assert np.linalg.norm(res.T[b][:3, 3] - T_gt[:3, 3]) < 0.3
    assert sum(len(a) >= 10 for a in res.assoc) >= 5


def test_explicit_association_lists_in_a_batch(ctx, orc):
    reg = registration_for("clipper+prune", cosine_min=0.5); reg.set_context(ctx)
    pairs = []
    for k in range(4):
        pr = synth.make_pair(30 + 3 * k, 28, 32, 200 + k)
        pairs.append((pr.map1, pr.map2))
    batch = rb.batch_from_pairs(reg, pairs)
    assert batch.assoc is not None and batch.assoc_off[-1] == len(batch.assoc)
    res = rb.run_batch(reg, batch)
    for b, (m1, m2) in enumerate(pairs):
        assert np.array_equal(res.assoc[b], oracle_one(orc, reg, m1, m2)["assoc"])


def test_kmax_truncation_is_flagged(ctx):
    reg = registration_for("clipper"); reg.set_context(ctx)
    pr = synth.make_pair(30, 30, 0, 1000)
    b = rb.batch_from_pairs(reg, [(pr.map1, pr.map2)])
    full = ctx.align_batch(reg._abi_params(), b.feats, b.off1, b.n1, b.off2, b.n2, kmax=30)
    cut = ctx.align_batch(reg._abi_params(), b.feats, b.off1, b.n1, b.off2, b.n2, kmax=5)
    assert len(full.assoc[0]) > 5 and not (full.status[0] & _abi.ROMAN_ST_ASSOC_TRUNCATED)
    assert cut.status[0] & _abi.ROMAN_ST_ASSOC_TRUNCATED and np.array_equal(cut.assoc[0], full.assoc[0][:5])
    assert np.allclose(cut.T[0], full.T[0])          # the pose always uses every selected association


def test_tie_fallback_matches_published_heap(ctx, orc):
    """K4 with weight 0.5 -> F = 2.5 -> omega = 3 of four identical u: exact heap emulation on device."""
    n = 4
    M = np.full((n, n), 0.5); np.fill_diagonal(M, 1.0)
    C = np.ones((n, n))
    P = _abi.RomanParams.default(); P.invariant = _abi.ROMAN_INV_EUCLIDEAN
    ctx.set_matrix_data(P, M, C); ctx.solve(None)
    nodes, u, score, _ = ctx.solution()
    ref = orc.solve(P, orc.matrix_from_dense(M, C))
    assert nodes.tolist() == ref["nodes"].tolist() == [2, 1, 0]
    assert abs(score - 2.5) < 1e-12
    # a context without sizing history sends so small a problem to the ONE-WAVE instantiation of the solver (the shared
    # context of this session may or may not, depending on which tests ran before): same rounding there
    from roman_amd.runtime import Context
    fresh = Context(0)
    try:
        fresh.set_matrix_data(P, M, C); fresh.solve(None)
        nodes1, _, score1, _ = fresh.solution()
    finally:
        fresh.close()
    assert nodes1.tolist() == [2, 1, 0] and abs(score1 - 2.5) < 1e-12


def test_dense_matrix_path_and_mno_clipper(ctx, orc):
    reg = registration_for("clipper"); reg.set_context(ctx)
    pr = synth.make_pair(14, 14, 0, 33)
    M, C, A = reg.get_MCA(pr.map1, pr.map2)
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    mat, _ = orc.build_matrix(reg._abi_params(), D1, D2)
    Mo, Co = mat.dense()
    assert np.array_equal(C, Co) and np.allclose(M, Mo, rtol=1e-13, atol=0) and np.array_equal(M != 0, Mo != 0)
    sols = reg.mno_clipper(pr.map1, pr.map2, num_solutions=2)
    # the same loop on the oracle (object_registration.py:57-86)
    Mw = Mo.copy(); P = reg._abi_params(); P.invariant = _abi.ROMAN_INV_EUCLIDEAN
    for k in range(2):
        s = orc.solve(P, orc.matrix_from_dense(Mw, Co))
        assert np.array_equal(sols[k][0], A[s["nodes"]].astype(np.int64))
        u_sol = np.zeros_like(s["u"]); u_sol[s["nodes"]] = s["u"][s["nodes"]]
        assert abs(sols[k][1] - u_sol @ Mo @ u_sol / (u_sol @ u_sol)) < 1e-9
        Mw[np.ix_(s["nodes"], s["nodes"])] = 0.0
    assert len(sols[0][0]) >= 5


def _random_dense_problem(n, density, seed, flag_fraction=0.2):
    """Symmetric M (weights in (0, 1]) and C over a random pattern; a fraction of the stored pairs has C == 0 with a
    non-zero weight (the `C flag`), a few have M == 0 with C == 1 (a consistent pair of weight 0)."""
    rng = np.random.default_rng(seed)
    iu = np.triu_indices(n, 1)
    keep = rng.random(len(iu[0])) < density
    clique = rng.choice(n, size=min(n, 12), replace=False)              # a planted consistent subset, so the solve has something to find
    w = np.where(keep, rng.uniform(0.05, 1.0, len(iu[0])), 0.0)
    M = np.zeros((n, n)); C = np.zeros((n, n))
    M[iu] = w; C[iu] = (w > 0).astype(float)
    r = rng.random(len(iu[0]))
    flagged = keep & (r < flag_fraction)
    zero_w = keep & (r > 0.97)
    Cu = C[iu]; Cu[flagged] = 0.0; C[iu] = Cu
    Mu = M[iu]; Mu[zero_w & ~flagged] = 0.0; M[iu] = Mu
    for a in clique:
        for b in clique:
            if a < b: M[a, b] = rng.uniform(0.8, 1.0); C[a, b] = 1.0
    M = M + M.T; C = C + C.T
    np.fill_diagonal(M, 1.0); np.fill_diagonal(C, 1.0)
    return M, C


@pytest.mark.parametrize("n,density,iters", [(5, 0.8, None), (64, 0.3, None), (65, 0.2, None), (300, 0.05, None), (700, 0.02, None), (300, 0.05, 0)],
                         ids=["n5", "n64", "n65", "n300", "n700", "n300_fallback_layout"])
def test_dense_conversion_on_the_device(ctx, orc, n, density, iters):
    """roman_set_matrix_data builds the layouts on the device: the matrix it holds — read back through
    roman_get_dense_matrices — is the caller's (strict upper triangles, C flags, zero weights), and the solve is the oracle's.
    maxiniters = 0 selects the fallback layout (symmetric SELL-64, 32-bit labels) for a small problem."""
    M, C = _random_dense_problem(n, density, 1000 + n)
    P = _abi.RomanParams.default(); P.invariant = _abi.ROMAN_INV_EUCLIDEAN
    if iters is not None:
        P.maxiniters = iters
    ctx.set_matrix_data(P, M, C)
    Mg, Cg = ctx.dense_matrices()
    assert np.array_equal(Mg, M) and np.array_equal(Cg, C)
    ctx.solve(None)
    nodes, u, score, st = ctx.solution()
    ref = orc.solve(P, orc.matrix_from_dense(M, C))
    assert np.allclose(u, ref["u"], rtol=0, atol=1e-9)
    # the selected SET is the oracle's; the ORDER (descending u) may differ only between entries whose u agree to within the
    # iteration's own tolerance (the two solvers' u agree to 1e-9, not bitwise: tests/test_u0_stability.py)
    assert sorted(nodes.tolist()) == sorted(ref["nodes"].tolist())
    uo = ref["u"][ref["nodes"]]
    for r in np.nonzero(nodes != ref["nodes"])[0]:
        assert abs(uo[r] - uo[min(r + 1, len(uo) - 1)]) < 1e-6 or abs(uo[r] - uo[max(r - 1, 0)]) < 1e-6
    assert st.nnz_upper == int(np.count_nonzero(np.triu((M != 0) | (C != 0), 1)))


def _same_selection_as_oracle(nodes, ref):
    assert sorted(nodes.tolist()) == sorted(ref["nodes"].tolist())
    uo = ref["u"][ref["nodes"]]
    for r in np.nonzero(nodes != ref["nodes"])[0]:              # order swaps only between entries equal to within the iteration's tolerance
        assert abs(uo[r] - uo[min(r + 1, len(uo) - 1)]) < 1e-6 or abs(uo[r] - uo[max(r - 1, 0)]) < 1e-6


@pytest.mark.parametrize("n,kind", [(40, "large"), (300, "large"), (300, "negative"), (64, "tiny+huge"), (129, "nan_free_inf")],
                         ids=["n40_w50", "n300_w50", "n300_negative", "n64_1e-4_to_30", "n129_w7.5"])
def test_dense_weights_outside_the_unit_interval_take_the_plain_double_solver(ctx, orc, n, kind):
    """roman_set_matrix_data accepts ANY caller matrix (the mno_clipper loop, [REF roman/align/object_registration.py:57-86],
    only ever passes scores in [0, 1], a C caller may not).  The stream solver's fixed-point sums assume 0 <= v <= 1: a
    matrix with a weight outside (weights up to 50, negative weights, five decades of dynamic range) is detected on the device and
    solved by the plain-double solver of the fallback layout — u within 1e-9 of the oracle's, the same selected set."""
    M, C = _random_dense_problem(n, 0.3 if n < 100 else 0.06, 4000 + n)
    rng = np.random.default_rng(n)
    iu = np.triu_indices(n, 1)
    w = M[iu]
    if kind == "large":
        w = w * rng.uniform(1.0, 50.0, w.shape)
    elif kind == "negative":
        w = np.where(rng.random(w.shape) < 0.3, -w, w) * rng.uniform(0.5, 3.0, w.shape)
    elif kind == "tiny+huge":
        w = np.where(w != 0, 10.0 ** rng.uniform(-4, 1.5, w.shape), 0.0)
    else:
        w = w * 7.5
    M = np.zeros((n, n)); M[iu] = w; M = M + M.T; np.fill_diagonal(M, 1.0)
    P = _abi.RomanParams.default(); P.invariant = _abi.ROMAN_INV_EUCLIDEAN
    ctx.set_matrix_data(P, M, C)
    Mg, Cg = ctx.dense_matrices()
    assert np.array_equal(Mg, M) and np.array_equal(Cg, C)
    ctx.solve(None)
    nodes, u, score, st = ctx.solution()
    ref = orc.solve(P, orc.matrix_from_dense(M, C))
    assert np.all(np.isfinite(u))
    assert np.max(np.abs(u - ref["u"])) < 1e-9, (np.max(np.abs(u - ref["u"])), st.n_pass, ref["stats"].n_pass)
    assert abs(score - ref["stats"].score) <= 1e-9 * max(1.0, abs(ref["stats"].score))
    _same_selection_as_oracle(nodes, ref)
    # the same matrix scaled into [0, 1] takes the stream layout again (and is still the oracle's)
    Ms = M / np.abs(M[iu]).max() if kind != "negative" else np.abs(M) / np.abs(M[iu]).max()
    np.fill_diagonal(Ms, 1.0)
    ctx.set_matrix_data(P, Ms, C); ctx.solve(None)
    nodes2, u2, _, _ = ctx.solution()
    ref2 = orc.solve(P, orc.matrix_from_dense(Ms, C))
    assert np.allclose(u2, ref2["u"], rtol=0, atol=1e-9)
    _same_selection_as_oracle(nodes2, ref2)


@pytest.mark.parametrize("compact", [None, "0x01FF10"], ids=["compaction_default", "compaction_every_pass"])
def test_dense_problem_beyond_the_stream_layout_on_the_whole_device_solver(ctx, orc, compact, monkeypatch):
    """A caller's dense matrix of 3200 rows (the mno_clipper loop on a large association set) is beyond the stream layout: the
    fallback layout with 32-bit labels and C flags, the whole-device solver k_solve_wide — and its column compaction, which must
    keep flagged entries (C == 0, weight non-zero), consistent pairs of weight 0 (C == 1) and drop the inert padding: u within
    1e-9 of the oracle's, the same selected set, as the library compacts and with a compaction forced at almost every pass."""
    if compact is not None:
        monkeypatch.setenv("ROMAN_WIDE_COMPACT", compact)
    n = 3200
    M, C = _random_dense_problem(n, 0.5, 9100)
    P = _abi.RomanParams.default(); P.invariant = _abi.ROMAN_INV_EUCLIDEAN
    ctx.set_matrix_data(P, M, C)
    ctx.solve(None)
    nodes, u, score, st = ctx.solution()
    ref = orc.solve(P, orc.matrix_from_dense(M, C))
    assert st.nnz_upper == int(np.count_nonzero(np.triu((M != 0) | (C != 0), 1)))
    assert np.max(np.abs(u - ref["u"])) < 1e-9, (np.max(np.abs(u - ref["u"])), st.n_pass, ref["stats"].n_pass)
    assert abs(score - ref["stats"].score) <= 1e-9 * max(1.0, abs(ref["stats"].score))
    _same_selection_as_oracle(nodes, ref)


@pytest.mark.parametrize("u0_kind", ["ones", "1e-12..1", "one_dominant", "1e-12..1+c_flags"])
def test_fixed_point_sums_at_the_stream_solvers_limit(ctx, orc, u0_kind):
    """Worst case of the stream solver's exact sums (kernels.hip.h "Exact accumulation"): the largest live set it takes
    (L = STREAM_MAXL = 3072), EVERY pair stored with the largest weight the path admits (v = 1), start vectors whose elements
    span twelve decades.  A term is rint(v x 2^s) < 2^49, a row's sum of 3071 of them < 2^61: no accumulator overflows; u
    within 1e-9 of the oracle's plain-double sums, the same selected set."""
    n = 3072
    M = np.ones((n, n)); C = np.ones((n, n))
    rng = np.random.default_rng(5)
    if u0_kind.endswith("c_flags"):
        # 0.5 % of the pairs inconsistent (C = 0 with a stored weight: the flagged-entry path): the iteration now runs ~2000
        # passes over the whole matrix (21 homotopy steps, a clique of ~700) instead of stopping after three
        Cu = np.triu((rng.random((n, n)) >= 0.005).astype(float), 1); C = Cu + Cu.T; np.fill_diagonal(C, 1.0)
    if u0_kind == "ones":
        u0 = None
    elif u0_kind.startswith("1e-12..1"):
        u0 = 10.0 ** rng.uniform(-12, 0, n); u0[7] = 1.0; u0[11] = 1e-12
    else:
        u0 = np.full(n, 1e-12); u0[1234] = 1.0
    P = _abi.RomanParams.default(); P.invariant = _abi.ROMAN_INV_EUCLIDEAN
    ctx.set_matrix_data(P, M, C)
    ctx.solve(u0)
    nodes, u, score, st = ctx.solution()
    assert st.n_live == n and st.nnz_upper == n * (n - 1) // 2
    ref = orc.solve(P, orc.matrix_from_dense(M, C), u0)
    assert np.all(np.isfinite(u)) and np.allclose(u, ref["u"], rtol=0, atol=1e-9)
    # F enters the result only through omega = round(F).  Its value at the last homotopy step carries d * u'(1 - C)u: when the
    # two sides stop one step of d apart (whether an element with u ~ 1e-9 still counts as active is decided by the last bits
    # of a sum), F differs by ~1e-2 with u equal to 1e-9 — compare F itself only when the step counts agree.
    assert round(score) == round(ref["stats"].score) and len(nodes) == len(ref["nodes"])
    if st.outer_iters == ref["stats"].outer_iters:
        assert abs(score - ref["stats"].score) < 1e-6
    assert sorted(nodes.tolist()) == sorted(ref["nodes"].tolist())
    if u0_kind.endswith("c_flags"):
        # The PATH to the fixed point is not unique on this degenerate problem (every weight equal: hundreds of near-equivalent
        # cliques): the oracle's sequential double sums take 21 homotopy steps / 2150 passes, a dense NumPy statement of the
        # same algorithm (BLAS summation order) 25 / 2334, the device more.  What is compared is what the caller gets: u and the
        # selected set.  For the record, the same matrix on the device's plain-double solver (one weight nudged above 1 takes
        # it off the fixed-point path: roman_set_matrix_data's range check):
        M2 = M.copy(); M2[0, 1] = M2[1, 0] = 1.0 + 2.0 ** -40
        ctx.set_matrix_data(P, M2, C); ctx.solve(u0)
        n2, u2, s2, st2 = ctx.solution()
        print(f"outer iterations / passes: fixed-point solver {st.outer_iters} / {st.n_pass}, plain-double solver {st2.outer_iters} / {st2.n_pass}, "
              f"oracle {ref['stats'].outer_iters} / {ref['stats'].n_pass}; max |u - u_oracle| {np.max(np.abs(u - ref['u'])):.2e} / {np.max(np.abs(u2 - ref['u'])):.2e}")
        # (no assertion on the plain-double run: on this matrix of 4.7 M EQUAL weights it wanders — observed: 1000 homotopy
        #  steps, u 3e-8 off, a few more nodes — where the exact sums stay within 4e-10 of the oracle: the torture case shows
        #  how flat the objective is, and that the order-free sums are the steadier of the two device solvers)
        assert st.n_pass > 1000


def test_mno_clipper_leaves_the_registration_untouched(ctx, orc):
    """mno_clipper() solves on a separate plain-CLIPPER problem ([REF roman/align/object_registration.py:60]); the
    registration's own invariant parameters must not change: register() before == register() after."""
    reg = registration_for("roman", semantics_dim=16); reg.set_context(ctx)
    pr = synth.make_pair(16, 15, 16, 35, tilt_deg=1.0)
    before = reg.register(pr.map1, pr.map2)
    inv0 = reg._abi_params().invariant
    sols = reg.mno_clipper(pr.map1, pr.map2, num_solutions=2)
    assert reg._abi_params().invariant == inv0 == _abi.ROMAN_INV_ROMAN
    after = reg.register(pr.map1, pr.map2)
    assert np.array_equal(before, after) and np.array_equal(after, oracle_one(orc, reg, pr.map1, pr.map2)["assoc"])
    assert len(sols) == 2 and len(sols[0][0]) >= 3
    res = reg.register_and_align_batch([(pr.map1, pr.map2)])
    assert np.array_equal(res.assoc[0], before)


def test_batch_entry_rejects_malformed_association_lists(ctx):
    """roman_align_batch range-checks explicit lists like roman_score does (ROMAN_E_INVALID, no device reads)."""
    from roman_amd import RomanHipError
    reg = registration_for("clipper"); reg.set_context(ctx)
    pr = synth.make_pair(6, 5, 0, 36)
    b = rb.batch_from_pairs(reg, [(pr.map1, pr.map2), (pr.map1, pr.map2)])
    P = reg._abi_params()
    ok_list = np.array([[0, 0], [1, 1], [2, 2], [3, 3]], dtype=np.int32)
    good = ctx.align_batch(P, b.feats, b.off1, b.n1, b.off2, b.n2, assoc=np.concatenate([ok_list, ok_list]), assoc_off=[0, 4, 8])
    assert len(good.assoc) == 2
    for assoc, off in [(np.concatenate([ok_list, [[6, 0]]]), [0, 4, 5]),          # i == n1
                       (np.concatenate([ok_list, [[0, -1]]]), [0, 4, 5]),         # negative j
                       (np.concatenate([ok_list, ok_list]), [0, 5, 4]),           # decreasing offsets
                       (np.concatenate([ok_list, ok_list]), [1, 4, 8])]:          # does not start at 0
        with pytest.raises(RomanHipError, match="out of range|non-decreasing|must be 0|bad assoc_off"):
            ctx.align_batch(P, b.feats, b.off1, b.n1, b.off2, b.n2, assoc=assoc, assoc_off=off)
    # an EMPTY list for one problem means all-to-all for that problem (clipperpy's convention)
    mixed = ctx.align_batch(P, b.feats, b.off1, b.n1, b.off2, b.n2, assoc=ok_list, assoc_off=[0, 0, 4])
    alltoall = ctx.align_batch(P, b.feats, b.off1, b.n1, b.off2, b.n2)
    assert np.array_equal(mixed.assoc[0], alltoall.assoc[0]) and mixed.stats["n_assoc_in"].tolist() == [30, 4]


def test_full_size_properties_cfg2_cfg3(ctx):
    """BASELINE configs 2/3 shape (n=m=200, d=512, semanticgrav): properties that need no oracle."""
    reg = registration_for("semanticgrav", semantics_dim=512); reg.set_context(ctx)
    B = 12
    prs = [synth.make_pair(200, 200, 512, 3000 + k) for k in range(B)]
    res = reg.register_and_align_batch([(p.map1, p.map2) for p in prs])
    for b, pr in enumerate(prs):
        got = set(map(tuple, res.assoc[b].tolist())); truth = set(map(tuple, pr.inliers.tolist()))
        assert res.status[b] == 0
        assert len(got & truth) >= 0.9 * len(truth) and len(got - truth) <= 2          # planted clique
        assert len(set(i for i, _ in got)) == len(got) == len(set(j for _, j in got))    # one-to-one
        assert np.linalg.norm(res.T[b][:3, :3] - pr.T_gt[:3, :3]) < 0.02 and np.linalg.norm(res.T[b][:3, 3] - pr.T_gt[:3, 3]) < 0.1
        R = res.T[b][:3, :3]
        assert abs(np.linalg.det(R) - 1) < 1e-12 and np.allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert res.stats["n_assoc_in"][b] == 40000 and res.stats["n_live"][b] < 8000
    # idempotence / determinism: a second run returns bit-identical results
    res2 = reg.register_and_align_batch([(p.map1, p.map2) for p in prs])
    for b in range(B):
        assert np.array_equal(res.assoc[b], res2.assoc[b]) and np.array_equal(res.T[b], res2.T[b])
    # batch composition does not matter: problem 0 alone == problem 0 in the batch
    solo = reg.register_and_align_batch([(prs[0].map1, prs[0].map2)])
    assert np.array_equal(solo.assoc[0], res.assoc[0]) and np.array_equal(solo.T[0], res.T[0])
    # permuting the objects of map 2 permutes the associations and leaves the pose unchanged
    perm = np.random.default_rng(1).permutation(200)
    m2p = [prs[1].map2[k] for k in perm]
    rp = reg.register_and_align_batch([(prs[1].map1, m2p)])
    assert set((int(i), int(perm[j])) for i, j in rp.assoc[0]) == set(map(tuple, res.assoc[1].tolist()))
    assert np.linalg.norm(rp.T[0] - res.T[1]) < 1e-9


def test_rigid_motion_of_map2_composes_with_the_pose(ctx):
    """Full size (n=m=200, d=512): a gravity-preserving rigid motion G of map 2 (yaw + translation) leaves every
    intra-map distance and height difference unchanged up to rounding, so the inlier set stays (almost) the same
    and the estimated pose composes: T(map1 <- G map2) = T(map1 <- map2) G^-1."""
    import copy
    reg = registration_for("semanticgrav", semantics_dim=512); reg.set_context(ctx)
    pr = synth.make_pair(200, 200, 512, 3001)
    G = synth.yaw_transform(1.1, [4.0, -7.0, 0.8])
    moved = copy.deepcopy(pr.map2)
    for o in moved:
        o.centroid = (G[:3, :3] @ o.centroid.reshape(3, 1) + G[:3, 3:4])
    res = reg.register_and_align_batch([(pr.map1, pr.map2), (pr.map1, moved)])
    assert res.status[0] == 0 and res.status[1] == 0
    a0, a1 = set(map(tuple, res.assoc[0].tolist())), set(map(tuple, res.assoc[1].tolist()))
    assert len(a0 ^ a1) <= 2
    assert np.linalg.norm(res.T[1] @ G - res.T[0]) < 1e-6 + 0.05 * (len(a0 ^ a1) > 0)


def test_full_size_properties_cfg4_submap_grid(ctx):
    """BASELINE config 4 shape (cross pairs of two robots' submaps, n=200, d=512, one shared feature pool):
    poses agree with the generator's ground truth and compose consistently, T_ij = T_ik T_kj^-1... checked as
    T(a,c) ~ T(a,b) T(b,c) through a third submap; every submap is packed once."""
    reg = registration_for("semanticgrav", semantics_dim=512); reg.set_context(ctx)
    S = 6
    subs, poses = synth.make_submap_grid(2 * S, n=200, d=512, seed0=4000)
    batch = rb.batch_from_submap_grid(reg, subs[:S], subs[S:])
    assert len(batch) == S * S and batch.feats.shape[0] == 2 * S * 200
    res = rb.run_batch(reg, batch)
    T = {}
    for b, (i, j) in enumerate(batch.pair_index):
        assert res.status[b] == 0
        a = res.assoc[b]
        assert len(a) >= 60 and len(set(a[:, 0].tolist())) == len(a) == len(set(a[:, 1].tolist()))
        T_gt = np.linalg.inv(poses[i]) @ poses[S + j]                     # map 2 (robot-1 submap j) -> map 1 (robot-0 submap i)
        assert np.linalg.norm(res.T[b][:3, :3] - T_gt[:3, :3]) < 0.03 and np.linalg.norm(res.T[b][:3, 3] - T_gt[:3, 3]) < 0.15
        T[(int(i), int(j))] = res.T[b]
    # composition through another pair of submaps: T(i,j) T(i',j)^-1 T(i',j') ~ T(i,j')
    for (i, j, i2, j2) in [(0, 0, 1, 1), (2, 3, 4, 5), (5, 1, 0, 4)]:
        lhs = T[(i, j)] @ np.linalg.inv(T[(i2, j)]) @ T[(i2, j2)]
        assert np.linalg.norm(lhs - T[(i, j2)]) < 0.3
    # the same pairs as independent problems (each submap packed per pair) give bit-identical results
    solo = reg.register_and_align_batch([(subs[0], subs[S + 1]), (subs[3], subs[S + 2])])
    assert np.array_equal(solo.assoc[0], res.assoc[0 * S + 1]) and np.array_equal(solo.T[0], res.T[0 * S + 1])
    assert np.array_equal(solo.assoc[1], res.assoc[3 * S + 2]) and np.array_equal(solo.T[1], res.T[3 * S + 2])


def test_large_live_set_uses_global_u_path(ctx, orc):
    """L = 150*150 = 22500 live associations exceed the LDS-resident vectors: same results."""
    reg = registration_for("clipper"); reg.set_context(ctx)
    pr = synth.make_pair(150, 150, 0, 61)
    res = reg.register_and_align_batch([(pr.map1, pr.map2)])
    o = oracle_one(orc, reg, pr.map1, pr.map2)
    assert np.array_equal(res.assoc[0], o["assoc"])
    assert res.stats["n_pass"][0] == o["stats"].n_pass and res.stats["nnz_upper"][0] == o["stats"].nnz_upper


class _Hip:
    """Minimal device-memory helper on the HIP runtime the library already loaded (no torch in this process:
    torch bundles its own HIP runtime, and two runtimes in one process do not share the device)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.bufs = []

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.lib.hipMalloc(self.C.byref(p), max(int(nbytes), 8)) == 0
        self.bufs.append(p)
        return p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        assert self.lib.hipMemcpy(p, arr.ctypes.data, arr.nbytes, 1) == 0      # hipMemcpyHostToDevice
        return p

    def download(self, ptr, shape, dtype):
        out = np.zeros(shape, dtype=dtype)
        assert self.lib.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0    # hipMemcpyDeviceToHost
        return out

    def free_all(self):
        for p in self.bufs:
            self.lib.hipFree(p)
        self.bufs = []


def test_two_batches_in_flight_equal_sequential_calls(ctx):
    """roman_ctx_set_pipeline(2): consecutive device-pointer batch calls alternate between two workspaces on
    internal streams; after roman_ctx_sync every batch's results equal the ones of plain sequential calls."""
    from roman_amd.runtime import stats_dtype
    reg = registration_for("semanticgrav", semantics_dim=32); reg.set_context(ctx)
    P = reg._abi_params(); F = P.feature_dim()
    hip = _Hip()
    batches = []
    for g in range(4):                                          # four different batches (different sizes and seeds)
        pairs = [synth.make_pair(40 + 5 * g, 35 + 3 * k, 32, 700 + 10 * g + k) for k in range(3 + g)]
        batches.append(rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs]))
    ref = [rb.run_batch(reg, b) for b in batches]               # sequential host-pointer calls
    outs = []
    for b in batches:
        B, kmax = len(b), b.kmax()
        outs.append(dict(feats=hip.upload(b.feats), kmax=kmax, assoc=hip.alloc(B * kmax * 2 * 4), n=hip.alloc(B * 4),
                         T=hip.alloc(B * 16 * 8), status=hip.alloc(B * 4), stats=hip.alloc(B * _abi.STATS_NBYTES)))
    ctx.set_pipeline(2)
    try:
        for b, o in zip(batches, outs):                         # issue all four without waiting in between
            ctx.align_batch_dev(P, o["feats"], F, b.off1, b.n1, b.off2, b.n2, o["kmax"], o["assoc"], o["n"], o["T"], o["status"], o["stats"])
        ctx.sync()
    finally:
        ctx.set_pipeline(1)
    try:
        for b, o, r in zip(batches, outs, ref):
            B, kmax = len(b), o["kmax"]
            n = hip.download(o["n"], (B,), np.int32); a = hip.download(o["assoc"], (B, kmax, 2), np.int32)
            T = hip.download(o["T"], (B, 16), np.float64); st = hip.download(o["status"], (B,), np.int32)
            stt = hip.download(o["stats"], (B,), stats_dtype())
            for k in range(B):
                assert np.array_equal(a[k, :n[k]], r.assoc[k]) and st[k] == r.status[k]
                assert stt["n_pass"][k] == r.stats["n_pass"][k] and stt["nnz_upper"][k] == r.stats["nnz_upper"][k]
                if st[k] == 0:
                    assert np.array_equal(T[k].reshape(4, 4), r.T[k])       # same kernels, same order: bit-identical pose
    finally:
        hip.free_all()


def test_align_sharded_device_path_single_rank():
    """align_sharded's HIP path (device-resident records, cost-balanced deal) on one rank equals run_batch.  torch is
    imported FIRST in a fresh process: its bundled HIP runtime must be the one libroman_hip binds to."""
    import subprocess, sys, textwrap
    from conftest import ROOT
    code = textwrap.dedent("""
        import torch, sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
        from conftest import registration_for
        from roman_amd import synth
        from roman_amd.align import batch as rb
        from roman_amd.align.distributed import align_sharded
        reg = registration_for('semanticgrav', semantics_dim=16)
        subs, _ = synth.make_submap_grid(6, n=30, d=16, seed0=71)
        subs[4] = subs[4][:17]
        batch = rb.batch_from_submap_grid(reg, subs[:3], subs[3:])
        ref = rb.run_batch(reg, batch)
        assoc, T, status = align_sharded(reg, batch)
        assert len(assoc) == len(batch) == 9
        for b in range(9):
            assert np.array_equal(assoc[b], ref.assoc[b]), b
            assert status[b] == ref.status[b]
            assert np.array_equal(np.nan_to_num(T[b]), np.nan_to_num(ref.T[b]))
        print('SHARDED_OK')
    """ % (ROOT, ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "SHARDED_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_live_set_larger_than_the_history_expects_is_retried_with_the_fallback_kernels(orc):
    """Once a context has seen batches of a parameter block it launches the fallback layout's kernels only when a live set
    beyond the stream layout is expected.  A problem that then turns out larger (here: every descriptor identical, so
    all 60 x 60 associations pass the cosine gate where earlier pairs kept 5 %) is skipped (ROMAN_ST_WORKSPACE on the
    device-pointer entry) and run again by the host-pointer entry with the corrected history: same result as the oracle."""
    from roman_amd.runtime import Context
    c = Context(0)
    try:
        reg = registration_for("semanticgrav", semantics_dim=16); reg.set_context(c)
        ordinary = [synth.make_pair(60, 60, 16, 5000 + k) for k in range(3)]
        for _ in range(2):                                   # the second call finds the totals of the first
            r0 = reg.register_and_align_batch([(p.map1, p.map2) for p in ordinary])
        assert (r0.status == 0).all() and r0.stats["n_live"].max() < 1000
        odd = synth.make_pair(60, 60, 16, 5100)
        e = np.zeros(16); e[0] = 1.0
        for o in list(odd.map1) + list(odd.map2):
            o.semantic_descriptor = e.copy()
        res = reg.register_and_align_batch([(ordinary[0].map1, ordinary[0].map2), (odd.map1, odd.map2)])
        assert res.stats["n_live"].tolist()[1] == 3600 and (res.status == 0).all()
        assert np.array_equal(res.assoc[0], r0.assoc[0])
        o = oracle_one(orc, reg, odd.map1, odd.map2)
        assert np.array_equal(res.assoc[1], o["assoc"]) and res.stats["nnz_upper"][1] == o["stats"].nnz_upper
    finally:
        c.close()


def test_zero_inner_iterations_take_the_fallback_layout_and_match_the_oracle(ctx, orc):
    """maxiniters < 1 (or maxlsiters < 1) makes every problem kind 1 (only the fallback solver implements the degenerate
    loop): its fill kernel must run for ordinary small problems too (it was launched only for live sets beyond
    the stream layout, so the solver read unwritten column indices).  Both batch entries and the stepwise pair."""
    reg = registration_for("semanticgrav", semantics_dim=24); reg.set_context(ctx)
    P = reg._abi_params()
    old = (P.maxiniters, P.maxoliters)
    try:
        P.maxiniters = 0; P.maxoliters = 7
        pairs = [synth.make_pair(40, 35 + k, 24, 820 + k, tilt_deg=1.0) for k in range(3)]
        res = reg.register_and_align_batch([(p.map1, p.map2) for p in pairs])
        for b, p in enumerate(pairs):
            o = oracle_one(orc, reg, p.map1, p.map2)
            assert np.array_equal(res.assoc[b], o["assoc"]), b
            assert res.stats["nnz_upper"][b] == o["stats"].nnz_upper and res.stats["outer_iters"][b] == o["stats"].outer_iters == 7
            assert res.stats["n_pass"][b] == o["stats"].n_pass == 2
        D1, D2 = reg.pack(pairs[0].map1), reg.pack(pairs[0].map2)
        ctx.score(P, D1, D2, None); ctx.solve(None)
        assert np.array_equal(ctx.selected_associations(), res.assoc[0])
    finally:
        P.maxiniters, P.maxoliters = old


def test_sizing_history_of_one_parameter_block_is_not_applied_to_another(orc):
    """The totals a batch leaves behind size the NEXT batch — of the same parameter block only.  A block that keeps far more
    associations, issued right after (its predecessor's totals still pending), must start from the first-call heuristics
    and the fallback kernels, not from the other block's ratios: no ROMAN_ST_WORKSPACE on its first call."""
    from roman_amd.runtime import Context
    c = Context(0)
    hip = _Hip()
    try:
        regA = registration_for("semanticgrav", semantics_dim=16, cosine_min=0.9, cosine_max=0.95); regA.set_context(c)
        regB = registration_for("semanticgrav", semantics_dim=16, cosine_min=-0.9, cosine_max=0.7); regB.set_context(c)
        pairs = [synth.make_pair(60, 60, 16, 5200 + k) for k in range(4)]
        batch = rb.batch_from_pairs(regA, [(p.map1, p.map2) for p in pairs])
        B, kmax, F = len(batch), batch.kmax(), batch.feats.shape[1]
        o = dict(feats=hip.upload(batch.feats), assoc=hip.alloc(B * kmax * 2 * 4), n=hip.alloc(B * 4), T=hip.alloc(B * 16 * 8),
                 status=hip.alloc(B * 4), stats=hip.alloc(B * _abi.STATS_NBYTES))
        for _ in range(2):
            c.align_batch_dev(regA._abi_params(), o["feats"], F, batch.off1, batch.n1, batch.off2, batch.n2, kmax, o["assoc"], o["n"], o["T"], o["status"], o["stats"])
        # no sync: block A's totals are still pending when block B is sized
        c.align_batch_dev(regB._abi_params(), o["feats"], F, batch.off1, batch.n1, batch.off2, batch.n2, kmax, o["assoc"], o["n"], o["T"], o["status"], o["stats"])
        c.sync()
        st = hip.download(o["status"], (B,), np.int32); n = hip.download(o["n"], (B,), np.int32); a = hip.download(o["assoc"], (B, kmax, 2), np.int32)
        assert not (st & _abi.ROMAN_ST_WORKSPACE).any(), st
        assert c.skipped() == 0
        for b, p in enumerate(pairs):
            ob = oracle_one(orc, regB, p.map1, p.map2)
            assert np.array_equal(a[b, :n[b]], ob["assoc"]), b
    finally:
        hip.free_all(); c.close()


def test_prefilter_on_the_device_equals_the_host_prefilter(ctx, orc):
    """DistRegWithPruning(prune_on_device=True) — the cosine / shape-ratio prefilter of
    [REF roman/align/dist_reg_with_pruning.py:71-97] evaluated by the single-score kernels (raw descriptor products on the f64
    matrix core, ratio gates), the pruned list never on the host — against (a) the same plugin with the NumPy prefilter and an
    explicit list and (b) the oracle, in ONE ragged batch that includes a pair where nothing survives (all-to-all then)."""
    from roman_amd.align.dist_reg_with_pruning import DistRegWithPruning
    specs = [({"cosine_min": 0.5, "epsilon_shape": 0.1}, 50, 50, 64, 1012), ({"cosine_min": 0.6}, 40, 30, 64, 1013), ({"cosine_min": 0.9999}, 24, 24, 64, 1015),
             ({"cosine_min": 0.5}, 60, 60, 64, 15)]
    for kw, n, m, d, seed in specs:
        host = registration_for("clipper+prune", **kw); host.set_context(ctx)
        dev = DistRegWithPruning(host.sigma, host.epsilon, host.mindist, host.shape_epsilon, host.cos_min, dim=3, use_gravity=True, prune_on_device=True)
        dev.set_context(ctx)
        pr = synth.make_pair(n, m, d, seed)
        want = host.register_and_align_batch([(pr.map1, pr.map2)])
        got = dev.register_and_align_batch([(pr.map1, pr.map2)])
        assert np.array_equal(got.assoc[0], want.assoc[0]) and got.status[0] == want.status[0]
        assert got.stats["n_live"][0] == want.stats["n_live"][0] and got.stats["nnz_upper"][0] == want.stats["nnz_upper"][0]
        assert got.stats["n_assoc_in"][0] == n * m                     # the device scored the all-to-all list itself
        if got.status[0] == 0:
            assert np.linalg.norm(got.T[0] - want.T[0]) < POSE_TOL
        o = orc.register(dev._abi_params(), dev.pack(pr.map1), dev.pack(pr.map2))
        assert np.array_equal(got.assoc[0], o["assoc"]) and got.stats["n_pass"][0] == o["stats"].n_pass
    # several pairs in one call (one parameter block: the descriptor length is part of it)
    kw = {"cosine_min": 0.55, "epsilon_shape": 0.05}
    host = registration_for("clipper+prune", **kw); host.set_context(ctx)
    dev = DistRegWithPruning(host.sigma, host.epsilon, host.mindist, host.shape_epsilon, host.cos_min, dim=3, use_gravity=True, prune_on_device=True); dev.set_context(ctx)
    prs = [synth.make_pair(30 + 5 * k, 28 + 3 * k, 32, 1700 + k) for k in range(5)]
    want = host.register_and_align_batch([(p.map1, p.map2) for p in prs]); got = dev.register_and_align_batch([(p.map1, p.map2) for p in prs])
    for b in range(5):
        assert np.array_equal(got.assoc[b], want.assoc[b]), b


def test_whole_device_solver_that_gives_up_says_so_for_every_problem(orc):
    """A bounded wait of the whole-device solver that expires (never expected; provoked here with a zero budget through the
    test hook ROMAN_WIDE_SPIN_MS: the first unsuccessful poll of any barrier wait gives up) must not leave stale output records behind: every fallback problem the launch did not
    finish reports ROMAN_ST_INTERNAL with no associations and a NaN pose (k_skipped pre-writes that record, the solver
    overwrites it on completion), the host-pointer entry returns ROMAN_E_INTERNAL, and a context with the normal budget
    solves the same batch correctly."""
    import os
    from roman_amd import RomanHipError
    from roman_amd.runtime import Context
    reg = registration_for("gravity")
    pairs = [synth.make_pair(70, 70, 0, 950 + k, tilt_deg=1.0) for k in range(3)]          # L = 4900 each: the fallback layout
    batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    os.environ["ROMAN_WIDE_SPIN_MS"] = "0"
    try:
        c1 = Context(0)
    finally:
        del os.environ["ROMAN_WIDE_SPIN_MS"]
    try:
        reg.set_context(c1)
        P = reg._abi_params()
        B = len(batch)
        with pytest.raises(RomanHipError, match="ROMAN_ST_INTERNAL") as ei:
            rb.run_batch(reg, batch, ctx=c1)
        res = ei.value.result                                     # the host-pointer entry copies the outputs before it reports ROMAN_E_INTERNAL
        st = res.status
        # (a problem can get through before a wait happens to exceed the budget: every record is either a finished result or
        #  the statement that there is none — never what the buffers held before)
        assert (st & _abi.ROMAN_ST_INTERNAL).any(), st
        for b in range(B):
            if st[b] & _abi.ROMAN_ST_INTERNAL:
                assert st[b] == _abi.ROMAN_ST_INTERNAL and len(res.assoc[b]) == 0 and np.isnan(res.T[b]).all()
            else:
                o = orc.register(P, reg.pack(pairs[b].map1), reg.pack(pairs[b].map2))
                assert st[b] == 0 and np.array_equal(res.assoc[b], o["assoc"])
        from roman_amd.align.distributed import check_records
        with pytest.raises(RomanHipError, match="ROMAN_ST_INTERNAL"):
            check_records(st)
    finally:
        c1.close()
    c2 = Context(0)
    try:
        reg.set_context(c2)
        res = rb.run_batch(reg, batch, ctx=c2)
        for b, p in enumerate(pairs):
            o = orc.register(reg._abi_params(), reg.pack(p.map1), reg.pack(p.map2))
            assert res.status[b] == 0 and np.array_equal(res.assoc[b], o["assoc"])
    finally:
        c2.close()


def test_problem_beyond_the_fused_small_path_after_a_history_of_small_ones(orc):
    """Once every problem of a parameter block has been finished by k_small, the general kernels are not launched (`small_only`).
    A batch that then contains a larger problem (more than 128 live associations) must still come out right: the problem is
    skipped on the first attempt (ROMAN_ST_WORKSPACE), the history learns that the general kernels are needed, and the
    host-pointer entry runs the batch again — every result equals the oracle's, the small ones included."""
    from roman_amd.runtime import Context
    c = Context(0)
    try:
        reg = registration_for("roman", semantics_dim=16); reg.set_context(c)
        small = [synth.make_pair(14 + k % 5, 12 + k % 7, 16, 8000 + k, tilt_deg=1.0) for k in range(24)]
        for _ in range(3):                                    # the history now says: k_small finishes everything
            r0 = reg.register_and_align_batch([(p.map1, p.map2) for p in small])
        assert (r0.stats["n_live"] <= 128).all() and ((r0.status == 0) | (r0.status == _abi.ROMAN_ST_INSUFFICIENT)).all()
        for b, p in enumerate(small):
            assert np.array_equal(r0.assoc[b], oracle_one(orc, reg, p.map1, p.map2)["assoc"])
        big = synth.make_pair(90, 90, 16, 8100, tilt_deg=1.0, desc_noise=0.2)
        mixed = small[:5] + [big] + small[5:9]
        res = reg.register_and_align_batch([(p.map1, p.map2) for p in mixed])
        assert res.stats["n_live"][5] > 128
        for b, p in enumerate(mixed):
            o = oracle_one(orc, reg, p.map1, p.map2)
            assert np.array_equal(res.assoc[b], o["assoc"]) and not (res.status[b] & _abi.ROMAN_ST_WORKSPACE), b
        res2 = reg.register_and_align_batch([(p.map1, p.map2) for p in small])          # and small batches keep working afterwards
        for b in range(len(small)):
            assert np.array_equal(res2.assoc[b], r0.assoc[b])
    finally:
        c.close()


@pytest.mark.parametrize("capnnz", [None, "3000"], ids=["sized_by_history", "first_calls_overflow"])
def test_large_host_batch_goes_out_in_calls_in_flight_and_equals_one_call(orc, capnnz, monkeypatch):
    """roman_align_batch (host pointers) issues a batch of more than `chunk` problems as calls of `chunk` problems with `depth`
    of them in flight (roman_ctx_set_host_batching: the pipelined loop of a device-pointer caller, done by the library for the
    caller of [REF roman/align/submap_align.py:93-200] who hands over every surviving pair at once): every result — associations
    incl. order, pose bits, status, statistics — equals the one-call result and the oracle's, with explicit association lists
    and start vectors sliced per call, on a fresh context (no sizing history: the first call is waited for) and when the
    first calls' workspace is too small (ROMAN_TEST_CAPNNZ: skipped problems are issued again, those only)."""
    from roman_amd.runtime import Context
    if capnnz is not None:
        monkeypatch.setenv("ROMAN_TEST_CAPNNZ", capnnz)
    reg = registration_for("semanticgrav", semantics_dim=16)
    pairs = [synth.make_pair(22 + (7 * k) % 19, 20 + (5 * k) % 23, 16, 9300 + k, tilt_deg=1.0) for k in range(150)]
    batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    rng = np.random.default_rng(3)
    u0 = rng.uniform(0.2, 1.0, int(np.sum(batch.n1.astype(np.int64) * batch.n2)))
    results = {}
    for name, (chunk, depth) in {"one_call": (100000, 1), "chunks_32x3": (32, 3), "chunks_50x2": (50, 2)}.items():
        c = Context(0)                                         # fresh: no sizing history, first-call heuristics
        try:
            reg.set_context(c)
            c.set_host_batching(chunk, depth)
            results[name] = (rb.run_batch(reg, batch), rb.run_batch(reg, batch, u0=u0))
            assert c.skipped() >= 0
        finally:
            c.close()
    ref, ref_u0 = results["one_call"]
    for name in ("chunks_32x3", "chunks_50x2"):
        for got, want in zip(results[name], (ref, ref_u0)):
            assert np.array_equal(got.status, want.status) and not (got.status & _abi.ROMAN_ST_WORKSPACE).any(), name
            for b in range(len(batch)):
                assert np.array_equal(got.assoc[b], want.assoc[b]), (name, b)
            assert np.array_equal(got.T, want.T, equal_nan=True)
            for f in ("n_live", "nnz_upper", "n_pass", "score"):
                assert np.array_equal(got.stats[f], want.stats[f]), (name, f)
    for b in range(0, len(pairs), 7):
        o = oracle_one(orc, reg, pairs[b].map1, pairs[b].map2)
        assert np.array_equal(ref.assoc[b], o["assoc"]), b
    # explicit association lists (the pruning plugin): the lists are sliced per call by their row offsets
    prune = registration_for("clipper+prune", cosine_min=0.4)
    pb = rb.batch_from_pairs(prune, [(p.map1, p.map2) for p in pairs[:90]])
    assert pb.assoc is not None
    out = []
    for chunk, depth in ((100000, 1), (16, 3)):
        c = Context(0)
        try:
            prune.set_context(c); c.set_host_batching(chunk, depth)
            out.append(rb.run_batch(prune, pb))
        finally:
            c.close()
    for b in range(len(pb)):
        assert np.array_equal(out[0].assoc[b], out[1].assoc[b]), b
    assert np.array_equal(out[0].T, out[1].T, equal_nan=True) and np.array_equal(out[0].status, out[1].status)


def test_align_resident_and_align_stream_equal_run_batch():
    """roman_amd.align.pipeline on the GPU (torch for device memory: its own process, torch imported first): (1) align_resident —
    a 150-pair batch over a resident pool as calls of 32 pairs, three in flight, on a fresh context with the first calls'
    workspace too small (ROMAN_TEST_CAPNNZ: skipped problems are issued again) — equals run_batch problem by problem, with
    explicit association lists too; (2) AlignStream — ten calls of distinct batches, three in flight, every call's output set
    collected by `on_collect` before the set is rewritten — equals the same calls one at a time."""
    import subprocess, sys, textwrap
    from conftest import ROOT
    code = textwrap.dedent("""
        import os, sys
        import numpy as np
        import torch
        sys.path.insert(0, %r)
        from roman_amd import _abi, synth
        from roman_amd.align import SubmapAlignParams, batch as rb
        from roman_amd.align.pipeline import AlignStream, align_resident
        from roman_amd.runtime import Context
        dev = torch.device("cuda", 0)
        stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
        reg = SubmapAlignParams(method="semanticgrav", semantics_dim=16).get_object_registration()
        pairs = [synth.make_pair(22 + (7 * k) %% 19, 20 + (5 * k) %% 23, 16, 9300 + k, tilt_deg=1.0) for k in range(150)]
        batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
        c0 = Context(0, stream=stream.cuda_stream); reg.set_context(c0)
        want = rb.run_batch(reg, batch)
        pool = torch.from_numpy(batch.feats).to(dev)
        os.environ["ROMAN_TEST_CAPNNZ"] = "3000"
        c1 = Context(0, stream=stream.cuda_stream); reg.set_context(c1)
        got = align_resident(reg, batch, pool, chunk=32, in_flight=3, ctx=c1)
        assert c1.skipped() > 0, "the test hook did not make the first calls overflow"
        del os.environ["ROMAN_TEST_CAPNNZ"]
        for b in range(len(batch)):
            assert np.array_equal(got.assoc[b], want.assoc[b]), b
        assert np.array_equal(got.T, want.T, equal_nan=True) and np.array_equal(got.status, want.status)
        assert np.array_equal(got.stats["n_pass"], want.stats["n_pass"])
        prune = SubmapAlignParams(method="clipper+prune", cosine_min=0.4).get_object_registration(); prune.set_context(c1)
        pb = rb.batch_from_pairs(prune, [(p.map1, p.map2) for p in pairs[:70]])
        wp = rb.run_batch(prune, pb)
        gp = align_resident(prune, pb, torch.from_numpy(pb.feats).to(dev), chunk=16, in_flight=3, ctx=c1)
        for b in range(len(pb)):
            assert np.array_equal(gp.assoc[b], wp.assoc[b]), b
        # the library entry itself (roman_align_batch_resident) with start vectors in device memory: equals the host-pointer entry
        sub = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs[:20]])
        u0 = np.concatenate([np.random.default_rng(77 + b).uniform(0.05, 1.0, int(sub.n1[b]) * int(sub.n2[b])) for b in range(20)])
        reg.set_context(c1)
        wu = rb.run_batch(reg, sub, u0=u0)
        spool = torch.from_numpy(sub.feats).to(dev); du0 = torch.from_numpy(u0).to(dev); torch.cuda.synchronize(dev)
        gu = c1.align_batch_resident(reg._abi_params(), spool.data_ptr(), sub.feats.shape[1], sub.off1, sub.n1, sub.off2, sub.n2, sub.kmax(), u0_ptr=du0.data_ptr())
        for b in range(20):
            assert np.array_equal(gu.assoc[b], wu.assoc[b]), b
        assert np.array_equal(gu.T, wu.T, equal_nan=True) and np.array_equal(gu.stats["n_pass"], wu.stats["n_pass"]) and np.array_equal(gu.stats["score"], wu.stats["score"])
        assert c1.has_history(reg._abi_params(), sub.feats.shape[1])             # (roman_ctx_has_history: this block has reported its needs)
        other = SubmapAlignParams(method="semanticgrav", semantics_dim=16, epsilon=0.55).get_object_registration()
        assert not c1.has_history(other._abi_params(), sub.feats.shape[1])
        # AlignStream: ten distinct calls of 15 problems, three in flight, collected before their set is rewritten
        reg.set_context(c0)
        kmax = batch.kmax(); F = batch.feats.shape[1]
        seen = {}
        def collect(k, tag):
            O = S.sets[k]
            seen[tag] = (O.n.clone(), O.assoc.clone(), O.T.clone(), O.status.clone())
        S = AlignStream(reg, c0, dev, rows=15, kmax=kmax, in_flight=3, stream=stream, on_collect=collect)
        for ci in range(10):
            sl = slice(15 * ci, 15 * ci + 15)
            S.submit(pool.data_ptr(), F, batch.off1[sl], batch.n1[sl], batch.off2[sl], batch.n2[sl], tag=ci)
        S.drain(); torch.cuda.synchronize(dev); S.close()
        assert sorted(seen) == list(range(10))
        for ci in range(10):
            n, a, T, st = (x.cpu().numpy() for x in seen[ci])
            for r in range(15):
                b = 15 * ci + r
                assert np.array_equal(a[r, :n[r]], want.assoc[b]), (ci, r)
                assert st[r] == want.status[b]
                if st[r] == 0:
                    assert np.array_equal(T[r].reshape(4, 4), want.T[b])
        c0.close(); c1.close()
        print("PIPELINE_OK")
    """ % (ROOT,))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PIPELINE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_team_that_cannot_hold_its_problem_is_solved_again_by_the_whole_device(orc, monkeypatch):
    """the host sizes a team's buffers from the XCD count the runtime reports, the kernel forms teams from the XCD
    every workgroup really runs on.  ROMAN_NUM_XCC=32 makes the host believe in XCDs of 8 compute units: the real teams (32
    workgroups) exceed their share of the partials buffer, the kernel leaves those problems ROMAN_ST_INTERNAL — and
    roman_align_batch runs them again with the whole device per problem: every result is the oracle's, no error surfaces.
    With teams forced off from the start (roman_ctx_set_wide_teams(0)) the same results come out in one attempt."""
    from roman_amd.runtime import Context
    monkeypatch.setenv("ROMAN_NUM_XCC", "32")
    c = Context(0)
    monkeypatch.delenv("ROMAN_NUM_XCC")
    c2 = Context(0)
    try:
        reg = registration_for("gravity")
        pairs = [synth.make_pair(60, 58 + k, 0, 9500 + k, tilt_deg=1.0) for k in range(3)]      # L = 3480 .. 3600: beyond the stream layout
        reg.set_context(c)
        res = reg.register_and_align_batch([(p.map1, p.map2) for p in pairs])
        c2.set_wide_teams(0); reg.set_context(c2)
        res0 = reg.register_and_align_batch([(p.map1, p.map2) for p in pairs])
        assert (res.stats["n_live"] > 3072).all()
        for b, p in enumerate(pairs):
            o = oracle_one(orc, reg, p.map1, p.map2)
            assert np.array_equal(res.assoc[b], o["assoc"]) and np.array_equal(res0.assoc[b], o["assoc"]), b
            assert not (res.status[b] & _abi.ROMAN_ST_INTERNAL) and res.stats["n_pass"][b] == o["stats"].n_pass == res0.stats["n_pass"][b]
    finally:
        c.close(); c2.close()


def test_join_on_the_null_stream_is_refused(ctx):
    """roman_ctx_join_on reads a NULL stream handle as "the context's own stream", and 0 is also the handle of the
    legacy default stream: Context.join(stream=0) would make the wrong stream wait.  It raises instead."""
    with pytest.raises(ValueError):
        ctx.join(skip_latest=False, stream=0)
    ctx.join(skip_latest=False)                                   # (the context's own stream: fine)


@pytest.mark.parametrize("cap,B", [("3", 24), ("1", 24), ("7", 100)], ids=["budget_3_passes", "budget_1_pass", "more_suspended_than_slots"])
def test_bounded_solver_launches_resume_suspended_problems_bit_for_bit(ctx, cap, B, monkeypatch):
    """The stream solver with a pass budget (SolveCont, kernels.hip.h: a problem still iterating after `cap` passes of a launch is
    suspended — iterate, fused product, the vector about to be multiplied and sixteen scalars to a slot — and a second launch resumes
    the suspended ones) against the same batch without one: EVERY output identical bit for bit — associations incl. order, poses,
    scores, d, pass / trial / iteration counts.  A budget of 1 suspends in front of the second pass (the rescale pass done), 3 inside
    the first line searches; 100 problems with a budget of 7 are more suspended problems than the launch has slots (64 for a batch of
    100): the rest runs on where it is.  The pairs of the reference's loop are independent ([REF roman/align/submap_align.py:93-200]):
    when and where a problem's iteration continues must not show."""
    reg = registration_for("semanticgrav", semantics_dim=48); reg.set_context(ctx)
    rng = np.random.default_rng(42)
    pairs = []
    for k in range(B):
        n, m = int(rng.integers(55, 90)), int(rng.integers(55, 90))
        pr = synth.make_pair(n, m, 48, 9000 + k, tilt_deg=1.0)
        pairs.append((pr.map1, pr.map2))
    batch = rb.batch_from_pairs(reg, pairs)
    got = {}
    for setting in ("0", cap):
        monkeypatch.setenv("ROMAN_SOLVE_CAP", setting)
        got[setting] = rb.run_batch(reg, batch)
    a, b_ = got["0"], got[cap]
    assert (a.stats["n_live"] > 128).sum() >= B // 2              # (the general instantiation's problems: the one-wave solver has no budget)
    assert (a.stats["n_pass"] > int(cap)).sum() >= B // 2         # ... and most of them run out of it
    assert np.array_equal(a.status, b_.status)
    for k in range(B):
        assert np.array_equal(a.assoc[k], b_.assoc[k]), k
    assert np.array_equal(a.T, b_.T, equal_nan=True)
    for f in ("n_live", "nnz_upper", "n_pass", "outer_iters", "inner_iters", "ls_trials", "score", "d_final"):
        assert np.array_equal(a.stats[f], b_.stats[f]), f
