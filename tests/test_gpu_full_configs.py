"""Oracle parity at BASELINE.json's OWN configurations (n = m = 200 objects, d = 512, method 'semanticgrav'):
EVERY problem of config 3 (the 256 pairs the benchmark times, seeds 3000..3255) and a 16 x 16 config-4 grid
(32 submaps packed once, 256 cross pairs) is compared with the CPU oracle — identical association arrays
(indices and order), identical n_live / nnz_upper / n_pass, pose within 1e-5 Frobenius of the oracle's
T_align on the oracle's associations.  The hot call these replace: [REF roman/align/submap_align.py:155-166]."""
import numpy as np
import pytest

from conftest import registration_for
from roman_amd import _abi, synth
from roman_amd.align import batch as rb

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-5


def _compare(orc, reg, res, problems):
    """problems: list of (D1, D2) packed feature matrices, in batch order.
    -> (problems whose RESULT differs, worst pose error, problems whose iteration counts differ).
    Results — association arrays incl. order, live set size, stored non-zeros, pose — must be identical for every
    problem.  Iteration counts are compared too but not required to be equal everywhere: the line search decides on
    `dF < -1e-9` with F ~ 100, so a last-bit difference of a sum (reduction order) can add or drop a trial; the
    oracle's own two arithmetic modes disagree on a few of these problems' pass counts (tools/gpu_diag_cfg3.py)."""
    P = reg._abi_params()
    bad, traj = [], []
    worst = 0.0
    # many problems: one host thread per problem (oracle.register_each); a few large ones: one after the other on all threads
    oracle_results = orc.register_each(P, problems) if len(problems) >= 64 else [orc.register(P, D1, D2, faithful=False) for D1, D2 in problems]
    for b, (D1, D2) in enumerate(problems):
        o = oracle_results[b]
        st = o["stats"]
        same = (np.array_equal(res.assoc[b], o["assoc"]) and res.stats["n_live"][b] == st.n_live
                and res.stats["nnz_upper"][b] == st.nnz_upper and res.stats["outer_iters"][b] == st.outer_iters
                and abs(res.stats["score"][b] - st.score) < 1e-6)
        if not (res.stats["n_pass"][b] == st.n_pass and res.stats["inner_iters"][b] == st.inner_iters):
            traj.append(b)
        if len(o["assoc"]) >= 3:
            T_o = orc.t_align(D1[o["assoc"][:, 0], :3], D2[o["assoc"][:, 1], :3])
            err = float(np.linalg.norm(res.T[b] - T_o))
            worst = max(worst, err)
            same = same and res.status[b] == _abi.ROMAN_ST_OK and err < POSE_TOL
        else:
            same = same and bool(res.status[b] & _abi.ROMAN_ST_INSUFFICIENT)
        if not same:
            bad.append(b)
    return bad, worst, traj


def _as_published(orc, P, feats, off1, n1, off2, n2, kmax, res):
    """The same problems once more through the oracle AS PUBLISHED — pass mode `carried` (every pass forms M x and C x, the
    accepted trial's products are carried to the d update: findDenseClique's own order) and PLAIN arithmetic (sequential
    descriptor dot, glibc exp / cbrt instead of the stated-order sequences the device reproduces) — i.e. with nothing of the
    oracle shaped after the device.  -> (problems whose association array — indices AND order — or pose differs, worst pose
    error, number of problems whose pass count differs from the device's (reported, not asserted: the device counts one
    split pass per d update))."""
    with orc.pass_mode("carried"), orc.plain_arith():
        many = orc.register_many(P, feats, off1, n1, off2, n2, kmax, faithful=False)
    bad, worst = [], 0.0
    for b in range(len(n1)):
        a = many[b]
        same = np.array_equal(res.assoc[b], a)
        if len(a) >= 3:
            T_o = orc.t_align(feats[off1[b] + a[:, 0], :3], feats[off2[b] + a[:, 1], :3])
            err = float(np.linalg.norm(res.T[b] - T_o)); worst = max(worst, err)
            same = same and err < POSE_TOL
        if not same:
            bad.append(b)
    return bad, worst


def test_config3_every_one_of_the_256_pairs_matches_the_oracle(ctx, orc):
    reg = registration_for("semanticgrav", semantics_dim=512); reg.set_context(ctx)
    pairs = [synth.make_pair(200, 200, 512, 3000 + k) for k in range(256)]
    batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    res = rb.run_batch(reg, batch)
    F = batch.feats.shape[1]
    problems = [(batch.feats[batch.off1[b]:batch.off1[b] + 200], batch.feats[batch.off2[b]:batch.off2[b] + 200]) for b in range(256)]
    assert problems[0][0].shape == (200, F)
    bad, worst, traj = _compare(orc, reg, res, problems)
    print(f"config 3: {256 - len(bad)}/256 identical results, worst pose error {worst:.2e}, iteration counts differ on {traj}")
    assert not bad, f"{len(bad)} of 256 problems differ from the oracle: {bad[:10]}"
    assert worst < POSE_TOL and len(traj) <= 12
    assert (res.stats["n_assoc_in"] == 40000).all()
    # ... and against the oracle as published (carried passes, libm, sequential dot): same associations incl. order, same poses
    bad_p, worst_p = _as_published(orc, reg._abi_params(), batch.feats, batch.off1, batch.n1, batch.off2, batch.n2, batch.kmax(), res)
    print(f"config 3 vs the oracle as published (carried + plain arithmetic): {256 - len(bad_p)}/256 identical, worst pose error {worst_p:.2e}")
    assert not bad_p, f"{len(bad_p)} of 256 problems differ from the published-order oracle: {bad_p[:10]}"


def test_config4_grid_16x16_every_pair_matches_the_oracle(ctx, orc):
    reg = registration_for("semanticgrav", semantics_dim=512); reg.set_context(ctx)
    S = 16
    subs, poses = synth.make_submap_grid(2 * S, n=200, d=512, seed0=4000)
    batch = rb.batch_from_submap_grid(reg, subs[:S], subs[S:])
    assert len(batch) == S * S and batch.feats.shape[0] == 2 * S * 200        # every submap packed once
    res = rb.run_batch(reg, batch)
    problems = [(batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]], batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]])
                for b in range(len(batch))]
    bad, worst, traj = _compare(orc, reg, res, problems)
    print(f"config 4 grid: {S * S - len(bad)}/{S * S} identical results, worst pose error {worst:.2e}, iteration counts differ on {traj}")
    assert not bad, f"{len(bad)} of {S * S} grid problems differ from the oracle: {bad[:10]}"
    assert worst < POSE_TOL and len(traj) <= 12
    ok = sum(int(res.status[b] == 0 and len(res.assoc[b]) >= 20) for b in range(len(batch)))
    assert ok >= 0.9 * len(batch)                                            # overlapping submaps do align


def test_config4_full_64x64_grid_all_4096_pairs_match_the_oracle(ctx, orc):
    """BASELINE config 4 in full: the 64 x 64 = 4096 cross pairs of 128 submaps (n = 200, d = 512) that bench.py's
    `grid_config4` leg times, EVERY pair compared with the oracle (one OpenMP thread per pair, oracle_register_many):
    identical association arrays incl. order, pose within 1e-5 Frobenius of the oracle's T_align on the oracle's associations."""
    reg = registration_for("semanticgrav", semantics_dim=512); reg.set_context(ctx)
    S = 64
    subs, poses = synth.make_submap_grid(2 * S, n=200, d=512, seed0=4000)
    batch = rb.batch_from_submap_grid(reg, subs[:S], subs[S:])
    assert len(batch) == S * S and batch.feats.shape[0] == 2 * S * 200
    res = rb.run_batch(reg, batch)
    many = orc.register_many(reg._abi_params(), batch.feats, batch.off1, batch.n1, batch.off2, batch.n2, batch.kmax(), faithful=False)
    bad, worst = [], 0.0
    for b in range(len(batch)):
        a = many[b]
        same = np.array_equal(res.assoc[b], a)
        if len(a) >= 3:
            T_o = orc.t_align(batch.feats[batch.off1[b] + a[:, 0], :3], batch.feats[batch.off2[b] + a[:, 1], :3])
            err = float(np.linalg.norm(res.T[b] - T_o)); worst = max(worst, err)
            same = same and res.status[b] == _abi.ROMAN_ST_OK and err < POSE_TOL
        else:
            same = same and bool(res.status[b] & _abi.ROMAN_ST_INSUFFICIENT)
        if not same:
            bad.append(b)
    print(f"config 4, full grid: {S * S - len(bad)}/{S * S} identical results, worst pose error {worst:.2e}, most passes {int(res.stats['n_pass'].max())}")
    assert not bad, f"{len(bad)} of {S * S} grid problems differ from the oracle: {bad[:10]}"
    assert worst < POSE_TOL
    bad_p, worst_p = _as_published(orc, reg._abi_params(), batch.feats, batch.off1, batch.n1, batch.off2, batch.n2, batch.kmax(), res)
    print(f"config 4 vs the oracle as published (carried + plain arithmetic): {S * S - len(bad_p)}/{S * S} identical, worst pose error {worst_p:.2e}")
    assert not bad_p, f"{len(bad_p)} of {S * S} grid problems differ from the published-order oracle: {bad_p[:10]}"


def test_demo_scale_1024_pairs_match_the_oracle(ctx, orc):
    """The scale the reference's demo runs at ([REF params/demo/submap_align.yaml:2,7,15]: submap_max_size 40, method 'roman',
    768-d descriptors): 1024 pairs with n, m uniform in [20, 40] in ONE call — the path of the small kernels (one-wave
    solver, one-block cosine tiles, parallel batch scans, small list / fill workgroups); every result is the oracle's."""
    from roman_amd.align import SubmapAlignParams
    reg = SubmapAlignParams(method="roman", semantics_dim=768).get_object_registration(); reg.set_context(ctx)
    rng = np.random.default_rng(5000)
    sizes = rng.integers(20, 41, size=(256, 2))
    # (every 16th pair: nine of ten objects are planted inliers — several hundred to a thousand stored pairs: beyond the
    #  coordinate-list registers of the one-wave solver, its quad stream takes those)
    base = [synth.make_pair(int(a), int(b), 768, 5000 + k, tilt_deg=1.0, **({"inlier_frac": 0.9, "noise": 0.03} if k % 16 == 5 else {}))
            for k, (a, b) in enumerate(sizes)]
    b256 = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in base])
    rep = 4                                           # the 256 distinct pairs four times: 1024 problems per call
    batch = rb.AlignmentBatch(b256.feats, np.tile(b256.off1, rep), np.tile(b256.n1, rep), np.tile(b256.off2, rep), np.tile(b256.n2, rep))
    for _ in range(2):                                # the second call runs with the sizing history (small workgroups, range claims)
        res = rb.run_batch(reg, batch)
    problems = [(batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]], batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]]) for b in range(256)]
    bad, worst, traj = _compare(orc, reg, res, problems)
    nz = res.stats["nnz_upper"][:256]; lv = res.stats["n_live"][:256]
    print(f"demo scale: {256 - len(bad)}/256 identical results, worst pose error {worst:.2e}, iteration counts differ on {traj}; "
          f"one-wave problems {int((lv <= 128).sum())}, of them in coordinate form {int(((lv <= 128) & (nz <= 384)).sum())}, on the quad stream {int(((lv <= 128) & (nz > 384)).sum())}")
    assert ((lv <= 128) & (nz <= 384)).sum() >= 100 and ((lv <= 128) & (nz > 384)).sum() >= 2
    assert not bad, f"{len(bad)} of 256 problems differ from the oracle: {bad[:10]}"
    assert worst < POSE_TOL and len(traj) <= 12
    for r in range(1, rep):                           # the replicas are the same problems: the same results
        for b in range(256):
            assert np.array_equal(res.assoc[r * 256 + b], res.assoc[b]) and res.status[r * 256 + b] == res.status[b]


def test_mixed_batch_large_and_small_live_sets(ctx, orc):
    """One problem whose live set exceeds the streaming solver's capacity next to ordinary ones: each problem
    takes the layout / solver that fits it, results equal the oracle's for all of them."""
    reg = registration_for("gravity"); reg.set_context(ctx)
    sizes = [(40, 40), (75, 80), (35, 30), (20, 20)]                           # L = n*m: 1600, 6000, 1050, 400
    pairs = [synth.make_pair(n, m, 0, 900 + k, tilt_deg=1.0) for k, (n, m) in enumerate(sizes)]
    batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    res = rb.run_batch(reg, batch)
    problems = [(batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]], batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]])
                for b in range(len(batch))]
    bad, worst, traj = _compare(orc, reg, res, problems)
    assert not bad and worst < POSE_TOL and not traj
    assert res.stats["n_live"].tolist() == [1600, 6000, 1050, 400]


# ROMAN_WIDE_COMPACT: 0 = k_solve_wide never compacts the matrix's columns; 0x01FF10 = a window of ONE pass, threshold 255/256,
# 16 compactions per problem: a copy is cut at almost every pass, the next vector's support leaves its columns again and again
# (the line search re-admits elements) — the way back to the full matrix and the compaction of a copy in place run many times
# ROMAN_WIDE_UPPER: the instantiation of k_solve_wide with pull + push passes over the HALF copy of the matrix (round 6: every stored pair
# once, column blocks with fixed-point accumulators in LDS, rows in a per-block order, until the first column compaction takes the mirror
# pools).  The library takes it for TEAMS on live sets of at least 8 000 associations (this batch: L up to 10 000) and not otherwise;
# "1" / "0" force it / the plain kernel.  With the compaction off the half copy serves EVERY pass of a problem, with one at every pass it
# serves the first pass only.
_COMPACT_CASES = [(None, None, None, None), ("2", None, None, None), ("0", None, None, None), (None, "0", None, None), (None, "0x01FF10", None, None), ("0", "0x01FF10", None, None),
                  ("2", "0x02C008", None, None), (None, "0x01FF10", "0", None), (None, None, "0", None),
                  ("0", None, None, "1"), ("1", "0", None, "1"), ("1", "0x040010", None, "1"),
                  (None, None, None, "0"), ("2", None, None, "0"), (None, "0x01FF10", None, "0"), ("2", "0x02C008", None, "0"), (None, "0", None, "0")]
# (no switch: the library's choice — the half copy for the team settings of this batch, the plain kernel for the whole device and for 32-bit labels)
_COMPACT_IDS = ["teams_auto", "two_teams_per_xcd", "whole_device", "teams_auto-no_compaction", "teams_auto-compaction_every_pass",
                "whole_device-compaction_every_pass", "two_teams_per_xcd-eager_compaction", "teams_auto-compaction_every_pass-32bit_labels",
                "teams_auto-32bit_labels",
                "whole_device-half_copy", "one_team_per_xcd-no_compaction-half_copy", "one_team_per_xcd-late_compaction-half_copy",
                "teams_auto-plain_kernel", "two_teams_per_xcd-plain_kernel", "teams_auto-compaction_every_pass-plain_kernel",
                "two_teams_per_xcd-eager_compaction-plain_kernel", "teams_auto-no_compaction-plain_kernel"]


@pytest.mark.parametrize("teams,compact,idx16,upper", _COMPACT_CASES, ids=_COMPACT_IDS)
def test_batch_of_mid_size_live_sets_matches_the_oracle(ctx, orc, teams, compact, idx16, upper, monkeypatch):
    """Methods without a semantic gate ('gravity', 'clipper', 'pcavolgrav': [REF roman/params/submap_align_params.py:98-116])
    make every association live: L = n * m, 3600 ... 10 000 at 60-100 objects per submap — beyond the stream layout.  A batch
    of such problems is solved by TEAMS of compute units (the workgroups of an XCD, or of half an XCD, on one problem each,
    k_solve_wide in team mode); every result equals the oracle's, as it does with the whole device on one problem at a time —
    with the solver's column compaction as the library sets it, off, and forced at (almost) every pass."""
    if teams is not None:
        monkeypatch.setenv("ROMAN_WIDE_TEAMS", teams)
    if compact is not None:
        monkeypatch.setenv("ROMAN_WIDE_COMPACT", compact)
    if idx16 is not None:
        monkeypatch.setenv("ROMAN_WIDE_IDX16", idx16)           # "0": 32-bit column labels (the layout of dense problems with C-flags)
    if upper is not None:
        monkeypatch.setenv("ROMAN_WIDE_UPPER", upper)
    reg = registration_for("gravity"); reg.set_context(ctx)
    rng = np.random.default_rng(77)
    sizes = [(int(a), int(b)) for a, b in rng.integers(60, 101, size=(18, 2))] + [(100, 100), (30, 30), (64, 48)]   # (30 x 30, 64 x 48: stream layout)
    pairs = [synth.make_pair(n, m, 0, 980 + k, tilt_deg=1.0) for k, (n, m) in enumerate(sizes)]
    batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    res = rb.run_batch(reg, batch)
    problems = [(batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]], batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]])
                for b in range(len(batch))]
    bad, worst, traj = _compare(orc, reg, res, problems)
    print(f"mid-size live sets ({teams}, {compact}): {len(batch) - len(bad)}/{len(batch)} identical, iteration counts differ on {traj}")
    assert not bad and worst < POSE_TOL and len(traj) <= 2
    assert res.stats["n_live"].tolist() == [n * m for n, m in sizes]


@pytest.mark.parametrize("count,lo,hi", [(28, 62, 76), (70, 56, 66), (3, 114, 120)],
                         ids=["28_problems_four_teams_per_xcd", "70_problems_one_workgroup_each", "3_problems_of_14000_half_copy_in_three_column_blocks"])
def test_many_mid_size_live_sets_take_the_solver_the_library_picks(ctx, orc, count, lo, hi):
    """The library's own choice for a batch of gate-less problems (roman_hip.hip, enqueue_score / the launch of the fallback solvers): more
    than 24 of them with at most 6 144 live associations go to FOUR teams per XCD (eight compute units per problem: the barrier and
    collect latencies of a pass overlap across 32 teams); more than a quarter of the compute units' worth of problems below 4 608
    live associations go to k_solve, one workgroup per problem.  Either way every result equals the oracle's."""
    reg = registration_for("gravity"); reg.set_context(ctx)
    rng = np.random.default_rng(count)
    sizes = [(int(a), int(b)) for a, b in rng.integers(lo, hi + 1, size=(count, 2))]
    pairs = [synth.make_pair(n, m, 0, 3300 + k, tilt_deg=1.0) for k, (n, m) in enumerate(sizes)]
    batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    res = rb.run_batch(reg, batch)
    problems = [(batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]], batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]])
                for b in range(len(batch))]
    bad, worst, traj = _compare(orc, reg, res, problems)
    assert not bad and worst < POSE_TOL and len(traj) <= 3, (bad, traj)
    assert res.stats["n_live"].tolist() == [n * m for n, m in sizes]


@pytest.mark.parametrize("upper", ["0", "1"], ids=["plain_kernel", "half_copy"])
def test_mid_size_live_sets_from_random_start_vectors(ctx, orc, upper, monkeypatch):
    """Explicit start vectors (part of the C ABI) for live sets beyond the stream layout: the first product of the whole-device solver
    is M u0 of the caller's RAW vector (rescale_u0) — the pull + push pass takes its fixed-point scale from that vector's largest
    element.  Starts over six decades; results equal the oracle's from the same starts."""
    if upper is not None:
        monkeypatch.setenv("ROMAN_WIDE_UPPER", upper)
    reg = registration_for("gravity"); reg.set_context(ctx)
    sizes = [(90, 88), (70, 100), (100, 64)]
    pairs = [synth.make_pair(n, m, 0, 1230 + k, tilt_deg=1.0) for k, (n, m) in enumerate(sizes)]
    batch = rb.batch_from_pairs(reg, [(p.map1, p.map2) for p in pairs])
    rng = np.random.default_rng(5)
    u0s = [rng.uniform(0.05, 1.0, n * m) * 10.0 ** rng.integers(-3, 4) for n, m in sizes]
    res = rb.run_batch(reg, batch, u0=np.concatenate(u0s))
    P = reg._abi_params()
    for b, (n, m) in enumerate(sizes):
        D1 = batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]]; D2 = batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]]
        o = orc.register(P, D1, D2, u0=u0s[b], faithful=False)
        assert res.status[b] == _abi.ROMAN_ST_OK and np.array_equal(res.assoc[b], o["assoc"]), b
        assert res.stats["outer_iters"][b] == o["stats"].outer_iters and abs(int(res.stats["n_pass"][b]) - int(o["stats"].n_pass)) <= 2


@pytest.mark.parametrize("upper", [None, "1"], ids=["plain", "half_copy"])
def test_gravity_200x200_all_associations_live(ctx, orc, upper, monkeypatch):
    """method 'gravity' has no semantic gate: at n = m = 200 every one of the 40 000 associations is live — far beyond
    the stream layout (L <= 3072).  The problem takes the symmetric SELL-64 layout and the COOPERATIVE fallback
    solver (all compute units on one problem, grid barriers; k_solve_wide): results and pass counts equal the oracle's —
    with the column compaction the library chooses (three or four copies over the 244 passes) and with one forced at almost
    every pass (copies of copies in place, many returns to the full matrix)."""
    if upper is not None:
        monkeypatch.setenv("ROMAN_WIDE_UPPER", upper)           # (seven column blocks, the whole device dealt to them)
    reg = registration_for("gravity"); reg.set_context(ctx)
    pr = synth.make_pair(200, 200, 0, 7001, tilt_deg=1.0)
    res = reg.register_and_align_batch([(pr.map1, pr.map2)])
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    bad, worst, traj = _compare(orc, reg, res, [(D1, D2)])
    assert not bad and worst < POSE_TOL
    monkeypatch.setenv("ROMAN_WIDE_COMPACT", "0x01FF10")
    res2 = reg.register_and_align_batch([(pr.map1, pr.map2)])
    assert np.array_equal(res2.assoc[0], res.assoc[0]) and int(res2.stats["n_pass"][0]) == int(res.stats["n_pass"][0])
    assert np.max(np.abs(res2.T[0] - res.T[0])) < POSE_TOL
    assert res.stats["n_live"][0] == 40000
    truth = set(map(tuple, pr.inliers.tolist()))
    assert len(truth & set(map(tuple, res.assoc[0].tolist()))) >= 0.9 * len(truth)
