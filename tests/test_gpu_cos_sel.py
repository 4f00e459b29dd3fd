"""k_cos_sel — the cosine stage of a batch whose cosines only matter behind the gate cos > cosine_min
([REF roman/align/roman_registration.py:52-59]: scores of 0 below cosine_min): a bf16 matrix-core screen of all pairs, then the exact f64
contraction (the oracle's stated order) for the pairs the screen cannot rule out.  What must hold:
  * the screen's matrix lies within 0.0042 of the exact one (the bound the kernel's margin 2^-6 is set against);
  * every pair whose exact cosine is not below cosine_min - 2^-6 + 0.0042 — in particular every pair that passes the gate — holds the
    oracle's BITS; every other pair holds either those bits or a value that is below cosine_min like the exact one;
  * descriptors whose norm the screen cannot bound (huge, tiny, inf, NaN), zero descriptors, more candidates than the list holds, maps beyond
    the kernel's block budget: bits of the dense kernel throughout;
  * a batch scored through it gives the same live set, scores, associations and transforms as through the dense kernel, bit for bit.
The switch ROMAN_COS_SEL is read per call (0 never, 1 whenever the results allow it; roman_debug_cosine: approx / gated)."""
import numpy as np
import pytest

from conftest import registration_for
from roman_amd import _abi, synth
from roman_amd.align import batch as rb

pytestmark = pytest.mark.gpu

BOUND = 0.0042
DELTA = 2.0 ** -6


def _descs(rng, n, d, dirn, lo=0.1, hi=1.2):
    """Descriptors that share a direction: cosines on both sides of the default cosine_min of 0.5 (3-15 % above it)."""
    w = rng.uniform(lo, hi, size=(n, 1))
    X = rng.standard_normal((n, 3 + d))
    X[:, 3:] += w * dirn
    return X


def _three(ctx, monkeypatch, P, D1, D2):
    monkeypatch.delenv("ROMAN_COS_SEL", raising=False)
    exact = ctx.debug_cosine(P, D1, D2)
    monkeypatch.setenv("ROMAN_COS_SEL", "approx")
    approx = ctx.debug_cosine(P, D1, D2)
    monkeypatch.setenv("ROMAN_COS_SEL", "gated")
    gated = ctx.debug_cosine(P, D1, D2)
    monkeypatch.delenv("ROMAN_COS_SEL", raising=False)
    return exact, approx, gated


SHAPES = [(200, 200, 512), (1, 1, 9), (5, 3, 1), (16, 16, 32), (17, 5, 31), (37, 53, 70), (200, 200, 48), (113, 97, 33), (256, 160, 64), (160, 256, 100),
          (208, 208, 515), (49, 49, 15), (64, 64, 768), (130, 40, 16), (200, 200, 7)]


@pytest.mark.parametrize("n1,n2,d", SHAPES)
def test_screen_is_within_its_bound_and_candidates_hold_the_exact_bits(ctx, monkeypatch, n1, n2, d):
    rng = np.random.default_rng(n1 * 991 + n2 * 7 + d)
    P = _abi.RomanParams.default(); P.cos_feature_dim = d
    dirn = rng.standard_normal(d)
    D1, D2 = _descs(rng, n1, d, dirn), _descs(rng, n2, d, dirn)
    D1[:, 3:] *= np.linspace(0.5, 2.0, d)               # asymmetric operands (a row/column swap shows)
    if n1 > 2:
        D1[2, 3:] = 0.0                                  # a zero descriptor: cosine 0 by the reference's guard
    exact, approx, gated = _three(ctx, monkeypatch, P, D1, D2)
    assert np.max(np.abs(approx - exact)) <= BOUND
    if n1 > 2:
        assert np.all(gated[2] == 0.0) and np.all(approx[2] == 0.0)
    same = gated.view(np.uint64) == exact.view(np.uint64)
    must = ~(exact < P.cosine_min - DELTA + BOUND)
    assert np.all(same[must])
    assert np.all(same[exact > P.cosine_min])
    rest = ~same
    assert np.all(exact[rest] < P.cosine_min) and np.all(gated[rest] < P.cosine_min)
    assert np.all(np.abs(gated[rest] - exact[rest]) <= BOUND)
    if d >= 31 and n1 >= 16:
        assert must.any()                                # the case has pairs on both sides of the gate ...
        if (exact >= P.cosine_min - DELTA - BOUND).sum() < 4000:
            assert rest.any()                            # ... and, unless its candidates overflow the list (dense kernel), screened-out ones


def test_rows_the_screen_cannot_bound_are_computed_exactly(ctx, monkeypatch):
    rng = np.random.default_rng(3)
    d = 96
    P = _abi.RomanParams.default(); P.cos_feature_dim = d
    dirn = rng.standard_normal(d)
    D1, D2 = _descs(rng, 60, d, dirn), _descs(rng, 70, d, dirn)
    D1[3, 3:] *= 1e200; D1[4, 3:] *= 1e-200; D1[5, 3 + 7] = np.inf; D1[6, 3 + 9] = np.nan; D1[7, 3:] *= 2.0 ** 41; D1[8, 3:] *= 2.0 ** -45
    D2[11, 3:] *= 1e150; D2[12, 3:] *= 1e-150; D2[13, 3 + 1] = -np.inf; D2[14, 3 + 2] = np.nan; D2[15, 3:] = 0.0
    D1[9, 3:] *= 1e160                                   # (1e160)^2 overflows: norm inf
    with np.errstate(all="ignore"):
        exact, approx, gated = _three(ctx, monkeypatch, P, D1, D2)
    eb, gb = exact.view(np.uint64), gated.view(np.uint64)
    for i in (3, 4, 5, 6, 7, 8, 9):
        assert np.array_equal(gb[i], eb[i]), i
    for j in (11, 12, 13, 14, 15):
        assert np.array_equal(gb[:, j], eb[:, j]), j
    ordinary = np.ones_like(exact, dtype=bool)
    ordinary[[3, 4, 5, 6, 7, 8, 9]] = False; ordinary[:, [11, 12, 13, 14, 15]] = False
    same = gb == eb
    assert np.all(same[ordinary & ~(exact < P.cosine_min - DELTA + BOUND)])
    assert np.all(gated[ordinary & ~same] < P.cosine_min)


def test_more_candidates_than_the_list_holds_and_maps_beyond_the_block_budget_go_to_the_dense_kernel(ctx, monkeypatch):
    rng = np.random.default_rng(4)
    d = 64
    P = _abi.RomanParams.default(); P.cos_feature_dim = d
    dirn = rng.standard_normal(d)
    D1, D2 = _descs(rng, 150, d, dirn, lo=3.0, hi=4.0), _descs(rng, 140, d, dirn, lo=3.0, hi=4.0)     # every cosine ~0.9: 21000 candidates
    exact, approx, gated = _three(ctx, monkeypatch, P, D1, D2)
    assert exact.min() > 0.7
    assert np.array_equal(gated.view(np.uint64), exact.view(np.uint64))
    D1, D2 = _descs(rng, 256, d, dirn), _descs(rng, 250, d, dirn)                         # 16 x 16 + 32 blocks > 208
    exact, approx, gated = _three(ctx, monkeypatch, P, D1, D2)
    assert np.array_equal(gated.view(np.uint64), exact.view(np.uint64))
    assert np.array_equal(approx.view(np.uint64), exact.view(np.uint64))


@pytest.mark.parametrize("B,nlo,nhi,d,seed", [(24, 200, 200, 512, 1), (40, 30, 90, 33, 2), (9, 100, 140, 128, 3)])
def test_a_batch_scored_through_the_screen_equals_the_dense_kernel_bit_for_bit(ctx, monkeypatch, B, nlo, nhi, d, seed):
    """Planted-inlier pairs of the bench's generator (class-clustered descriptors: the gate has pairs on both sides) through the batched
    call with and without ROMAN_COS_SEL=1: status, associations incl. order, poses, live counts, nnz, pass counts, scores — every output
    identical bit for bit."""
    reg = registration_for("semanticgrav", semantics_dim=d); reg.set_context(ctx)
    rng = np.random.default_rng(seed)
    pairs = []
    for k in range(B):
        n, m = int(rng.integers(nlo, nhi + 1)), int(rng.integers(nlo, nhi + 1))
        pr = synth.make_pair(n, m, d, 7000 + 10 * seed + k, tilt_deg=1.0)
        pairs.append((pr.map1, pr.map2))
    batch = rb.batch_from_pairs(reg, pairs)
    got = {}
    for setting in ("0", "0", "1"):                  # (the first call of a parameter block sizes its workspace without a history: once before the two compared)
        monkeypatch.setenv("ROMAN_COS_SEL", setting)
        got[setting] = rb.run_batch(reg, batch)
    monkeypatch.delenv("ROMAN_COS_SEL", raising=False)
    a, b_ = got["0"], got["1"]
    assert (a.stats["n_live"] > 0).all()
    assert np.array_equal(a.status, b_.status)
    for k in range(B):
        assert np.array_equal(a.assoc[k], b_.assoc[k]), k
    assert np.array_equal(a.T, b_.T, equal_nan=True)
    for f in ("n_live", "nnz_upper", "n_pass", "outer_iters", "inner_iters", "ls_trials", "score", "d_final"):
        assert np.array_equal(a.stats[f], b_.stats[f]), f


def test_the_library_takes_the_screen_for_large_batches_and_drops_it_when_it_rules_out_too_little(monkeypatch):
    """Without the switch: a batch that gives half the compute units a problem takes the screen; descriptors that are all alike (every
    cosine above cosine_min: every problem overflows the candidate list and is left to the dense kernel) make the NEXT batches of that
    parameter block take the dense kernels directly; results are the dense kernel's either way."""
    from roman_amd.runtime import Context
    monkeypatch.delenv("ROMAN_COS_SEL", raising=False)
    d = 64
    reg = registration_for("semanticgrav", semantics_dim=d)
    c = Context(0)
    try:
        reg.set_context(c)
        B = 160
        pairs = []
        for k in range(B):
            pr = synth.make_pair(70, 72, d, 9100 + k, tilt_deg=1.0)
            pairs.append((pr.map1, pr.map2))
        batch = rb.batch_from_pairs(reg, pairs)
        r1 = rb.run_batch(reg, batch); c.sync()
        s1 = c.cosine_screen_stats()
        assert s1[0] >= 1 and s1[2] == 0.0                 # screened, nothing left to the dense kernel
        monkeypatch.setenv("ROMAN_COS_SEL", "0")
        r0 = rb.run_batch(reg, batch); c.sync()
        monkeypatch.delenv("ROMAN_COS_SEL", raising=False)
        assert np.array_equal(r0.status, r1.status) and all(np.array_equal(a, b) for a, b in zip(r0.assoc, r1.assoc))
        # the same maps with descriptors that are all alike
        alike = rb.AlignmentBatch(batch.feats.copy(), batch.off1, batch.n1, batch.off2, batch.n2, batch.assoc, batch.assoc_off)
        P = reg._abi_params(); lo = P.point_dim + P.ratio_feature_dim
        alike.feats[:, lo:lo + d] = 1.0 + 0.01 * np.random.default_rng(1).standard_normal((alike.feats.shape[0], d))
        before = c.cosine_screen_stats()
        ra = rb.run_batch(reg, alike); c.sync()
        mid = c.cosine_screen_stats()
        assert mid[0] == before[0] + 1 and mid[2] == 1.0   # screened once more: every problem fell back
        rb2 = rb.run_batch(reg, alike); c.sync()
        after = c.cosine_screen_stats()
        assert after[0] == mid[0] and after[1] == mid[1] + 1        # ... so this batch took the dense kernels directly
        assert np.array_equal(ra.status, rb2.status) and all(np.array_equal(a, b) for a, b in zip(ra.assoc, rb2.assoc))
        assert np.array_equal(ra.T, rb2.T, equal_nan=True)
    finally:
        c.close()
