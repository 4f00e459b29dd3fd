"""The device's selection of the omega largest elements (kernels.hip.h, finish_one) restated in numpy — ranks by (value, association
index), and the test that sends a tie at the cut to the sequential heap emulation — in the form of rounds 1-4 (the ranks of the entries
equal to the omega-th value, computed again) and in the form of round 5 (one count): the two must agree on every input, ties included."""
import numpy as np
import pytest


def ranks(v, a):
    """rank of e = number of entries greater in (value, association index) order"""
    return np.array([int(np.sum((v > v[e]) | ((v == v[e]) & (a > a[e])))) for e in range(len(v))])


def tie_old(v, a, omega):
    r = ranks(v, a)
    vstar = v[r == omega - 1][0]
    return bool(np.any((v == vstar) & (r >= omega)))


def tie_new(v, a, omega):
    r = ranks(v, a)
    vstar = v[r == omega - 1][0]
    return int(np.sum(v >= vstar)) > omega


@pytest.mark.parametrize("seed", range(40))
def test_tie_at_the_cut_is_a_count(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 60))
    levels = int(rng.integers(1, 8))
    v = rng.choice(np.linspace(0.1, 1.0, levels), size=n) if seed % 2 else rng.random(n)      # few distinct values: many ties / none
    a = rng.permutation(10 * n)[:n]                                                           # association indices are unique
    r = ranks(v, a)
    assert sorted(r.tolist()) == list(range(n))                                               # a total order: every rank once
    for omega in range(1, n + 1):
        assert tie_old(v, a, omega) == tie_new(v, a, omega)
        top = a[np.argsort(r)][:omega]                                                        # what nodesLive[rank] = association index collects
        want = sorted(zip(-v, -a))[:omega]
        assert [int(-x[1]) for x in want] == top.tolist()


def place_keys_numpy(deg, rng):
    """kernels.hip.h place_keys() restated: histogram of the degrees, exclusive scan with the LARGEST degree first, the rows scattered
    into their degree's range in an ARBITRARY order (what the atomics grant: here a random permutation), a row's rank inside its range =
    the number of smaller row indices there.  -> keys array as the bitonic sort of ((deg + 1) << 12 | (4095 - row)) leaves it."""
    L = len(deg)
    hcnt = np.bincount(deg, minlength=L)
    start = np.zeros(L, dtype=np.int64)
    run = 0
    for d in range(L - 1, -1, -1):
        start[d] = run; run += hcnt[d]
    cur = start.copy()
    tmp = np.zeros(L, dtype=np.int64)
    for k in rng.permutation(L):
        tmp[cur[deg[k]]] = k; cur[deg[k]] += 1
    keys = np.zeros(L, dtype=np.int64)
    for i in range(L):
        k = tmp[i]; d = deg[k]
        lo, hi = cur[d] - hcnt[d], cur[d]
        r = int(np.sum(tmp[lo:hi] < k))
        keys[lo + r] = ((d + 1) << 12) | (4095 - k)
    return keys


@pytest.mark.parametrize("seed", range(30))
def test_positions_from_a_histogram_equal_the_sorted_keys(seed):
    """Position of row k = #(rows of larger degree) + #(rows of the same degree and smaller index): the array place_keys() writes is the
    descending sort of the unique keys, whatever order the scatter's atomics were granted in (the GPU test compares the kernels)."""
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 400))
    spread = [1, 3, 17, max(L, 1)][seed % 4]                        # all rows of one degree ... degrees spread over the whole range
    deg = rng.integers(0, min(spread, L), size=L)
    keys = place_keys_numpy(deg, rng)
    want = np.sort(((deg.astype(np.int64) + 1) << 12) | (4095 - np.arange(L)))[::-1]
    assert np.array_equal(keys, want)
