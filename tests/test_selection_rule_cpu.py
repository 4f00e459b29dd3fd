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
