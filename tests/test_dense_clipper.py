"""Cross-check of the C oracle (oracle/clipper_oracle.c) against the independent dense NumPy statement of
the published algorithm (tests/_dense_clipper.py) on problems of <= 40 objects per map.  Two restatements
written from different sources that agree on trajectory and selection make a transcription slip in either
unlikely; neither pins the absent clipperpy."""
import numpy as np
import pytest

from conftest import registration_for
from roman_amd import synth
from _dense_clipper import find_dense_clique

CASES = [  # (id, method, kwargs, n, m, d, seed)
    ("clipper30", "clipper", {}, 30, 30, 0, 1000),
    ("clipper_ragged", "clipper", {}, 17, 38, 0, 21),
    ("clipper2d", "clipper", {"dim": 2}, 25, 25, 0, 22),
    ("gravity", "gravity", {}, 36, 30, 0, 11),
    ("pcavolgrav", "pcavolgrav", {"epsilon_shape": 0.2}, 32, 30, 0, 23),
    ("semgrav", "semanticgrav", {"semantics_dim": 24}, 40, 36, 24, 12),
    ("roman", "roman", {"semantics_dim": 16}, 34, 34, 16, 13),
    ("prune", "clipper+prune", {"cosine_min": 0.5}, 40, 40, 32, 15),
    ("outliers_only", "clipper", {}, 20, 20, 0, 77),
]


def _pair(case):
    _, method, kw, n, m, d, seed = case
    reg = registration_for(method, **kw)
    if case[0] == "outliers_only":
        pr = synth.make_pair(n, m, d, seed, inlier_frac=0.0)
    else:
        pr = synth.make_pair(n, m, d, seed, tilt_deg=1.0 if reg._abi_params().gravity_guided else 0.0)
    if kw.get("dim") == 2:
        for o in pr.map1 + pr.map2:
            o.centroid = o.centroid[:2]; o.dim = 2
    return reg, pr


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_c_oracle_matches_dense_numpy_statement(orc, case):
    reg, pr = _pair(case)
    P = reg._abi_params()
    D1, D2 = reg.pack(pr.map1), reg.pack(pr.map2)
    A = reg._association_list(pr.map1, pr.map2)
    mat, _ = orc.build_matrix(P, D1, D2, A)
    M, C = mat.dense()
    sol_d = find_dense_clique(M, C, tol_u=P.tol_u, tol_F=P.tol_F, maxiniters=P.maxiniters, maxoliters=P.maxoliters,
                              beta=P.beta, maxlsiters=P.maxlsiters, eps=P.eps, rescale_u0=bool(P.rescale_u0))
    # both organisations of the oracle's passes (published: products carried to the d update; the device stream solver's:
    # fused line-search products + a split pass per d update) against the one dense statement
    for mode, key in (("carried", "n_pass"), ("fused", "n_pass_fused")):
        with orc.pass_mode(mode):
            sol_c = orc.solve(P, mat)
        st = sol_c["stats"]
        assert np.array_equal(sol_c["nodes"], sol_d["nodes"]), "selected nodes / order differ"
        assert (st.inner_iters, st.ls_trials, st.outer_iters, st.n_pass) == \
            (sol_d["inner_iters"], sol_d["ls_trials"], sol_d["outer_iters"], sol_d[key]), mode
        assert abs(st.score - sol_d["F"]) <= 1e-9 * max(1.0, abs(sol_d["F"]))
        assert abs(st.d_final - sol_d["d"]) <= 1e-9 * max(1.0, abs(sol_d["d"]))
        assert np.allclose(sol_c["u"], sol_d["u"], rtol=0, atol=1e-10)


def test_dense_statement_with_explicit_u0_and_no_rescale(orc):
    reg, pr = _pair(CASES[0])
    P = reg._abi_params()
    mat, _ = orc.build_matrix(P, reg.pack(pr.map1), reg.pack(pr.map2))
    M, C = mat.dense()
    rng = np.random.default_rng(5)
    u0 = rng.uniform(0.1, 1.0, mat.n)
    P2 = type(P).from_buffer_copy(P); P2.rescale_u0 = 0
    for PP in (P, P2):
        sd = find_dense_clique(M, C, u0=u0, rescale_u0=bool(PP.rescale_u0))
        for mode, key in (("carried", "n_pass"), ("fused", "n_pass_fused")):
            with orc.pass_mode(mode):
                sc = orc.solve(PP, mat, u0)
            assert np.array_equal(sc["nodes"], sd["nodes"])
            assert sc["stats"].n_pass == sd[key]


def test_dense_statement_recovers_a_planted_clique():
    """The independent statement on its own: a weighted clique planted in noise is what it returns."""
    rng = np.random.default_rng(9)
    n, k = 60, 12
    M = np.zeros((n, n)); C = np.zeros((n, n))
    idx = rng.choice(n, k, replace=False)
    for a in idx:
        for b in idx:
            if a != b:
                M[a, b] = 1.0; C[a, b] = 1.0           # unit weights: F = k exactly, so round(F) = k
    noise = np.triu(rng.uniform(0, 1, (n, n)) < 0.05, 1)
    W = np.triu(rng.uniform(0.3, 0.8, (n, n)), 1) * noise
    M = np.maximum(M, W + W.T); C = np.maximum(C, (noise + noise.T).astype(float))
    np.fill_diagonal(M, 1.0); np.fill_diagonal(C, 1.0)
    sol = find_dense_clique(M, C)
    assert set(sol["nodes"].tolist()) == set(idx.tolist())
