"""The clipperpy-compatible shim on the GPU: the call sequences the reference's files make
(object_registration.py:22-29,50-86; roman_registration.py:82-96; dist_reg_with_pruning.py:48-57)."""
import numpy as np
import pytest

import roman_amd
from conftest import registration_for
from roman_amd import synth

pytestmark = pytest.mark.gpu


class _Recorder:
    """Logs the name of every C-ABI entry called through it (the real libroman_hip.so underneath)."""

    def __init__(self, lib):
        self._lib = lib; self.calls = []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("roman_"):
            return fn

        def wrapped(*a):
            self.calls.append(name)
            return fn(*a)
        return wrapped


def test_reference_call_sequence_through_shim(ctx, orc):
    clipperpy = roman_amd.install_clipperpy_shim(force=True)
    reg = registration_for("roman", semantics_dim=24)
    pr = synth.make_pair(40, 36, 24, 70)
    map1_cl, map2_cl = reg.pack(pr.map1), reg.pack(pr.map2)

    ip = clipperpy.invariants.ROMANParams()                       # roman_registration.py:55-78
    ip.point_dim = 3; ip.ratio_feature_dim = 4; ip.cos_feature_dim = 24
    ip.sigma, ip.epsilon, ip.mindist = 0.4, 0.6, 0.2
    ip.distance_weight = ip.ratio_weight = ip.cosine_weight = 1.0
    ip.ratio_epsilon = np.ones(4) * 0.0
    ip.cosine_min, ip.cosine_max = 0.5, 0.7
    ip.gravity_guided = True; ip.drift_aware = False; ip.gravity_unc_ang_rad = 0.0872665
    clipper = clipperpy.CLIPPERPairwiseAndSingle(clipperpy.invariants.ROMAN(ip), clipperpy.Params())
    clipper._ctx = ctx
    A_init = clipperpy.utils.create_all_to_all(len(pr.map1), len(pr.map2))
    rec = _Recorder(ctx._lib); ctx._lib = rec                     # the C-ABI calls of ONE register(): the same sequence the
    try:                                                          # reference's unmodified files make on the CPU box (tests/test_reference_files_on_shim.py)
        clipper.score_pairwise_and_single_consistency(map1_cl.T, map2_cl.T, A_init)   # F-ordered (F,n) views
        clipper.solve()
        Ain = clipper.get_selected_associations()
    finally:
        ctx._lib = rec._lib
    from _recording_lib import EXPECTED_REGISTER_CALLS
    assert [c for c in rec.calls if c != "roman_last_error"] == EXPECTED_REGISTER_CALLS, rec.calls
    ref = orc.register(reg._abi_params(), map1_cl, map2_cl)
    assert np.array_equal(Ain, ref["assoc"])
    sol = clipper.get_solution()
    assert np.array_equal(A_init[sol.nodes], Ain) and sol.u.shape == (len(A_init),) and abs(sol.score - ref["stats"].score) < 1e-8
    M = clipper.get_affinity_matrix(); C = clipper.get_constraint_matrix()
    assert M.shape == C.shape == (len(A_init),) * 2 and np.array_equal(M, M.T) and np.all(np.diag(C) == 1)
    assert np.array_equal((M != 0) & ~np.eye(len(M), dtype=bool), (C != 0) & ~np.eye(len(M), dtype=bool))   # C has M's pattern

    # plain CLIPPER with the Euclidean invariant and a pruned list (dist_reg_with_pruning.py:48-57,96)
    ep = clipperpy.invariants.EuclideanDistanceParams(); ep.sigma, ep.epsilon, ep.mindist = 0.4, 0.6, 0.2
    c2 = clipperpy.CLIPPER(clipperpy.invariants.EuclideanDistance(ep), clipperpy.Params()); c2._ctx = ctx
    A_put = np.delete(A_init, np.arange(0, len(A_init), 3), axis=0)
    pts1, pts2 = map1_cl[:, :3], map2_cl[:, :3]
    c2.score_pairwise_consistency(pts1.T, pts2.T, A_put)
    c2.solve()
    preg = registration_for("clipper+prune")
    assert np.array_equal(c2.get_selected_associations(), orc.register(preg._abi_params(), pts1, pts2, A_put.astype(np.int32))["assoc"])
    assert np.all(np.diag(c2.get_affinity_matrix()) == 1.0)              # implicit identity exported

    # set_matrix_data / get_solution loop of mno_clipper (object_registration.py:60-72)
    c3 = clipperpy.CLIPPER(clipperpy.invariants.PairwiseInvariant(), clipperpy.Params()); c3._ctx = ctx
    Md = c2.get_affinity_matrix(); Cd = c2.get_constraint_matrix()
    c3.set_matrix_data(M=Md, C=Cd); c3.solve()
    assert np.array_equal(A_put[c3.get_solution().nodes], c2.get_selected_associations())


def test_context_sharing_between_shim_objects_and_plugins(ctx, orc):
    """One context holds ONE stepwise problem.  A shim CLIPPER object whose problem was displaced — by another
    CLIPPER object, by a registration plugin's register(), or by a batch call on the same context — reloads its
    own inputs before solve() / get_*_matrix() instead of silently operating on the newcomer's problem."""
    clipperpy = roman_amd.install_clipperpy_shim(force=True)
    ep = clipperpy.invariants.EuclideanDistanceParams(); ep.sigma, ep.epsilon, ep.mindist = 0.4, 0.6, 0.2
    prA, prB = synth.make_pair(20, 22, 0, 801), synth.make_pair(26, 18, 0, 802)
    reg = registration_for("clipper"); reg.set_context(ctx)
    ptsA = (reg.pack(prA.map1), reg.pack(prA.map2)); ptsB = (reg.pack(prB.map1), reg.pack(prB.map2))
    refA = orc.register(reg._abi_params(), *ptsA)["assoc"]; refB = orc.register(reg._abi_params(), *ptsB)["assoc"]
    assert not np.array_equal(refA, refB)
    cA = clipperpy.CLIPPER(clipperpy.invariants.EuclideanDistance(ep), clipperpy.Params()); cA._ctx = ctx
    cB = clipperpy.CLIPPER(clipperpy.invariants.EuclideanDistance(ep), clipperpy.Params()); cB._ctx = ctx
    cA.score_pairwise_consistency(ptsA[0].T, ptsA[1].T)
    cB.score_pairwise_consistency(ptsB[0].T, ptsB[1].T)            # displaces A's problem
    cA.solve()
    assert np.array_equal(cA.get_selected_associations(), refA)
    assert np.array_equal(reg.register(prB.map1, prB.map2), refB)   # a plugin call on the same context ...
    assert cA.get_affinity_matrix().shape == (20 * 22,) * 2         # ... and A still answers for its own problem
    reg.register_and_align_batch([(prB.map1, prB.map2)])            # so does a batch call
    cA.solve()
    assert np.array_equal(cA.get_selected_associations(), refA)
    cB.solve()
    assert np.array_equal(cB.get_selected_associations(), refB)
