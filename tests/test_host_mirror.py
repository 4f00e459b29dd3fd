"""Host-side mirror of the reference's plugin surface (no GPU): factory table, empty-map and
exception protocol, clipperpy shim surface."""
import sys

import numpy as np
import pytest

import roman_amd
from roman_amd import _abi, synth
from roman_amd.align import (DistRegWithPruning, GravityConstraintError, InsufficientAssociationsException,
                             ObjectRegistration, ROMANRegistration, SubmapAlignParams)

# SURVEY.md Appendix E == /root/reference/roman/params/submap_align_params.py:98-116
TABLE = {  # method: (gravity, ratio_dim, semantics)
    "clipper": (0, 0, False), "gravity": (1, 0, False), "pcavolgrav": (1, 4, False), "extentvolgrav": (1, 4, False),
    "roman": (1, 4, True), "sevg": (1, 4, True), "spv": (0, 4, False), "semanticgrav": (1, 0, True),
}


@pytest.mark.parametrize("method", sorted(TABLE))
def test_factory_flag_table(method):
    sp = SubmapAlignParams(method=method, semantics_dim=12)
    reg = sp.get_object_registration()
    assert isinstance(reg, ROMANRegistration)
    g, r, s = TABLE[method]
    p = reg._abi_params()
    assert (p.gravity_guided, p.ratio_feature_dim, p.cos_feature_dim) == (g, r, 12 if s else 0)
    assert p.invariant == _abi.ROMAN_INV_ROMAN and p.drift_aware == 0
    assert (p.distance_weight, p.ratio_weight, p.cosine_weight) == (1.0, 1.0, 1.0)
    assert (p.cosine_min, p.cosine_max) == (0.5, 0.7)          # SubmapAlignParams overrides ROMANParams' .85/1.0
    assert p.fusion_method == _abi.ROMAN_FUSE_GEOMETRIC_MEAN   # fusion_method never reaches the invariant


def test_factory_aliases_rewritten_in_place():
    sp = SubmapAlignParams(method="spvg"); sp.get_object_registration(); assert sp.method == "roman"
    sp = SubmapAlignParams(method="roman_no_semantics"); sp.get_object_registration(); assert sp.method == "pcavolgrav"
    sp = SubmapAlignParams(method="clipper+prune")
    reg = sp.get_object_registration()
    assert isinstance(reg, DistRegWithPruning) and reg.use_gravity and reg._abi_params().invariant == _abi.ROMAN_INV_EUCLIDEAN
    with pytest.raises(AssertionError):
        SubmapAlignParams(method="nope").get_object_registration()
    assert SubmapAlignParams(submap_descriptor="None").submap_descriptor is None


def test_feature_row_layout():
    """[x y z] ++ pca(3) ++ volume ++ sorted extent(3) ++ descriptor (roman_registration.py:98-108)."""
    pr = synth.make_pair(4, 4, 6, 3)
    o = pr.map1[0]
    row = SubmapAlignParams(method="roman", semantics_dim=6).get_object_registration()._object_to_clipper_list(o)
    assert row == o.center.ravel().tolist() + [o.linearity, o.planarity, o.scattering, o.volume] + list(o.semantic_descriptor)
    row = SubmapAlignParams(method="sevg", semantics_dim=6).get_object_registration()._object_to_clipper_list(o)
    assert row == o.center.ravel().tolist() + [o.volume] + sorted(o.extent) + list(o.semantic_descriptor)
    assert SubmapAlignParams(method="clipper", dim=2).get_object_registration()._object_to_clipper_list(o) == o.center.ravel().tolist()[:2]


def test_empty_map_and_exception_protocol():
    reg = SubmapAlignParams(method="clipper").get_object_registration()
    pr = synth.make_pair(5, 5, 0, 1)
    out = reg.register([], pr.map2)                    # object_registration.py:23-24
    assert out.shape == (1, 0) and out.dtype == np.float64 and len(out) == 1
    with pytest.raises(InsufficientAssociationsException) as ei:
        reg.T_align([], pr.map2)                       # :102-103
    assert (ei.value.map1_len, ei.value.map2_len, ei.value.n_associations) == (0, 5, None)
    with pytest.raises(InsufficientAssociationsException) as ei:
        reg.T_align(pr.map1, pr.map2, np.array([[0, 0], [1, 1]]))     # :107-108, k < dim
    assert ei.value.n_associations == 2 and "Insufficient associations" in str(ei.value)
    with pytest.raises(InsufficientAssociationsException):
        reg.T_align(pr.map1, pr.map2, reg.register([], pr.map2))     # (1,0) array has len 1 < dim
    assert issubclass(GravityConstraintError, Exception) and ObjectRegistration(dim=2).dim == 2


def test_fully_pruned_list_means_all_to_all():
    """Every association pruned -> the reference hands clipperpy an empty A, which it replaces by the all-to-all
    list ([REF roman/align/dist_reg_with_pruning.py:94-96]); the mirror maps the empty list to None likewise
    (pinned by the last register_golden case, generated through the reference's own class)."""
    reg = DistRegWithPruning(0.4, 0.6, 0.2, shape_epsilon=0.0, cos_min=0.9999, use_gravity=True)
    pr = synth.make_pair(12, 10, 16, 5)
    assert reg._associations_to_score(pr.map1, pr.map2).shape == (0, 2)
    assert reg._association_list(pr.map1, pr.map2) is None
    from roman_amd.align.batch import batch_from_pairs
    assert batch_from_pairs(reg, [(pr.map1, pr.map2)]).assoc is None


def test_prune_prefilter_matches_formula():
    reg = DistRegWithPruning(0.4, 0.6, 0.2, shape_epsilon=0.3, cos_min=0.5, use_gravity=True)
    pr = synth.make_pair(12, 10, 16, 5)
    A = reg._associations_to_score(pr.map1, pr.map2)
    keep = set(map(tuple, A.tolist()))
    for i, a in enumerate(pr.map1):
        for j, b in enumerate(pr.map2):
            cos_ok = not (float(a.semantic_descriptor @ b.semantic_descriptor) < 0.5)
            sa = np.array([a.volume, a.linearity, a.planarity, a.scattering]); sb = np.array([b.volume, b.linearity, b.planarity, b.scattering])
            ratio_ok = not np.any(np.minimum(sa, sb) / np.maximum(sa, sb) < 0.3)
            assert ((i, j) in keep) == (cos_ok and ratio_ok)
    assert np.all(np.diff(A[:, 0] * 10 + A[:, 1]) > 0)           # order of create_all_to_all preserved


def test_clipperpy_shim_surface():
    shim = roman_amd.install_clipperpy_shim(force=True)
    import clipperpy
    assert clipperpy is shim and sys.modules["clipperpy.invariants"] is shim.invariants
    p = clipperpy.Params()
    assert (p.tol_u, p.tol_F, p.maxiniters, p.maxoliters, p.beta, p.maxlsiters, p.eps, p.affinityeps, p.rescale_u0) == \
        (1e-8, 1e-9, 200, 1000, 0.25, 99, 1e-9, 1e-4, True)
    ip = clipperpy.invariants.ROMANParams()
    for attr in ["point_dim", "ratio_feature_dim", "cos_feature_dim", "sigma", "epsilon", "mindist", "distance_weight",
                 "ratio_weight", "cosine_weight", "ratio_epsilon", "cosine_min", "cosine_max", "gravity_guided",
                 "drift_aware", "gravity_unc_ang_rad"]:
        assert hasattr(ip, attr)
    R = clipperpy.invariants.ROMAN
    assert len({R.GEOMETRIC_MEAN, R.ARITHMETIC_MEAN, R.PRODUCT}) == 3
    ep = clipperpy.invariants.EuclideanDistanceParams(); ep.sigma, ep.epsilon, ep.mindist = 0.4, 0.6, 0.2
    c = clipperpy.CLIPPER(clipperpy.invariants.EuclideanDistance(ep), clipperpy.Params())
    for m in ["score_pairwise_consistency", "solve", "get_selected_associations", "get_affinity_matrix",
              "get_constraint_matrix", "set_matrix_data", "get_solution"]:
        assert callable(getattr(c, m))
    cs = clipperpy.CLIPPERPairwiseAndSingle(R(ip), clipperpy.Params())
    assert callable(cs.score_pairwise_and_single_consistency) and isinstance(cs, clipperpy.CLIPPER)
    clipperpy.invariants.PairwiseInvariant()
    ab = cs._abi_params()
    assert ab.invariant == _abi.ROMAN_INV_ROMAN and ab.maxlsiters == 99
    ip.ratio_feature_dim = 2; ip.ratio_epsilon = np.ones(3)
    with pytest.raises(ValueError):
        R(ip)._to_abi()
    with pytest.raises(RuntimeError):
        c.solve()


def test_synthetic_generator_is_deterministic_and_well_formed():
    a = synth.make_pair(30, 25, 8, 77); b = synth.make_pair(30, 25, 8, 77)
    assert all(np.array_equal(x.center, y.center) for x, y in zip(a.map1 + a.map2, b.map1 + b.map2))
    assert len(a.map1) == 30 and len(a.map2) == 25 and a.inliers.shape == (12, 2)
    c1 = np.array([o.center.ravel() for o in a.map1])
    dist = np.linalg.norm(c1[:, None] - c1[None], axis=2) + 10 * np.eye(30)
    assert dist.min() >= 0.5
    for i, j in a.inliers:                       # T_gt maps map-2 inliers onto their map-1 twins (noise 0.1 m)
        p = a.T_gt[:3, :3] @ a.map2[j].center.ravel() + a.T_gt[:3, 3]
        assert np.linalg.norm(p - a.map1[i].center.ravel()) < 0.6
    assert abs(np.linalg.norm(a.map1[0].semantic_descriptor) - 1.0) < 1e-12
    subs, poses = synth.make_submap_grid(3, n=20, d=4, seed0=5)
    assert len(subs) == 3 and all(len(s) == 20 for s in subs) and poses[0].shape == (4, 4)


def test_vectorised_pack_equals_the_per_object_lists():
    """ROMANRegistration.pack assembles column blocks; it must be the very matrix the reference builds row by row."""
    from conftest import registration_for
    from roman_amd.align.object_registration import ObjectRegistration
    for method, kw, d in [("semanticgrav", {"semantics_dim": 40}, 40), ("roman", {"semantics_dim": 24}, 24),
                          ("sevg", {"semantics_dim": 16}, 16), ("pcavolgrav", {}, 0), ("clipper", {"dim": 2}, 0)]:
        reg = registration_for(method, **kw)
        subs, _ = synth.make_submap_grid(3, n=17, d=d, seed0=77)
        if kw.get("dim") == 2:
            for sm in subs:
                for o in sm:
                    o.centroid = o.centroid[:2]; o.dim = 2
        for sm in subs + [[]]:
            a, b = reg.pack(sm), ObjectRegistration.pack(reg, sm)
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b) and a.flags.c_contiguous


def test_device_prefilter_needs_the_descriptor_length_before_an_empty_map():
    """An empty map packed before the descriptor length is known would get rows of another width than the maps packed after it
    (one pool per batch): refused; with semantics_dim given the widths agree."""
    reg = DistRegWithPruning(0.4, 0.6, 0.2, cos_min=0.5, use_gravity=True, prune_on_device=True)
    with pytest.raises(ValueError, match="semantics_dim"):
        reg.pack([])
    pr = synth.make_pair(6, 5, 16, 9)
    reg = DistRegWithPruning(0.4, 0.6, 0.2, cos_min=0.5, use_gravity=True, prune_on_device=True, semantics_dim=16)
    assert reg.pack([]).shape == (0, 3 + 4 + 16) and reg.pack(pr.map1).shape == (6, 3 + 4 + 16)
    assert reg._abi_params().cos_feature_dim == 16
