"""TEST-ONLY `clipperpy` stand-in backed by the CPU oracle (oracle/).

Lets the reference's UNMODIFIED Python files (/root/reference/roman/align/*.py,
/root/reference/roman/params/submap_align_params.py) run in this container, where the real
clipperpy cannot be built, so that golden fixtures can be generated through the reference's own
feature packing / factory / pruning code (tests/golden/make_golden.py).  Never imported by roman_amd.
"""
import sys
import types

import numpy as np

from oracle import oracle as orc
from roman_amd import _abi


class Params:
    def __init__(self):
        self.tol_u = 1e-8; self.tol_F = 1e-9; self.tol_Fop = 1e-10
        self.maxiniters = 200; self.maxoliters = 1000
        self.beta = 0.25; self.maxlsiters = 99; self.eps = 1e-9; self.affinityeps = 1e-4
        self.rescale_u0 = True


class _Solution:
    def __init__(self, nodes, u, score):
        self.nodes = nodes; self.u = u; self.score = score


class _EuclideanDistanceParams:
    def __init__(self):
        self.sigma = 0.01; self.epsilon = 0.06; self.mindist = 0.0


class _PairwiseInvariant:
    def _abi(self):
        p = orc.default_params(); p.invariant = _abi.ROMAN_INV_EUCLIDEAN
        return p


class _EuclideanDistance(_PairwiseInvariant):
    def __init__(self, params):
        self.params = params

    def _abi(self):
        p = orc.default_params(); p.invariant = _abi.ROMAN_INV_EUCLIDEAN
        p.sigma, p.epsilon, p.mindist = self.params.sigma, self.params.epsilon, self.params.mindist
        return p


class _ROMANParams:
    def __init__(self):
        self.point_dim = 3; self.ratio_feature_dim = 0; self.cos_feature_dim = 0
        self.sigma = 0.4; self.epsilon = 0.6; self.mindist = 0.2
        self.distance_weight = self.ratio_weight = self.cosine_weight = 1.0
        self.ratio_epsilon = np.zeros(0); self.cosine_min = 0.85; self.cosine_max = 1.0
        self.gravity_guided = False; self.drift_aware = False; self.gravity_unc_ang_rad = 0.0


class _ROMAN(_PairwiseInvariant):
    GEOMETRIC_MEAN = 0; ARITHMETIC_MEAN = 1; PRODUCT = 2

    def __init__(self, params):
        self.params = params

    def _abi(self):
        ip = self.params
        p = orc.default_params(); p.invariant = _abi.ROMAN_INV_ROMAN
        p.point_dim = ip.point_dim; p.ratio_feature_dim = ip.ratio_feature_dim; p.cos_feature_dim = ip.cos_feature_dim
        p.sigma, p.epsilon, p.mindist = ip.sigma, ip.epsilon, ip.mindist
        p.distance_weight, p.ratio_weight, p.cosine_weight = ip.distance_weight, ip.ratio_weight, ip.cosine_weight
        re = np.asarray(ip.ratio_epsilon, dtype=np.float64).ravel()
        for f in range(ip.ratio_feature_dim):
            p.ratio_epsilon[f] = re[f] if re.size else 0.0
        p.cosine_min, p.cosine_max = ip.cosine_min, ip.cosine_max
        p.gravity_guided = int(bool(ip.gravity_guided)); p.drift_aware = int(bool(ip.drift_aware))
        p.gravity_unc_ang_rad = ip.gravity_unc_ang_rad
        return p


LAST = {"A": None}     # association list handed to the most recent scoring call (golden generation)


class CLIPPER:
    def __init__(self, invariant, params):
        self.invariant = invariant; self.params = params
        self._mat = None; self._A = None; self._sol = None

    def _p(self):
        p = self.invariant._abi(); sp = self.params
        p.tol_u, p.tol_F, p.beta, p.eps, p.affinityeps = sp.tol_u, sp.tol_F, sp.beta, sp.eps, sp.affinityeps
        p.maxiniters, p.maxoliters, p.maxlsiters = sp.maxiniters, sp.maxoliters, sp.maxlsiters
        p.rescale_u0 = int(bool(sp.rescale_u0))
        return p

    def _score(self, D1, D2, A):
        D1 = np.ascontiguousarray(np.asarray(D1, dtype=np.float64).T)
        D2 = np.ascontiguousarray(np.asarray(D2, dtype=np.float64).T)
        A = None if (A is None or np.asarray(A).size == 0) else np.asarray(A, dtype=np.int32)
        self._mat, self._A = orc.build_matrix(self._p(), D1, D2, A)
        LAST["A"] = self._A.copy()

    def score_pairwise_consistency(self, D1, D2, A=None):
        self._score(D1, D2, A)

    def set_matrix_data(self, M, C):
        self._mat = orc.matrix_from_dense(M, C); self._A = None

    def solve(self, u0=None):
        self._sol = orc.solve(self._p(), self._mat, u0)

    def get_selected_associations(self):
        return self._A[self._sol["nodes"]]

    def get_solution(self):
        return _Solution(self._sol["nodes"], self._sol["u"], self._sol["stats"].score)

    def get_affinity_matrix(self):
        return self._mat.dense()[0]

    def get_constraint_matrix(self):
        return self._mat.dense()[1]


class CLIPPERPairwiseAndSingle(CLIPPER):
    def score_pairwise_and_single_consistency(self, D1, D2, A=None):
        self._score(D1, D2, A)


def install():
    """Register this module tree as `clipperpy` (+ stubs for the reference's other absent imports)."""
    me = types.ModuleType("clipperpy")
    me.Params = Params; me.CLIPPER = CLIPPER; me.CLIPPERPairwiseAndSingle = CLIPPERPairwiseAndSingle
    inv = types.ModuleType("clipperpy.invariants")
    inv.ROMANParams = _ROMANParams; inv.ROMAN = _ROMAN
    inv.EuclideanDistanceParams = _EuclideanDistanceParams; inv.EuclideanDistance = _EuclideanDistance
    inv.PairwiseInvariant = _PairwiseInvariant
    ut = types.ModuleType("clipperpy.utils")
    ut.create_all_to_all = lambda n1, n2: orc.create_all_to_all(int(n1), int(n2))
    me.invariants = inv; me.utils = ut
    sys.modules["clipperpy"] = me; sys.modules["clipperpy.invariants"] = inv; sys.modules["clipperpy.utils"] = ut
    return me
