"""clipperpy.invariants — parameter holders for the invariants the reference instantiates."""
import numpy as np

from .. import _abi


class PairwiseInvariant:
    """Default-constructible base ([REF roman/align/object_registration.py:60]); used together with
    set_matrix_data, where no scoring happens."""

    def _to_abi(self):
        p = _abi.RomanParams.default()
        p.invariant = _abi.ROMAN_INV_EUCLIDEAN
        return p


class EuclideanDistanceParams:
    """[REF roman/align/dist_reg_with_pruning.py:49-52]"""

    def __init__(self):
        self.sigma = 0.01
        self.epsilon = 0.06
        self.mindist = 0.0


class EuclideanDistance(PairwiseInvariant):
    """[REF roman/align/dist_reg_with_pruning.py:54]"""

    def __init__(self, params):
        self.params = params

    def _to_abi(self):
        p = _abi.RomanParams.default()
        p.invariant = _abi.ROMAN_INV_EUCLIDEAN
        p.sigma, p.epsilon, p.mindist = float(self.params.sigma), float(self.params.epsilon), float(self.params.mindist)
        p.point_dim = int(getattr(self.params, "point_dim", 3))
        return p


class ROMANParams:
    """Attributes set at [REF roman/align/roman_registration.py:55-78]."""

    def __init__(self):
        self.point_dim = 3
        self.ratio_feature_dim = 0
        self.cos_feature_dim = 0
        self.sigma = 0.4
        self.epsilon = 0.6
        self.mindist = 0.2
        self.distance_weight = 1.0
        self.ratio_weight = 1.0
        self.cosine_weight = 1.0
        self.ratio_epsilon = np.zeros(0)
        self.cosine_min = 0.85
        self.cosine_max = 1.0
        self.gravity_guided = False
        self.drift_aware = False
        self.gravity_unc_ang_rad = 0.0
        self.fusion_method = 0
        # NOT upstream attributes: switches of the two formulas pinned by decision (include/roman_hip.h
        # ROMAN_GRAV_*, ROMAN_SINGLE_*); the reference never sets them, 0 = the pinned default
        self.gravity_mode = 0
        self.single_mode = 0


class ROMAN(PairwiseInvariant):
    """clipperpy.invariants.ROMAN(iparams) ([REF roman/align/roman_registration.py:83]); the enum
    values are only compared/stored by the reference ([REF roman/align/roman_registration.py:11-14],
    [REF roman/params/submap_align_params.py:87-92])."""
    GEOMETRIC_MEAN = _abi.ROMAN_FUSE_GEOMETRIC_MEAN
    ARITHMETIC_MEAN = _abi.ROMAN_FUSE_ARITHMETIC_MEAN
    PRODUCT = _abi.ROMAN_FUSE_PRODUCT

    def __init__(self, params):
        self.params = params

    def _to_abi(self):
        ip = self.params
        p = _abi.RomanParams.default()
        p.invariant = _abi.ROMAN_INV_ROMAN
        p.point_dim = int(ip.point_dim)
        p.ratio_feature_dim = int(ip.ratio_feature_dim)
        p.cos_feature_dim = int(ip.cos_feature_dim)
        p.sigma, p.epsilon, p.mindist = float(ip.sigma), float(ip.epsilon), float(ip.mindist)
        p.distance_weight, p.ratio_weight, p.cosine_weight = float(ip.distance_weight), float(ip.ratio_weight), float(ip.cosine_weight)
        re = np.asarray(ip.ratio_epsilon, dtype=np.float64).ravel()
        if p.ratio_feature_dim > _abi.ROMAN_MAX_RATIO_FEATURES:
            raise ValueError(f"ratio_feature_dim > {_abi.ROMAN_MAX_RATIO_FEATURES}")
        if re.size not in (0, p.ratio_feature_dim):
            raise ValueError("ratio_epsilon must have ratio_feature_dim entries")
        for f in range(p.ratio_feature_dim):
            p.ratio_epsilon[f] = float(re[f]) if re.size else 0.0
        p.cosine_min, p.cosine_max = float(ip.cosine_min), float(ip.cosine_max)
        p.gravity_guided = int(bool(ip.gravity_guided))
        p.drift_aware = int(bool(ip.drift_aware))
        p.gravity_unc_ang_rad = float(ip.gravity_unc_ang_rad)
        p.fusion_method = int(getattr(ip, "fusion_method", 0))
        p.gravity_mode = int(getattr(ip, "gravity_mode", 0))
        p.single_mode = int(getattr(ip, "single_mode", 0))
        return p
