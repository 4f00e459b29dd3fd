"""DistRegWithPruning — `method='clipper+prune'`: prune associations by semantic cosine and shape
ratios, then plain Euclidean-distance CLIPPER; reject results with roll/pitch above a threshold.
Mirrors [REF roman/align/dist_reg_with_pruning.py:12-97]."""
from typing import List

import numpy as np

from .. import _abi
from ..clipperpy.utils import create_all_to_all
from .object_registration import ObjectRegistration


class GravityConstraintError(Exception):
    """[REF roman/align/dist_reg_with_pruning.py:12-13]"""
    pass


def _zyx_euler(R):
    """(yaw, pitch, roll) of R = Rz(yaw) Ry(pitch) Rx(roll) — what
    scipy `Rot.from_matrix(R).as_euler('ZYX')` returns at [REF dist_reg_with_pruning.py:41]."""
    pitch = -np.arcsin(np.clip(R[2, 0], -1.0, 1.0))
    yaw = np.arctan2(R[1, 0], R[0, 0])
    roll = np.arctan2(R[2, 1], R[2, 2])
    return yaw, pitch, roll


class DistRegWithPruning(ObjectRegistration):
    """The reference's class ([REF roman/align/dist_reg_with_pruning.py:15-97]).  With `prune_on_device=True` the descriptor
    length is part of the feature row: pass `semantics_dim` when the FIRST map packed may be empty (a sequence whose first
    submap holds no objects is legitimate input — register() returns the empty association set for it without packing
    anything; only the batched entries pack every map of a pool, and an empty one cannot tell them its row width)."""

    def __init__(self, sigma, epsilon, mindist=0.0, shape_epsilon=0.0, cos_min=0.85,
                 dim=3, use_gravity=False, roll_pitch_thresh=np.deg2rad(5), prune_on_device=False, semantics_dim=None):
        """The reference's arguments ([REF roman/align/dist_reg_with_pruning.py:17-27]) plus two of this package's:
        prune_on_device — evaluate the prefilter on the GPU instead of in NumPy (ROMAN_INV_EUCLIDEAN_PRUNED: the descriptors
        and shape attributes travel with the centroids, the pruned list never exists on the host);
        semantics_dim — descriptor length for that mode (default: taken from the first object packed)."""
        super().__init__(dim)
        self.prune_on_device = bool(prune_on_device)
        self.semantics_dim = semantics_dim
        self.sigma = sigma
        self.epsilon = epsilon
        self.mindist = mindist
        self.shape_epsilon = shape_epsilon
        self.cos_min = cos_min
        self.use_gravity = use_gravity
        self.roll_pitch_thresh = roll_pitch_thresh
        assert not self.use_gravity or self.dim == 3, "Gravity can only be used with 3D points"

    def _abi_params(self):
        """clipperpy.invariants.EuclideanDistance + default clipperpy.Params,
        [REF roman/align/dist_reg_with_pruning.py:48-57]."""
        p = _abi.RomanParams.default()
        p.invariant = _abi.ROMAN_INV_EUCLIDEAN
        p.point_dim = self.dim
        p.sigma, p.epsilon, p.mindist = self.sigma, self.epsilon, self.mindist
        if self.prune_on_device:                              # the prefilter's thresholds ride in the ROMAN-invariant fields
            if self.semantics_dim is None:
                raise ValueError("prune_on_device needs semantics_dim (or pack a map first)")
            p.invariant = _abi.ROMAN_INV_EUCLIDEAN_PRUNED
            p.ratio_feature_dim = 4
            p.cos_feature_dim = int(self.semantics_dim)
            p.cosine_min = self.cos_min
            for f in range(4):
                p.ratio_epsilon[f] = self.shape_epsilon
        return p

    def _object_to_clipper_list(self, object):
        return object.center.reshape(-1)[:self.dim].tolist()

    def pack(self, object_map):
        """Host prefilter: centroids only, as the reference hands them over ([REF :92-96]).  Device prefilter: one row per
        object [centroid | volume, linearity, planarity, scattering | descriptor] — the inputs of [REF :75-90]."""
        if not self.prune_on_device:
            return super().pack(object_map)
        if self.semantics_dim is None and len(object_map):
            self.semantics_dim = int(np.asarray(object_map[0].semantic_descriptor).size)
        if self.semantics_dim is None:
            # an empty map packed before the descriptor length is known would get rows of another width than the maps that
            # follow it in the same pool
            raise ValueError("prune_on_device: pass semantics_dim=... (or pack a non-empty map first) before packing an empty map")
        F = self.dim + 4 + int(self.semantics_dim)
        if len(object_map) == 0:
            return np.zeros((0, F), dtype=np.float64)
        return np.array([np.concatenate([o.center.reshape(-1)[:self.dim], self._object_shape_attributes(o),
                                         np.asarray(o.semantic_descriptor, dtype=np.float64).flatten()]) for o in object_map], dtype=np.float64)

    def _object_shape_attributes(self, object):
        return np.array([object.volume, object.linearity, object.planarity, object.scattering])

    def _associations_to_score(self, map1, map2):
        """The NumPy prefilter of [REF roman/align/dist_reg_with_pruning.py:71-90] (a9): cosine of the
        raw descriptors below cos_min, or any min/max shape ratio below shape_epsilon, removes an
        association.  Host-side index logic; the pruned list goes to the device as-is."""
        if self.prune_on_device:
            return None                                       # all-to-all: the device applies the prefilter (ROMAN_INV_EUCLIDEAN_PRUNED)
        A_all = create_all_to_all(len(map1), len(map2))
        descriptors1 = np.array([p.semantic_descriptor.flatten() for p in map1])
        descriptors2 = np.array([p.semantic_descriptor.flatten() for p in map2])
        semantic_cos_sim = descriptors1 @ descriptors2.T
        keep = ~(semantic_cos_sim < self.cos_min)[A_all[:, 0], A_all[:, 1]]
        A_put = A_all[keep]
        s1 = np.array([self._object_shape_attributes(o) for o in map1])[A_put[:, 0], :]
        s2 = np.array([self._object_shape_attributes(o) for o in map2])[A_put[:, 1], :]
        both = np.stack([s1, s2], axis=2)
        violates = (np.min(both, axis=2) / np.max(both, axis=2)) < self.shape_epsilon
        return np.ascontiguousarray(A_put[~np.any(violates, axis=1)], dtype=np.int32)

    def register(self, map1: List, map2: List):
        """[REF roman/align/dist_reg_with_pruning.py:29-46]"""
        if len(map1) == 0 or len(map2) == 0:
            return np.array([[]])
        Ain = super().register(map1, map2)
        if self.use_gravity:
            T_align = self.T_align(map1, map2, Ain)
            yaw, pitch, roll = _zyx_euler(T_align[:self.dim, :self.dim])
            if not (np.abs(roll) < self.roll_pitch_thresh and np.abs(pitch) < self.roll_pitch_thresh):
                raise GravityConstraintError(f"Roll and pitch must be less than {self.roll_pitch_thresh} rad")
        return Ain
