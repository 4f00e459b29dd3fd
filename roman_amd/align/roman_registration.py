"""ROMANRegistration — the default `method='roman'` plugin.
Mirrors [REF roman/align/roman_registration.py:10-114]."""
from dataclasses import dataclass
from enum import Enum
from typing import List

import numpy as np

from .. import _abi
from .object_registration import ObjectRegistration


class FusionMethod(Enum):
    """[REF roman/align/roman_registration.py:10-14]"""
    GEOMETRIC_MEAN = _abi.ROMAN_FUSE_GEOMETRIC_MEAN
    ARITHMETIC_MEAN = _abi.ROMAN_FUSE_ARITHMETIC_MEAN
    PRODUCT = _abi.ROMAN_FUSE_PRODUCT


@dataclass
class ROMANParams:
    """Field for field [REF roman/align/roman_registration.py:16-35] (fusion_method is a plain class
    attribute there too, and — as in the reference — never reaches the invariant: Appendix D)."""
    point_dim: int = 3
    fusion_method = FusionMethod.GEOMETRIC_MEAN

    sigma: float = 0.4
    epsilon: float = 0.6
    mindist: float = 0.2

    gravity: bool = False
    volume: bool = False
    pca: bool = False
    extent: bool = False
    semantics_dim: int = 0
    gravity_unc_ang_rad: float = 0.0872665

    cos_min: float = 0.85
    cos_max: float = 1.0
    epsilon_shape: float = None


class ROMANRegistration(ObjectRegistration):

    def __init__(self, params: ROMANParams):
        super().__init__(dim=params.point_dim)
        self.volume = params.volume
        self.extent = params.extent
        self.pca = params.pca
        self.semantics = params.semantics_dim > 0

        ratio_feature_dim = 0                       # [REF roman_registration.py:42-53]
        if self.pca:
            ratio_feature_dim += 3
        if self.volume:
            ratio_feature_dim += 1
        if self.extent:
            ratio_feature_dim += 3

        p = _abi.RomanParams.default()              # [REF roman_registration.py:55-78]
        p.invariant = _abi.ROMAN_INV_ROMAN
        p.point_dim = params.point_dim
        p.ratio_feature_dim = ratio_feature_dim
        p.cos_feature_dim = params.semantics_dim
        p.sigma, p.epsilon, p.mindist = params.sigma, params.epsilon, params.mindist
        p.distance_weight = p.ratio_weight = p.cosine_weight = 1.0
        eps_shape = 0.0 if params.epsilon_shape is None else params.epsilon_shape
        for f in range(ratio_feature_dim):
            p.ratio_epsilon[f] = eps_shape
        p.cosine_min, p.cosine_max = params.cos_min, params.cos_max
        p.gravity_guided = int(bool(params.gravity))
        p.drift_aware = 0
        if params.gravity:
            p.gravity_unc_ang_rad = params.gravity_unc_ang_rad
        p.fusion_method = _abi.ROMAN_FUSE_GEOMETRIC_MEAN   # the reference never forwards fusion_method
        self.iparams = p

    def _abi_params(self):
        return self.iparams

    def _object_to_clipper_list(self, object):
        """Feature row [x y (z)] ++ pca(3) ++ volume(1) ++ sorted extent(3) ++ descriptor,
        [REF roman/align/roman_registration.py:98-108]."""
        object_as_list = object.center.reshape(-1).tolist()[:self.dim]
        if self.pca:
            object_as_list += [object.linearity, object.planarity, object.scattering]
        if self.volume:
            object_as_list.append(object.volume)
        if self.extent:
            object_as_list += sorted(object.extent)
        if self.semantics:
            object_as_list += np.array(object.semantic_descriptor).tolist()
        return object_as_list

    def pack(self, object_map):
        """Same (n, F) matrix as the per-object lists above, assembled column block by column block (the list form
        costs ~60 us per object with a 512-d descriptor: 1.5 s for the 128 submaps of a 64 x 64 grid, 30x the device
        time of its 4096 alignments)."""
        n = len(object_map)
        if n == 0:
            return super().pack(object_map)
        try:
            blocks = [np.stack([np.asarray(o.center, dtype=np.float64).reshape(-1)[:self.dim] for o in object_map])]
            if self.pca:
                blocks.append(np.array([[o.linearity, o.planarity, o.scattering] for o in object_map], dtype=np.float64))
            if self.volume:
                blocks.append(np.array([[o.volume] for o in object_map], dtype=np.float64))
            if self.extent:
                blocks.append(np.array([sorted(o.extent) for o in object_map], dtype=np.float64))
            if self.semantics:
                blocks.append(np.stack([np.asarray(o.semantic_descriptor, dtype=np.float64).reshape(-1) for o in object_map]))
            out = np.ascontiguousarray(np.hstack(blocks))
            if out.shape != (n, self._abi_params().feature_dim()):
                raise ValueError("feature width")
            return out
        except (ValueError, TypeError):
            return super().pack(object_map)          # ragged input: let the generic path raise what it always raised

    def _check_clipper_arrays(self, map1_cl, map2_cl):
        assert map1_cl.shape[1] == map2_cl.shape[1]
