"""ObjectRegistration plugin base class on the HIP library.

Mirrors [REF roman/align/object_registration.py:17-129]: `register(map1, map2)` returns the (k,2)
inlier association array, `T_align(map1, map2, correspondences=None)` the rigid transform taking
map 2 into map 1.  Everything numerical runs in libroman_hip.so; there is no NumPy fallback.
"""
from typing import List

import numpy as np

from .. import _abi
from ..runtime import default_context


class InsufficientAssociationsException(Exception):
    """Same fields and message as [REF roman/align/object_registration.py:8-15]."""

    def __init__(self, map1_len, map2_len, n_associations=None):
        self.map1_len = map1_len
        self.map2_len = map2_len
        self.n_associations = n_associations
        message = (f"Insufficient associations. Map 1 length: {map1_len}. Map 2 length: {map2_len}. "
                   f"Associations: {n_associations}")
        super().__init__(message)


class ObjectRegistration:
    """Base class: subclasses provide `_abi_params()` (invariant + solver parameters) and
    `_object_to_clipper_list()` (feature row of one object)."""

    def __init__(self, dim=3):
        self.dim = dim
        self._ctx = None

    # ------------------------------------------------------------------ plumbing
    def _context(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def set_context(self, ctx):
        """Use a specific runtime.Context (device/stream) instead of the process default."""
        self._ctx = ctx

    def _abi_params(self) -> _abi.RomanParams:
        raise NotImplementedError

    def _object_to_clipper_list(self, object):
        raise NotImplementedError

    def _check_clipper_arrays(self, map1_cl, map2_cl):
        return

    def pack(self, object_map) -> np.ndarray:
        """(n, F) float64 object-major feature matrix of one map — what the reference builds per
        call at [REF roman/align/object_registration.py:43-44] (and hands over transposed)."""
        F = self._abi_params().feature_dim() if self._abi_params().invariant != _abi.ROMAN_INV_EUCLIDEAN else self.dim
        if len(object_map) == 0:
            return np.zeros((0, F), dtype=np.float64)
        return np.array([self._object_to_clipper_list(p) for p in object_map], dtype=np.float64)

    def _associations_to_score(self, map1, map2):
        """None = all-to-all ([REF roman/align/object_registration.py:41]); subclasses may prune."""
        return None

    def _association_list(self, map1, map2):
        """The list handed to the scorer.  An EMPTY pruned list means all-to-all, as it does for
        clipperpy's `score_pairwise_consistency(D1, D2, A)` (an empty A is replaced by the all-to-all
        list), which is what the reference's pruning plugin ends up calling when every association
        was pruned ([REF roman/align/dist_reg_with_pruning.py:94-96])."""
        A = self._associations_to_score(map1, map2)
        return None if (A is None or len(A) == 0) else A

    # ------------------------------------------------------------------ reference API
    def register(self, map1: List, map2: List):
        """[REF roman/align/object_registration.py:22-29]"""
        if len(map1) == 0 or len(map2) == 0:
            return np.array([[]])                                # (1,0) float64, as the reference
        m1, m2 = self.pack(map1), self.pack(map2)
        self._check_clipper_arrays(m1, m2)
        A = self._association_list(map1, map2)
        ctx = self._context()
        ctx.score(self._abi_params(), m1, m2, A)
        ctx.solve(None)
        return ctx.selected_associations()

    def get_MCA(self, map1: List, map2: List):
        """[REF roman/align/object_registration.py:50-55]: dense M, C and the association list."""
        m1, m2 = self.pack(map1), self.pack(map2)
        A = self._association_list(map1, map2)
        ctx = self._context()
        ctx.score(self._abi_params(), m1, m2, A)
        M, C = ctx.dense_matrices()
        if A is None:
            from ..clipperpy.utils import create_all_to_all
            A = create_all_to_all(len(map1), len(map2))
        return M, C, A

    def mno_clipper(self, map1: List, map2: List, num_solutions=2):
        """Multi-solution extraction, [REF roman/align/object_registration.py:57-86]: solve, record the
        Rayleigh quotient of the selected nodes on the original M, zero their block, repeat."""
        M, C, A = self.get_MCA(map1, map2)
        M_orig = M.copy()
        # a COPY: the reference builds a separate CLIPPER(PairwiseInvariant) for this loop
        # ([REF roman/align/object_registration.py:60]) and never touches the registration's own parameters
        params = _abi.RomanParams.from_buffer_copy(self._abi_params())
        params.invariant = _abi.ROMAN_INV_EUCLIDEAN
        ctx = self._context()
        solutions = []
        for k in range(num_solutions):
            ctx.set_matrix_data(params, M, C)
            ctx.solve(None)
            nodes, u, _, _ = ctx.solution()
            Ain = np.asarray(A)[nodes, :].astype(np.int64).reshape(len(nodes), 2)
            u_sol = np.zeros_like(u)
            u_sol[nodes] = u[nodes]
            score = 0 if len(nodes) == 0 else u_sol.T @ M_orig @ u_sol / (u_sol.T @ u_sol)
            solutions.append((Ain.copy(), score))
            if k + 1 < num_solutions and len(nodes) != 0:
                M[np.ix_(nodes, nodes)] = 0.0
        return solutions

    def T_align(self, map1: List, map2: List, correspondences: np.array = None):
        """Transformation that aligns map2 to map1 (Arun's method),
        [REF roman/align/object_registration.py:88-129]."""
        if len(map1) == 0 or len(map2) == 0:
            raise InsufficientAssociationsException(len(map1), len(map2))
        if correspondences is None:
            correspondences = self.register(map1, map2)
        if len(correspondences) < self.dim:
            raise InsufficientAssociationsException(len(map1), len(map2), len(correspondences))
        pts1 = np.array([map1[corr[0]].center.reshape(-1)[:self.dim] for corr in correspondences], dtype=np.float64)
        pts2 = np.array([map2[corr[1]].center.reshape(-1)[:self.dim] for corr in correspondences], dtype=np.float64)
        T, status = self._context().pose_batch(self.dim, pts1, pts2, np.array([0, len(pts1)], dtype=np.int64))
        if status[0] & _abi.ROMAN_ST_INSUFFICIENT:
            raise InsufficientAssociationsException(len(map1), len(map2), len(correspondences))
        return T[0]

    # ------------------------------------------------------------------ batched extension
    def register_and_align_batch(self, pairs, u0=None):
        """All of `register()` + `T_align()` for many (map1, map2) pairs in one device call.
        Returns runtime.BatchResult (assoc list, T, status, stats).  Pairs where T_align would raise
        carry ROMAN_ST_INSUFFICIENT / ROMAN_ST_EMPTY_MAP in `status` and a NaN pose — the sentinel the
        reference's caller writes at [REF roman/align/submap_align.py:179-184]."""
        from .batch import align_pairs
        return align_pairs(self, pairs, u0=u0)
