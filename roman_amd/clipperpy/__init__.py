"""`clipperpy`-compatible surface backed by libroman_hip.so.

Exactly the subset of mit-acl/clipper's pybind11 module that mit-acl/roman uses (SURVEY.md
Appendix A).  Each class cites the reference call sites it serves.  Importing this module does not
touch the GPU; methods that compute do (and raise RomanHipError without one).
"""
import numpy as np

from .. import _abi
from . import invariants, utils  # noqa: F401


class Params:
    """clipperpy.Params — always default-constructed by the reference
    ([REF roman/align/roman_registration.py:84], [REF roman/align/dist_reg_with_pruning.py:55],
    [REF roman/align/object_registration.py:60])."""

    def __init__(self):
        self.tol_u = 1e-8
        self.tol_F = 1e-9
        self.tol_Fop = 1e-10        # kept for attribute compatibility; unused by the solver
        self.maxiniters = 200
        self.maxoliters = 1000
        self.beta = 0.25
        self.maxlsiters = 99
        self.eps = 1e-9
        self.affinityeps = 1e-4
        self.rescale_u0 = True


class Solution:
    """clipper.get_solution() -> .nodes, .u, .score ([REF roman/align/object_registration.py:67-71])."""

    def __init__(self, nodes, u, score, u0=None, ifinal=0, t=0.0):
        self.nodes = nodes
        self.u = u
        self.score = score
        self.u0 = u0
        self.ifinal = ifinal
        self.t = t


def _object_major(D):
    """The reference passes `map_cl.T`: an (F, n) float64 view, one column per object
    ([REF roman/align/roman_registration.py:91-95]).  Return the (n, F) C-contiguous array."""
    D = np.asarray(D, dtype=np.float64)
    if D.ndim != 2:
        raise ValueError("data must be 2-D (features x objects)")
    return np.ascontiguousarray(D.T)


class CLIPPER:
    """clipperpy.CLIPPER(invariant, params)
    ([REF roman/align/dist_reg_with_pruning.py:56], [REF roman/align/object_registration.py:60])."""

    def __init__(self, invariant, params):
        self.invariant = invariant
        self.params = params
        self._ctx = None
        self._A = None
        self._scored = False
        self._u0 = None
        self._inputs = None          # what was scored: re-sent if another object used the context since
        self._solution = None        # cached at solve(): upstream keeps results per CLIPPER object
        self._selected = None
        self._gen = None             # Context._generation right after this object's inputs were loaded

    # -- plumbing ---------------------------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            from ..runtime import default_context
            self._ctx = default_context()
        return self._ctx

    def _abi_params(self):
        p = self.invariant._to_abi()
        sp = self.params
        p.tol_u, p.tol_F, p.beta, p.eps, p.affinityeps = sp.tol_u, sp.tol_F, sp.beta, sp.eps, sp.affinityeps
        p.maxiniters, p.maxoliters, p.maxlsiters = int(sp.maxiniters), int(sp.maxoliters), int(sp.maxlsiters)
        p.rescale_u0 = int(bool(sp.rescale_u0))
        return p

    def _score(self, D1, D2, A):
        D1, D2 = _object_major(D1), _object_major(D2)
        A = None if A is None else np.asarray(A)
        if A is not None and A.size == 0:
            A = None                     # upstream: an empty A means all-to-all
        if A is None:
            self._A = utils.create_all_to_all(D1.shape[0], D2.shape[0])
            self._inputs = ("score", D1, D2, None)
        else:
            self._A = np.ascontiguousarray(A, dtype=np.int32).reshape(-1, 2)
            self._inputs = ("score", D1, D2, self._A)
        self._solution = self._selected = None
        self._send()
        self._scored = True

    def _send(self):
        """(Re)load this object's matrices into the context: one context holds one problem."""
        ctx = self._context()
        kind = self._inputs[0]
        if kind == "score":
            ctx.score(self._abi_params(), self._inputs[1], self._inputs[2], self._inputs[3])
        else:
            ctx.set_matrix_data(self._abi_params(), self._inputs[1], self._inputs[2])
        self._gen = ctx._generation

    def _own(self):
        """Anything else that used the context since (another CLIPPER object, a registration plugin, a batch
        call) moved its generation on: reload this object's problem before operating on it."""
        if self._context()._generation != self._gen:
            self._send()

    # -- clipperpy API ----------------------------------------------------------------------------
    def score_pairwise_consistency(self, D1, D2, A=None):
        """[REF roman/align/object_registration.py:47], [REF roman/align/dist_reg_with_pruning.py:96]"""
        self._score(D1, D2, A)

    def set_matrix_data(self, M, C):
        """[REF roman/align/object_registration.py:64]"""
        self._A = None
        self._inputs = ("dense", np.array(M, dtype=np.float64), np.array(C, dtype=np.float64))
        self._solution = self._selected = None
        self._send()
        self._scored = True

    def solve(self, u0=None):
        """[REF roman/align/object_registration.py:27,65].  u0=None -> all ones."""
        if not self._scored:
            raise RuntimeError("solve() called before scoring / set_matrix_data")
        self._u0 = None if u0 is None else np.asarray(u0, dtype=np.float64)
        self._own()
        ctx = self._context()
        ctx.solve(self._u0)
        nodes, u, score, st = ctx.solution()
        self._solution = Solution(nodes, u, score, self._u0, st.outer_iters)
        self._selected = ctx.selected_associations()

    def get_selected_associations(self):
        """[REF roman/align/object_registration.py:28] -> (k,2) int32 rows of A in `nodes` order."""
        if self._selected is None:
            raise RuntimeError("get_selected_associations() before solve()")
        return self._selected.copy()

    def get_solution(self):
        """[REF roman/align/object_registration.py:67-71]"""
        if self._solution is None:
            raise RuntimeError("get_solution() before solve()")
        return self._solution

    def get_affinity_matrix(self):
        """[REF roman/align/object_registration.py:53]"""
        self._own()
        return self._context().dense_matrices()[0]

    def get_constraint_matrix(self):
        """[REF roman/align/object_registration.py:54]"""
        self._own()
        return self._context().dense_matrices()[1]

    def get_initial_associations(self):
        return self._A


class CLIPPERPairwiseAndSingle(CLIPPER):
    """clipperpy.CLIPPERPairwiseAndSingle(invariant, params)
    ([REF roman/align/roman_registration.py:85])."""

    def score_pairwise_and_single_consistency(self, D1, D2, A=None):
        """[REF roman/align/roman_registration.py:95]"""
        self._score(D1, D2, A)
