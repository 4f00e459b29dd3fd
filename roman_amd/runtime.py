"""Thin Python wrapper over the C ABI of libroman_hip.so (include/roman_hip.h).

Everything numerical happens in the HIP library; this module only marshals NumPy arrays (or
device pointers of torch tensors) across ctypes.  There is no CPU fallback: creating a Context
without a gfx950 device raises RomanHipError.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi
from ._abi import RomanHipError, RomanParams, RomanStats


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def stats_dtype():
    return np.dtype([("n_assoc_in", np.int32), ("n_live", np.int32), ("nnz_upper", np.int64),
                     ("n_pass", np.int32), ("outer_iters", np.int32), ("inner_iters", np.int32),
                     ("ls_trials", np.int32), ("score", np.float64), ("d_final", np.float64)])


@dataclass
class BatchResult:
    """Results of one batched call, one entry per problem."""
    assoc: list            # list of (k_b, 2) int32 arrays (map-1 index, map-2 index), clipperpy order
    T: np.ndarray          # (B, dim+1, dim+1) float64, NaN where status has INSUFFICIENT/EMPTY_MAP
    status: np.ndarray     # (B,) int32 ROMAN_ST_* flags
    stats: np.ndarray      # (B,) structured array (stats_dtype)


class Context:
    """One roman_ctx: a HIP device + stream + the library's HBM workspace."""

    def __init__(self, device=0, stream=None):
        self._lib = _abi.load_library()
        self._h = C.c_void_p()
        rc = self._lib.roman_ctx_create(C.byref(self._h), int(device), C.c_void_p(stream) if stream else None)
        if rc != 0:
            msg = self._lib.roman_last_error(None)
            raise RomanHipError(f"roman_ctx_create failed ({rc}): {msg.decode() if msg else ''}")
        self.device = int(device)
        # One context holds ONE stepwise problem (the matrices of the last score()/set_matrix_data()).  Every call
        # that replaces or invalidates it bumps this counter; holders of a problem (the clipperpy shim's CLIPPER
        # objects) remember the value they loaded at and re-send their inputs when it has moved on.
        self._generation = 0
        self.pipeline_depth = 1                 # what set_pipeline() last set (callers that change it restore it)
        self.wide_teams = -1                    # what set_wide_teams() last set
        self.host_batching = (2048, 3)          # what set_host_batching() last set (the library's defaults)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.roman_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pipeline(self, depth):
        """Batches in flight (1 ... 6), see roman_ctx_set_pipeline in include/roman_hip.h.  With depth 2 the
        results of align_batch_dev calls are complete after sync() (or a device-wide synchronise)."""
        self._check(self._lib.roman_ctx_set_pipeline(self._h, int(depth)), "roman_ctx_set_pipeline")
        self.pipeline_depth = int(depth)

    def sync(self):
        self._check(self._lib.roman_ctx_sync(self._h), "roman_ctx_sync")

    def set_host_batching(self, chunk=2048, depth=3):
        """How align_batch() (host pointers) issues a large batch: more than `chunk` problems go to the device as calls of
        `chunk` problems with `depth` of them in flight (roman_ctx_set_host_batching; depth 1 = one call for everything)."""
        self._check(self._lib.roman_ctx_set_host_batching(self._h, int(chunk), int(depth)), "roman_ctx_set_host_batching")
        self.host_batching = (int(chunk), int(depth))

    def set_wide_teams(self, teams_per_xcd=-1):
        """Team mode of the whole-device solver for large live sets (roman_ctx_set_wide_teams): -1 automatic, 0 never, 1 / 2 / 4
        teams per XCD.  A device-pointer caller that gets ROMAN_ST_INTERNAL records back from a batch with several large live
        sets issues those problems again with 0 (pipeline.issue_chunked does)."""
        self._check(self._lib.roman_ctx_set_wide_teams(self._h, int(teams_per_xcd)), "roman_ctx_set_wide_teams")
        self.wide_teams = int(teams_per_xcd)

    def has_history(self, params, F):
        """Does the library hold a sizing history for this parameter block (roman_ctx_has_history)?  Without one the first of
        several queued calls should be waited for, so that the others size their pools from what it needed."""
        yes = C.c_int32(0)
        self._check(self._lib.roman_ctx_has_history(self._h, C.byref(params), int(F), C.byref(yes)), "roman_ctx_has_history")
        return bool(yes.value)

    def cosine_screen_stats(self):
        """(batches whose cosine stage took the bf16 screen + exact candidates, batches that took the dense product, share of the latest
        screened batch left to the dense kernel) — roman_ctx_cosine_screen_stats."""
        a, b, f = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        self._check(self._lib.roman_ctx_cosine_screen_stats(self._h, C.byref(a), C.byref(b), C.byref(f)), "roman_ctx_cosine_screen_stats")
        return int(a.value), int(b.value), float(f.value)

    def join(self, skip_latest=False, stream=None):
        """Make the context's stream — or `stream` (a hipStream_t handle, e.g. torch.cuda.Stream.cuda_stream) — wait for the
        pipelined batches issued so far (optionally all but the latest)."""
        if stream is None:
            self._check(self._lib.roman_ctx_join(self._h, int(bool(skip_latest))), "roman_ctx_join")
        elif int(stream) == 0:
            # roman_ctx_join_on reads a NULL handle as "the context's own stream" — and 0 is also the handle of the legacy default
            # stream (torch.cuda.default_stream().cuda_stream): the waits would land on the wrong stream and a collective queued on
            # the default stream could read records that are not complete
            raise ValueError("join(stream=0): 0 is the legacy default stream's handle, which the C ABI reads as 'the context's stream'; "
                             "queue the collective on an explicit stream (torch.cuda.Stream) and pass its handle")
        else:
            self._check(self._lib.roman_ctx_join_on(self._h, int(bool(skip_latest)), C.c_void_p(int(stream))), "roman_ctx_join_on")

    def skipped(self, wait=True):
        """Running total of problems that batch calls on this context reported with ROMAN_ST_WORKSPACE
        (roman_ctx_skipped); wait=True synchronises the context first so that every issued batch counts."""
        n = C.c_int64(0)
        self._check(self._lib.roman_ctx_skipped(self._h, int(bool(wait)), C.byref(n)), "roman_ctx_skipped")
        return int(n.value)

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.roman_last_error(self._h)
            raise RomanHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # ------------------------------------------------------------------ batched hot path
    def align_batch(self, params, feats, off1, n1, off2, n2, assoc=None, assoc_off=None, u0=None,
                    kmax=None):
        """Host-pointer batch call (roman_align_batch).  feats: (n_objects, F) float64."""
        feats = _f64(feats)
        if feats.ndim != 2:
            raise ValueError("feats must be (n_objects, F)")
        n_obj, F = feats.shape
        off1 = np.ascontiguousarray(off1, dtype=np.int64); off2 = np.ascontiguousarray(off2, dtype=np.int64)
        n1 = np.ascontiguousarray(n1, dtype=np.int32); n2 = np.ascontiguousarray(n2, dtype=np.int32)
        B = int(n1.shape[0])
        if assoc is not None:
            assoc = np.ascontiguousarray(assoc, dtype=np.int32).reshape(-1, 2)
            assoc_off = np.ascontiguousarray(assoc_off, dtype=np.int64)
        if u0 is not None:
            u0 = _f64(u0)
        if kmax is None:
            kmax = int(max(1, np.max(np.minimum(n1, n2)))) if B else 1
        dim = params.point_dim
        a_out = np.zeros((B, kmax, 2), dtype=np.int32)
        n_out = np.zeros(B, dtype=np.int32)
        T = np.zeros((B, 16), dtype=np.float64)
        status = np.zeros(B, dtype=np.int32)
        stats = np.zeros(B, dtype=stats_dtype())
        assert stats.dtype.itemsize == _abi.STATS_NBYTES
        self._generation += 1
        rc = self._lib.roman_align_batch(self._h, C.byref(params), B, _ptr(feats), n_obj, _ptr(off1), _ptr(n1),
                                         _ptr(off2), _ptr(n2), F, _ptr(assoc), _ptr(assoc_off), _ptr(u0), kmax,
                                         _ptr(a_out), _ptr(n_out), _ptr(T), _ptr(status), _ptr(stats))
        s = dim + 1
        if rc == _abi.ROMAN_E_INTERNAL:                          # outputs were copied: the error says which problems have no result
            msg = self._lib.roman_last_error(self._h)
            err = RomanHipError(f"roman_align_batch failed ({rc}): {msg.decode() if msg else ''}")
            err.result = BatchResult([a_out[b, :n_out[b]].copy() for b in range(B)], T[:, :s * s].reshape(B, s, s).copy(), status, stats)
            raise err
        self._check(rc, "roman_align_batch")
        Ts = T[:, :s * s].reshape(B, s, s).copy()
        return BatchResult([a_out[b, :n_out[b]].copy() for b in range(B)], Ts, status, stats)

    def align_batch_resident(self, params, feats_ptr, F, off1, n1, off2, n2, kmax=None, assoc_ptr=None, assoc_off=None, u0_ptr=None):
        """Inputs in HBM (device pointers as integers, e.g. torch.Tensor.data_ptr()), results on the host
        (roman_align_batch_resident): -> BatchResult.  Synchronous; the library chunks, pipelines and retries like align_batch(),
        and the whole result comes back with one copy."""
        off1 = np.ascontiguousarray(off1, dtype=np.int64); off2 = np.ascontiguousarray(off2, dtype=np.int64)
        n1 = np.ascontiguousarray(n1, dtype=np.int32); n2 = np.ascontiguousarray(n2, dtype=np.int32)
        B = int(n1.shape[0])
        if assoc_off is not None:
            assoc_off = np.ascontiguousarray(assoc_off, dtype=np.int64)
        if kmax is None:
            kmax = int(max(1, np.max(np.minimum(n1, n2)))) if B else 1
        a_out = np.empty((B, kmax, 2), dtype=np.int32); n_out = np.empty(B, dtype=np.int32)
        T = np.empty((B, 16), dtype=np.float64); status = np.empty(B, dtype=np.int32); stats = np.empty(B, dtype=stats_dtype())
        vp = lambda x: C.c_void_p(int(x)) if x else None
        self._generation += 1
        rc = self._lib.roman_align_batch_resident(self._h, C.byref(params), B, vp(feats_ptr), _ptr(off1), _ptr(n1), _ptr(off2), _ptr(n2),
                                                  int(F), vp(assoc_ptr), _ptr(assoc_off), vp(u0_ptr), int(kmax),
                                                  _ptr(a_out), _ptr(n_out), _ptr(T), _ptr(status), _ptr(stats))
        s = params.point_dim + 1
        res = lambda: BatchResult([a_out[b, :n_out[b]].copy() for b in range(B)], T[:, :s * s].reshape(B, s, s).copy(), status, stats)
        if rc == _abi.ROMAN_E_INTERNAL:                          # outputs were copied: the error says which problems have no result
            msg = self._lib.roman_last_error(self._h)
            err = RomanHipError(f"roman_align_batch_resident failed ({rc}): {msg.decode() if msg else ''}")
            err.result = res()
            raise err
        self._check(rc, "roman_align_batch_resident")
        return res()

    def align_batch_dev(self, params, feats_ptr, F, off1, n1, off2, n2, kmax, assoc_out_ptr, n_assoc_out_ptr,
                        T_out_ptr, status_out_ptr, stats_out_ptr=None, assoc_ptr=None, assoc_off=None,
                        u0_ptr=None):
        """Device-pointer batch call (roman_align_batch_dev).  Pointers are integers (e.g.
        torch.Tensor.data_ptr()); metadata arrays are host NumPy arrays.  A pure enqueue: nothing is
        waited for or read back; problems that found no workspace come back with ROMAN_ST_WORKSPACE
        (run them again)."""
        off1 = np.ascontiguousarray(off1, dtype=np.int64); off2 = np.ascontiguousarray(off2, dtype=np.int64)
        n1 = np.ascontiguousarray(n1, dtype=np.int32); n2 = np.ascontiguousarray(n2, dtype=np.int32)
        if assoc_off is not None:
            assoc_off = np.ascontiguousarray(assoc_off, dtype=np.int64)
        vp = lambda x: C.c_void_p(int(x)) if x else None
        self._generation += 1
        rc = self._lib.roman_align_batch_dev(self._h, C.byref(params), int(n1.shape[0]), vp(feats_ptr), _ptr(off1),
                                             _ptr(n1), _ptr(off2), _ptr(n2), int(F), vp(assoc_ptr), _ptr(assoc_off),
                                             vp(u0_ptr), int(kmax), vp(assoc_out_ptr), vp(n_assoc_out_ptr),
                                             vp(T_out_ptr), vp(status_out_ptr), vp(stats_out_ptr))
        self._check(rc, "roman_align_batch_dev")

    # ------------------------------------------------------------------ stepwise (clipperpy shim)
    def score(self, params, D1, D2, assoc=None):
        D1, D2 = _f64(D1), _f64(D2)
        n1, n2 = D1.shape[0], D2.shape[0]
        F = D1.shape[1] if D1.ndim == 2 else 0
        if D2.ndim == 2 and D2.shape[1] != F and n1 > 0 and n2 > 0:
            raise ValueError("D1 and D2 must have the same number of features")
        if n1 == 0 and D2.ndim == 2:
            F = D2.shape[1]
        na = 0
        if assoc is not None:
            assoc = np.ascontiguousarray(assoc, dtype=np.int32).reshape(-1, 2)
            na = assoc.shape[0]
        self._generation += 1
        rc = self._lib.roman_score(self._h, C.byref(params), _ptr(D1), n1, _ptr(D2), n2, F, _ptr(assoc), na)
        self._check(rc, "roman_score")

    def set_matrix_data(self, params, M, Cm):
        M, Cm = _f64(M), _f64(Cm)
        if M.shape != Cm.shape or M.ndim != 2 or M.shape[0] != M.shape[1]:
            raise ValueError("M and C must be square matrices of the same shape")
        self._generation += 1
        self._check(self._lib.roman_set_matrix_data(self._h, C.byref(params), _ptr(M), _ptr(Cm), M.shape[0]),
                    "roman_set_matrix_data")

    def solve(self, u0=None):
        u0a = None if u0 is None else _f64(u0)
        self._check(self._lib.roman_solve(self._h, _ptr(u0a)), "roman_solve")

    def num_associations(self):
        n = C.c_int32(0)
        self._check(self._lib.roman_num_associations(self._h, C.byref(n)), "roman_num_associations")
        return n.value

    def selected_associations(self):
        n = C.c_int32(0)
        self._check(self._lib.roman_num_selected(self._h, C.byref(n)), "roman_num_selected")
        out = np.zeros((max(n.value, 1), 2), dtype=np.int32)
        self._check(self._lib.roman_get_selected_associations(self._h, _ptr(out)), "roman_get_selected_associations")
        return out[:n.value].copy()

    def solution(self):
        """-> (nodes int32 (k,), u float64 (A,), score, RomanStats)"""
        n = C.c_int32(0); na = C.c_int32(0)
        self._check(self._lib.roman_num_selected(self._h, C.byref(n)), "roman_num_selected")
        self._check(self._lib.roman_num_associations(self._h, C.byref(na)), "roman_num_associations")
        nodes = np.zeros(max(n.value, 1), dtype=np.int32)
        u = np.zeros(max(na.value, 1), dtype=np.float64)
        score = C.c_double(0.0); st = RomanStats()
        self._check(self._lib.roman_get_solution(self._h, _ptr(nodes), _ptr(u), C.byref(score), C.byref(st)),
                    "roman_get_solution")
        return nodes[:n.value].copy(), u[:na.value].copy(), score.value, st

    def dense_matrices(self):
        na = self.num_associations()
        M = np.zeros((na, na), dtype=np.float64); Cm = np.zeros((na, na), dtype=np.float64)
        if na > 0:
            self._check(self._lib.roman_get_dense_matrices(self._h, _ptr(M), _ptr(Cm)), "roman_get_dense_matrices")
        return M, Cm

    def upper_csr(self):
        """-> (rowptr int64 (A+1,), cols int32, vals float64, diag float64 (A,))"""
        na = self.num_associations()
        nnz = C.c_int64(0)
        self._check(self._lib.roman_get_upper_csr(self._h, C.byref(nnz), None, None, None, None), "roman_get_upper_csr")
        rowptr = np.zeros(na + 1, dtype=np.int64)
        cols = np.zeros(max(nnz.value, 1), dtype=np.int32); vals = np.zeros(max(nnz.value, 1), dtype=np.float64)
        diag = np.zeros(max(na, 1), dtype=np.float64)
        self._check(self._lib.roman_get_upper_csr(self._h, C.byref(nnz), _ptr(rowptr), _ptr(cols), _ptr(vals), _ptr(diag)),
                    "roman_get_upper_csr")
        return rowptr, cols[:nnz.value].copy(), vals[:nnz.value].copy(), diag[:na].copy()

    def live(self):
        n = C.c_int32(0)
        self._check(self._lib.roman_debug_live(self._h, C.byref(n), None, None), "roman_debug_live")
        idx = np.zeros(max(n.value, 1), dtype=np.int32); sc = np.zeros(max(n.value, 1), dtype=np.float64)
        self._check(self._lib.roman_debug_live(self._h, C.byref(n), _ptr(idx), _ptr(sc)), "roman_debug_live")
        return idx[:n.value].copy(), sc[:n.value].copy()

    # ------------------------------------------------------------------ pose
    def pose_batch(self, dim, pts1, pts2, corr_off):
        pts1 = _f64(pts1).reshape(-1, dim); pts2 = _f64(pts2).reshape(-1, dim)
        corr_off = np.ascontiguousarray(corr_off, dtype=np.int64)
        B = corr_off.shape[0] - 1
        T = np.zeros((max(B, 1), 16), dtype=np.float64); status = np.zeros(max(B, 1), dtype=np.int32)
        self._check(self._lib.roman_pose_batch(self._h, int(dim), B, _ptr(pts1), _ptr(pts2), _ptr(corr_off), _ptr(T),
                                               _ptr(status)), "roman_pose_batch")
        s = dim + 1
        return T[:B, :s * s].reshape(B, s, s).copy(), status[:B].copy()

    # ------------------------------------------------------------------ instrumentation / diagnostics
    def profile_enable(self, on=True):
        self._check(self._lib.roman_profile_enable(self._h, int(bool(on))), "roman_profile_enable")

    def profile_reset(self):
        self._check(self._lib.roman_profile_reset(self._h), "roman_profile_reset")

    def profile_get(self):
        ms = (C.c_double * _abi.ROMAN_STAGE_COUNT)(); n = (C.c_int64 * _abi.ROMAN_STAGE_COUNT)()
        self._check(self._lib.roman_profile_get(self._h, ms, n), "roman_profile_get")
        return {name: (ms[i], n[i]) for i, name in enumerate(_abi.STAGE_NAMES)}

    def debug_math(self, kind, x, y=None):
        x = _f64(x).ravel(); ya = None if y is None else _f64(y).ravel()
        out = np.zeros_like(x)
        self._check(self._lib.roman_debug_math(self._h, int(kind), _ptr(x), _ptr(ya), x.size, _ptr(out)), "roman_debug_math")
        return out

    def debug_cosine(self, params, D1, D2):
        D1, D2 = _f64(D1), _f64(D2)
        out = np.zeros((D1.shape[0], D2.shape[0]), dtype=np.float64)
        self._generation += 1
        self._check(self._lib.roman_debug_cosine(self._h, C.byref(params), _ptr(D1), D1.shape[0], _ptr(D2), D2.shape[0],
                                                 D1.shape[1], _ptr(out)), "roman_debug_cosine")
        return out


    def cosine_matrix(self, A, B):
        """Normalised cosine of every row of A (n1, d) with every row of B (n2, d) on the f64 matrix core (k_cos);
        0 where a row has zero norm.  Used for the submap-descriptor gate of the pair loop."""
        A, B = _f64(A), _f64(B)
        if A.ndim != 2 or B.ndim != 2 or A.shape[1] != B.shape[1]:
            raise ValueError("cosine_matrix needs two (n, d) matrices of the same d")
        P = _abi.RomanParams.default()
        P.cos_feature_dim = A.shape[1]
        pad = lambda M: np.ascontiguousarray(np.hstack([np.zeros((M.shape[0], P.point_dim)), M]))     # [xyz | descriptor]
        if A.shape[0] == 0 or B.shape[0] == 0:
            return np.zeros((A.shape[0], B.shape[0]))
        return self.debug_cosine(P, pad(A), pad(B))


_DEFAULT_CTX = None


def default_context():
    """Process-wide context on HIP device 0 (created on first use; fails loudly without a GPU)."""
    global _DEFAULT_CTX
    if _DEFAULT_CTX is None:
        _DEFAULT_CTX = Context(0)
    return _DEFAULT_CTX


def version():
    return _abi.load_library().roman_version().decode()
