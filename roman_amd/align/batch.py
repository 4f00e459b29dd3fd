"""Batched alignment: many submap pairs in one device call.

This replaces the reference's serial `for i ... for j ...: register(); T_align()` loop
([REF roman/align/submap_align.py:93-200]) and its per-pair Python feature
packing ([REF roman/align/object_registration.py:43-44], row f2): every submap is packed once into a
shared feature pool and each problem refers to two slices of it.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .. import _abi


@dataclass
class AlignmentBatch:
    """A batch of independent problems over one feature pool (the C-ABI's argument layout)."""
    feats: np.ndarray                 # (n_objects, F) float64, object-major
    off1: np.ndarray                  # (B,) int64  first map-1 object of each problem
    n1: np.ndarray                    # (B,) int32
    off2: np.ndarray                  # (B,) int64
    n2: np.ndarray                    # (B,) int32
    assoc: Optional[np.ndarray] = None        # (sum A_b, 2) int32 explicit association lists
    assoc_off: Optional[np.ndarray] = None    # (B+1,) int64
    pair_index: Optional[np.ndarray] = None   # (B,2) (i,j) submap indices, for grid batches

    def __len__(self):
        return int(self.n1.shape[0])

    def kmax(self):
        return int(max(1, np.max(np.minimum(self.n1, self.n2)))) if len(self) else 1

    def subset(self, lo, hi):
        """Problems [lo, hi) over the same pool (used for sharding across ranks)."""
        a, ao = None, None
        if self.assoc is not None:
            ao = self.assoc_off[lo:hi + 1] - self.assoc_off[lo]
            a = self.assoc[self.assoc_off[lo]:self.assoc_off[hi]]
        return AlignmentBatch(self.feats, self.off1[lo:hi], self.n1[lo:hi], self.off2[lo:hi], self.n2[lo:hi], a, ao,
                              None if self.pair_index is None else self.pair_index[lo:hi])


def pack_submaps(registration, submaps: Sequence[Sequence]) -> Tuple[np.ndarray, np.ndarray]:
    """Pack every submap ONCE: -> (feats (sum n_s, F), offsets (S+1,) int64)."""
    mats = [registration.pack(sm) for sm in submaps]
    F = mats[0].shape[1] if mats else registration._abi_params().feature_dim()
    offsets = np.zeros(len(mats) + 1, dtype=np.int64)
    for s, m in enumerate(mats):
        offsets[s + 1] = offsets[s] + m.shape[0]
    feats = np.concatenate(mats, axis=0) if mats else np.zeros((0, F))
    return np.ascontiguousarray(feats, dtype=np.float64), offsets


def all_pairs_problems(offsets0: np.ndarray, offsets1: np.ndarray, base1: int = 0,
                       mask: Optional[np.ndarray] = None):
    """Problem list of the |S0| x |S1| grid (row-major i*|S1|+j), optionally restricted to
    mask[i,j] == True (the distance / descriptor gates of [REF roman/align/submap_align.py:136-149]).
    offsets1 index a pool that starts `base1` objects into the shared feature pool."""
    S0, S1 = len(offsets0) - 1, len(offsets1) - 1
    ii, jj = np.meshgrid(np.arange(S0), np.arange(S1), indexing='ij')
    ii, jj = ii.ravel(), jj.ravel()
    if mask is not None:
        keep = np.asarray(mask, dtype=bool).ravel()
        ii, jj = ii[keep], jj[keep]
    off1 = offsets0[ii].astype(np.int64)
    n1 = (offsets0[ii + 1] - offsets0[ii]).astype(np.int32)
    off2 = (offsets1[jj] + base1).astype(np.int64)
    n2 = (offsets1[jj + 1] - offsets1[jj]).astype(np.int32)
    return off1, n1, off2, n2, np.stack([ii, jj], axis=1)


def batch_from_pairs(registration, pairs) -> AlignmentBatch:
    """pairs: sequence of (map1, map2) object lists."""
    maps = []
    for m1, m2 in pairs:
        maps.append(m1); maps.append(m2)
    feats, offs = pack_submaps(registration, maps)
    B = len(pairs)
    off1 = offs[0:2 * B:2].copy(); off2 = offs[1:2 * B:2].copy()
    n1 = (offs[1:2 * B + 1:2] - offs[0:2 * B:2]).astype(np.int32)
    n2 = (offs[2:2 * B + 2:2] - offs[1:2 * B:2]).astype(np.int32)
    assoc, assoc_off = None, None
    lists = [registration._association_list(m1, m2) if (len(m1) and len(m2)) else None for m1, m2 in pairs]
    if any(a is not None for a in lists):
        from ..clipperpy.utils import create_all_to_all
        lists = [a if a is not None else create_all_to_all(len(m1), len(m2)) for a, (m1, m2) in zip(lists, pairs)]
        assoc_off = np.zeros(B + 1, dtype=np.int64)
        for b, a in enumerate(lists):
            assoc_off[b + 1] = assoc_off[b] + len(a)
        assoc = np.concatenate(lists, axis=0).astype(np.int32) if B else np.zeros((0, 2), dtype=np.int32)
    return AlignmentBatch(feats, off1, n1, off2, n2, assoc, assoc_off)


def batch_from_submap_grid(registration, submaps0, submaps1, mask=None) -> AlignmentBatch:
    """All (or the masked) cross pairs of two robots' submaps over one shared feature pool."""
    f0, o0 = pack_submaps(registration, submaps0)
    f1, o1 = pack_submaps(registration, submaps1)
    feats = np.concatenate([f0, f1], axis=0)
    off1, n1, off2, n2, idx = all_pairs_problems(o0, o1, base1=f0.shape[0], mask=mask)
    return AlignmentBatch(feats, off1, n1, off2, n2, pair_index=idx)


def run_batch(registration, batch: AlignmentBatch, u0=None, ctx=None):
    """One roman_align_batch call for `batch` -> runtime.BatchResult."""
    ctx = ctx or registration._context()
    return ctx.align_batch(registration._abi_params(), batch.feats, batch.off1, batch.n1, batch.off2, batch.n2,
                           assoc=batch.assoc, assoc_off=batch.assoc_off, u0=u0, kmax=batch.kmax())


def align_pairs(registration, pairs, u0=None):
    """register() + T_align() for every (map1, map2) in `pairs`, one device call."""
    return run_batch(registration, batch_from_pairs(registration, pairs), u0=u0)


# ------------------------------------------------------------------------------------------------
# fixed-size result records (what ranks exchange with one all_gather)
# ------------------------------------------------------------------------------------------------
def pack_records(result, kmax):
    """-> (ints (B, 2+2*kmax) int32 [count, status, i0, j0, i1, j1, ...], poses (B,16) float64)."""
    B = len(result.assoc)
    ints = np.full((B, 2 + 2 * kmax), -1, dtype=np.int32)
    poses = np.full((B, 16), np.nan, dtype=np.float64)
    for b in range(B):
        a = np.asarray(result.assoc[b]).reshape(-1, 2)[:kmax]
        ints[b, 0] = a.shape[0]
        ints[b, 1] = result.status[b]
        ints[b, 2:2 + 2 * a.shape[0]] = a.ravel()
        t = np.asarray(result.T[b]).ravel()
        poses[b, :t.size] = t
    return ints, poses


def record_bytes(kmax):
    """Bytes of ONE fixed-size result record (count, status, <= kmax index pairs, 16 doubles): the int32 part
    [count, status, i0, j0, ...] — 8 * (1 + kmax) bytes, so the pose behind it is 8-byte aligned — then the pose."""
    return 8 * (1 + int(kmax)) + 128


def records_as_bytes(ints, poses):
    """(ints (B, 2+2*kmax) int32, poses (B,16) float64) -> (B, record_bytes(kmax)) uint8: what ONE all_gather moves."""
    ints = np.ascontiguousarray(ints, dtype=np.int32); poses = np.ascontiguousarray(poses, dtype=np.float64)
    return np.concatenate([ints.view(np.uint8).reshape(ints.shape[0], -1), poses.view(np.uint8).reshape(poses.shape[0], -1)], axis=1)


def records_from_bytes(rec, kmax):
    """Inverse of records_as_bytes -> (ints, poses)."""
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    ib = 8 * (1 + int(kmax))
    return (np.ascontiguousarray(rec[:, :ib]).view(np.int32).reshape(rec.shape[0], -1),
            np.ascontiguousarray(rec[:, ib:]).view(np.float64).reshape(rec.shape[0], 16))


def unpack_records(ints, poses, dim=3):
    """Inverse of pack_records -> (assoc list, T (B,dim+1,dim+1), status)."""
    ints = np.asarray(ints); poses = np.asarray(poses)
    s = dim + 1
    assoc = [ints[b, 2:2 + 2 * max(int(ints[b, 0]), 0)].reshape(-1, 2).copy() for b in range(ints.shape[0])]
    return assoc, poses[:, :s * s].reshape(-1, s, s).copy(), ints[:, 1].copy()
