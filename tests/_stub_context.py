"""A stand-in for roman_amd.runtime.Context on a box without a GPU — TEST INFRASTRUCTURE (tests/test_pipeline_cpu.py).

`align_batch_dev` takes the same raw addresses the real entry takes (here: of torch CPU tensors / NumPy arrays), computes every
problem with the CPU oracle and writes associations, counts, poses and statuses through the pointers, so that the package's
chunking / re-issue logic (roman_amd.align.pipeline.issue_chunked, distributed._device_records) runs unmodified.  Faults can
be injected the way the library produces them: a problem reported ROMAN_ST_WORKSPACE on its first `skip_times` appearances,
ROMAN_ST_INTERNAL for chosen problems while the team mode is not switched off."""
import ctypes as C

import numpy as np

from roman_amd import _abi


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(int(ptr))
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class OracleContext:
    def __init__(self, n_objects, dim=3, skip_times=None, internal_with_teams=()):
        from oracle import oracle
        self.orc = oracle
        self.n_objects, self.dim = int(n_objects), dim
        self.skip_times = dict(skip_times or {})       # (off1, off2) -> appearances still to be skipped
        self.internal_with_teams = set(internal_with_teams)
        self.calls = []                                # (pipeline depth, problems) of every call
        self.pipeline, self.wide_teams, self.syncs = 1, -1, 0
        self._block, self._hist = None, None           # parameter block of the latest call / the ONE block the sizing history is for

    def set_pipeline(self, depth):
        self.pipeline = int(depth)

    def set_wide_teams(self, t):
        self.wide_teams = int(t)

    def sync(self):
        self.syncs += 1
        if self._block is not None:                    # a finished batch has reported what it needed (roman_ctx_has_history)
            self._hist = self._block

    def join(self, skip_latest=False, stream=None):      # every call of this stand-in is complete when it returns
        pass

    def has_history(self, P, F):
        return self._hist == bytes(P) + int(F).to_bytes(4, "little")

    def align_batch_dev(self, P, feats_ptr, F, off1, n1, off2, n2, kmax, a_ptr, n_ptr, T_ptr, st_ptr, stats_ptr=None,
                        assoc_ptr=None, assoc_off=None, u0_ptr=None):
        B = len(n1)
        self.calls.append((self.pipeline, B, self.syncs))
        self._block = bytes(P) + int(F).to_bytes(4, "little")
        feats = _view(feats_ptr, (self.n_objects, F), np.float64)
        a_out = _view(a_ptr, (B, kmax, 2), np.int32); n_out = _view(n_ptr, (B,), np.int32)
        T_out = _view(T_ptr, (B, 16), np.float64); st_out = _view(st_ptr, (B,), np.int32)
        rows = int(assoc_off[-1]) if assoc_off is not None else 0
        assoc = _view(assoc_ptr, (rows, 2), np.int32) if assoc_ptr else None
        d = self.dim
        for b in range(B):
            key = (int(off1[b]), int(off2[b]))
            n_out[b] = 0; T_out[b] = np.nan; a_out[b] = -1
            if self.skip_times.get(key, 0) > 0:
                self.skip_times[key] -= 1
                st_out[b] = _abi.ROMAN_ST_WORKSPACE
                continue
            if key in self.internal_with_teams and self.wide_teams != 0:
                st_out[b] = _abi.ROMAN_ST_INTERNAL
                continue
            D1 = feats[off1[b]:off1[b] + n1[b]]; D2 = feats[off2[b]:off2[b] + n2[b]]
            if len(D1) == 0 or len(D2) == 0:
                st_out[b] = _abi.ROMAN_ST_EMPTY_MAP | _abi.ROMAN_ST_INSUFFICIENT
                continue
            A = None
            if assoc is not None and assoc_off[b + 1] > assoc_off[b]:
                A = assoc[assoc_off[b]:assoc_off[b + 1]]
            a = self.orc.register(P, D1, D2, A)["assoc"][:kmax]
            n_out[b] = len(a); a_out[b, :len(a)] = a
            if len(a) >= d:
                T = self.orc.t_align(D1[a[:, 0], :d], D2[a[:, 1], :d], d)
                T_out[b, :(d + 1) ** 2] = T.ravel(); st_out[b] = 0
            else:
                st_out[b] = _abi.ROMAN_ST_INSUFFICIENT
