"""A recording stand-in for libroman_hip.so, for the CPU box: every C-ABI entry the clipperpy shim reaches is present, logs
its name, decodes its ctypes arguments exactly as the C side would (raw pointers + sizes, the roman_params_t block) and
answers from the CPU oracle.  TEST INFRASTRUCTURE: it lets the reference's UNMODIFIED roman/align/*.py run end to end
through roman_amd.clipperpy -> roman_amd.runtime -> (this object) where no GPU exists, so that the module-swap adoption
of INTEGRATION.md §1 is executed on the actual files and the ctypes marshalling (transposed views, association lists, the
parameter block) is checked, not argued.  `EXPECTED_REGISTER_CALLS` is the C-ABI sequence of ONE register(); the GPU suite
(tests/test_gpu_shim.py) records the same sequence against the real library."""
import ctypes as C

import numpy as np

from roman_amd import _abi

# one `registration.register(map1, map2)` of the reference ([REF roman/align/object_registration.py:22-29]) on a live context
EXPECTED_REGISTER_CALLS = ["roman_score", "roman_solve", "roman_num_selected", "roman_num_associations", "roman_get_solution",
                           "roman_num_selected", "roman_get_selected_associations"]


def _arr(ptr, n, dtype):
    """n elements of `dtype` at a raw pointer (c_void_p / int / None) -> numpy copy."""
    if ptr is None or n == 0:
        return np.zeros(0, dtype=dtype)
    addr = ptr.value if isinstance(ptr, C.c_void_p) else int(ptr)
    buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).copy()


def _put(ptr, values, dtype):
    values = np.ascontiguousarray(values, dtype=dtype)
    if ptr is None or values.size == 0:
        return
    addr = ptr.value if isinstance(ptr, C.c_void_p) else int(ptr)
    C.memmove(addr, values.ctypes.data, values.nbytes)


def _obj(ref):
    """ctypes byref(x) -> x"""
    return ref._obj


class RecordingLib:
    def __init__(self, orc):
        self.orc = orc
        self.calls = []
        self.P = None; self.mat = None; self.A = None; self.sol = None; self.dense = False
        self._roman_symbols = _abi.EXPORTED_SYMBOLS

    def _log(self, name):
        self.calls.append(name)

    # ---- context -----------------------------------------------------------------------------------------
    def roman_ctx_create(self, href, device, stream):
        self._log("roman_ctx_create"); _obj(href).value = 0x1234; return 0

    def roman_ctx_destroy(self, h):
        self._log("roman_ctx_destroy"); return 0

    def roman_last_error(self, h):
        return b""

    def roman_version(self):
        return b"recording stand-in (CPU oracle behind the C ABI's argument layout)"

    # ---- stepwise surface ----------------------------------------------------------------------------------
    def roman_score(self, h, pref, D1p, n1, D2p, n2, F, Ap, nA):
        self._log("roman_score")
        P = type(_obj(pref)).from_buffer_copy(_obj(pref))
        D1 = _arr(D1p, n1 * F, np.float64).reshape(n1, F); D2 = _arr(D2p, n2 * F, np.float64).reshape(n2, F)
        A = None if Ap is None or nA == 0 else _arr(Ap, 2 * nA, np.int32).reshape(nA, 2)
        if P.invariant == _abi.ROMAN_INV_EUCLIDEAN:           # the C side ignores the ROMAN-only fields then
            P.ratio_feature_dim = 0; P.cos_feature_dim = 0; P.gravity_guided = 0
        self.P = P
        self.mat, self.A = self.orc.build_matrix(P, D1, D2, A)
        self.n2 = n2; self.dense = False; self.sol = None
        self.scored = dict(D1=D1, D2=D2, A=A, F=F, params=P.as_dict())
        return 0

    def roman_set_matrix_data(self, h, pref, Mp, Cp, n):
        self._log("roman_set_matrix_data")
        P = type(_obj(pref)).from_buffer_copy(_obj(pref)); P.invariant = _abi.ROMAN_INV_EUCLIDEAN
        M = _arr(Mp, n * n, np.float64).reshape(n, n); Cm = _arr(Cp, n * n, np.float64).reshape(n, n)
        self.P = P; self.mat = self.orc.matrix_from_dense(M, Cm); self.A = np.stack([np.arange(n), np.arange(n)], axis=1).astype(np.int32)
        self.dense = True; self.sol = None
        return 0

    def roman_solve(self, h, u0p):
        self._log("roman_solve")
        u0 = None if u0p is None else _arr(u0p, self.mat.n, np.float64)
        self.sol = self.orc.solve(self.P, self.mat, u0)
        return 0

    def roman_num_associations(self, h, nref):
        self._log("roman_num_associations"); _obj(nref).value = self.mat.n; return 0

    def roman_num_selected(self, h, nref):
        self._log("roman_num_selected"); _obj(nref).value = len(self.sol["nodes"]); return 0

    def roman_get_selected_associations(self, h, outp):
        self._log("roman_get_selected_associations"); _put(outp, self.A[self.sol["nodes"]], np.int32); return 0

    def roman_get_solution(self, h, nodesp, up, scoreref, statsref):
        self._log("roman_get_solution")
        _put(nodesp, self.sol["nodes"], np.int32); _put(up, self.sol["u"], np.float64)
        st = self.sol["stats"]
        _obj(scoreref).value = st.score
        C.memmove(C.addressof(_obj(statsref)), C.addressof(st), C.sizeof(st))
        return 0

    def roman_get_dense_matrices(self, h, Mp, Cp):
        self._log("roman_get_dense_matrices")
        M, Cm = self.mat.dense()
        _put(Mp, M, np.float64); _put(Cp, Cm, np.float64)
        return 0

    def roman_pose_batch(self, h, dim, B, p1p, p2p, offp, Tp, statp):
        """[include/roman_hip.h roman_pose_batch]: B ragged correspondence sets -> (dim+1)^2 poses in 16-double records."""
        self._log("roman_pose_batch")
        off = _arr(offp, B + 1, np.int64)
        p1 = _arr(p1p, int(off[-1]) * dim, np.float64).reshape(-1, dim); p2 = _arr(p2p, int(off[-1]) * dim, np.float64).reshape(-1, dim)
        T = np.full((B, 16), np.nan); status = np.zeros(B, dtype=np.int32)
        for b in range(B):
            k = int(off[b + 1] - off[b])
            if k < dim:
                status[b] = _abi.ROMAN_ST_INSUFFICIENT
            else:
                T[b, :(dim + 1) ** 2] = self.orc.t_align(p1[off[b]:off[b + 1]], p2[off[b]:off[b + 1]]).ravel()
        _put(Tp, T, np.float64); _put(statp, status, np.int32)
        return 0

    def __getattr__(self, name):                                   # any other entry: a loud failure naming it
        if name.startswith("roman_"):
            def missing(*a):
                raise AssertionError(f"the reference path reached {name}, which the recording stand-in does not model")
            return missing
        raise AttributeError(name)
