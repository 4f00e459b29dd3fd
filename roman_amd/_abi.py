"""ctypes mirror of include/roman_hip.h (structs + loader of libroman_hip.so).

The structs are shared with the test oracle (oracle/oracle.py) so both sides of a parity test
are driven by the same parameter block.  The library loader fails loudly: there is no Python /
CPU fallback behind the C ABI.
"""
import ctypes as C
import os

ROMAN_MAX_RATIO_FEATURES = 8

# error codes / status flags (include/roman_hip.h)
ROMAN_OK = 0
ROMAN_E_INTERNAL = -7
ROMAN_ST_OK = 0
ROMAN_ST_EMPTY_MAP = 1
ROMAN_ST_INSUFFICIENT = 2
ROMAN_ST_MAXITER = 4
ROMAN_ST_ASSOC_TRUNCATED = 8
ROMAN_ST_TIE_FALLBACK = 16
ROMAN_ST_WORKSPACE = 32
ROMAN_ST_INTERNAL = 64

ROMAN_INV_EUCLIDEAN = 0
ROMAN_INV_ROMAN = 1
ROMAN_INV_EUCLIDEAN_PRUNED = 2

ROMAN_FUSE_GEOMETRIC_MEAN = 0
ROMAN_FUSE_ARITHMETIC_MEAN = 1
ROMAN_FUSE_PRODUCT = 2

ROMAN_GRAV_COMBINED = 0
ROMAN_GRAV_SEPARATE = 1
ROMAN_GRAV_ZGATE = 2
ROMAN_SINGLE_BOTH = 0
ROMAN_SINGLE_OFFDIAG = 1
ROMAN_SINGLE_DIAG = 2
ROMAN_SINGLE_DIAG_KEEP = 3

ROMAN_STAGE_SINGLE = 0
ROMAN_STAGE_COUNT_PASS = 1
ROMAN_STAGE_FILL = 2
ROMAN_STAGE_SOLVE = 3
ROMAN_STAGE_COUNT = 4
STAGE_NAMES = ("single", "count", "fill", "solve")


class RomanParams(C.Structure):
    """roman_params_t"""
    _fields_ = [
        ("invariant", C.c_int32),
        ("point_dim", C.c_int32),
        ("ratio_feature_dim", C.c_int32),
        ("cos_feature_dim", C.c_int32),
        ("fusion_method", C.c_int32),
        ("gravity_guided", C.c_int32),
        ("drift_aware", C.c_int32),
        ("rescale_u0", C.c_int32),
        ("sigma", C.c_double),
        ("epsilon", C.c_double),
        ("mindist", C.c_double),
        ("distance_weight", C.c_double),
        ("ratio_weight", C.c_double),
        ("cosine_weight", C.c_double),
        ("cosine_min", C.c_double),
        ("cosine_max", C.c_double),
        ("gravity_unc_ang_rad", C.c_double),
        ("ratio_epsilon", C.c_double * ROMAN_MAX_RATIO_FEATURES),
        ("tol_u", C.c_double),
        ("tol_F", C.c_double),
        ("beta", C.c_double),
        ("eps", C.c_double),
        ("affinityeps", C.c_double),
        ("maxiniters", C.c_int32),
        ("maxoliters", C.c_int32),
        ("maxlsiters", C.c_int32),
        ("gravity_mode", C.c_int32),
        ("single_mode", C.c_int32),
        ("reserved", C.c_int32),
    ]

    @classmethod
    def default(cls):
        """Same values as roman_params_default() (kept in Python so parameter blocks can be
        built without a GPU; tests check the two agree)."""
        p = cls()
        p.invariant = ROMAN_INV_ROMAN
        p.point_dim = 3
        p.fusion_method = ROMAN_FUSE_GEOMETRIC_MEAN
        p.rescale_u0 = 1
        p.sigma, p.epsilon, p.mindist = 0.4, 0.6, 0.2
        p.distance_weight = p.ratio_weight = p.cosine_weight = 1.0
        p.cosine_min, p.cosine_max = 0.5, 0.7
        p.gravity_unc_ang_rad = 0.0872665
        p.tol_u, p.tol_F, p.beta, p.eps, p.affinityeps = 1e-8, 1e-9, 0.25, 1e-9, 1e-4
        p.maxiniters, p.maxoliters, p.maxlsiters = 200, 1000, 99
        return p

    def feature_dim(self):
        return self.point_dim + self.ratio_feature_dim + self.cos_feature_dim

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if name == "ratio_epsilon" else v
        return d


class RomanStats(C.Structure):
    """roman_stats_t"""
    _fields_ = [
        ("n_assoc_in", C.c_int32),
        ("n_live", C.c_int32),
        ("nnz_upper", C.c_int64),
        ("n_pass", C.c_int32),
        ("outer_iters", C.c_int32),
        ("inner_iters", C.c_int32),
        ("ls_trials", C.c_int32),
        ("score", C.c_double),
        ("d_final", C.c_double),
    ]


STATS_NBYTES = C.sizeof(RomanStats)
PARAMS_NBYTES = C.sizeof(RomanParams)

_LIB = None
# in-tree build product (roman_amd/csrc/Makefile); ROMAN_HIP_LIBRARY overrides the location
_LIB_PATH = os.environ.get("ROMAN_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libroman_hip.so")


class RomanHipError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load_library():
    """dlopen libroman_hip.so and declare the prototypes of every symbol in roman_hip.h."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise RomanHipError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for roman_amd.")
    lib = C.CDLL(_LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    P = C.POINTER
    ctxp = vp
    protos = {
        "roman_params_default": (C.c_int, [P(RomanParams)]),
        "roman_ctx_create": (C.c_int, [P(ctxp), C.c_int, vp]),
        "roman_ctx_destroy": (C.c_int, [ctxp]),
        "roman_ctx_set_pipeline": (C.c_int, [ctxp, C.c_int]),
        "roman_ctx_sync": (C.c_int, [ctxp]),
        "roman_ctx_set_host_batching": (C.c_int, [ctxp, C.c_int, C.c_int]),
        "roman_ctx_set_wide_teams": (C.c_int, [ctxp, C.c_int]),
        "roman_ctx_join": (C.c_int, [ctxp, C.c_int]),
        "roman_ctx_join_on": (C.c_int, [ctxp, C.c_int, C.c_void_p]),
        "roman_ctx_skipped": (C.c_int, [ctxp, C.c_int, P(i64)]),
        "roman_last_error": (C.c_char_p, [ctxp]),
        "roman_align_batch_dev": (C.c_int, [ctxp, P(RomanParams), i32, vp, vp, vp, vp, vp, i32,
                                            vp, vp, vp, i32, vp, vp, vp, vp, vp]),
        "roman_align_batch": (C.c_int, [ctxp, P(RomanParams), i32, vp, i64, vp, vp, vp, vp, i32,
                                        vp, vp, vp, i32, vp, vp, vp, vp, vp]),
        "roman_align_batch_resident": (C.c_int, [ctxp, P(RomanParams), i32, vp, vp, vp, vp, vp, i32,
                                                 vp, vp, vp, i32, vp, vp, vp, vp, vp]),
        "roman_ctx_has_history": (C.c_int, [ctxp, P(RomanParams), i32, P(i32)]),
        "roman_ctx_cosine_screen_stats": (C.c_int, [ctxp, P(C.c_int64), P(C.c_int64), P(C.c_double)]),
        "roman_create_all_to_all": (C.c_int, [i32, i32, vp]),
        "roman_deal_problems": (C.c_int, [i32, vp, vp, vp, i32, i32, vp, P(i32)]),
        "roman_score": (C.c_int, [ctxp, P(RomanParams), vp, i32, vp, i32, i32, vp, i32]),
        "roman_set_matrix_data": (C.c_int, [ctxp, P(RomanParams), vp, vp, i32]),
        "roman_solve": (C.c_int, [ctxp, vp]),
        "roman_num_associations": (C.c_int, [ctxp, P(i32)]),
        "roman_num_selected": (C.c_int, [ctxp, P(i32)]),
        "roman_get_selected_associations": (C.c_int, [ctxp, vp]),
        "roman_get_solution": (C.c_int, [ctxp, vp, vp, P(dbl), P(RomanStats)]),
        "roman_get_dense_matrices": (C.c_int, [ctxp, vp, vp]),
        "roman_get_upper_csr": (C.c_int, [ctxp, P(i64), vp, vp, vp, vp]),
        "roman_pose_batch": (C.c_int, [ctxp, i32, i32, vp, vp, vp, vp, vp]),
        "roman_profile_enable": (C.c_int, [ctxp, C.c_int]),
        "roman_profile_reset": (C.c_int, [ctxp]),
        "roman_profile_get": (C.c_int, [ctxp, P(dbl), P(i64)]),
        "roman_debug_math": (C.c_int, [ctxp, C.c_int, vp, vp, i64, vp]),
        "roman_debug_cosine": (C.c_int, [ctxp, P(RomanParams), vp, i32, vp, i32, i32, vp]),
        "roman_debug_live": (C.c_int, [ctxp, P(i32), vp, vp]),
        "roman_version": (C.c_char_p, []),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)          # AttributeError here == header/library drift
        fn.restype = res
        fn.argtypes = args
    lib._roman_symbols = tuple(protos)
    _LIB = lib
    return lib


EXPORTED_SYMBOLS = (
    "roman_params_default", "roman_ctx_create", "roman_ctx_destroy", "roman_ctx_set_pipeline", "roman_ctx_sync", "roman_ctx_set_host_batching", "roman_ctx_set_wide_teams", "roman_ctx_join", "roman_ctx_join_on",
    "roman_ctx_skipped", "roman_last_error",
    "roman_align_batch_dev", "roman_align_batch", "roman_align_batch_resident", "roman_ctx_has_history", "roman_ctx_cosine_screen_stats", "roman_deal_problems", "roman_create_all_to_all", "roman_score",
    "roman_set_matrix_data", "roman_solve", "roman_num_associations", "roman_num_selected",
    "roman_get_selected_associations", "roman_get_solution", "roman_get_dense_matrices",
    "roman_get_upper_csr", "roman_pose_batch", "roman_profile_enable", "roman_profile_reset",
    "roman_profile_get", "roman_debug_math", "roman_debug_cosine", "roman_debug_live",
    "roman_version",
)
