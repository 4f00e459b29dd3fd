"""SURVEY.md §8 row a10 on the CPU: the roll/pitch check of DistRegWithPruning.register
([REF roman/align/dist_reg_with_pruning.py:38-44]).  (1) `_zyx_euler` against scipy's as_euler('ZYX') — the call the
reference makes — incl. near gimbal lock; (2) the mirror's register() over an oracle-backed stand-in of the C ABI
(tests/_recording_lib.py) raises GravityConstraintError exactly where the reference's own class did
(tests/golden/gravity_golden.npz, made by tests/golden/make_golden.py::gen_gravity).  The GPU twin of (2) is
tests/test_gpu_golden.py::test_gravity_constraint_error_raised_where_the_reference_raises."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from conftest import golden_gravity_cases, golden_gravity_pair, registration_for
from roman_amd.align import GravityConstraintError
from roman_amd.align.dist_reg_with_pruning import _zyx_euler

GCASES = golden_gravity_cases()


def test_zyx_euler_equals_scipy_on_random_rotations():
    R = Rot.random(2000, random_state=7).as_matrix()
    want = Rot.from_matrix(R).as_euler('ZYX')
    got = np.array([_zyx_euler(r) for r in R])
    assert np.max(np.abs(got - want)) < 1e-12


@pytest.mark.parametrize("pitch_deg", [89.0, 89.9, 89.999, -89.0, -89.9, -89.999, 60.0, -60.0])
def test_zyx_euler_near_gimbal_lock(pitch_deg):
    """|pitch| -> 90 degrees: yaw and roll become ill-conditioned (compare them through the rotation they rebuild), the pitch
    itself — one of the two angles the check reads — stays accurate."""
    rng = np.random.default_rng(int(abs(pitch_deg) * 1000))
    for _ in range(50):
        yaw, roll = rng.uniform(-np.pi, np.pi, 2)
        R = Rot.from_euler('ZYX', [yaw, np.deg2rad(pitch_deg), roll]).as_matrix()
        want = Rot.from_matrix(R).as_euler('ZYX')
        got = np.array(_zyx_euler(R))
        assert abs(got[1] - want[1]) < 1e-9
        assert np.allclose(Rot.from_euler('ZYX', got).as_matrix(), R, atol=1e-9)
        assert np.max(np.abs(got - want)) < 1e-6


def test_zyx_euler_on_the_thresholds_neighbourhood():
    """Angles within 1e-9 rad of the 5 degree threshold resolve on the same side as scipy."""
    th = np.deg2rad(5)
    for delta in (-1e-9, 1e-9, -1e-12, 1e-12):
        for which in (0, 1):
            ang = [0.3, 0.0, 0.0]
            ang[1 + which] = th + delta
            R = Rot.from_euler('ZYX', ang).as_matrix()
            want = Rot.from_matrix(R).as_euler('ZYX'); got = _zyx_euler(R)
            assert (abs(got[1]) < th) == (abs(want[1]) < th) and (abs(got[2]) < th) == (abs(want[2]) < th)


def test_zyx_euler_matches_the_golden_angles():
    for c in GCASES:
        assert np.max(np.abs(np.array(_zyx_euler(c["T"][:3, :3])) - c["ypr"])) < 1e-12


@pytest.fixture()
def oracle_backed_context(orc, monkeypatch):
    """A runtime.Context over the recording stand-in of libroman_hip (CPU oracle behind the C ABI's argument layout)."""
    from _recording_lib import RecordingLib
    from roman_amd import _abi, runtime
    lib = RecordingLib(orc)
    monkeypatch.setattr(_abi, "_LIB", lib)
    ctx = runtime.Context(0)
    yield ctx, lib


@pytest.mark.parametrize("case", GCASES, ids=[f"roll{c['roll']:+g}_pitch{c['pitch']:+g}" for c in GCASES])
def test_register_raises_where_the_reference_raises(oracle_backed_context, case):
    ctx, lib = oracle_backed_context
    reg = registration_for("clipper+prune", **case["kw"]); reg.set_context(ctx)
    assert reg.use_gravity and reg.roll_pitch_thresh == np.deg2rad(5)
    pr = golden_gravity_pair(case)
    if case["raised"]:
        with pytest.raises(GravityConstraintError, match="Roll and pitch must be less than"):
            reg.register(pr.map1, pr.map2)
    else:
        assert np.array_equal(np.asarray(reg.register(pr.map1, pr.map2), dtype=np.int64), case["assoc"])
    assert "roman_pose_batch" in lib.calls                       # the check ran T_align on the selected associations
    reg.use_gravity = False                                      # the solve itself: the reference's selection
    assoc = np.asarray(reg.register(pr.map1, pr.map2), dtype=np.int64)
    assert np.array_equal(assoc, case["assoc"])
    assert np.linalg.norm(reg.T_align(pr.map1, pr.map2, assoc) - case["T"]) < 1e-9
