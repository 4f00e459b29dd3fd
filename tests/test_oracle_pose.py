"""The pose oracle (oracle.t_align) against the REFERENCE'S OWN T_align outputs
(tests/golden/t_align_golden.npz, made by tests/golden/make_golden.py from
/root/reference/roman/align/object_registration.py:88-129).  This is the pinned part of the oracle."""
import numpy as np
import pytest

from conftest import golden_t_align_cases

CASES = golden_t_align_cases()


@pytest.mark.parametrize("case", CASES, ids=[f"{i}-{c['tag']}" for i, c in enumerate(CASES)])
def test_pose_oracle_matches_reference(orc, case):
    if not case["ok"]:
        with pytest.raises(orc.InsufficientAssociations):
            orc.t_align(case["p1"], case["p2"], case["dim"])
        return
    T = orc.t_align(case["p1"], case["p2"], case["dim"])
    assert T.shape == case["T"].shape
    assert np.linalg.norm(T - case["T"]) <= 1e-12          # same LAPACK: bitwise-close


def test_golden_covers_reference_branches():
    tags = " ".join(c["tag"] for c in CASES)
    for needed in ("rigid3d", "exact3d", "reflect3d", "planar3d", "rigid2d", "reflect2d", "insufficient3d", "insufficient2d"):
        assert needed in tags
    # the reflection cases really exercise the det == -1 branch: the golden R is a proper rotation
    for c in CASES:
        if c["ok"]:
            d = c["dim"]
            R = c["T"][:d, :d]
            assert abs(np.linalg.det(R) - 1.0) < 1e-9
            assert np.allclose(R @ R.T, np.eye(d), atol=1e-9)
