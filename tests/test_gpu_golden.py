"""HIP path against the committed golden fixtures made from the reference's own Python
(tests/golden/make_golden.py): T_align outputs of the reference function, and register()/T_align()
results of the reference's unmodified plugin classes."""
import numpy as np
import pytest

from conftest import (golden_gravity_cases, golden_gravity_pair, golden_pair, golden_register_cases, golden_t_align_cases,
                      registration_for)
from roman_amd import _abi
from roman_amd.align import GravityConstraintError, InsufficientAssociationsException

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-5
TCASES = golden_t_align_cases()
RCASES = golden_register_cases()
GCASES = golden_gravity_cases()


def test_pose_kernel_matches_reference_t_align(ctx):
    """roman_pose_batch on every golden case in ONE ragged batch (k from 1 to 200, 2-D and 3-D
    batched separately), incl. reflection (det=-1 branch), planar clouds and k<dim."""
    for dim in (2, 3):
        cs = [c for c in TCASES if c["dim"] == dim]
        off = np.zeros(len(cs) + 1, dtype=np.int64)
        for i, c in enumerate(cs):
            off[i + 1] = off[i] + len(c["p1"])
        p1 = np.concatenate([c["p1"].reshape(-1, dim) for c in cs]); p2 = np.concatenate([c["p2"].reshape(-1, dim) for c in cs])
        T, status = ctx.pose_batch(dim, p1, p2, off)
        for i, c in enumerate(cs):
            if c["ok"]:
                assert status[i] == 0, c["tag"]
                assert np.linalg.norm(T[i] - c["T"]) < POSE_TOL, (c["tag"], np.linalg.norm(T[i] - c["T"]))
            else:
                assert status[i] & _abi.ROMAN_ST_INSUFFICIENT and np.all(np.isnan(T[i])), c["tag"]


def test_pose_kernel_accuracy_is_near_machine_precision(ctx):
    worst = 0.0
    for dim in (2, 3):
        cs = [c for c in TCASES if c["dim"] == dim and c["ok"] and "planar" not in c["tag"]]
        off = np.cumsum([0] + [len(c["p1"]) for c in cs]).astype(np.int64)
        T, _ = ctx.pose_batch(dim, np.concatenate([c["p1"] for c in cs]), np.concatenate([c["p2"] for c in cs]), off)
        worst = max(worst, max(np.linalg.norm(T[i] - c["T"]) for i, c in enumerate(cs)))
    assert worst < 1e-11


def test_pose_of_collinear_points_is_a_proper_rigid_fit(ctx, orc):
    """SURVEY.md §8(c)(5) degenerate case: all correspondences on ONE line.  H has rank 1, the rotation about the line is
    not determined: numpy's SVD (the reference) and the device's Jacobi pick different — equally optimal — members of the
    family.  What is determined is checked: a proper rotation (orthonormal, det +1), the same residual as the reference
    function's fit, and (noise-free) the points mapped onto each other; with k == dim points as well."""
    rng = np.random.default_rng(99)
    cases = []
    for k in (3, 7, 40):
        dirn = rng.standard_normal(3); dirn /= np.linalg.norm(dirn)
        s = np.sort(rng.uniform(-12, 12, k))
        p2 = np.outer(s, dirn) + np.array([1.0, -2.0, 0.5])
        th = rng.uniform(-np.pi, np.pi); R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
        t = rng.uniform(-4, 4, 3)
        cases.append(((R @ p2.T).T + t, p2))
    off = np.cumsum([0] + [len(c[0]) for c in cases]).astype(np.int64)
    T, status = ctx.pose_batch(3, np.concatenate([c[0] for c in cases]), np.concatenate([c[1] for c in cases]), off)
    for i, (p1, p2) in enumerate(cases):
        assert status[i] == 0
        Rg, tg = T[i][:3, :3], T[i][:3, 3]
        assert np.allclose(Rg @ Rg.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(Rg) - 1.0) < 1e-12
        res_gpu = np.linalg.norm((Rg @ p2.T).T + tg - p1)
        To = orc.t_align(p1, p2)
        res_ref = np.linalg.norm((To[:3, :3] @ p2.T).T + To[:3, 3] - p1)
        assert res_gpu < 1e-9 and abs(res_gpu - res_ref) < 1e-9


@pytest.mark.parametrize("case", RCASES, ids=[f"{i}-{c['method']}" for i, c in enumerate(RCASES)])
def test_register_and_t_align_match_reference_plugins(ctx, case):
    """Same calls the reference's caller makes (submap_align.py:156-166): register() then T_align()."""
    reg = registration_for(case["method"], **case["kw"]); reg.set_context(ctx)
    pr = golden_pair(case)
    assert np.array_equal(reg.pack(pr.map1), case["pack1"])          # generator reproducibility guard
    assoc = reg.register(pr.map1, pr.map2)
    assert assoc.dtype.kind == "i" and assoc.shape[1] == 2
    assert np.array_equal(assoc.astype(np.int64), case["assoc"])      # bit-exact association indices
    if len(case["assoc"]) >= reg.dim:
        T = reg.T_align(pr.map1, pr.map2, assoc)
        assert T.shape == case["T"].shape and np.linalg.norm(T - case["T"]) < POSE_TOL
        assert np.linalg.norm(reg.T_align(pr.map1, pr.map2) - case["T"]) < POSE_TOL   # correspondences=None path
    else:
        with pytest.raises(InsufficientAssociationsException):
            reg.T_align(pr.map1, pr.map2, assoc)


@pytest.mark.parametrize("case", RCASES, ids=[f"{i}-{c['method']}" for i, c in enumerate(RCASES)])
def test_batch_entry_matches_reference_plugins(ctx, case):
    """The same golden cases through the BATCHED entry (register_and_align_batch -> roman_align_batch): problems of at most 128
    live associations are finished by the fused small-problem kernel (k_small) — every method string of the factory, 2-D points,
    explicit (pruned) association lists, the all-pruned case — larger ones by the general path; twice, so that the second call
    runs with the sizing history (and, where every problem was k_small's, without the general kernels)."""
    reg = registration_for(case["method"], **case["kw"]); reg.set_context(ctx)
    pr = golden_pair(case)
    for _ in range(2):
        res = reg.register_and_align_batch([(pr.map1, pr.map2), (pr.map2, pr.map1)])
        assert np.array_equal(res.assoc[0].astype(np.int64), case["assoc"])
        if len(case["assoc"]) >= reg.dim:
            assert res.status[0] == 0 and np.linalg.norm(res.T[0] - case["T"]) < POSE_TOL
        else:
            assert res.status[0] & _abi.ROMAN_ST_INSUFFICIENT and np.all(np.isnan(res.T[0]))


@pytest.mark.parametrize("case", GCASES, ids=[f"roll{c['roll']:+g}_pitch{c['pitch']:+g}" for c in GCASES])
def test_gravity_constraint_error_raised_where_the_reference_raises(ctx, case):
    """Row a10 on the device path: `method='clipper+prune'` through roman_amd.align raises GravityConstraintError for exactly
    the pairs the reference's own class raised on ([REF roman/align/dist_reg_with_pruning.py:38-44]; planted roll / pitch 8,
    5.1, 4.9 degrees ...), with the host prefilter and with the prefilter on the device; without the check the selected
    associations and the pose are the reference's."""
    pr = golden_gravity_pair(case)
    for on_device in (False, True):
        reg = registration_for("clipper+prune", **case["kw"]); reg.set_context(ctx)
        if on_device:
            reg.prune_on_device = True; reg.semantics_dim = case["d"]
        if case["raised"]:
            with pytest.raises(GravityConstraintError, match="Roll and pitch must be less than"):
                reg.register(pr.map1, pr.map2)
        else:
            assert np.array_equal(np.asarray(reg.register(pr.map1, pr.map2), dtype=np.int64), case["assoc"])
        reg.use_gravity = False
        assoc = np.asarray(reg.register(pr.map1, pr.map2), dtype=np.int64)
        assert np.array_equal(assoc, case["assoc"])
        assert np.linalg.norm(reg.T_align(pr.map1, pr.map2, assoc) - case["T"]) < POSE_TOL
