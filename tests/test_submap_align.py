"""Rows f1/f3 of SURVEY.md §8: the batched submap-pair loop and the result writers against fixtures produced by
the reference's own `submap_align()` / `save_submap_align_results()` (tests/golden/make_golden.py
gen_submap_align; [REF roman/align/submap_align.py:28-220], [REF roman/align/results.py:122-243]).

CPU tests inject the oracle as the compute step (host logic under test); the `gpu` test runs the same
scenarios through libroman_hip.so."""
import json
import os
import pickle

import numpy as np
import pytest

from roman_amd import synth
from roman_amd.align import SubmapAlignParams
from roman_amd.align import submap_align as sa
from roman_amd.runtime import BatchResult, stats_dtype

GOLD = os.path.join(os.path.dirname(__file__), "golden", "submap_align_golden.npz")
TOL = 1e-8          # float outputs: poses via two different SVD implementations, angles through arccos


def oracle_compute(registration, batch):
    """CPU double for run_batch: every problem of the batch through the oracle (explicit lists included)."""
    from oracle import oracle as orc
    P, d = registration._abi_params(), registration.dim
    assoc, Ts, status = [], [], []
    for b in range(len(batch)):
        D1 = batch.feats[batch.off1[b]:batch.off1[b] + batch.n1[b]]; D2 = batch.feats[batch.off2[b]:batch.off2[b] + batch.n2[b]]
        if len(D1) == 0 or len(D2) == 0:
            assoc.append(np.zeros((0, 2), np.int32)); Ts.append(np.full((d + 1, d + 1), np.nan)); status.append(3); continue
        A = None if batch.assoc is None else batch.assoc[batch.assoc_off[b]:batch.assoc_off[b + 1]]
        a = orc.register(P, D1, D2, A=A)["assoc"]
        assoc.append(a)
        if len(a) >= d:
            Ts.append(orc.t_align(D1[a[:, 0], :d], D2[a[:, 1], :d], d)); status.append(0)
        else:
            Ts.append(np.full((d + 1, d + 1), np.nan)); status.append(2)
    return BatchResult(assoc, np.array(Ts).reshape(-1, d + 1, d + 1), np.array(status, np.int32), np.zeros(len(batch), stats_dtype()))


def build(name):
    pk, iok, robots, trajs = synth.make_align_scenario(name)
    submaps = [[sa.Submap(**s) for s in rob] for rob in robots]
    return SubmapAlignParams(**pk), sa.SubmapAlignIO(**iok), submaps, trajs


def numbers_close(text_a, text_b):
    """Same token structure; numeric tokens within TOL, everything else identical."""
    ta, tb = text_a.split(), text_b.split()
    assert len(ta) == len(tb)
    for x, y in zip(ta, tb):
        try:
            fx, fy = float(x), float(y)
        except ValueError:
            assert x == y
            continue
        assert abs(fx - fy) <= TOL * max(1.0, abs(fy)), (x, y)


def json_close(a, b):
    if isinstance(a, dict):
        assert a.keys() == b.keys()
        for k in a:
            json_close(a[k], b[k])
    elif isinstance(a, list):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            json_close(x, y)
    elif isinstance(a, float):
        assert abs(a - b) <= TOL * max(1.0, abs(b))
    else:
        assert a == b


def check_scenario(name, compute, tmp_path):
    g = np.load(GOLD, allow_pickle=False)
    params, io, submaps, trajs = build(name)
    res = sa.submap_align(params, submaps, io, compute=compute)
    for k in ["robots_nearby_mat", "clipper_num_associations", "submap_yaw_diff_mat", "T_ij_mat"]:
        np.testing.assert_allclose(getattr(res, k), g[f"{name}/{k}"], rtol=0, atol=TOL, equal_nan=True, err_msg=k)
    np.testing.assert_allclose(res.T_ij_hat_mat, g[f"{name}/T_ij_hat_mat"], rtol=0, atol=TOL, equal_nan=True)
    np.testing.assert_allclose(res.clipper_dist_mat, g[f"{name}/clipper_dist_mat"], rtol=0, atol=1e-7, equal_nan=True)
    np.testing.assert_allclose(res.clipper_angle_mat, g[f"{name}/clipper_angle_mat"], rtol=0, atol=1e-5, equal_nan=True)  # arccos near 0
    if bool(g[f"{name}/has_similarity"]):
        np.testing.assert_allclose(res.similarity_mat, g[f"{name}/similarity_mat"], rtol=0, atol=1e-12, equal_nan=True)
    else:
        assert res.similarity_mat is None
    n0, n1 = res.clipper_num_associations.shape
    for i in range(n0):
        for j in range(n1):
            mine = np.asarray(res.associated_objs_mat[i][j], dtype=np.int64).reshape(-1, 2)
            assert np.array_equal(mine, g[f"{name}/assoc_{i}_{j}"]), (i, j)       # bit-exact index work

    # writers
    sa.write_g2o(tmp_path / "run.g2o", res, submaps, [t[0] for t in trajs])
    gold_g2o, mine_g2o = str(g[f"{name}/g2o"]), (tmp_path / "run.g2o").read_text()
    assert [l.split()[:3] for l in mine_g2o.splitlines()] == [l.split()[:3] for l in gold_g2o.splitlines()]   # "# LC: n" / vertex keys
    assert mine_g2o.count("\t") == gold_g2o.count("\t")
    numbers_close(mine_g2o, gold_g2o)
    sa.write_lc_json(tmp_path / "run.json", res, submaps)
    json_close(json.loads((tmp_path / "run.json").read_text()), json.loads(str(g[f"{name}/json"])))
    sa.write_timing(tmp_path / "run.timing.txt", res, submaps)
    mine_t, gold_t = (tmp_path / "run.timing.txt").read_text().splitlines(), str(g[f"{name}/timing"]).splitlines()
    assert [mine_t[k] for k in (0, 3, 4)] == [gold_t[k] for k in (0, 3, 4)] and len(mine_t) == len(gold_t)
    sa.write_matrix_pickle(tmp_path / "run.matrix.pkl", res)
    with open(tmp_path / "run.matrix.pkl", "rb") as f:
        mats = pickle.load(f)
    assert len(mats) == int(g[f"{name}/matrix_pkl_len"]) == 5
    np.testing.assert_array_equal(mats[3], res.clipper_num_associations)
    for r in range(2):
        segs = synth.map_segments_of([s.segments for s in submaps[r]])
        sa.write_submaps_json(tmp_path / f"{r}.sm.json", io.robot_names[r], segs, submaps[r])
        json_close(json.loads((tmp_path / f"{r}.sm.json").read_text()), json.loads(str(g[f"{name}/sm_json_{r}"])))
    return res


@pytest.mark.parametrize("name", list(synth.ALIGN_SCENARIOS))
def test_pair_loop_and_writers_match_reference_fixture(name, tmp_path):
    res = check_scenario(name, oracle_compute, tmp_path)
    assert len(res.timing_list) <= res.clipper_num_associations.size


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(synth.ALIGN_SCENARIOS))
def test_pair_loop_on_hip_matches_reference_fixture(name, tmp_path):
    check_scenario(name, None, tmp_path)                 # default compute: one roman_align_batch call


def test_gates_and_sentinels():
    """The three ways a pair skips registration ([REF roman/align/submap_align.py:136-149,179-184])."""
    g = np.load(GOLD, allow_pickle=False)
    params, io, submaps, _ = build("roman_descriptor")
    calls = []

    def compute(reg, batch):
        calls.append(len(batch))
        return oracle_compute(reg, batch)
    res = sa.submap_align(params, submaps, io, compute=compute)
    assert calls == [3]                                  # ONE batched call, only the ungated pairs
    assert np.all(res.clipper_num_associations[:, 0] == 0) and np.all(np.isnan(res.T_ij_hat_mat[:, 0]))     # descriptor gate
    assert np.all(res.clipper_dist_mat[:, 0] == 1e6) and np.all(res.clipper_angle_mat[:, 0] == np.rad2deg(180.0))   # the reference converts its 180 sentinel too
    assert np.all(res.clipper_num_associations[:, 2] == 0) and np.all(np.isnan(res.similarity_mat[:, 2]))   # skip_distance
    assert np.all(np.isnan(res.clipper_dist_mat[:, 2]))
    params, io, submaps, _ = build("gravity_gt")
    res = sa.submap_align(params, submaps, io, compute=oracle_compute)
    assert np.all(res.clipper_num_associations[:, 1] == 0)            # the empty submap: T_align raises in the reference
    assert np.all(np.isnan(res.robots_nearby_mat[:, 2])) and np.all(np.isnan(res.clipper_angle_mat[:, 2]))  # far: not "nearby"
    assert np.all(res.clipper_num_associations[:, 2] > 0)                                                   # but still registered


def test_each_submap_is_packed_once():
    params, io, submaps, _ = build("prune")
    reg = params.get_object_registration()
    packed = []
    orig = reg.pack
    reg.pack = lambda m: (packed.append(len(m)), orig(m))[1]
    sa.submap_align(params, submaps, io, registration=reg, compute=oracle_compute)
    assert len(packed) == 6                              # 3 + 3 submaps for 9 pairs (the reference packs 18 times)


def test_similarity_and_helpers():
    a = sa.Submap(0, 0.0, [], np.eye(4), descriptor=np.array([1.0, 0.0, 0.0]))
    b = sa.Submap(1, 0.0, [], np.eye(4), descriptor=np.array([1.0, 1.0, 0.0]))
    z = sa.Submap(2, 0.0, [], np.eye(4), descriptor=np.zeros(3))
    assert abs(sa.Submap.similarity(a, b) - 1 / np.sqrt(2)) < 1e-15 and sa.Submap.similarity(a, z) == 0.0
    s1 = sa.Submap(3, 0.0, [], np.eye(4), descriptor=np.array([[1.0, 0, 0], [0, 1.0, 0]]))
    s2 = sa.Submap(4, 0.0, [], np.eye(4), descriptor=np.array([[0, 0, 2.0], [0, 3.0, 0], [0, 0, 0]]))
    assert sa.Submap.similarity(s1, s2) == 1.0           # stacked descriptors: best pairwise cosine, zero rows ignored
    T = synth.yaw_transform(0.7, [1, 2, 3], roll=0.2, pitch=-0.1)
    flat = sa.transform_rm_roll_pitch(T)
    assert flat is T and abs(T[2, 2] - 1.0) < 1e-15 and abs(np.arctan2(T[1, 0], T[0, 0]) - 0.7) < 1e-12   # in place, yaw kept
    assert sa.aabb_intersects(np.array([[0, 0, 0], [1, 1, 1.0]]), np.array([[1, 1, 1], [2, 2, 2.0]]))
    assert not sa.aabb_intersects(np.array([[0, 0, 0], [1, 1, 1.0]]), np.array([[1.1, 0, 0], [2, 2, 2.0]]))
    assert sa.nearest_index([0.0, 0.5, 1.0], 0.7) == 1


@pytest.mark.gpu
def test_device_descriptor_gate_matches_pairwise_similarity():
    """Row f2: the S0 x S1 submap-descriptor cosines of one device call against Submap.similarity per pair."""
    from roman_amd.runtime import default_context
    rng = np.random.default_rng(5)
    A = rng.standard_normal((7, 48)); B = rng.standard_normal((5, 48))
    A[3] = 0.0                                           # a zero descriptor: similarity 0 by the guard
    S = default_context().cosine_matrix(A, B)
    for i in range(7):
        for j in range(5):
            ref = sa.Submap.similarity(sa.Submap(0, 0.0, [], np.eye(4), descriptor=A[i]), sa.Submap(1, 0.0, [], np.eye(4), descriptor=B[j]))
            assert abs(S[i, j] - ref) < 1e-12


@pytest.mark.gpu
def test_device_stacked_descriptor_gate_matches_pairwise_similarity():
    """Stacked per-frame descriptors ([REF roman/map/map.py:152-162]): one cosine kernel over all frames of all
    submaps + a segmented maximum equals Submap.similarity pair by pair (zero frames score 0)."""
    from roman_amd.runtime import default_context
    rng = np.random.default_rng(6)
    d0 = [rng.standard_normal((k, 24)) for k in (3, 1, 5, 2)]
    d1 = [rng.standard_normal((k, 24)) for k in (2, 4, 1)]
    d0[2][1] = 0.0; d1[1][3] = 0.0                       # zero frames
    d0[1][:] = 0.0                                       # a submap with only zero frames: similarity 0 with everything
    S = sa.stacked_similarity(default_context(), d0, d1)
    assert S.shape == (4, 3)
    for i, a in enumerate(d0):
        for j, b in enumerate(d1):
            ref = sa.Submap.similarity(sa.Submap(0, 0.0, [], np.eye(4), descriptor=a), sa.Submap(1, 0.0, [], np.eye(4), descriptor=b))
            assert abs(S[i, j] - ref) < 1e-12
    assert np.all(S[1] == 0.0)
