#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE'S OWN Python code
(/root/reference, read-only) in this container.  Re-run with:  python tests/golden/make_golden.py

1. t_align_golden.npz — PINS the pose step: outputs of the reference's unmodified
   ObjectRegistration.T_align (/root/reference/roman/align/object_registration.py:88-129), imported
   with stub modules for the packages that are absent here (clipperpy, robotdatapy, open3d).
2. register_golden.npz — the reference's unmodified plugin classes and factory
   (ROMANRegistration, DistRegWithPruning, SubmapAlignParams.get_object_registration) driven over
   synthetic maps with the test-only oracle-backed `clipperpy` (tests/_oracle_clipperpy.py).  This
   pins the host-side mirror in roman_amd.align (feature packing, method->flag table, pruning)
   against the reference's Python; the CLIPPER arithmetic itself remains a restatement
   (PARITY UNPINNED, see oracle/clipper_oracle.c).

3. gravity_golden.npz — the roll/pitch check of DistRegWithPruning.register (row a10), see gen_gravity().
4. submap_align_golden.npz — the pair loop and the writers (rows f1/f3), see gen_submap_align().

/root/reference does not exist on the GPU box: tests only read the committed .npz files.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def install_reference_stubs():
    import _oracle_clipperpy
    _oracle_clipperpy.install()
    for name in ["robotdatapy", "robotdatapy.transform", "robotdatapy.data", "robotdatapy.data.pose_data", "open3d"]:
        sys.modules[name] = types.ModuleType(name)
    install_robotdatapy_restatement(sys.modules["robotdatapy.transform"], sys.modules["robotdatapy.data.pose_data"])
    sys.modules["robotdatapy"].transform = sys.modules["robotdatapy.transform"]
    # roman.object.segment / pointcloud_object pull cv2, shapely, ... : the plugins only use them
    # for type hints, so give dist_reg_with_pruning importable placeholders.
    for name, attrs in [("roman.object.pointcloud_object", ["PointCloudObject"]), ("roman.object.segment", ["Segment", "SegmentMinimalData"])]:
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, type(a, (), {}))
        sys.modules[name] = m
    sys.path.insert(0, REF)


def install_robotdatapy_restatement(tf, pd):
    """robotdatapy (pip dependency of the reference, [REF setup.py], absent here and not vendored) — the five
    helpers the pair loop and the writers call, restated from its published behaviour:
    transform(T, pts) applies a homogeneous transform to row-vector points; transform_to_xyzrpy = translation +
    scipy 'xyz' Euler angles; transform_to_xyz_quat = translation + scipy xyzw quaternion; transform_to_xytheta
    = planar pose; PoseData.idx(t, force_single=True) = index of the sample closest in time."""
    from scipy.spatial.transform import Rotation as Rot

    def transform(T, pts, **kw):
        pts = np.asarray(pts, dtype=np.float64)
        d = T.shape[0] - 1
        return pts @ T[:d, :d].T + T[:d, d]

    def transform_to_xyzrpy(T, degrees=False):
        return np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_euler('xyz', degrees=degrees)])

    def transform_to_xyz_quat(T, separate=False):
        t, q = T[:3, 3].copy(), Rot.from_matrix(T[:3, :3]).as_quat()
        return (t, q) if separate else np.concatenate([t, q])

    def transform_to_xytheta(T):
        return T[0, -1], T[1, -1], np.arctan2(T[1, 0], T[0, 0])

    class PoseData:
        def __init__(self, times, poses):
            self.times = np.asarray(times, dtype=np.float64); self.poses = poses

        @classmethod
        def from_times_and_poses(cls, times, poses, **kw):
            return cls(times, poses)

        @classmethod
        def from_csv(cls, **kw):                         # only `is not None` is tested on the ground-truth source
            return cls([], [])

        def idx(self, t, force_single=False):
            return int(np.argmin(np.abs(self.times - t)))

    tf.transform, tf.transform_to_xyzrpy, tf.transform_to_xyz_quat, tf.transform_to_xytheta = \
        transform, transform_to_xyzrpy, transform_to_xyz_quat, transform_to_xytheta
    pd.PoseData = PoseData


class Pt:
    """Minimal object with the attribute T_align reads (.center)."""

    def __init__(self, c):
        self.center = np.asarray(c, dtype=np.float64).reshape(-1, 1)


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def gen_t_align():
    from roman.align.object_registration import ObjectRegistration, InsufficientAssociationsException
    rng = np.random.default_rng(20240601)
    cases = []

    def add(dim, p1, p2, tag):
        reg = ObjectRegistration(dim=dim)
        m1 = [Pt(p) for p in p1]; m2 = [Pt(p) for p in p2]
        corr = np.stack([np.arange(len(p1)), np.arange(len(p2))], axis=1)
        try:
            T = reg.T_align(m1, m2, corr)
            ok = 1
        except InsufficientAssociationsException:
            T = np.full((dim + 1, dim + 1), np.nan); ok = 0
        cases.append((dim, np.asarray(p1, float), np.asarray(p2, float), T, ok, tag))

    for k in [3, 4, 5, 8, 20, 57, 100, 200]:                       # noisy rigid 3-D
        for rep in range(3):
            R = rot(rng.standard_normal(3), rng.uniform(-np.pi, np.pi)); t = rng.uniform(-5, 5, 3)
            p2 = rng.uniform(-15, 15, (k, 3))
            p1 = (R @ p2.T).T + t + 0.05 * rng.standard_normal((k, 3))
            add(3, p1, p2, f"rigid3d_k{k}")
    for k in [3, 6, 40]:                                           # exact (noise-free)
        R = rot(rng.standard_normal(3), rng.uniform(-np.pi, np.pi)); t = rng.uniform(-5, 5, 3)
        p2 = rng.uniform(-15, 15, (k, 3)); add(3, (R @ p2.T).T + t, p2, f"exact3d_k{k}")
    for k in [4, 10, 50]:                                          # reflected cloud -> det(U Vh) = -1 branch
        p2 = rng.uniform(-10, 10, (k, 3)); p1 = p2.copy(); p1[:, 2] *= -1.0
        p1 += 0.01 * rng.standard_normal((k, 3)); add(3, p1, p2, f"reflect3d_k{k}")
    for k in [5, 30]:                                              # planar clouds (rank-2 H)
        R = rot([0, 0, 1], rng.uniform(-np.pi, np.pi)); p2 = rng.uniform(-10, 10, (k, 3)); p2[:, 2] = 1.5
        add(3, (R @ p2.T).T + np.array([1.0, -2.0, 0.3]), p2, f"planar3d_k{k}")
    for k in [2, 3, 9, 64]:                                        # 2-D
        th = rng.uniform(-np.pi, np.pi); R2 = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        p2 = rng.uniform(-15, 15, (k, 2)); add(2, (R2 @ p2.T).T + rng.uniform(-3, 3, 2) + 0.02 * rng.standard_normal((k, 2)), p2, f"rigid2d_k{k}")
    p2 = rng.uniform(-5, 5, (6, 2)); p1 = p2.copy(); p1[:, 1] *= -1.0
    add(2, p1, p2, "reflect2d_k6")
    for k in [0, 1, 2]:                                            # insufficient (k < dim) in 3-D
        if k == 0:
            continue
        p = rng.uniform(-1, 1, (k, 3)); add(3, p, p, f"insufficient3d_k{k}")
    add(2, rng.uniform(-1, 1, (1, 2)), rng.uniform(-1, 1, (1, 2)), "insufficient2d_k1")

    out = {"n": len(cases)}
    for i, (dim, p1, p2, T, ok, tag) in enumerate(cases):
        out[f"dim{i}"] = dim; out[f"p1_{i}"] = p1; out[f"p2_{i}"] = p2; out[f"T_{i}"] = T; out[f"ok{i}"] = ok; out[f"tag{i}"] = tag
    np.savez_compressed(os.path.join(HERE, "t_align_golden.npz"), **out)
    print(f"t_align_golden.npz: {len(cases)} cases")


def gen_register():
    from roman.params.submap_align_params import SubmapAlignParams as RefParams
    from roman.align.object_registration import InsufficientAssociationsException
    from roman.align.dist_reg_with_pruning import GravityConstraintError
    from roman_amd import synth
    cases = []
    specs = [
        # (method, n, m, d, seed, extra SubmapAlignParams kwargs)
        ("clipper", 30, 30, 0, 1000, {}),
        ("clipper", 25, 40, 0, 1001, {}),
        ("gravity", 35, 30, 0, 1002, {}),
        ("pcavolgrav", 40, 40, 0, 1003, {"epsilon_shape": 0.2}),
        ("extentvolgrav", 30, 36, 0, 1004, {"epsilon_shape": 0.1}),
        ("roman", 40, 40, 48, 1005, {"semantics_dim": 48}),
        ("spvg", 30, 30, 32, 1006, {"semantics_dim": 32}),
        ("sevg", 36, 30, 24, 1007, {"semantics_dim": 24, "epsilon_shape": 0.25}),
        ("spv", 30, 30, 16, 1008, {}),
        ("semanticgrav", 60, 60, 64, 1009, {"semantics_dim": 64}),
        ("semanticgrav", 30, 45, 40, 1010, {"semantics_dim": 40, "cosine_min": 0.6, "cosine_max": 0.9}),
        ("roman_no_semantics", 30, 30, 0, 1011, {}),
        ("clipper+prune", 50, 50, 64, 1012, {"cosine_min": 0.5, "epsilon_shape": 0.1}),
        ("clipper+prune", 40, 30, 32, 1013, {"cosine_min": 0.6}),
        ("clipper", 20, 20, 0, 1014, {"dim": 2}),
        # the prefilter prunes EVERY association: the reference then hands clipperpy an empty list, which means all-to-all
        ("clipper+prune", 24, 24, 16, 1015, {"cosine_min": 0.9999}),
    ]
    for method, n, m, d, seed, kw in specs:
        sp = RefParams(method=method, **kw)
        reg = sp.get_object_registration()
        pr = synth.make_pair(n, m, max(d, 8) if method in ("clipper+prune",) else d, seed, tilt_deg=1.0 if "grav" in method or method in ("roman", "spvg", "sevg") else 0.0)
        if kw.get("dim") == 2:
            for o in pr.map1 + pr.map2:
                o.centroid = o.centroid[:2]; o.dim = 2
        status = "ok"
        try:
            assoc = np.asarray(reg.register(pr.map1, pr.map2)).astype(np.int64).reshape(-1, 2)
            T = reg.T_align(pr.map1, pr.map2, assoc) if len(assoc) >= reg.dim else np.full((reg.dim + 1,) * 2, np.nan)
        except InsufficientAssociationsException:
            assoc = np.zeros((0, 2), np.int64); T = np.full((reg.dim + 1,) * 2, np.nan); status = "insufficient"
        except GravityConstraintError:
            assoc = np.zeros((0, 2), np.int64); T = np.full((reg.dim + 1,) * 2, np.nan); status = "gravity"
        # the reference's own feature packing, for the host-mirror test
        m1 = np.array([reg._object_to_clipper_list(p) for p in pr.map1], dtype=np.float64)
        m2 = np.array([reg._object_to_clipper_list(p) for p in pr.map2], dtype=np.float64)
        import _oracle_clipperpy
        cases.append(dict(method=method, n=n, m=m, d=d, seed=seed, kw=repr(kw), assoc=assoc, T=T, status=status, pack1=m1, pack2=m2,
                          A_scored=_oracle_clipperpy.LAST["A"].astype(np.int32),
                          tilt=1.0 if "grav" in method or method in ("roman", "spvg", "sevg") else 0.0))
        print(f"  {method:20s} n={n} m={m} d={d} seed={seed}: k={len(assoc)} status={status}")
    out = {"n": len(cases)}
    for i, c in enumerate(cases):
        for k, v in c.items():
            out[f"{k}_{i}"] = v
    np.savez_compressed(os.path.join(HERE, "register_golden.npz"), **out)
    print(f"register_golden.npz: {len(cases)} cases")


GRAVITY_SPECS = [
    # (n, m, d, seed, SubmapAlignParams kwargs, (roll, pitch) in degrees planted in T_gt, centroid noise)
    (40, 40, 32, 1100, {"cosine_min": 0.5}, (8.0, 0.0), 0.05),        # roll above the 5 degree threshold -> raises
    (40, 40, 32, 1101, {"cosine_min": 0.5}, (0.0, 8.0), 0.05),        # pitch above -> raises
    (40, 40, 32, 1102, {"cosine_min": 0.5}, (0.0, -8.0), 0.05),       # negative pitch, |.| above -> raises
    (40, 36, 32, 1103, {"cosine_min": 0.5}, (4.9, 0.0), 0.0),         # just below (noise-free: the estimate IS 4.9) -> passes
    (40, 36, 32, 1104, {"cosine_min": 0.5}, (0.0, -4.9), 0.0),        # -> passes
    (36, 40, 32, 1105, {"cosine_min": 0.5}, (5.1, 0.0), 0.0),         # just above -> raises
    (36, 40, 32, 1106, {"cosine_min": 0.5}, (-4.9, 4.9), 0.0),        # both just below -> passes
    (36, 40, 32, 1107, {"cosine_min": 0.5}, (3.0, -7.0), 0.05),       # one below, one above -> raises
    (50, 50, 64, 1108, {"cosine_min": 0.5, "epsilon_shape": 0.1}, (0.0, 0.0), 0.1),   # no tilt -> passes
    (30, 30, 16, 1109, {"cosine_min": 0.6}, (12.0, 12.0), 0.02),      # both far above -> raises
]


def gen_gravity():
    """gravity_golden.npz — PINS row a10: the reference's unmodified DistRegWithPruning.register
    (/root/reference/roman/align/dist_reg_with_pruning.py:29-46, the roll/pitch check at :38-44 with scipy's
    as_euler('ZYX')) through the factory's method='clipper+prune' (use_gravity=True), on pairs whose planted
    roll / pitch lie on both sides of the 5 degree threshold.  Records whether the reference raised, the selected
    associations (solve without the check), the pose and the Euler angles scipy reported."""
    from scipy.spatial.transform import Rotation as Rot
    from roman.params.submap_align_params import SubmapAlignParams as RefParams
    from roman.align.dist_reg_with_pruning import GravityConstraintError
    from roman_amd import synth
    out = {"n": len(GRAVITY_SPECS)}
    for i, (n, m, d, seed, kw, rp, noise) in enumerate(GRAVITY_SPECS):
        reg = RefParams(method="clipper+prune", **kw).get_object_registration()
        assert reg.use_gravity
        pr = synth.make_pair(n, m, d, seed, noise=noise, roll_pitch_deg=rp)
        try:
            reg.register(pr.map1, pr.map2)
            raised = 0
        except GravityConstraintError:
            raised = 1
        reg.use_gravity = False                                    # the same solve without the check: what was selected
        assoc = np.asarray(reg.register(pr.map1, pr.map2)).astype(np.int64).reshape(-1, 2)
        T = reg.T_align(pr.map1, pr.map2, assoc)
        ypr = Rot.from_matrix(T[:3, :3]).as_euler('ZYX')
        assert raised == int(not (abs(ypr[2]) < reg.roll_pitch_thresh and abs(ypr[1]) < reg.roll_pitch_thresh))
        for k, v in dict(n=n, m=m, d=d, seed=seed, kw=repr(kw), roll=rp[0], pitch=rp[1], noise=noise, raised=raised,
                         assoc=assoc, T=T, ypr=ypr).items():
            out[f"{k}_{i}"] = v
        print(f"  clipper+prune planted roll/pitch {rp}: k={len(assoc)} estimated roll/pitch "
              f"{np.rad2deg(ypr[2]):+.3f}/{np.rad2deg(ypr[1]):+.3f} deg raised={raised}")
    np.savez_compressed(os.path.join(HERE, "gravity_golden.npz"), **out)
    print(f"gravity_golden.npz: {len(GRAVITY_SPECS)} cases")


def gen_submap_align():
    """submap_align_golden.npz — PINS rows f1/f3: the reference's unmodified pair loop
    (/root/reference/roman/align/submap_align.py:28-220) and result writers (roman/align/results.py:122-198)
    run over the scenarios of roman_amd.synth.make_align_scenario, with the map loader replaced by the
    synthetic submaps (reference `Submap` instances) and the oracle-backed clipperpy underneath."""
    import tempfile
    import pickle
    import yaml
    import matplotlib
    matplotlib.use("Agg")
    import roman.align.submap_align as ref_sa
    import roman.align.results as ref_res
    from roman.map.map import Submap as RefSubmap
    from roman.params.submap_align_params import SubmapAlignParams as RefParams, SubmapAlignInputOutput as RefIO
    from roman_amd import synth

    class FakeMap:                                       # what load_roman_map would have returned
        def __init__(self, submaps, times, traj):
            self.submaps, self.times, self.trajectory = submaps, list(times), traj
            self.segments = synth.map_segments_of([s.segments for s in submaps])

    ref_sa.load_roman_map = lambda m: m
    ref_sa.submaps_from_roman_map = lambda rm, sp, gt: rm.submaps
    ref_res.plot_align_results = lambda *a, **k: None
    ref_res.plt.savefig = lambda *a, **k: None
    out = {"names": np.array(list(synth.ALIGN_SCENARIOS))}
    for name in synth.ALIGN_SCENARIOS:
        pk, iok, robots, trajs = synth.make_align_scenario(name)
        tmp = tempfile.mkdtemp()
        gt = iok.pop("gt_available", (False, False))
        yamls = [None, None]
        for r in range(2):
            if gt[r]:
                yamls[r] = os.path.join(tmp, f"gt{r}.yaml")
                with open(yamls[r], "w") as f:
                    yaml.safe_dump({"type": "csv"}, f)
        maps = [FakeMap([RefSubmap(id=s["id"], time=s["time"], segments=s["segments"], pose_flu=s["pose_flu"],
                                   pose_flu_gt=s["pose_flu_gt"], descriptor=s["descriptor"]) for s in robots[r]],
                        *trajs[r]) for r in range(2)]
        io = RefIO(inputs=maps, output_dir=tmp, run_name="run", input_gt_pose_yaml=yamls, **iok)
        sp = RefParams(**pk)
        captured = {}
        real_save = ref_res.save_submap_align_results

        def save(results, submaps, roman_maps):
            captured["r"] = results
            # the results pickle embeds reference classes and the fake maps: not part of the fixture
            orig_dump = pickle.dump
            ref_res.pickle.dump = lambda obj, f: orig_dump(obj, f) if isinstance(obj, list) else None
            try:
                real_save(results, submaps, roman_maps)
            finally:
                ref_res.pickle.dump = orig_dump
        ref_sa.save_submap_align_results = save
        ref_sa.submap_align(sp, io)
        R = captured["r"]
        for k in ["robots_nearby_mat", "clipper_angle_mat", "clipper_dist_mat", "clipper_num_associations",
                  "submap_yaw_diff_mat", "T_ij_mat", "T_ij_hat_mat"]:
            out[f"{name}/{k}"] = getattr(R, k)
        out[f"{name}/similarity_mat"] = R.similarity_mat if R.similarity_mat is not None else np.zeros(0)
        out[f"{name}/has_similarity"] = R.similarity_mat is not None
        n0, n1 = R.clipper_num_associations.shape
        for i in range(n0):
            for j in range(n1):
                out[f"{name}/assoc_{i}_{j}"] = np.asarray(R.associated_objs_mat[i][j], dtype=np.int64).reshape(-1, 2) \
                    if np.size(R.associated_objs_mat[i][j]) else np.zeros((0, 2), np.int64)
        out[f"{name}/n_timed"] = len(R.timing_list)
        out[f"{name}/g2o"] = open(io.output_g2o).read()
        out[f"{name}/json"] = open(io.output_lc_json).read()
        out[f"{name}/timing"] = open(io.output_timing).read()
        for r in range(2):
            out[f"{name}/sm_json_{r}"] = open(io.output_submaps[r]).read()
        with open(io.output_matrix, "rb") as f:
            mats = pickle.load(f)
        out[f"{name}/matrix_pkl_len"] = len(mats)
        print(f"  {name:20s} pairs registered={len(R.timing_list)} edges={out[f'{name}/g2o'].count('EDGE_SE3')}"
              f" assoc counts={R.clipper_num_associations.astype(int).ravel().tolist()}")
    np.savez_compressed(os.path.join(HERE, "submap_align_golden.npz"), **out)
    print("submap_align_golden.npz written")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present: golden fixtures can only be regenerated where /root/reference exists")
    install_reference_stubs()
    which = sys.argv[1:] or ["t_align", "register", "gravity", "submap_align"]
    if "t_align" in which:
        gen_t_align()
    if "register" in which:
        gen_register()
    if "gravity" in which:
        gen_gravity()
    if "submap_align" in which:
        gen_submap_align()
