"""Synthetic submap-pair generator used by tests and bench.py.

Objects are exposed as attribute records with the duck-typed contract the reference's
registration plugins read from ``SegmentMinimalData``
([REF roman/object/segment.py:19-59]): ``center`` ((3,1) ndarray), ``volume``, ``linearity``,
``planarity``, ``scattering`` (properties), ``extent`` (3,), ``semantic_descriptor`` (unit norm,
[REF roman/object/segment.py:489]).  Scales follow the reference defaults: submap radius 15 m
[REF roman/params/submap_align_params.py:42], objects in a gravity-aligned submap-centred frame
[REF roman/map/map.py:328-330].
"""
from dataclasses import dataclass
import numpy as np


class SyntheticSegment:
    """Attribute-only stand-in for roman.object.segment.SegmentMinimalData."""

    def __init__(self, id, center, volume, linearity, planarity, scattering, extent,
                 semantic_descriptor):
        self.id = int(id)
        self.dim = 3
        self.centroid = np.asarray(center, dtype=np.float64).reshape(3, 1)
        self._volume = float(volume)
        self._linearity = float(linearity)
        self._planarity = float(planarity)
        self._scattering = float(scattering)
        self.extent = np.asarray(extent, dtype=np.float64)
        self.semantic_descriptor = semantic_descriptor
        self.first_seen = 0.0
        self.last_seen = 0.0

    @property
    def center(self):
        return self.centroid

    @property
    def volume(self):
        return self._volume

    @property
    def linearity(self):
        return self._linearity

    @property
    def planarity(self):
        return self._planarity

    @property
    def scattering(self):
        return self._scattering

    def normalized_eigenvalues(self):
        """Called (result unused) by the reference's clipper+prune plugin
        [REF roman/align/dist_reg_with_pruning.py:63]; a method there, as on PointCloudObject."""
        e0 = 1.0 / (1.0 + (1.0 - self._linearity) + self._scattering)
        return np.array([e0, e0 * (1.0 - self._linearity), e0 * self._scattering])


@dataclass
class SyntheticPair:
    map1: list            # List[SyntheticSegment]
    map2: list
    T_gt: np.ndarray      # 4x4, maps map-2 coordinates into map-1 coordinates
    inliers: np.ndarray   # (k,2) planted (map1 index, map2 index) pairs
    seed: int


def _sample_centroids(rng, n, radius=15.0, zlo=-1.0, zhi=3.0, min_sep=0.5, existing=None):
    """xy uniform in a disc, z uniform; rejection-sampled so all intra-map distances >= min_sep."""
    n0 = 0 if existing is None else len(existing)
    pts = np.zeros((n0 + n, 3))
    if n0:
        pts[:n0] = np.asarray(existing, dtype=np.float64).reshape(n0, 3)
    cnt = n0
    while cnt < n0 + n:
        r = radius * np.sqrt(rng.uniform())
        th = rng.uniform(-np.pi, np.pi)
        p = np.array([r * np.cos(th), r * np.sin(th), rng.uniform(zlo, zhi)])
        if cnt == 0 or np.min(np.linalg.norm(pts[:cnt] - p, axis=1)) >= min_sep:
            pts[cnt] = p
            cnt += 1
    return pts[n0:]


def _shape_attrs(rng):
    vol = float(rng.lognormal(0.0, 1.0))
    e = np.sort(rng.dirichlet([1.0, 1.0, 1.0]))[::-1]           # descending eigenvalues
    e = np.maximum(e, 1e-6)
    lin, pla, sca = (e[0] - e[1]) / e[0], (e[1] - e[2]) / e[0], e[2] / e[0]   # segment.py:445-472
    ext = np.sort(rng.uniform(0.2, 3.0, size=3))
    return vol, lin, pla, sca, ext


def _perturb_shape(rng, attrs):
    vol, lin, pla, sca, ext = attrs
    f = lambda x: float(max(x * (1.0 + 0.1 * rng.standard_normal()), 1e-6))
    return f(vol), f(lin), f(pla), f(sca), np.array([f(x) for x in ext])


def yaw_transform(yaw, t, roll=0.0, pitch=0.0):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T


def make_pair(n=200, m=200, d=512, seed=2000, inlier_frac=0.5, noise=0.1, n_classes=20,
              desc_noise=0.35, tilt_deg=0.0, roll_pitch_deg=None):
    """One synthetic submap pair.  `tilt_deg` draws a random roll / pitch of that
    scale; `roll_pitch_deg=(roll, pitch)` plants exact angles instead (gravity-check cases,
    [REF roman/align/dist_reg_with_pruning.py:38-44]) without consuming random draws."""
    rng = np.random.default_rng(seed)
    c1 = _sample_centroids(rng, n)
    yaw = rng.uniform(-np.pi, np.pi)
    roll = pitch = 0.0
    if tilt_deg > 0.0:
        roll, pitch = np.deg2rad(tilt_deg) * rng.standard_normal(2)
    if roll_pitch_deg is not None:
        roll, pitch = np.deg2rad(float(roll_pitch_deg[0])), np.deg2rad(float(roll_pitch_deg[1]))
    t = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-0.5, 0.5)])
    T_gt = yaw_transform(yaw, t, roll, pitch)
    T_inv = np.linalg.inv(T_gt)

    k = int(round(inlier_frac * min(n, m)))
    in1 = rng.choice(n, size=k, replace=False) if k > 0 else np.zeros(0, dtype=np.int64)
    c2_in = (T_inv[:3, :3] @ c1[in1].T).T + T_inv[:3, 3] + noise * rng.standard_normal((k, 3))
    c2_out = _sample_centroids(rng, m - k, existing=list(c2_in))
    c2 = np.vstack([c2_in, c2_out]) if m - k > 0 else c2_in

    protos = None
    if d > 0:
        protos = rng.standard_normal((n_classes, d))
        protos /= np.linalg.norm(protos, axis=1, keepdims=True)

    def descriptor(cls):
        if d <= 0:
            return None
        v = protos[cls] + desc_noise * rng.standard_normal(d) / np.sqrt(d)
        return v / np.linalg.norm(v)

    cls1 = rng.integers(0, n_classes, size=n)
    attrs1 = [_shape_attrs(rng) for _ in range(n)]
    map1 = [SyntheticSegment(i, c1[i], *attrs1[i][:4], attrs1[i][4], descriptor(cls1[i])) for i in range(n)]

    cls2 = np.concatenate([cls1[in1], rng.integers(0, n_classes, size=m - k)])
    attrs2 = [_perturb_shape(rng, attrs1[i]) for i in in1] + [_shape_attrs(rng) for _ in range(m - k)]
    perm = rng.permutation(m)                      # map2 position -> pre-permutation index
    map2 = [SyntheticSegment(jj, c2[src], *attrs2[src][:4], attrs2[src][4], descriptor(cls2[src]))
            for jj, src in enumerate(perm)]
    inv = np.empty(m, dtype=np.int64)
    inv[perm] = np.arange(m)
    inliers = np.stack([in1, inv[:k]], axis=1) if k > 0 else np.zeros((0, 2), dtype=np.int64)
    return SyntheticPair(map1, map2, T_gt, inliers, seed)


def make_submap_grid(n_submaps, n=200, d=512, seed0=4000, overlap=0.5, noise=0.1, n_classes=20,
                     desc_noise=0.35):
    """cfg4 building block: `n_submaps` submaps of one robot.  Every submap shares an
    `overlap` fraction of a common landmark set (so cross pairs have planted inliers) and is
    expressed in its own randomly yawed/translated frame.  Returns (list of object lists, list
    of 4x4 world-from-submap poses)."""
    rng0 = np.random.default_rng(seed0)
    n_common = int(round(overlap * n))
    world = _sample_centroids(rng0, n_common)
    protos = None
    if d > 0:
        protos = rng0.standard_normal((n_classes, d))
        protos /= np.linalg.norm(protos, axis=1, keepdims=True)
    w_cls = rng0.integers(0, n_classes, size=n_common)
    w_attrs = [_shape_attrs(rng0) for _ in range(n_common)]
    submaps, poses = [], []
    for s in range(n_submaps):
        rng = np.random.default_rng(seed0 + 1 + s)
        T_ws = yaw_transform(rng.uniform(-np.pi, np.pi),
                             [rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-0.5, 0.5)])
        T_sw = np.linalg.inv(T_ws)
        pts_common = (T_sw[:3, :3] @ world.T).T + T_sw[:3, 3] + noise * rng.standard_normal((n_common, 3))
        pts_own = _sample_centroids(rng, n - n_common, existing=list(pts_common))
        pts = np.vstack([pts_common, pts_own]) if n - n_common > 0 else pts_common
        cls = np.concatenate([w_cls, rng.integers(0, n_classes, size=n - n_common)])
        attrs = [_perturb_shape(rng, a) for a in w_attrs] + [_shape_attrs(rng) for _ in range(n - n_common)]
        perm = rng.permutation(n)
        objs = []
        for jj, src in enumerate(perm):
            dsc = None
            if d > 0:
                v = protos[cls[src]] + desc_noise * rng.standard_normal(d) / np.sqrt(d)
                dsc = v / np.linalg.norm(v)
            objs.append(SyntheticSegment(jj, pts[src], *attrs[src][:4], attrs[src][4], dsc))
        submaps.append(objs)
        poses.append(T_ws)
    return submaps, poses


# --------------------------------------------------------------------------------------------------
# scenarios for the submap-pair loop: two robots' submaps with odometry and
# ground-truth poses, timestamps and submap descriptors
# --------------------------------------------------------------------------------------------------
ALIGN_SCENARIOS = {
    # name: (params kwargs, io kwargs, generator options)
    "gravity_gt": (dict(method="gravity"), dict(gt_available=(True, True)),
                   dict(d=0, n=24, empty=True, far=True)),
    "roman_descriptor": (dict(method="roman", semantics_dim=16, submap_descriptor="mean_semantic", submap_descriptor_thresh=0.55),
                         dict(skip_distance=25.0, lc_association_thresh=5), dict(d=16, n=28, far=True)),
    "single_robot_fill": (dict(method="pcavolgrav", single_robot_lc=True, single_robot_lc_time_thresh=50.0,
                               force_fill_submaps=True, epsilon_shape=0.1, force_rm_lc_roll_pitch=False),
                          dict(), dict(d=0, n=26, shared_ids=True)),
    "prune": (dict(method="clipper+prune", cosine_min=0.5, epsilon_shape=0.05), dict(), dict(d=24, n=22)),
    "stacked_descriptors": (dict(method="semanticgrav", semantics_dim=12, submap_descriptor="stacked_frame_descriptors",
                                 submap_descriptor_thresh=0.6), dict(), dict(d=12, n=20, stacked=True)),
}


def make_align_scenario(name, seed0=7100):
    """-> (params_kwargs, io_kwargs, robots, trajectories): robots[r] is a list of dicts with the fields of the
    reference's Submap ([REF roman/map/map.py:94-103]: id, time, segments, pose_flu, pose_flu_gt, descriptor),
    trajectories[r] the (times, poses) of robot r's odometry that the `.g2o` writer indexes into."""
    pk, iok, opt = ALIGN_SCENARIOS[name]
    n_sub = (3, 3)
    subs, poses = make_submap_grid(sum(n_sub), n=opt["n"], d=opt["d"], seed0=seed0)
    rng = np.random.default_rng(seed0 + 999)
    robots, trajectories = [], []
    place = rng.standard_normal(max(opt["d"], 1)); place /= np.linalg.norm(place)
    s = 0
    for r in range(2):
        rob = []
        for k in range(n_sub[r]):
            T_gt = poses[s].copy()
            if opt.get("far") and r == 1 and k == 2:
                T_gt[:3, 3] += np.array([60.0, -45.0, 0.0])           # beyond 2 x submap_radius and skip_distance
            T_odom = T_gt @ yaw_transform(rng.normal(0, 0.05), rng.normal(0, 0.3, 3))      # drifted odometry
            tilt = yaw_transform(0.0, [0, 0, 0], roll=rng.normal(0, 0.03), pitch=rng.normal(0, 0.03))
            segs = subs[s]
            if opt.get("empty") and r == 1 and k == 1:
                segs = []
            if opt.get("shared_ids"):                                  # one robot revisiting: overlapping id ranges
                for q, sg in enumerate(segs):
                    sg.id = 10 * (3 * r + k) + q                       # neighbours share ids 10..(n-1) apart
            desc = None
            if opt["d"] > 0:                                           # same place seen twice: place vector + noise
                desc = place + 0.45 * rng.standard_normal(opt["d"]) / np.sqrt(opt["d"])
                if r == 1 and k == 0:
                    desc = rng.standard_normal(opt["d"])               # an unrelated place: the descriptor gate fires
                if opt.get("stacked"):                                 # several frame descriptors per submap: best pairwise cosine
                    rows = [desc] + [rng.standard_normal(opt["d"]) for _ in range(1 + (k + r) % 3)]
                    if k == 2:
                        rows.append(np.zeros(opt["d"]))                # a zero row is ignored by the similarity
                    desc = np.stack(rows)
            rob.append(dict(id=k, time=100.0 * r + 30.0 * k + 0.25 * (k + 1), segments=segs,
                            pose_flu=T_odom @ tilt, pose_flu_gt=T_gt @ tilt, descriptor=desc))
            s += 1
        robots.append(rob)
        times = np.arange(0.0, 400.0, 0.5) + 0.01 * r
        trajectories.append((times, [np.eye(4) for _ in times]))
    return dict(pk), dict(iok), robots, trajectories


def map_segments_of(submap_segment_lists):
    """The robot's whole-map segment list for the `sm.json` writer: every distinct segment of its submaps; every
    third one carries a small point cloud (`points`), the others do not (minimal data) and are skipped by the writer."""
    seen, out = set(), []
    for segs in submap_segment_lists:
        for sg in segs:
            if id(sg) in seen:
                continue
            seen.add(id(sg))
            if len(out) % 3 == 0:
                c = sg.center.reshape(-1)
                sg.points = c + 0.1 * np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, -1, -1.0]])
            out.append(sg)
    return out
