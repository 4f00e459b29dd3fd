"""Submap-pair loop and result writers on the batched HIP path.

Mirrors the part of the reference's caller that surrounds the hot path:

  * `submap_align()` [REF roman/align/submap_align.py:74-220] — gating (distance / AABB, submap descriptor
    similarity, `skip_distance`, shared-segment removal for single-robot loop closures), `register()` +
    `T_align()` for every surviving pair, the gravity post-filters, the error metrics against the reference
    transform, and the result matrices.  Instead of the serial double loop with a device round trip per pair,
    all surviving pairs go to ONE `roman_align_batch` call and every submap is packed once.
  * `save_submap_align_results()` [REF roman/align/results.py:122-194] — the `.g2o` loop-closure edges
    (`# LC: <n>` + `EDGE_SE3:QUAT`), the loop-closure json, the matrix pickle and the timing text, byte for byte
    in the reference's formats, and the per-robot `sm.json` dump ([REF :200-243]).  Plots and the pickled results
    object (which embeds reference classes) are left to the reference.

Loading ROMAN maps and ground-truth trajectories (robotdatapy, `roman.map`) is out of scope: the caller hands
over submap objects exposing the attributes of [REF roman/map/map.py:94-141] (`segments`, `pose_flu`,
`pose_flu_gt`, `descriptor`, `time`); `Submap` below is a minimal stand-in with the same semantics, including
the reference's quirk that `pose_gravity_aligned` flattens `pose_flu` IN PLACE [REF roman/utils.py:128-130].
"""
import json
import pickle
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np
from scipy.spatial.transform import Rotation as Rot

from .. import _abi
from .batch import AlignmentBatch, pack_submaps, run_batch
from .dist_reg_with_pruning import _zyx_euler


# ---------------------------------------------------------------------------------------------
# small SE(3) helpers (the reference takes them from robotdatapy.transform / roman.utils)
# ---------------------------------------------------------------------------------------------
def transform_rm_roll_pitch(T):
    """[REF roman/utils.py:128-130] — keeps yaw only; MUTATES and returns its argument, like the reference."""
    T[:3, :3] = Rot.from_euler('z', Rot.from_matrix(T[:3, :3]).as_euler('ZYX')[0]).as_matrix()
    return T


def transform_to_xyzrpy(T):
    """x, y, z, roll, pitch, yaw (fixed-axis xyz Euler angles, radians)."""
    return np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_euler('xyz')])


def transform_to_xyz_quat(T):
    """translation (3,), quaternion (4,) in xyzw order."""
    return T[:3, 3].copy(), Rot.from_matrix(T[:3, :3]).as_quat()


def aabb_intersects(p1, p2):
    """[REF roman/utils.py:160-169]"""
    p1_min, p1_max, p2_min, p2_max = np.min(p1, axis=0), np.max(p1, axis=0), np.min(p2, axis=0), np.max(p2, axis=0)
    return bool(np.all(p1_min[:3] <= p2_max[:3]) and np.all(p1_max[:3] >= p2_min[:3]))


@dataclass
class Submap:
    """Minimal stand-in for [REF roman/map/map.py:94-141]."""
    id: int
    time: float
    segments: List
    pose_flu: np.ndarray
    pose_flu_gt: Optional[np.ndarray] = None
    descriptor: Optional[np.ndarray] = None

    @property
    def pose_gravity_aligned(self):
        return transform_rm_roll_pitch(self.pose_flu)

    @property
    def pose_gravity_aligned_gt(self):
        return transform_rm_roll_pitch(self.pose_flu_gt)

    @property
    def position(self):
        return self.pose_flu[:3, 3]

    @property
    def position_gt(self):
        return self.pose_flu_gt[:3, 3]

    @property
    def has_gt(self):
        return self.pose_flu_gt is not None

    @property
    def segments_as_global_points(self):
        T = self.pose_gravity_aligned_gt if self.has_gt else self.pose_gravity_aligned
        pts = np.vstack([np.asarray(seg.center).reshape(1, -1)[:, :3] for seg in self.segments])
        return pts @ T[:3, :3].T + T[:3, 3]

    def __len__(self):
        return len(self.segments)

    @classmethod
    def similarity(cls, submap1, submap2):
        """[REF roman/map/map.py:144-162]: cosine, or maximum pairwise cosine for stacked descriptors."""
        desc1, desc2 = np.asarray(submap1.descriptor), np.asarray(submap2.descriptor)
        if desc1.ndim == desc2.ndim == 1:
            norm_prod = np.linalg.norm(desc1) * np.linalg.norm(desc2)
            if np.isclose(norm_prod, 0.0, atol=1e-9, rtol=0.0):
                return 0.0
            return np.dot(desc1, desc2) / norm_prod
        d1 = desc1.reshape(desc1.shape[0], 1, desc1.shape[1]); d2 = desc2.reshape(1, desc2.shape[0], desc2.shape[1])
        norm_prods = np.linalg.norm(d1, axis=2) * np.linalg.norm(d2, axis=2)
        with np.errstate(invalid="ignore", divide="ignore"):
            sims = np.sum(d1 * d2, axis=2) / norm_prods
        sims[np.isclose(norm_prods, 0.0, atol=1e-9, rtol=0.0)] = 0.0
        return np.max(sims)


def stacked_similarity(ctx, descs0, descs1):
    """Best pairwise cosine between the frame descriptors of every submap of robot 0 and every submap of robot 1
    ([REF roman/map/map.py:152-162]) -> (S0, S1).  One cosine kernel over all frames, then a segmented maximum."""
    o0 = np.concatenate([[0], np.cumsum([d.shape[0] for d in descs0])]).astype(np.int64)
    o1 = np.concatenate([[0], np.cumsum([d.shape[0] for d in descs1])]).astype(np.int64)
    frames = ctx.cosine_matrix(np.concatenate(descs0, axis=0), np.concatenate(descs1, axis=0))     # zero-norm frames: 0
    return np.maximum.reduceat(np.maximum.reduceat(frames, o0[:-1], axis=0), o1[:-1], axis=1)


@dataclass
class SubmapAlignIO:
    """The fields of SubmapAlignInputOutput [REF roman/params/submap_align_params.py:153-198] the loop and the
    writers read."""
    robot_names: List[str] = field(default_factory=lambda: ["0", "1"])
    lc_association_thresh: int = 4
    g2o_t_std: float = 0.5
    g2o_r_std: float = float(np.deg2rad(0.5))
    skip_distance: float = np.inf
    gt_available: Sequence[bool] = (False, False)       # the reference tests `gt_pose_data[i] is not None`


@dataclass
class SubmapAlignResults:
    """Same fields as [REF roman/align/results.py:18-31]."""
    robots_nearby_mat: np.ndarray
    clipper_angle_mat: np.ndarray
    clipper_dist_mat: np.ndarray
    clipper_num_associations: np.ndarray
    similarity_mat: Optional[np.ndarray]
    submap_yaw_diff_mat: np.ndarray
    associated_objs_mat: list
    T_ij_mat: np.ndarray
    T_ij_hat_mat: np.ndarray
    timing_list: List[float]
    submap_align_params: object
    submap_io: object
    total_time: float = -np.inf


def submap_align(sm_params, submaps, sm_io: Optional[SubmapAlignIO] = None, registration=None,
                 compute: Optional[Callable] = None) -> SubmapAlignResults:
    """The pair loop of [REF roman/align/submap_align.py:74-220] over two lists of submaps, with ONE batched
    device call for all pairs that reach `register()`.

    `compute(registration, AlignmentBatch) -> runtime.BatchResult` defaults to the HIP path (`run_batch`); tests
    inject a CPU double.  `timing_list` gets the batch wall time divided evenly over the registered pairs (the
    reference times each `register()` call, [REF :155-157])."""
    sm_io = sm_io or SubmapAlignIO()
    registration = registration or sm_params.get_object_registration()
    compute = compute or run_batch
    n0, n1 = len(submaps[0]), len(submaps[1])
    nan = lambda *s: np.zeros(s) * np.nan
    clipper_angle_mat, clipper_dist_mat, clipper_num_associations = nan(n0, n1), nan(n0, n1), nan(n0, n1)
    similarity_mat, robots_nearby_mat, submap_yaw_diff_mat = nan(n0, n1), nan(n0, n1), nan(n0, n1)
    T_ij_mat, T_ij_hat_mat = nan(n0, n1, 4, 4), nan(n0, n1, 4, 4)
    associated_objs_mat = [[[] for _ in range(n1)] for _ in range(n0)]
    total_time_t0 = time.time()

    # ---- submap-descriptor gate (row f2): every cosine of the S0 x S1 gate in ONE device call (k_cos, f64 matrix
    # core).  Plain vector descriptors: the S0 x S1 cosine matrix itself.  Stacked per-frame descriptors
    # ([REF roman/map/map.py:152-162]: the best cosine over all frame pairs, zero-norm frames scoring 0): all frames
    # of all submaps go through the same kernel at once and the per-pair maximum is a segmented reduction of its
    # output.  (The CPU test double uses the per-pair numpy form below.) -----------------------------------------
    sim_all = None
    if sm_params.submap_descriptor is not None and compute is run_batch and n0 and n1:
        descs = [[np.asarray(sm.descriptor) for sm in submaps[r]] for r in range(2)]
        flat = [d for r in range(2) for d in descs[r]]
        if all(d.ndim == 1 for d in flat):
            sim_all = registration._context().cosine_matrix(np.stack(descs[0]), np.stack(descs[1]))
        elif all(d.ndim == 2 and d.shape[0] > 0 and d.shape[1] == flat[0].shape[1] for d in flat):
            sim_all = stacked_similarity(registration._context(), descs[0], descs[1])

    # ---- pass 1: gating, reference transforms, the list of pairs to register ([REF :93-149]) -------------------
    todo = []                                            # (i, j, segs_i, segs_j)
    skipped_sim = []
    for i in range(n0):
        for j in range(n1):
            si, sj = submaps[0][i], submaps[1][j]
            if si.has_gt and sj.has_gt:
                submap_distance = np.linalg.norm(si.position_gt - sj.position_gt)
            else:
                submap_distance = np.linalg.norm(si.position - sj.position)
            if (not sm_params.force_fill_submaps and sm_params.submap_radius is not None and submap_distance < sm_params.submap_radius * 2) or \
                    ((sm_params.force_fill_submaps or sm_params.submap_radius is None) and len(si) and len(sj)
                     and aabb_intersects(si.segments_as_global_points, sj.segments_as_global_points)):
                robots_nearby_mat[i, j] = submap_distance
            segs_i, segs_j = list(si.segments), list(sj.segments)
            if sm_params.single_robot_lc:                # self loop closures: drop the segments both submaps hold
                common = {seg.id for seg in segs_i} & {seg.id for seg in segs_j}
                segs_i = [s for s in segs_i if s.id not in common]; segs_j = [s for s in segs_j if s.id not in common]
            T_wi = si.pose_gravity_aligned_gt if sm_io.gt_available[0] else si.pose_gravity_aligned
            T_wj = sj.pose_gravity_aligned_gt if sm_io.gt_available[1] else sj.pose_gravity_aligned
            T_ij = np.linalg.inv(T_wi) @ T_wj
            if not np.isnan(robots_nearby_mat[i, j]):
                submap_yaw_diff_mat[i, j] = np.abs(np.rad2deg(transform_to_xyzrpy(T_ij)[5]))
            submap_sim = np.inf if sm_params.submap_descriptor is None else (sim_all[i, j] if sim_all is not None else Submap.similarity(si, sj))
            T_ij_mat[i, j] = T_ij
            if submap_distance > sm_io.skip_distance:
                clipper_num_associations[i, j] = 0
                T_ij_hat_mat[i, j] = nan(4, 4)
                continue
            similarity_mat[i, j] = submap_sim
            if submap_sim < sm_params.submap_descriptor_thresh:
                skipped_sim.append((i, j, len(segs_i), len(segs_j)))
            else:
                todo.append((i, j, segs_i, segs_j))

    # ---- the hot path: every submap (variant) packed once, one batched call ---------------------------------
    timing_list = []
    if todo:
        pool, index = [], {}

        def slot(key, segs):
            if key not in index:
                index[key] = len(pool); pool.append(segs)
            return index[key]
        shared = not sm_params.single_robot_lc           # without id removal a submap has ONE segment list
        ii = [slot((0, i) if shared else (0, i, j), si_) for (i, j, si_, _) in todo]
        jj = [slot((1, j) if shared else (1, i, j), sj_) for (i, j, _, sj_) in todo]
        feats, offs = pack_submaps(registration, pool)
        lens = np.diff(offs).astype(np.int32)
        batch = AlignmentBatch(feats, offs[ii].astype(np.int64), lens[ii], offs[jj].astype(np.int64), lens[jj])
        lists = [registration._association_list(a, b) if (len(a) and len(b)) else None for (_, _, a, b) in todo]
        if any(l is not None for l in lists):            # pruning plugins score explicit association lists
            from ..clipperpy.utils import create_all_to_all
            lists = [l if l is not None else create_all_to_all(len(a), len(b)) for l, (_, _, a, b) in zip(lists, todo)]
            batch.assoc_off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int64)
            batch.assoc = np.concatenate(lists, axis=0).astype(np.int32)
        t0 = time.time()
        res = compute(registration, batch)
        timing_list = [(time.time() - t0) / len(todo)] * len(todo)

    # ---- pass 2: post-filters, error metrics, result matrices ([REF :160-200]) ---------------------------------
    def report(i, j, T_ij_hat, theta, dist, associations, len_i, len_j):
        if not np.isnan(robots_nearby_mat[i, j]):
            clipper_angle_mat[i, j] = np.abs(np.rad2deg(theta)); clipper_dist_mat[i, j] = dist
        clipper_num_associations[i, j] = len(associations)
        T_ij_hat_mat[i, j] = T_ij_hat
        associated_objs_mat[i][j] = associations

    prune_tilt = registration.roll_pitch_thresh if getattr(registration, "use_gravity", False) else None
    for (i, j, li, lj) in skipped_sim:
        report(i, j, nan(4, 4), 180.0, 1e6, [], li, lj)
    for b, (i, j, segs_i, segs_j) in enumerate(todo):
        T_ij = T_ij_mat[i, j]
        failed = bool(res.status[b] & (_abi.ROMAN_ST_INSUFFICIENT | _abi.ROMAN_ST_EMPTY_MAP))   # T_align would raise
        associations = res.assoc[b]
        if not failed:
            T_ij_hat = np.array(res.T[b], dtype=np.float64)
            if prune_tilt is not None:                   # DistRegWithPruning.register's own check [REF dist_reg_with_pruning.py:38-45]
                _, pitch, roll = _zyx_euler(T_ij_hat[:sm_params.dim, :sm_params.dim])
                failed = not (np.abs(roll) < prune_tilt and np.abs(pitch) < prune_tilt)
        if not failed:
            if sm_params.dim == 2:
                # The reference multiplies a 3x3 estimate into the 4x4 reference transform here and cannot run
                # ([REF :159-162]); the planar estimate is lifted to SE(3) (identity in z) instead.
                T2 = T_ij_hat; T_ij_hat = np.eye(4); T_ij_hat[:2, :2] = T2[:2, :2]; T_ij_hat[:2, 3] = T2[:2, 2]
                T_error = np.linalg.inv(T_ij_hat) @ T_ij
                theta = np.arctan2(T_error[1, 0], T_error[0, 0]); dist = np.linalg.norm(T_error[:2, 3])
            else:
                if sm_params.force_rm_upside_down:       # GravityConstraintError branch [REF :167-170]
                    xyzrpy = transform_to_xyzrpy(T_ij_hat)
                    failed = bool(np.abs(xyzrpy[3]) > np.deg2rad(90.) or np.abs(xyzrpy[4]) > np.deg2rad(90.))
                if not failed:
                    if sm_params.force_rm_lc_roll_pitch:
                        T_ij_hat = transform_rm_roll_pitch(T_ij_hat)
                    T_error = np.linalg.inv(T_ij_hat) @ T_ij
                    theta = Rot.from_matrix(T_error[:3, :3]).magnitude(); dist = np.linalg.norm(T_error[:3, 3])
        if failed:                                       # the except-branch sentinel [REF :179-184]
            T_ij_hat, theta, dist, associations = nan(4, 4), 180.0, 1e6, []
        report(i, j, T_ij_hat, theta, dist, associations, len(segs_i), len(segs_j))

    return SubmapAlignResults(
        robots_nearby_mat=robots_nearby_mat, clipper_angle_mat=clipper_angle_mat, clipper_dist_mat=clipper_dist_mat,
        clipper_num_associations=clipper_num_associations,
        similarity_mat=similarity_mat if sm_params.submap_descriptor is not None else None,
        submap_yaw_diff_mat=submap_yaw_diff_mat, T_ij_mat=T_ij_mat, T_ij_hat_mat=T_ij_hat_mat,
        associated_objs_mat=associated_objs_mat, timing_list=timing_list, submap_align_params=sm_params,
        submap_io=sm_io, total_time=time.time() - total_time_t0)


# ---------------------------------------------------------------------------------------------
# writers (row f3): the wire formats g2o_file_fusion / Kimera-RPGO consume
# ---------------------------------------------------------------------------------------------
def nearest_index(times, t):
    """Index of the trajectory sample closest to time t (what robotdatapy's PoseData.idx(t, force_single=True) returns)."""
    times = np.asarray(times, dtype=np.float64)
    return int(np.argmin(np.abs(times - t)))


def loop_closure_edges(results: SubmapAlignResults, submaps):
    """The (i, j, T_pi_pj) triples the reference writes ([REF roman/align/results.py:156-171]): pairs with at
    least `lc_association_thresh` associations (and far enough apart in time for single-robot runs), with the
    estimated submap-centre transform composed into pose-frame i -> pose-frame j."""
    out = []
    p, io = results.submap_align_params, results.submap_io
    for i in range(len(submaps[0])):
        for j in range(len(submaps[1])):
            if not (results.clipper_num_associations[i, j] >= io.lc_association_thresh):
                continue
            if np.abs(submaps[0][i].time - submaps[1][j].time) < p.single_robot_lc_time_thresh and p.single_robot_lc:
                continue
            T_ci_cj = results.T_ij_hat_mat[i, j]
            T_odomi_ci = submaps[0][i].pose_gravity_aligned
            T_odomj_cj = submaps[1][j].pose_gravity_aligned
            T_odomi_pi = submaps[0][i].pose_flu
            T_odomj_pj = submaps[1][j].pose_flu
            T_pi_pj = np.linalg.inv(T_odomi_pi) @ T_odomi_ci @ T_ci_cj @ np.linalg.inv(T_odomj_cj) @ T_odomj_pj
            out.append((i, j, T_pi_pj))
    return out


def write_g2o(path, results: SubmapAlignResults, submaps, trajectory_times):
    """`.g2o` loop closures, same text as [REF roman/align/results.py:156-194]: per edge a `# LC: <n>` comment
    (read by g2o_file_fusion, [REF roman/offline_rpgo/g2o_file_fusion.py:54-68]) and an `EDGE_SE3:QUAT` line with
    the upper triangle of the information matrix.  trajectory_times[r]: pose timestamps of robot r's odometry."""
    io = results.submap_io
    I_t, I_r = 1 / (io.g2o_t_std ** 2), 1 / (io.g2o_r_std ** 2)
    I = np.diag([I_t, I_t, I_t, I_r, I_r, I_r])
    with open(path, 'w') as f:
        for (i, j, T) in loop_closure_edges(results, submaps):
            t, q = transform_to_xyz_quat(T)
            idx_a = nearest_index(trajectory_times[0], submaps[0][i].time)
            idx_b = nearest_index(trajectory_times[1], submaps[1][j].time)
            f.write(f"# LC: {int(results.clipper_num_associations[i, j])}\n")
            f.write(f"EDGE_SE3:QUAT a{idx_a} b{idx_b} \t")
            f.write(f"{t[0]} {t[1]} {t[2]} \t")
            f.write(f"{q[0]} {q[1]} {q[2]} {q[3]} \t")
            for ii in range(6):
                for jj in range(6):
                    if jj < ii:
                        continue
                    f.write(f"{I[ii, jj]} ")
                f.write("\t")
            f.write("\n")


def write_lc_json(path, results: SubmapAlignResults, submaps):
    """Loop-closure json, same records as [REF roman/align/results.py:172-179,196-198]."""
    out = []
    for (i, j, T) in loop_closure_edges(results, submaps):
        t, q = transform_to_xyz_quat(T)
        out.append({
            'seconds': [int(submaps[0][i].time), int(submaps[1][j].time)],
            'nanoseconds': [int((submaps[0][i].time % 1) * 1e9), int((submaps[1][j].time % 1) * 1e9)],
            'names': results.submap_io.robot_names,
            'translation': t.tolist(),
            'rotation': q.tolist(),
            'rotation_convention': 'xyzw',
        })
    with open(path, 'w') as f:
        json.dump(out, f, indent=4)


def write_matrix_pickle(path, results: SubmapAlignResults):
    """[REF roman/align/results.py:128-132]: the five result matrices as one pickled list."""
    with open(path, 'wb') as f:
        pickle.dump([results.robots_nearby_mat, results.clipper_angle_mat, results.clipper_dist_mat,
                     results.clipper_num_associations, results.submap_yaw_diff_mat], f)


def write_timing(path, results: SubmapAlignResults, submaps):
    """[REF roman/align/results.py:139-144]"""
    with open(path, 'w') as f:
        f.write(f"Total number of submaps: {len(submaps[0])} x {len(submaps[1])} = {len(submaps[0])*len(submaps[1])}\n")
        f.write(f"Average time per registration: {np.mean(results.timing_list):.4f} seconds\n")
        f.write(f"Total time: {np.sum(results.timing_list):.4f} seconds\n")
        f.write(f"Total number of objects: {np.sum([len(submap) for submap in submaps[0] + submaps[1]])}\n")
        f.write(f"Average number of obects per map: {np.mean([len(submap) for submap in submaps[0] + submaps[1]]):.2f}\n")


def write_submaps_json(path, robot_name, map_segments, robot_submaps):
    """Per-robot `<name>.sm.json` ([REF roman/align/results.py:200-243]): one record per map segment that carries
    a point cloud (segments without one are skipped, as the reference's bare `except: continue` does) and one
    per submap with its gravity-aligned pose."""
    sm_json = {'segments': [], 'submaps': []}
    secs_nsecs = lambda t: {'seconds': int(t), 'nanoseconds': int((t - int(t)) * 1e9)}
    for segment in map_segments:
        try:
            sm_json['segments'].append({
                'robot_name': robot_name,
                'segment_index': segment.id,
                'centroid_odom': np.mean(segment.points, axis=0).tolist(),
                'shape_attributes': {'volume': segment.volume, 'linearity': segment.linearity,
                                     'planarity': segment.planarity, 'scattering': segment.scattering},
                'first_seen': secs_nsecs(segment.first_seen),
                'last_seen': secs_nsecs(segment.last_seen)})
        except Exception:
            continue
    for j, sm in enumerate(robot_submaps):
        x = np.concatenate(transform_to_xyz_quat(sm.pose_gravity_aligned))
        sm_json['submaps'].append({
            'submap_index': j,
            'T_odom_submap': {'tx': x[0], 'ty': x[1], 'tz': x[2], 'qx': x[3], 'qy': x[4], 'qz': x[5], 'qw': x[6]},
            'robot_name': robot_name,
            'seconds': int(sm.time),
            'nanoseconds': int((sm.time % 1) * 1e9),
            'segment_indices': [segment.id for segment in sm.segments]})
    with open(path, 'w') as f:
        json.dump(sm_json, f, indent=4)
