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
    for name in ["robotdatapy", "robotdatapy.transform", "open3d"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["robotdatapy.transform"].transform = lambda T, x, **k: x
    sys.modules["robotdatapy"].transform = sys.modules["robotdatapy.transform"]
    # roman.object.segment / pointcloud_object pull cv2, shapely, ... : the plugins only use them
    # for type hints, so give dist_reg_with_pruning importable placeholders.
    for name, attrs in [("roman.object.pointcloud_object", ["PointCloudObject"]), ("roman.object.segment", ["Segment"])]:
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, type(a, (), {}))
        sys.modules[name] = m
    sys.path.insert(0, REF)


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


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present: golden fixtures can only be regenerated where /root/reference exists")
    install_reference_stubs()
    gen_t_align()
    gen_register()
