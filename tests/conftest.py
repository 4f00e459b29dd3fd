import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (built on demand with gcc)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(autouse=True)
def _gpu_tests_count_passes_the_devices_way(request):
    """The oracle's DEFAULT solver is the published organisation of the passes (CARRIED, oracle/clipper_oracle.c).  The GPU tests
    compare pass COUNTS with the device, whose stream solver fuses the line-search products and takes a split pass per d update:
    they opt into `auto` (fused for what the stream solver takes, carried otherwise) — explicitly, here, for every test marked
    `gpu`.  Selections do not depend on the mode; tests/test_gpu_full_configs.py also checks every config-3 / config-4 result against
    the default (carried) mode with plain arithmetic."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from oracle import oracle
    with oracle.pass_mode("auto"):
        yield


@pytest.fixture(scope="session")
def ctx():
    """A libroman_hip context on device 0.  No fallback: a GPU test without a GPU must fail."""
    from roman_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def load_npz_cases(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    n = int(z["n"])
    keys = sorted({k.rsplit("_", 1)[0] if k[-1].isdigit() and "_" in k else k for k in z.files if k != "n"})
    return z, n, keys


def registration_for(method, **kw):
    from roman_amd.align import SubmapAlignParams
    return SubmapAlignParams(method=method, **kw).get_object_registration()


def golden_register_cases():
    z = np.load(os.path.join(GOLDEN, "register_golden.npz"), allow_pickle=False)
    out = []
    for i in range(int(z["n"])):
        out.append(dict(method=str(z[f"method_{i}"]), n=int(z[f"n_{i}"]), m=int(z[f"m_{i}"]), d=int(z[f"d_{i}"]),
                        seed=int(z[f"seed_{i}"]), kw=ast.literal_eval(str(z[f"kw_{i}"])), assoc=z[f"assoc_{i}"], T=z[f"T_{i}"],
                        status=str(z[f"status_{i}"]), pack1=z[f"pack1_{i}"], pack2=z[f"pack2_{i}"],
                        A_scored=z[f"A_scored_{i}"], tilt=float(z[f"tilt_{i}"])))
    return out


def golden_pair(case):
    """Re-create the synthetic pair a register_golden case was generated from."""
    from roman_amd import synth
    d = max(case["d"], 8) if case["method"] == "clipper+prune" else case["d"]
    pr = synth.make_pair(case["n"], case["m"], d, case["seed"], tilt_deg=case["tilt"])
    if case["kw"].get("dim") == 2:
        for o in pr.map1 + pr.map2:
            o.centroid = o.centroid[:2]; o.dim = 2
    return pr


def golden_gravity_cases():
    """tests/golden/gravity_golden.npz: the reference's DistRegWithPruning.register on both sides of the roll/pitch threshold."""
    z = np.load(os.path.join(GOLDEN, "gravity_golden.npz"), allow_pickle=False)
    return [dict(n=int(z[f"n_{i}"]), m=int(z[f"m_{i}"]), d=int(z[f"d_{i}"]), seed=int(z[f"seed_{i}"]), kw=ast.literal_eval(str(z[f"kw_{i}"])),
                 roll=float(z[f"roll_{i}"]), pitch=float(z[f"pitch_{i}"]), noise=float(z[f"noise_{i}"]), raised=int(z[f"raised_{i}"]),
                 assoc=z[f"assoc_{i}"], T=z[f"T_{i}"], ypr=z[f"ypr_{i}"]) for i in range(int(z["n"]))]


def golden_gravity_pair(case):
    from roman_amd import synth
    return synth.make_pair(case["n"], case["m"], case["d"], case["seed"], noise=case["noise"], roll_pitch_deg=(case["roll"], case["pitch"]))


def golden_t_align_cases():
    z = np.load(os.path.join(GOLDEN, "t_align_golden.npz"), allow_pickle=False)
    return [dict(dim=int(z[f"dim{i}"]), p1=z[f"p1_{i}"], p2=z[f"p2_{i}"], T=z[f"T_{i}"], ok=int(z[f"ok{i}"]), tag=str(z[f"tag{i}"]))
            for i in range(int(z["n"]))]


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy(); ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2**63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2**63) - ib[ib < 0]
    return np.abs(ia - ib)
