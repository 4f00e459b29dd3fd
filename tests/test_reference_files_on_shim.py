"""INTEGRATION.md §1 executed, not argued: the reference's UNMODIFIED files
    /root/reference/roman/params/submap_align_params.py   (the factory)
    /root/reference/roman/align/roman_registration.py     /root/reference/roman/align/dist_reg_with_pruning.py
    /root/reference/roman/align/object_registration.py    (register, T_align, get_MCA, mno_clipper)
are imported in a fresh interpreter with `roman_amd.clipperpy` installed as `clipperpy` and run over the synthetic pairs
of tests/golden/register_golden.npz.  Behind the shim sits tests/_recording_lib.py: the C ABI's argument layout answered
by the CPU oracle (this box has no GPU), logging every entry point.  Checked: (1) the reference files import and run
against the shim's surface at all; (2) the associations and poses equal the committed golden ones (which the same files
produced over the oracle-backed test module: so the shim + runtime marshalling lose nothing); (3) one register() is
exactly the C-ABI sequence EXPECTED_REGISTER_CALLS — the sequence tests/test_gpu_shim.py records against the real
library on the GPU box.  Skipped where /root/reference does not exist (the GPU box)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, golden_register_cases

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "roman", "align")), reason="the reference checkout is not on this box")

DRIVER = textwrap.dedent("""
    import json, sys, types
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tests'); sys.path.insert(0, %(root)r + '/tests/golden')
    import make_golden                                            # stubs for robotdatapy / open3d / roman.object.* only
    make_golden.install_reference_stubs()
    import roman_amd
    roman_amd.install_clipperpy_shim(force=True)                 # `import clipperpy` in the reference files now finds the shim
    from roman_amd import _abi, runtime
    from oracle import oracle as orc
    from _recording_lib import RecordingLib
    lib = RecordingLib(orc)
    _abi._LIB = lib                                               # what load_library() returns: the C ABI, recorded
    import clipperpy
    assert clipperpy is roman_amd.clipperpy
    from roman.params.submap_align_params import SubmapAlignParams as RefParams           # the reference's own, unmodified
    from roman.align.object_registration import InsufficientAssociationsException
    from roman.align.dist_reg_with_pruning import GravityConstraintError
    import roman.align.object_registration as ref_or
    assert ref_or.__file__.startswith('/root/reference/'), ref_or.__file__
    from conftest import golden_register_cases, golden_pair
    out = []
    for case in golden_register_cases():
        reg = RefParams(method=case['method'], **case['kw']).get_object_registration()
        pr = golden_pair(case)
        lib.calls.clear()
        status = 'ok'
        try:
            assoc = np.asarray(reg.register(pr.map1, pr.map2)).astype(np.int64).reshape(-1, 2)
            calls = list(lib.calls)
            T = reg.T_align(pr.map1, pr.map2, assoc) if len(assoc) >= reg.dim else np.full((reg.dim + 1,) * 2, np.nan)
        except InsufficientAssociationsException:
            assoc = np.zeros((0, 2), np.int64); T = np.full((reg.dim + 1,) * 2, np.nan); status = 'insufficient'; calls = list(lib.calls)
        except GravityConstraintError:
            assoc = np.zeros((0, 2), np.int64); T = np.full((reg.dim + 1,) * 2, np.nan); status = 'gravity'; calls = list(lib.calls)
        sc = lib.scored
        out.append(dict(method=case['method'], assoc=assoc.tolist(), T=np.nan_to_num(T, nan=-7.0).tolist(), status=status, calls=calls,
                        D1_equals_pack=bool(np.array_equal(sc['D1'], case['pack1'])), D2_equals_pack=bool(np.array_equal(sc['D2'], case['pack2'])),
                        A_scored=(None if sc['A'] is None else sc['A'].tolist()), invariant=sc['params']['invariant']))
    # get_MCA + mno_clipper on the reference's base class ([REF roman/align/object_registration.py:50-86])
    reg = RefParams(method='clipper').get_object_registration()
    from roman_amd import synth
    pr = synth.make_pair(14, 12, 0, 77)
    reg.register(pr.map1, pr.map2)
    lib.calls.clear()
    M, Cm, A = reg.get_MCA(pr.map1, pr.map2)
    mca_calls = list(lib.calls)
    lib.calls.clear()
    sols = reg.mno_clipper(pr.map1, pr.map2, num_solutions=2)
    print('RESULT' + json.dumps(dict(cases=out, mca_calls=mca_calls, mca_shape=list(np.shape(M)), mno_calls=list(lib.calls),
                                     mno=[[np.asarray(s_[0]).tolist(), float(s_[1])] if isinstance(s_, (tuple, list)) else None for s_ in (sols if isinstance(sols, (list, tuple)) else [])])))
""")


@pytest.fixture(scope="module")
def run():
    code = DRIVER % dict(root=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert lines, out.stdout[-3000:] + out.stderr[-6000:]
    return json.loads(lines[-1][len("RESULT"):])


def test_reference_files_reproduce_the_golden_results_through_the_shim(run):
    gold = golden_register_cases()
    assert len(run["cases"]) == len(gold)
    for got, want in zip(run["cases"], gold):
        assert got["method"] == want["method"] and got["status"] == want["status"], (got["method"], got["status"], want["status"])
        assert np.array_equal(np.array(got["assoc"]).reshape(-1, 2), want["assoc"]), got["method"]
        assert np.allclose(np.array(got["T"]), np.nan_to_num(want["T"], nan=-7.0), atol=1e-12), got["method"]
        # what reached the C ABI is what the reference packed (object-major copies of its transposed views) and its association list
        assert got["D1_equals_pack"] and got["D2_equals_pack"], got["method"]
        if got["A_scored"] is not None:
            assert np.array_equal(np.array(got["A_scored"]).reshape(-1, 2), want["A_scored"]), got["method"]


def test_one_register_is_the_documented_c_abi_sequence(run):
    from _recording_lib import EXPECTED_REGISTER_CALLS
    first = True
    for got in run["cases"]:
        calls = [c for c in got["calls"]]
        if first:                                                 # the process-wide context is created by the first call
            assert calls[0] == "roman_ctx_create"; calls = calls[1:]; first = False
        if got["method"] == "clipper+prune" and got["status"] != "insufficient":
            # DistRegWithPruning.register = the base register() + T_align for the gravity check (numpy in the reference)
            assert calls == EXPECTED_REGISTER_CALLS, (got["method"], calls)
        else:
            assert calls == EXPECTED_REGISTER_CALLS, (got["method"], calls)


def test_get_mca_and_mno_clipper_of_the_reference_base_class(run):
    assert run["mca_shape"] == [14 * 12, 14 * 12]
    assert run["mca_calls"].count("roman_score") == 1 and run["mca_calls"].count("roman_get_dense_matrices") == 2
    assert run["mno_calls"].count("roman_set_matrix_data") == 2 and run["mno_calls"].count("roman_solve") == 2
