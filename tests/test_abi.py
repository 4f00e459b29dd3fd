"""The C-ABI boundary without a GPU: the library loads, exports every symbol the header declares,
the ctypes mirror has the C layout, and GPU-needing entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from roman_amd import _abi

HEADER = os.path.join(ROOT, "include", "roman_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"ROMAN_API\s+(?:const\s+char\*|int)\s+(roman_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _abi.load_library()                     # raises if the .so is missing
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in roman_hip.h but not exported"
    assert set(syms) == set(_abi.EXPORTED_SYMBOLS)
    assert b"gfx950" in lib.roman_version()


def test_every_entry_point_cites_the_reference():
    src = open(HEADER).read()
    assert src.count("[REF ") >= 25


def test_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof from a C translation unit vs the ctypes mirror."""
    prog = tmp_path / "layout.c"
    fields_p = [f for f, _ in _abi.RomanParams._fields_]
    fields_s = [f for f, _ in _abi.RomanStats._fields_]
    body = "\n".join(f'printf("p.{f} %zu\\n", offsetof(roman_params_t, {f}));' for f in fields_p)
    body += "\n" + "\n".join(f'printf("s.{f} %zu\\n", offsetof(roman_stats_t, {f}));' for f in fields_s)
    prog.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(void){{\n'
                    f'printf("sizeof_p %zu\\nsizeof_s %zu\\n", sizeof(roman_params_t), sizeof(roman_stats_t));\n{body}\nreturn 0;}}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-o", str(exe), str(prog)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(out["sizeof_p"]) == C.sizeof(_abi.RomanParams)
    assert int(out["sizeof_s"]) == C.sizeof(_abi.RomanStats)
    for f in fields_p:
        assert int(out[f"p.{f}"]) == getattr(_abi.RomanParams, f).offset, f
    for f in fields_s:
        assert int(out[f"s.{f}"]) == getattr(_abi.RomanStats, f).offset, f


def test_params_default_matches_python_mirror_and_oracle(orc):
    lib = _abi.load_library()
    p = _abi.RomanParams()
    assert lib.roman_params_default(C.byref(p)) == 0
    assert p.as_dict() == _abi.RomanParams.default().as_dict() == orc.default_params().as_dict()
    # reference defaults: /root/reference/roman/params/submap_align_params.py:66-74, clipperpy.Params()
    assert (p.sigma, p.epsilon, p.mindist, p.cosine_min, p.cosine_max) == (0.4, 0.6, 0.2, 0.5, 0.7)
    assert (p.tol_u, p.tol_F, p.beta, p.maxiniters, p.maxoliters, p.maxlsiters, p.eps, p.affinityeps) == \
        (1e-8, 1e-9, 0.25, 200, 1000, 99, 1e-9, 1e-4)


def test_create_all_to_all_host_helper(orc):
    lib = _abi.load_library()
    out = np.zeros((12, 2), dtype=np.int32)
    assert lib.roman_create_all_to_all(3, 4, C.c_void_p(out.ctypes.data)) == 0
    assert np.array_equal(out, orc.create_all_to_all(3, 4))
    from roman_amd.clipperpy.utils import create_all_to_all
    assert np.array_equal(create_all_to_all(3, 4), out)
    assert create_all_to_all(0, 5).shape == (0, 2)


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback():
    from roman_amd.runtime import Context
    from roman_amd import RomanHipError
    with pytest.raises(RomanHipError, match="no HIP device|no CPU fallback"):
        Context(0)
    from roman_amd.align import SubmapAlignParams
    from roman_amd import synth
    reg = SubmapAlignParams(method="clipper").get_object_registration()
    pr = synth.make_pair(5, 5, 0, 1)
    with pytest.raises(RomanHipError):
        reg.register(pr.map1, pr.map2)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "roman_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, os.path.join(dp, f)
