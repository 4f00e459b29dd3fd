"""roman_amd — MI355X-native implementation of mit-acl/roman's object-map alignment hot path
(`roman.align`: affinity build -> CLIPPER dense-subgraph solve -> Umeyama pose).

Layout
  roman_amd.csrc        HIP kernels + the C ABI (libroman_hip.so, include/roman_hip.h)
  roman_amd.runtime     ctypes wrapper of the C ABI
  roman_amd.clipperpy   drop-in for the `clipperpy` module the reference imports
  roman_amd.align       the reference's ObjectRegistration plugin surface + batched / multi-GPU API
  roman_amd.synth       synthetic submap generator (tests, bench)

Importing this package never touches the GPU; the first numerical call creates a context and
fails loudly (RomanHipError) when libroman_hip.so or a gfx950 device is missing.
"""
import sys as _sys

from ._abi import RomanHipError, RomanParams, RomanStats  # noqa: F401

__version__ = "0.1.0"


def install_clipperpy_shim(force=False):
    """Register roman_amd.clipperpy as the top-level module `clipperpy`, so that the reference's
    unmodified files ([REF roman/align/object_registration.py:4], [REF roman/align/roman_registration.py:6],
    [REF roman/params/submap_align_params.py:20]) import the MI355X implementation."""
    from . import clipperpy as _shim
    if "clipperpy" in _sys.modules and not force and _sys.modules["clipperpy"] is not _shim:
        raise RuntimeError("a different `clipperpy` is already imported; pass force=True to replace it")
    _sys.modules["clipperpy"] = _shim
    _sys.modules["clipperpy.invariants"] = _shim.invariants
    _sys.modules["clipperpy.utils"] = _shim.utils
    return _shim
