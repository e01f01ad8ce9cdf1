"""The C-ABI boundary without a GPU: the HIP library loads, exports every entry point that
include/tedd.h declares, and the ctypes stub binds exactly that set.  No compute call is made."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tedd.h")


def declared_entry_points():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tedd_[a-z_0-9]+)\s*\(", text)))


def hip_library_path():
    from tedd import _ffi
    path = _ffi.DEFAULT_LIBRARY
    if not os.path.exists(path):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return path


def test_header_declares_entry_points():
    names = declared_entry_points()
    assert len(names) >= 30
    for must in ("tedd_create", "tedd_step_pc", "tedd_step_feuler", "tedd_advance", "tedd_error_norms",
                 "tedd_predict_cp_cs", "tedd_newton_step", "tedd_correct_cp_cs"):
        assert must in names


def test_hip_library_exports_every_declared_entry_point():
    lib = ctypes.CDLL(hip_library_path())
    missing = [n for n in declared_entry_points() if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_stub_binds_the_declared_set():
    from tedd import _ffi
    assert sorted(_ffi.PROTOTYPES) == declared_entry_points()


def test_product_path_fails_loudly_without_the_library(tmp_path):
    from tedd import _ffi
    with pytest.raises((OSError, RuntimeError)):
        _ffi.use_library(str(tmp_path / "no_such_libtedd_hip.so"))
        _ffi.library()
