"""The C-ABI shared library loads and exports every symbol the header declares.
No compute calls (there is no GPU here); argument validation is exercised."""
import ctypes
import os
import re

import pytest

from cimrgp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "cimrgp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cimrgp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 17
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_workspace_size():
    lib = _lib.load()
    assert lib.cimrgp_version() >= 100
    # 64x64 inverse slabs + 256x256 inverse blocks of the diagonal
    assert lib.cimrgp_potrf_workspace_bytes(_lib.F64, 8192) == (128 * 64 * 64 + 32 * 256 * 256) * 8
    assert lib.cimrgp_potrf_workspace_bytes(_lib.F32, 65) == (2 * 64 * 64 + 1 * 256 * 256) * 4
    assert lib.cimrgp_potrf_workspace_bytes(_lib.F64, 0) == 0


def test_argument_errors_are_reported_not_crashed():
    lib = _lib.load()
    rc = lib.cimrgp_potrf(_lib.F64, None, 8, 8, None, 0, None, None)
    assert rc < 0 and "null pointer" in _lib.last_error()
    buf = (ctypes.c_double * 64)()
    p = ctypes.addressof(buf)
    rc = lib.cimrgp_potrf(7, p, 4, 4, p, 1 << 20, p, None)
    assert rc < 0 and "dtype" in _lib.last_error()
    rc = lib.cimrgp_potrf(_lib.F64, p, 4, 3, p, 1 << 20, p, None)
    assert rc < 0 and "dimension" in _lib.last_error()
    rc = lib.cimrgp_rbf_gram(_lib.F64, p, 4, 9, 1.0, 1.0, 0.0, p, 4, 0, None)
    assert rc < 0 and "dimension" in _lib.last_error()
    rc = lib.cimrgp_rbf_gram(_lib.F64, p, 4, 1, -1.0, 1.0, 0.0, p, 4, 0, None)
    assert rc < 0 and "length-scale" in _lib.last_error()
    with pytest.raises(_lib.CimrgpError):
        _lib.check(rc, "cimrgp_rbf_gram")
