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
    # 64 x 64 inverses, 256 x 256 inverses, off-diagonal blocks of the 512 x 512 inverses (one per pair of full panels)
    assert lib.cimrgp_potrf_workspace_bytes(_lib.F64, 8192) == (128 * 64 * 64 + (32 + 16) * 256 * 256) * 8
    assert lib.cimrgp_potrf_workspace_bytes(_lib.F32, 700) == (11 * 64 * 64 + (3 + 1) * 256 * 256) * 4
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


def test_schedule_watchdog_is_not_reported_as_not_positive_definite():
    """include/cimrgp.h: CIMRGP_INFO_WATCHDOG in *info says the factorisation's internal schedule gave up waiting
    (a bounded device-side wait), not that a leading minor failed.  The reference's PD guard catches LinAlgError
    and repairs the matrix (src/SanityCheck.py:59-65): a schedule failure must surface as RuntimeError instead."""
    import numpy as np
    import torch
    from cimrgp_amd import device as dev
    text = open(os.path.join(ROOT, "include", "cimrgp.h")).read()
    m = re.search(r"#define\s+CIMRGP_INFO_WATCHDOG\s+(0x[0-9a-fA-F]+)", text)
    assert m and int(m.group(1), 16) == dev.INFO_WATCHDOG == 2 ** 31 - 1
    dev.raise_if_not_pd(0)
    dev.raise_if_not_pd(torch.zeros(1, dtype=torch.int32))
    with pytest.raises(np.linalg.LinAlgError, match="leading minor of order 5"):
        dev.raise_if_not_pd(5)
    with pytest.raises(RuntimeError, match="schedule watchdog"):
        dev.raise_if_not_pd(dev.INFO_WATCHDOG)
    with pytest.raises(RuntimeError, match="schedule watchdog"):
        dev.raise_if_not_pd(torch.tensor([dev.INFO_WATCHDOG], dtype=torch.int32))
    try:
        dev.raise_if_not_pd(dev.INFO_WATCHDOG)
    except np.linalg.LinAlgError:                      # RuntimeError is not a LinAlgError: the PD guard must not see it
        raise AssertionError("watchdog mapped to LinAlgError")
    except RuntimeError:
        pass
    # the flag crosses ranks inside a floating-point all-reduce (MRGP._fit): float32 rounds 2^31 - 1 up, sums grow
    assert dev.is_watchdog(np.float32(dev.INFO_WATCHDOG)) and dev.is_watchdog(float(dev.INFO_WATCHDOG) + 4096.0)
    assert not dev.is_watchdog(0) and not dev.is_watchdog(8 * 262144)


def test_collective_entry_points_validate_their_arguments():
    """cimrgp_comm_* / cimrgp_allreduce_sum (include/cimrgp.h): argument errors are reported before RCCL is touched
    (no GPU here; the reduce itself runs in tests/test_gpu_configs.py with a world of one)."""
    lib = _lib.load()
    assert lib.cimrgp_comm_unique_id(None) < 0 and "null pointer" in _lib.last_error()
    h = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    assert lib.cimrgp_comm_create(2, 2, ctypes.cast(ident, ctypes.c_void_p), ctypes.byref(h)) < 0
    assert "rank" in _lib.last_error()
    assert lib.cimrgp_comm_create(1, 0, None, ctypes.byref(h)) < 0 and "null pointer" in _lib.last_error()
    assert lib.cimrgp_allreduce_sum(None, _lib.F64, None, 4, None) < 0 and "communicator" in _lib.last_error()
    assert lib.cimrgp_comm_destroy(None) < 0
    with pytest.raises(ValueError):
        _lib.Comm(1, 0, b"short")
