"""ctypes loader of ``libcimrgp.so`` (the C ABI declared in include/cimrgp.h).

The product path has no CPU fallback: if the HIP library is missing or a call
fails, an exception is raised -- never a silent NumPy substitute.
"""
import ctypes as C
import os

F32, F64 = 0, 1
NB = 256

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CIMRGP_LIB_PATH", os.path.join(_HERE, "libcimrgp.so"))   # override: A/B builds of the library

_vp, _i64, _i32, _dbl, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/cimrgp.h
SIGNATURES = {
    "cimrgp_version": (_i32, []),
    "cimrgp_last_error": (C.c_char_p, []),
    "cimrgp_device_count": (_i32, []),
    "cimrgp_rbf_gram": (_i32, [_i32, _vp, _i64, _i32, _dbl, _dbl, _dbl, _vp, _i64, _i32, _vp]),
    "cimrgp_rbf_cross": (_i32, [_i32, _vp, _i64, _vp, _i64, _i32, _dbl, _dbl, _vp, _i64, _vp]),
    "cimrgp_potrf_workspace_bytes": (_sz, [_i32, _i64]),
    "cimrgp_potrf": (_i32, [_i32, _vp, _i64, _i64, _vp, _sz, _vp, _vp]),
    "cimrgp_potrf_rows": (_i32, [_i32, _vp, _i64, _i64, _vp, _sz, _vp, _vp, _i64, _i64, _vp]),
    "cimrgp_potrf_rows_batched": (_i32, [_i32, _vp, _i64, _i64, _i64, _vp, _sz, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "cimrgp_solve_lt_batched": (_i32, [_i32, _vp, _i64, _i64, _i64, _vp, _sz, _vp, _i32, _vp, _i32, _vp]),
    "cimrgp_solve_lt": (_i32, [_i32, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp]),
    "cimrgp_potrs": (_i32, [_i32, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp]),
    "cimrgp_trsm_rows": (_i32, [_i32, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp]),
    "cimrgp_predict_mean": (_i32, [_i32, _vp, _i64, _i32, _vp, _i32, _vp, _i64, _dbl, _dbl, _vp, _vp, _i32, _vp]),
    "cimrgp_predict_from_w": (_i32, [_i32, _vp, _i64, _i64, _i64, _vp, _i32, _dbl, _dbl, _vp, _vp, _vp, _vp, _i32, _vp]),
    "cimrgp_block_stats": (_i32, [_i32, _vp, _vp, _i64, _i32, _vp, _vp]),
    "cimrgp_residual": (_i32, [_i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "cimrgp_train_mean": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "cimrgp_add_diag": (_i32, [_i32, _vp, _i64, _i64, _vp, _vp]),
    "cimrgp_noise_from_stats": (_i32, [_i32, _vp, _i32, _dbl, _dbl, _vp, _vp]),
    "cimrgp_logdet_half": (_i32, [_i32, _vp, _i64, _i64, _vp, _vp]),
    "cimrgp_syrk_lower": (_i32, [_i32, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "cimrgp_lml_grad_scratch_bytes": (_sz, [_i64]),
    "cimrgp_lml_grad": (_i32, [_i32, _vp, _i64, _i32, _vp, _i64, _vp, _i32, _dbl, _dbl, _dbl, _vp, _vp, _vp]),
    "cimrgp_lml_grad_ard": (_i32, [_i32, _vp, _i64, _i32, _vp, _i64, _vp, _i32, _dbl, _dbl, _vp, _vp, _vp]),
    "cimrgp_laplace_basis": (_i32, [_i32, _vp, _i64, _i32, _vp, _i32, _vp, _vp]),
    "cimrgp_basis_moments_scratch_bytes": (_sz, [_i64, _i32, _i32]),
    "cimrgp_basis_moments": (_i32, [_i32, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "cimrgp_basis_apply": (_i32, [_i32, _vp, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _dbl, _vp, _vp, _i32, _vp]),
    "cimrgp_profile_begin": (_i32, []),
    "cimrgp_profile_collect": (_i32, [C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_i64)]),
}

_lib = None


class CimrgpError(RuntimeError):
    """A C-ABI call returned a negative status."""


def load():
    """Load the shared library once; raise ImportError loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "cimrgp_amd: %s not found. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`bash cimrgp_amd/csrc/build.sh` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
        # One process of several (torch.distributed launchers export WORLD_SIZE): keep this process within
        # the runtime's 4 hardware queues -- caller, panel chain, carried rows and the collective's stream.
        # The factorisation's second carried-rows queue would be a fifth stream; the
        # library reads the switch at its first factorisation.
        os.environ.setdefault("CIMRGP_ROWS_ONE_QUEUE", "1")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().cimrgp_last_error()
    return msg.decode() if msg else ""


def check(rc, what):
    if rc != 0:
        raise CimrgpError("%s failed (status %d): %s" % (what, rc, last_error()))
