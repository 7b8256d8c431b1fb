"""ctypes loader of ``libcimrgp.so`` (the C ABI declared in include/cimrgp.h).

The product path has no CPU fallback: if the HIP library is missing or a call
fails, an exception is raised -- never a silent NumPy substitute.
"""
import atexit
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
    "cimrgp_block_posterior": (_i32, [_i32, _vp, _i64, _i32, _vp, _i32, _vp, _i64, _dbl, _dbl, _dbl, _vp, _i64, _vp, _sz, _vp, _vp, _i64,
                                      _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "cimrgp_block_posterior_staged": (_i32, [_i32, _vp, _i64, _i32, _vp, _i32, _vp, _i64, _dbl, _dbl, _dbl, _vp, _i64, _vp, _sz, _vp, _vp,
                                             _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "cimrgp_solve_queue": (_i32, [_vp, C.POINTER(C.c_void_p)]),
    "cimrgp_front_queue": (_i32, [_vp, C.POINTER(C.c_void_p)]),
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
    "cimrgp_layer_fit": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _dbl, _dbl, _dbl, _dbl, _dbl, _vp, _vp,
                                _vp, _i64, _i64, _vp, _sz, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cimrgp_layer_predict": (_i32, [_i32, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _dbl, _dbl, _vp, _i64, _i64, _vp, _sz,
                                    _vp, _i32, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "cimrgp_comm_unique_id": (_i32, [_vp]),
    "cimrgp_comm_create": (_i32, [_i32, _i32, _vp, C.POINTER(_vp)]),
    "cimrgp_comm_destroy": (_i32, [_vp]),
    "cimrgp_allreduce_sum": (_i32, [_vp, _i32, _vp, _i64, _vp]),
    "cimrgp_set_rows_queues": (_i32, [_i32]),
    "cimrgp_get_rows_queues": (_i32, []),
    "cimrgp_tuning_build": (_i32, []),
    "cimrgp_shutdown": (_i32, []),
    "cimrgp_profile_begin": (_i32, []),
    "cimrgp_profile_pause": (_i32, []),
    "cimrgp_profile_collect": (_i32, [C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_i64)]),
    "cimrgp_profile_collect_bytes": (_i32, [C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_i64)]),
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
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # The factorisations keep streams and events per caller stream and device; release them before the
    # HIP runtime's own exit handlers run (cimrgp_shutdown in include/cimrgp.h).
    atexit.register(_shutdown)
    return lib


def _shutdown():
    if _lib is not None:
        try:
            _lib.cimrgp_shutdown()
        except Exception:
            pass


COMM_ID_BYTES = 128


class Comm(object):
    """The C ABI's communicator (include/cimrgp.h: cimrgp_comm_*), for callers without torch.distributed:
    ``ident = Comm.unique_id()`` on one rank, shipped to the others by the caller; ``Comm(world, rank, ident)`` on
    every rank (collective); ``allreduce_sum(tensor_or_ptr, ...)``; ``close()``."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        check(load().cimrgp_comm_unique_id(C.cast(buf, _vp)), "cimrgp_comm_unique_id")
        return buf.raw

    def __init__(self, world_size, rank, ident):
        if len(ident) != COMM_ID_BYTES:
            raise ValueError("a communicator id is %d bytes" % COMM_ID_BYTES)
        self._h = _vp()
        buf = C.create_string_buffer(bytes(ident), COMM_ID_BYTES)
        check(load().cimrgp_comm_create(int(world_size), int(rank), C.cast(buf, _vp), C.byref(self._h)), "cimrgp_comm_create")
        self.world_size, self.rank = int(world_size), int(rank)

    def allreduce_sum(self, dtype, ptr, count, stream=None):
        check(load().cimrgp_allreduce_sum(self._h, int(dtype), ptr, int(count), stream), "cimrgp_allreduce_sum")

    def close(self):
        if self._h:
            check(load().cimrgp_comm_destroy(self._h), "cimrgp_comm_destroy")
            self._h = _vp()


def set_rows_queues(queues):
    """1 or 2 low-priority queues for the carried rows of ``cimrgp_potrf_rows`` (include/cimrgp.h):
    1 keeps a process within four streams (caller, panel chain, rows, collective)."""
    check(load().cimrgp_set_rows_queues(int(queues)), "cimrgp_set_rows_queues")


def last_error():
    msg = load().cimrgp_last_error()
    return msg.decode() if msg else ""


def check(rc, what):
    if rc != 0:
        raise CimrgpError("%s failed (status %d): %s" % (what, rc, last_error()))
