"""Thin, typed wrappers around the C ABI operating on torch CUDA tensors.

torch is used for device memory, streams and (elsewhere) torch.distributed --
plumbing only; every arithmetic operation of the path below is a hand-written
HIP kernel reached through ``libcimrgp.so``.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def as_torch_dtype(dtype):
    if isinstance(dtype, torch.dtype):
        if dtype not in _DT:
            raise ValueError("dtype must be float32 or float64")
        return dtype
    key = str(dtype).lower()
    if key in ("f64", "float64", "fp64", "double", "<class 'numpy.float64'>"):
        return torch.float64
    if key in ("f32", "float32", "fp32", "float", "<class 'numpy.float32'>"):
        return torch.float32
    raise ValueError("dtype must be 'f32' or 'f64', got %r" % (dtype,))


def require_gpu(device=None):
    """Resolve the device of the product path; there is no CPU fallback."""
    _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("cimrgp_amd: no HIP device visible; the dense GP path has no CPU fallback "
                           "(the NumPy restatement under oracle/ is test infrastructure only)")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def padded_ld(n):
    """Row pitch (elements): a multiple of 16 (every row starts on a 128-byte line) that is NOT a
    multiple of 512 elements: with a power-of-two pitch all rows of a tile start in the same HBM
    channel / cache set and the strided tile accesses of the update kernels serialise there."""
    ld = max(16, (int(n) + 15) // 16 * 16)
    if ld % 512 == 0:
        ld += 16
    return ld


def alloc_matrix(rows, cols, dtype, device):
    """(rows x cols) row-major view of a buffer with a padded leading dimension."""
    ld = padded_ld(cols)
    buf = torch.empty((max(int(rows), 1), ld), dtype=dtype, device=device)
    return buf


def to_device(a, dtype, device):
    t = torch.as_tensor(np.ascontiguousarray(a)) if not isinstance(a, torch.Tensor) else a
    return t.to(device=device, dtype=dtype).contiguous()


def rbf_gram(x, ell, sf2, diag_add=0.0, lower_only=False, out=None):
    """D1.  x: (n x d) device tensor.  Returns the (n x ld) buffer; [:, :n] is K."""
    n, d = x.shape
    if out is None:
        out = alloc_matrix(n, n, x.dtype, x.device)
    lib = _lib.load()
    _lib.check(lib.cimrgp_rbf_gram(_DT[x.dtype], _p(x), n, d, float(ell), float(sf2), float(diag_add),
                                   _p(out), out.stride(0), int(bool(lower_only)), _stream()), "cimrgp_rbf_gram")
    return out


def rbf_cross(xa, xb, ell, sf2, out=None):
    """Cross-Gram (na x nb) into a padded buffer."""
    na, d = xa.shape
    nb = xb.shape[0]
    if out is None:
        out = alloc_matrix(na, nb, xa.dtype, xa.device)
    lib = _lib.load()
    _lib.check(lib.cimrgp_rbf_cross(_DT[xa.dtype], _p(xa), na, _p(xb), nb, d, float(ell), float(sf2),
                                    _p(out), out.stride(0), _stream()), "cimrgp_rbf_cross")
    return out


def potrf_workspace_bytes(n, dtype):
    return int(_lib.load().cimrgp_potrf_workspace_bytes(_DT[dtype], int(n)))


def potrf_workspace(n, dtype, device):
    return torch.empty(max(potrf_workspace_bytes(n, dtype), 16), dtype=torch.uint8, device=device)


def potrf(kbuf, n, ws=None, info=None):
    """D2 in place on kbuf[:n, :n] (lower).  Returns (ws, info) -- info is a
    device int32 scalar written asynchronously (LAPACK convention)."""
    lib = _lib.load()
    if ws is None:
        ws = potrf_workspace(n, kbuf.dtype, kbuf.device)
    if info is None:
        info = torch.zeros(1, dtype=torch.int32, device=kbuf.device)
    _lib.check(lib.cimrgp_potrf(_DT[kbuf.dtype], _p(kbuf), int(n), kbuf.stride(0), _p(ws), ws.numel(),
                                _p(info), _stream()), "cimrgp_potrf")
    return ws, info


def potrf_rows(kbuf, n, bbuf, m, ws=None, info=None):
    """D2 with m carried rows: kbuf[:n,:n] = L L^T in place and bbuf[:m,:n] <- bbuf L^-T."""
    lib = _lib.load()
    if ws is None:
        ws = potrf_workspace(n, kbuf.dtype, kbuf.device)
    if info is None:
        info = torch.zeros(1, dtype=torch.int32, device=kbuf.device)
    _lib.check(lib.cimrgp_potrf_rows(_DT[kbuf.dtype], _p(kbuf), int(n), kbuf.stride(0), _p(ws), ws.numel(),
                                     _p(info), _p(bbuf), int(m), bbuf.stride(0), _stream()), "cimrgp_potrf_rows")
    return ws, info


def block_posterior(x, y, xs, ell, sf2, noise, kbuf, wbuf, ws, info, alpha, z, mean, var, add_noise=False, accumulate=False,
                    scratch=None, streams=None):
    """One block's whole posterior in ONE call (cimrgp_block_posterior, include/cimrgp.h): Gram matrix, factorisation with
    the cross-Gram rows and the targets carried, backward solve, predictive mean and variance -- the same kernels as the
    separate calls.  kbuf (n x ld), wbuf ((ns + q) x ld), ws / info as for potrf; alpha, z (n x q),
    mean (ns x q), var (ns) are filled.  ``scratch``: 2 q n elements (allocated here when not given: a caller that
    must not allocate between two events -- bench.py's timed step -- passes its own).
    ``streams`` = (front, factor, solve) torch streams: the three stages on three streams
    (cimrgp_block_posterior_staged), for callers that pipeline independent blocks over rotating buffer sets and order
    the reuse of a set themselves; results are then ordered on ``solve``."""
    lib = _lib.load()
    n, d = x.shape
    q = y.shape[1]
    ns = 0 if xs is None else xs.shape[0]
    if scratch is None:
        scratch = torch.empty(2 * q * max(int(n), 1), dtype=x.dtype, device=x.device)
    elif scratch.numel() < 2 * q * max(int(n), 1) or scratch.dtype != x.dtype:
        raise ValueError("scratch must hold 2 q n elements of the block's dtype")
    if streams is not None:
        sf, sp, sq = (s.cuda_stream for s in streams)
        _lib.check(lib.cimrgp_block_posterior_staged(_DT[x.dtype], _p(x), int(n), int(d), _p(y), int(q), _p(xs), int(ns), float(ell),
                                                     float(sf2), float(noise), _p(kbuf), kbuf.stride(0), _p(ws), ws.numel(), _p(info),
                                                     _p(wbuf), wbuf.stride(0), _p(alpha), _p(z), _p(scratch), _p(mean), _p(var),
                                                     int(bool(add_noise)), int(bool(accumulate)), sf, sp, sq),
                   "cimrgp_block_posterior_staged")
        return mean, var
    _lib.check(lib.cimrgp_block_posterior(_DT[x.dtype], _p(x), int(n), int(d), _p(y), int(q), _p(xs), int(ns), float(ell), float(sf2),
                                          float(noise), _p(kbuf), kbuf.stride(0), _p(ws), ws.numel(), _p(info), _p(wbuf),
                                          wbuf.stride(0), _p(alpha), _p(z), _p(scratch), _p(mean), _p(var), int(bool(add_noise)),
                                          int(bool(accumulate)), _stream()), "cimrgp_block_posterior")
    return mean, var


def solve_queue(stream=None):
    """The stream to pass as ``streams[2]`` of :func:`block_posterior` (cimrgp_solve_queue, include/cimrgp.h): the
    look-ahead context's queue that is idle between two factorisations on ``stream`` (default: the current stream),
    wrapped as a torch stream; ``stream`` itself when it owns no context."""
    lib = _lib.load()
    cur = torch.cuda.current_stream() if stream is None else stream
    out = ctypes.c_void_p()
    _lib.check(lib.cimrgp_solve_queue(cur.cuda_stream, ctypes.byref(out)), "cimrgp_solve_queue")
    if (out.value or 0) == (cur.cuda_stream or 0):
        return cur
    return torch.cuda.ExternalStream(out.value, device=cur.device)


def front_queue(stream=None):
    """The stream to pass as ``streams[0]`` of :func:`block_posterior` when consecutive calls are independent blocks
    (cimrgp_front_queue, include/cimrgp.h): the look-ahead context's queue that falls idle before a factorisation on
    ``stream`` ends, wrapped as a torch stream; ``stream`` itself when it owns no context."""
    lib = _lib.load()
    cur = torch.cuda.current_stream() if stream is None else stream
    out = ctypes.c_void_p()
    _lib.check(lib.cimrgp_front_queue(cur.cuda_stream, ctypes.byref(out)), "cimrgp_front_queue")
    if (out.value or 0) == (cur.cuda_stream or 0):
        return cur
    return torch.cuda.ExternalStream(out.value, device=cur.device)


def potrf_rows_batched(karena, n, ld, ws_arena, info, barena=None, m=0, ldb=0):
    """``batch`` equal-sized factorisations in the same launches.  karena: (batch, n, ld) tensor,
    ws_arena: (batch, ws_bytes) uint8, info: (batch,) int32, barena: (batch, m, ldb) carried rows."""
    lib = _lib.load()
    batch = karena.shape[0]
    _lib.check(lib.cimrgp_potrf_rows_batched(_DT[karena.dtype], _p(karena), int(n), int(ld), karena.stride(0), _p(ws_arena),
                                             ws_arena.stride(0), _p(info), _p(barena), int(m), int(ldb),
                                             0 if barena is None else barena.stride(0), int(batch), _stream()),
               "cimrgp_potrf_rows_batched")


def solve_lt_batched(karena, n, ld, ws_arena, z):
    """Backward halves for a batch: z (batch, n, q) = L^-1 R is overwritten with alpha."""
    lib = _lib.load()
    batch, _, q = z.shape
    scratch = torch.empty((batch, 2 * q * max(int(n), 1)), dtype=karena.dtype, device=karena.device)
    _lib.check(lib.cimrgp_solve_lt_batched(_DT[karena.dtype], _p(karena), int(n), int(ld), karena.stride(0), _p(ws_arena),
                                           ws_arena.stride(0), _p(z), int(q), _p(scratch), int(batch), _stream()),
               "cimrgp_solve_lt_batched")
    return z


#: work areas of cimrgp_layer_fit (carried target rows, solve scratch) per (device, dtype, STREAM): the fit of a layer
#: is enqueue-only, so they are not allocated per call; a new shape replaces the cached one (O(batch q n) elements).
#: Keyed by the stream the call is enqueued on (round 5): two fits on two streams -- two models, two
#: threads, a caller's stream pool -- must not share the carried rows; within one stream the calls are ordered, and the
#: caching allocator frees a replaced area in that same stream's order.
_LAYER_WORK = {}


def _layer_work_areas(batch, q, n, ldr, dtype, device):
    key = (str(device), dtype, int(_stream() or 0))
    shape = (batch, q, n)
    hit = _LAYER_WORK.get(key)
    if hit is None or hit[0] != shape:
        rows = torch.empty((batch, q, ldr), dtype=dtype, device=device)
        scratch = torch.empty((batch, 2 * q * max(n, 1)), dtype=dtype, device=device)
        hit = (shape, rows, scratch)
        _LAYER_WORK[key] = hit
    return hit[1], hit[2]


def layer_fit(x, y, fbar, train_out, starts, n, ell, sf2, noise_fixed, noise_frac, noise_floor, shared_bias, shared_noise,
              karena, ws_arena, info, bias, noise, z, alpha):
    """The fit of ``batch`` equal-sized blocks of one layer in ONE call (cimrgp_layer_fit, include/cimrgp.h).
    x, y, fbar, train_out: the LAYER's arrays (N x d / N x q); starts: device int64 (batch,) row offsets.
    karena (batch, n, ld), ws_arena (batch, ws_bytes) uint8, info (batch,) int32, bias (batch, q), noise (batch,),
    z / alpha (batch, n, q) are filled.  noise_fixed < 0: from the statistics (or ``shared_noise``)."""
    lib = _lib.load()
    batch = int(karena.shape[0])
    q = int(y.shape[1])
    d = int(x.shape[1])
    ldr = padded_ld(n)
    rows, scratch = _layer_work_areas(batch, q, int(n), ldr, y.dtype, y.device)
    _lib.check(lib.cimrgp_layer_fit(_DT[y.dtype], _p(x), _p(y), _p(fbar), _p(train_out), _p(starts), batch, int(n), d, q,
                                    float(ell), float(sf2), float(noise_fixed), float(noise_frac), float(noise_floor),
                                    _p(shared_bias), _p(shared_noise), _p(karena), karena.stride(1), karena.stride(0),
                                    _p(ws_arena), ws_arena.stride(0), _p(info), _p(rows), ldr, _p(z), _p(alpha), _p(bias),
                                    _p(noise), _p(scratch), _stream()), "cimrgp_layer_fit")


def layer_predict(x, starts, n, xs, t_starts, ns, ell, sf2, larena, ws_arena, z, bias, noise, mean, var):
    """Predictive mean and variance of ``batch`` equal-sized blocks at ``ns`` test points each in ONE call
    (cimrgp_layer_predict): accumulates into mean (N* x q) / var (N*,) at rows t_starts[b] ...; ``noise``
    (batch,) or None is added to the variance."""
    lib = _lib.load()
    batch = int(larena.shape[0])
    ldw = padded_ld(n)
    w = torch.empty((batch, max(int(ns), 1), ldw), dtype=x.dtype, device=x.device)
    _lib.check(lib.cimrgp_layer_predict(_DT[x.dtype], _p(x), _p(starts), int(n), int(x.shape[1]), _p(xs), _p(t_starts), int(ns),
                                        batch, float(ell), float(sf2), _p(larena), larena.stride(1), larena.stride(0),
                                        _p(ws_arena), ws_arena.stride(0), _p(z), int(z.shape[2]), _p(bias), _p(noise),
                                        _p(w), ldw, w.stride(0), _p(mean), _p(var), _stream()), "cimrgp_layer_predict")


def solve_lt(lbuf, n, ws, z):
    """Backward half of D3: z (n x q) = L^-1 R is overwritten with alpha = L^-T z."""
    lib = _lib.load()
    q = z.shape[1]
    scratch = torch.empty(2 * q * max(int(n), 1), dtype=lbuf.dtype, device=lbuf.device)
    _lib.check(lib.cimrgp_solve_lt(_DT[lbuf.dtype], _p(lbuf), int(n), lbuf.stride(0), _p(ws), _p(z), q,
                                   _p(scratch), _stream()), "cimrgp_solve_lt")
    return z


#: include/cimrgp.h CIMRGP_INFO_WATCHDOG: the factorisation's schedule failed, not the matrix
INFO_WATCHDOG = 0x7fffffff


def is_watchdog(code):
    """``code``: an ``info`` word, possibly after a trip through a floating-point collective (float32 rounds
    2^31 - 1 up to 2^31; sums over ranks only grow)."""
    return float(code) >= float(INFO_WATCHDOG) - 128.0


def raise_if_not_pd(info):
    """LinAlgError is the exception the reference's PD guard keys on (SanityCheck.py:59-65); a schedule
    watchdog (CIMRGP_INFO_WATCHDOG) is NOT a numerical failure and must not be repaired as one: RuntimeError."""
    code = int(info.item()) if isinstance(info, torch.Tensor) else int(info)
    if is_watchdog(code):
        raise RuntimeError("cimrgp_potrf: schedule watchdog (a device-side wait inside the factorisation expired; "
                           "the result is undefined and the matrix was not judged)")
    if code != 0:
        raise np.linalg.LinAlgError("Matrix is not positive definite (leading minor of order %d)" % code)


def potrs(lbuf, n, ws, rhs, want_z=False):
    """D3: rhs (n x q) is overwritten with alpha.  Returns z = L^-1 rhs if asked."""
    lib = _lib.load()
    q = rhs.shape[1]
    scratch = torch.empty(2 * q * max(int(n), 1), dtype=lbuf.dtype, device=lbuf.device)
    z = torch.empty_like(rhs) if want_z else None
    _lib.check(lib.cimrgp_potrs(_DT[lbuf.dtype], _p(lbuf), int(n), lbuf.stride(0), _p(ws), _p(rhs), q,
                                _p(z), _p(scratch), _stream()), "cimrgp_potrs")
    return z


def trsm_rows(lbuf, n, ws, bbuf, m):
    """B <- B L^-T on bbuf[:m, :n]."""
    lib = _lib.load()
    _lib.check(lib.cimrgp_trsm_rows(_DT[lbuf.dtype], _p(lbuf), int(n), lbuf.stride(0), _p(ws), _p(bbuf),
                                    int(m), bbuf.stride(0), _stream()), "cimrgp_trsm_rows")
    return bbuf


def predict_mean(x, alpha, xs, ell, sf2, bias=None, out=None, accumulate=False):
    """D4 fused mean: out (ns x q) (+)= bias + K(xs, x) alpha."""
    lib = _lib.load()
    n, d = x.shape
    ns = xs.shape[0]
    q = alpha.shape[1]
    if out is None:
        out = torch.empty((ns, q), dtype=x.dtype, device=x.device)
        accumulate = False
    _lib.check(lib.cimrgp_predict_mean(_DT[x.dtype], _p(x), n, d, _p(alpha), q, _p(xs), ns, float(ell),
                                       float(sf2), _p(bias), _p(out), int(bool(accumulate)), _stream()),
               "cimrgp_predict_mean")
    return out


def predict_from_w(wbuf, ns, n, z, sf2, extra_var=0.0, bias=None, mean_out=None, var_out=None, accumulate=False,
                   extra_var_dev=None):
    """``extra_var``: host number; ``extra_var_dev``: device scalar (e.g. a block's noise) -- both are added."""
    lib = _lib.load()
    q = 0 if z is None else z.shape[1]
    _lib.check(lib.cimrgp_predict_from_w(_DT[wbuf.dtype], _p(wbuf), int(ns), int(n), wbuf.stride(0), _p(z), q,
                                         float(sf2), float(extra_var), _p(extra_var_dev), _p(bias), _p(mean_out),
                                         _p(var_out), int(bool(accumulate)), _stream()), "cimrgp_predict_from_w")


def block_stats(y, fbar, stats_out=None):
    lib = _lib.load()
    n, q = y.shape
    if stats_out is None:
        stats_out = torch.empty(q + 1, dtype=y.dtype, device=y.device)
    _lib.check(lib.cimrgp_block_stats(_DT[y.dtype], _p(y), _p(fbar), n, q, _p(stats_out), _stream()),
               "cimrgp_block_stats")
    return stats_out


def residual(y, fbar, bias, out=None):
    lib = _lib.load()
    n, q = y.shape
    if out is None:
        out = torch.empty_like(y)
    _lib.check(lib.cimrgp_residual(_DT[y.dtype], _p(y), _p(fbar), _p(bias), n, q, _p(out), _stream()),
               "cimrgp_residual")
    return out


def train_mean(r, alpha, bias, noise, out, accumulate=False):
    lib = _lib.load()
    n, q = r.shape
    _lib.check(lib.cimrgp_train_mean(_DT[r.dtype], _p(r), _p(alpha), _p(bias), _p(noise), n, q, _p(out),
                                     int(bool(accumulate)), _stream()), "cimrgp_train_mean")
    return out


def add_diag(kbuf, n, noise):
    lib = _lib.load()
    _lib.check(lib.cimrgp_add_diag(_DT[kbuf.dtype], _p(kbuf), int(n), kbuf.stride(0), _p(noise), _stream()),
               "cimrgp_add_diag")


def noise_from_stats(stats, q, frac, floor_value, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty(1, dtype=stats.dtype, device=stats.device)
    _lib.check(lib.cimrgp_noise_from_stats(_DT[stats.dtype], _p(stats), int(q), float(frac), float(floor_value),
                                           _p(out), _stream()), "cimrgp_noise_from_stats")
    return out


def logdet_half(lbuf, n):
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float64, device=lbuf.device)
    _lib.check(lib.cimrgp_logdet_half(_DT[lbuf.dtype], _p(lbuf), int(n), lbuf.stride(0), _p(out), _stream()),
               "cimrgp_logdet_half")
    return out


def syrk_lower(cbuf, abuf, n, k):
    """cbuf[:n,:n] (lower) -= abuf[:n,:k] abuf[:n,:k]^T."""
    lib = _lib.load()
    _lib.check(lib.cimrgp_syrk_lower(_DT[cbuf.dtype], _p(cbuf), cbuf.stride(0), _p(abuf), abuf.stride(0), int(n), int(k),
                                     _stream()), "cimrgp_syrk_lower")
    return cbuf


def lml_grad(x, kinv, n, alpha, ell, sf2, noise):
    """Gradient of the log marginal likelihood w.r.t. (log sf, log l, log noise): device float64[3]."""
    lib = _lib.load()
    out = torch.empty(3, dtype=torch.float64, device=x.device)
    scratch = torch.empty(max(lib.cimrgp_lml_grad_scratch_bytes(int(n)), 8) // 8, dtype=torch.float64, device=x.device)
    _lib.check(lib.cimrgp_lml_grad(_DT[x.dtype], _p(x), int(n), x.shape[1], _p(kinv), kinv.stride(0), _p(alpha),
                                   alpha.shape[1], float(ell), float(sf2), float(noise), _p(out), _p(scratch), _stream()),
               "cimrgp_lml_grad")
    return out


def lml_grad_ard(x_scaled, kinv, n, alpha, sf2, noise):
    """ARD gradient w.r.t. (log sf, log l_1..l_d, log noise); ``x_scaled`` = inputs / length-scales."""
    lib = _lib.load()
    d = x_scaled.shape[1]
    out = torch.empty(d + 2, dtype=torch.float64, device=x_scaled.device)
    scratch = torch.empty(max(lib.cimrgp_lml_grad_scratch_bytes(int(n)), 8) // 8, dtype=torch.float64, device=x_scaled.device)
    _lib.check(lib.cimrgp_lml_grad_ard(_DT[x_scaled.dtype], _p(x_scaled), int(n), d, _p(kinv), kinv.stride(0), _p(alpha),
                                       alpha.shape[1], float(sf2), float(noise), _p(out), _p(scratch), _stream()),
               "cimrgp_lml_grad_ard")
    return out


# ---- reduced-rank (Laplacian basis) block path ------------------------------------------------
def _f64dev(a, device):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def laplace_basis(x, interval, n_basis, out=None):
    """Phi (n x m) of the Dirichlet-Laplacian eigenfunctions on [-L, L]^d for device points x."""
    n, d = x.shape
    if out is None:
        out = torch.empty((n, int(n_basis)), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    iv = _f64dev(np.asarray(interval, dtype=np.float64).reshape(-1), x.device)
    if iv.numel() != d:
        raise ValueError('Basis interval should have the same dimensionality as the input.')
    if not x.is_contiguous():
        raise ValueError("laplace_basis needs contiguous inputs")
    _lib.check(lib.cimrgp_laplace_basis(_DT[x.dtype], _p(x), n, d, _p(iv), int(n_basis), _p(out), _stream()),
               "cimrgp_laplace_basis")
    return out


class BlockMoments(object):
    """Host copy of the sums one block's updates need (see include/cimrgp.h, cimrgp_basis_moments)."""
    __slots__ = ("proj", "colsum", "colsum2", "resid_sum", "resid_sq", "fvar_sum", "n")

    def __init__(self, rec, m, q, n):
        self.proj = rec[:m * q].reshape(m, q)           # Phi^T r0
        self.colsum = rec[m * q:m * q + m]
        self.colsum2 = rec[m * q + m:m * q + 2 * m]
        self.resid_sum = rec[m * q + 2 * m:m * q + 2 * m + q]
        self.resid_sq = float(rec[m * q + 2 * m + q])
        self.fvar_sum = float(rec[m * q + 2 * m + q + 1])
        self.n = int(n)


def _interval_dev(interval, d, device):
    iv = _f64dev(np.asarray(interval, dtype=np.float64).reshape(-1), device)
    if iv.numel() != d:
        raise ValueError('Basis interval should have the same dimensionality as the input.')
    return iv


def basis_moments(x, interval, n_basis, y, fbar, fvar, eau):
    """Every N-dependent sum of one block's variational updates in one pass over the block's
    points, with r0 = y - fbar - Phi E[au]^T; Phi is regenerated from ``x`` (n x d, device) and
    the interval, never read.  ``eau``: (q x m) host array.  Returns BlockMoments (host)."""
    n, d = x.shape
    m = int(n_basis)
    q = y.shape[1]
    lib = _lib.load()
    rec_len = m * q + 2 * m + q + 2
    out = torch.empty(rec_len, dtype=torch.float64, device=x.device)
    scratch = torch.empty(max(lib.cimrgp_basis_moments_scratch_bytes(n, m, q), 8) // 8, dtype=torch.float64,
                          device=x.device)
    e = _f64dev(eau, x.device)
    iv = _interval_dev(interval, d, x.device)
    for t in (x, y, fbar, fvar):
        if t is not None and not t.is_contiguous():
            raise ValueError("basis_moments needs contiguous operands")
    _lib.check(lib.cimrgp_basis_moments(_DT[x.dtype], _p(x), n, d, _p(iv), m, _p(y), _p(fbar), _p(fvar), _p(e), q, _p(out),
                                        _p(scratch), _stream()), "cimrgp_basis_moments")
    return BlockMoments(out.cpu().numpy(), m, q, n)


def basis_apply(x, interval, n_basis, eau, bias=None, c2=None, bias_var=0.0, mean=None, var=None, accumulate=False):
    """mean (+)= bias + Phi E[au]^T ;  var (+)= bias_var + Phi^2 c2, Phi regenerated from x."""
    n, d = x.shape
    q = np.asarray(eau).shape[0]
    lib = _lib.load()
    e = _f64dev(eau, x.device)
    b = None if bias is None else _f64dev(bias, x.device)
    c = None if c2 is None else _f64dev(c2, x.device)
    iv = _interval_dev(interval, d, x.device)
    for t in (x, mean, var):
        if t is not None and not t.is_contiguous():
            raise ValueError("basis_apply needs contiguous operands")
    _lib.check(lib.cimrgp_basis_apply(_DT[x.dtype], _p(x), n, d, _p(iv), int(n_basis), _p(e), q, _p(b), _p(c),
                                      float(bias_var), _p(mean), _p(var), int(bool(accumulate)), _stream()),
               "cimrgp_basis_apply")
