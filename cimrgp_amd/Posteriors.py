"""Per-layer exact-GP posterior over the regions of one resolution.

Dense twin of the reference's ``Posterior`` (Posteriors.py:9-211): constructed
per layer, holds per-region state in lists indexed by region, and is updated
IN PLACE from lists indexed by region (Posteriors.py:35,81,113).  Where the
reference's ``update_scale_given_axis`` forms a diagonal precision and the
projected targets y~ (Posteriors.py:35-78), this class builds the RBF Gram
matrix of the region, factors it and solves for the weights -- D1, D2, D3 of
SURVEY.md 8a' -- through the hand-written HIP kernels.
"""
import os

import numpy as np
import torch

from . import device as dev

NOISE_FRACTION = 0.01    # RegressionInput.py:62: labels.var() * 0.01
NOISE_FLOOR = 1e-8       # x sf: keeps a constant-target block positive definite

#: equal-sized blocks of a layer up to this size are factored as ONE batch (same launches, grid.y =
#: block); larger blocks fill the machine on their own and keep the look-ahead schedule
BATCH_MAX_N = int(os.environ.get("CIMRGP_BATCH_MAX_N", "16384"))

# ---- independent blocks of one layer run concurrently (the reference's independent-over-l loop,
# Posteriors.py:35-59): a small pool of streams per device; blocks are dealt round-robin.
_POOLS = {}


def block_streams(device, n_blocks, n_max):
    """How many blocks of a layer are in flight at once, and on which streams.  Small blocks are
    latency-bound (one queue each, ~10 % of the machine) and overlap almost perfectly; large ones
    already fill the machine with their trailing updates.  ``CIMRGP_BLOCK_STREAMS`` overrides."""
    env = os.environ.get("CIMRGP_BLOCK_STREAMS")
    if env is not None:
        want = int(env)
    elif n_max <= 5120:
        want = 8
    elif n_max <= 12288:
        want = 3
    elif n_max <= 20480:
        want = 2
    else:
        want = 1
    want = max(1, min(want, n_blocks))
    if want == 1:
        return []
    key = (torch.device(device).index, )
    pool = _POOLS.setdefault(key, [])
    while len(pool) < want:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:want]


class _Fanout(object):
    """Deal work items to pool streams: every stream starts after what is queued on the caller's
    stream, the caller's stream continues after all of them (events only, no host wait)."""

    def __init__(self, device, n_items, n_max):
        self.main = torch.cuda.current_stream(device)
        self.pool = block_streams(device, n_items, n_max)
        self.i = 0
        if self.pool:
            start = self.main.record_event()
            for s in self.pool:
                s.wait_event(start)

    def stream(self):
        if not self.pool:
            return self.main
        s = self.pool[self.i % len(self.pool)]
        self.i += 1
        return s

    def join(self):
        for s in self.pool:
            self.main.wait_stream(s)


class DenseBlock(object):
    """Device-resident state of one (resolution, region) block."""

    def __init__(self, x, kernel):
        self.x = x                       # (n x d) device view, normalised inputs
        self.n = int(x.shape[0])
        self.kernel = kernel
        self.lbuf = None                 # padded (n x ld) buffer holding L in its lower triangle
        self.ws = None                   # inverted diagonal blocks (potrf workspace)
        self.info = None                 # device int32, LAPACK convention
        self.alpha = None                # (n x q)  (K + noise I)^-1 r
        self.z = None                    # (n x q)  L^-1 r
        self.bias = None                 # (q,) device
        self.noise = None                # (1,) device
        self.r = None

    def fit(self, y, f_bar, train_out, shared_bias=None, shared_noise=None, keep_factor=True):
        """Fit on targets ``y - f_bar`` (both (n x q) device views); adds this
        block's training-point prediction into ``train_out`` (n x q)."""
        q = y.shape[1]
        k = self.kernel
        stats = None
        if shared_bias is None or (shared_noise is None and k.noise is None):
            stats = dev.block_stats(y, f_bar)
        self.bias = stats[:q] if shared_bias is None else shared_bias
        if k.noise is not None:
            self.noise = torch.full((1,), k.noise, dtype=y.dtype, device=y.device)
        elif shared_noise is not None:
            self.noise = shared_noise
        else:
            self.noise = dev.noise_from_stats(stats, q, NOISE_FRACTION, NOISE_FLOOR * k.sf)
        self.r = dev.residual(y, f_bar, self.bias)
        self.lbuf = dev.rbf_gram(self.x, k.l, k.sf, 0.0, lower_only=True)
        dev.add_diag(self.lbuf, self.n, self.noise)
        # the targets ride through the factorisation as q extra rows: z = L^-1 r comes out of the
        # same panel sweep (no separate forward solve), then one backward solve gives alpha
        q_rows = dev.alloc_matrix(q, self.n, y.dtype, y.device)
        q_rows[:q, :self.n] = self.r.t()
        self.ws, self.info = dev.potrf_rows(self.lbuf, self.n, q_rows, q)
        self.z = q_rows[:q, :self.n].t().contiguous()
        self.alpha = dev.solve_lt(self.lbuf, self.n, self.ws, self.z.clone())
        # K_noiseless alpha = r - noise * alpha: no second pass over the Gram matrix
        dev.train_mean(self.r, self.alpha, self.bias, self.noise, train_out, accumulate=True)
        if not keep_factor:
            self.lbuf = None
            self.ws = None
            self.z = None
        self.r = None

    def hand_over_to(self, stream):
        """The block was fitted on a pool stream; its tensors are used on ``stream`` from now on."""
        for t in (self.lbuf, self.ws, self.info, self.alpha, self.z, self.bias, self.noise):
            if t is not None:
                t.record_stream(stream)

    def predict(self, xs, mean_out, var_out=None, extra_var=0.0, chunk=16384, add_noise=False):
        """Accumulate this block's predictive mean (and latent variance) at ``xs``
        into ``mean_out`` (ns x q) / ``var_out`` (ns,).  ``add_noise``: add the block's noise
        variance, read from the device (no host round trip)."""
        k = self.kernel
        if var_out is None:
            dev.predict_mean(self.x, self.alpha, xs, k.l, k.sf, self.bias, out=mean_out, accumulate=True)
            return
        if self.lbuf is None:
            raise RuntimeError('predictive variance needs the Cholesky factor: fit with keep_factors=True')
        ns = xs.shape[0]
        for s0 in range(0, ns, chunk):
            s1 = min(ns, s0 + chunk)
            w = dev.rbf_cross(xs[s0:s1], self.x, k.l, k.sf)
            dev.trsm_rows(self.lbuf, self.n, self.ws, w, s1 - s0)
            dev.predict_from_w(w, s1 - s0, self.n, self.z, k.sf, extra_var, self.bias,
                               mean_out[s0:s1], var_out[s0:s1], accumulate=True,
                               extra_var_dev=self.noise if add_noise else None)

    def log_marginal_likelihood(self, r_dot_alpha):
        """-1/2 r^T alpha - sum log L_ii - n/2 log 2pi, per output column summed."""
        half_logdet = float(dev.logdet_half(self.lbuf, self.n).item())
        q = self.alpha.shape[1]
        return -0.5 * r_dot_alpha - q * half_logdet - 0.5 * q * self.n * np.log(2 * np.pi)


class DensePosterior(object):
    """One resolution: a list of :class:`DenseBlock`, updated in place."""

    def __init__(self, n_regions, dy, kernel, noise_region_specific=True, bias_region_specific=True):
        self.n_regions = int(n_regions)
        self.dy = int(dy)
        self.kernel = kernel
        self.noise_region_specific = noise_region_specific
        self.bias_region_specific = bias_region_specific
        self.blocks = [None] * self.n_regions

    def update_scale_given_axis(self, y_mean, x, f_bar, train_out, owned=None, keep_factors=True):
        """``y_mean``, ``x``, ``f_bar``, ``train_out``: lists indexed by region of device
        views (the reference passes lists indexed by region too, Posteriors.py:35).
        ``owned``: iterable of the region ids this process computes (default all)."""
        regions = range(self.n_regions) if owned is None else owned
        shared_bias = shared_noise = None
        if not (self.bias_region_specific and (self.noise_region_specific or self.kernel.noise is not None)):
            # layer-wide statistics over the concatenation of all regions (regions are
            # contiguous slices of one array: region 0's base with the total length)
            y_all, f_all = self._whole_layer(y_mean), self._whole_layer(f_bar)
            stats = dev.block_stats(y_all, f_all)
            if not self.bias_region_specific:
                shared_bias = stats[:self.dy]
            if not self.noise_region_specific and self.kernel.noise is None:
                shared_noise = dev.noise_from_stats(stats, self.dy, NOISE_FRACTION, NOISE_FLOOR * self.kernel.sf)
        regions = list(regions)
        if not regions:
            return
        # equal-sized small blocks: one batch per size (a uniform index set has at most two sizes per
        # layer, the last region taking the remainder, IndexSetGenerator.py:51-65)
        by_size = {}
        for l in regions:
            by_size.setdefault(int(x[l].shape[0]), []).append(l)
        single = []
        for n_l, group in by_size.items():
            if len(group) >= 2 and n_l <= BATCH_MAX_N:
                self._fit_batched(group, y_mean, x, f_bar, train_out, shared_bias, shared_noise, keep_factors)
            else:
                single.extend(group)
        if not single:
            return
        fan = _Fanout(y_mean[single[0]].device, len(single), max(int(x[l].shape[0]) for l in single))
        for l in single:
            with torch.cuda.stream(fan.stream()):
                blk = DenseBlock(x[l], self.kernel)
                blk.fit(y_mean[l], f_bar[l], train_out[l], shared_bias, shared_noise, keep_factor=keep_factors)
                if fan.pool:
                    blk.hand_over_to(fan.main)
            self.blocks[l] = blk
        fan.join()

    def _fit_batched(self, group, y_mean, x, f_bar, train_out, shared_bias, shared_noise, keep_factors):
        """Fit ``group`` (regions of equal size) with ONE batched factorisation and ONE batched
        backward solve: the matrices live in one arena (batch x n x ld) and every kernel of the panel
        sweep is launched once for all of them (cimrgp_potrf_rows_batched)."""
        k = self.kernel
        n = int(x[group[0]].shape[0])
        q = self.dy
        device, dtype = y_mean[group[0]].device, y_mean[group[0]].dtype
        nb = len(group)
        ld = dev.padded_ld(n)
        karena = torch.empty((nb, n, ld), dtype=dtype, device=device)
        ws_bytes = (dev.potrf_workspace_bytes(n, dtype) + 15) // 16 * 16
        ws_arena = torch.empty((nb, max(ws_bytes, 16)), dtype=torch.uint8, device=device)
        info = torch.zeros(nb, dtype=torch.int32, device=device)
        ldq = dev.padded_ld(n)
        rows = torch.empty((nb, q, ldq), dtype=dtype, device=device)
        blocks, resid = [], []
        for i, l in enumerate(group):
            blk = DenseBlock(x[l], k)
            stats = None
            if shared_bias is None or (shared_noise is None and k.noise is None):
                stats = dev.block_stats(y_mean[l], f_bar[l])
            blk.bias = stats[:q] if shared_bias is None else shared_bias
            if k.noise is not None:
                blk.noise = torch.full((1,), k.noise, dtype=dtype, device=device)
            elif shared_noise is not None:
                blk.noise = shared_noise
            else:
                blk.noise = dev.noise_from_stats(stats, q, NOISE_FRACTION, NOISE_FLOOR * k.sf)
            r = dev.residual(y_mean[l], f_bar[l], blk.bias)
            dev.rbf_gram(blk.x, k.l, k.sf, 0.0, lower_only=True, out=karena[i])
            dev.add_diag(karena[i], n, blk.noise)
            blocks.append(blk)
            resid.append(r)
        rows[:, :, :n] = torch.stack(resid).transpose(1, 2)
        # the targets ride through the factorisation as q extra rows per block: z = L^-1 r
        dev.potrf_rows_batched(karena, n, ld, ws_arena, info, rows, q, ldq)
        z = rows[:, :, :n].transpose(1, 2).contiguous()
        alpha = dev.solve_lt_batched(karena, n, ld, ws_arena, z.clone())
        for i, l in enumerate(group):
            blk = blocks[i]
            blk.info = info[i:i + 1]
            blk.alpha = alpha[i]
            if keep_factors:
                blk.lbuf, blk.ws, blk.z = karena[i], ws_arena[i], z[i]
            dev.train_mean(resid[i], blk.alpha, blk.bias, blk.noise, train_out[l], accumulate=True)
            self.blocks[l] = blk

    @staticmethod
    def _whole_layer(views):
        base = views[0]
        total = sum(int(v.shape[0]) for v in views)
        return torch.as_strided(base, (total, base.shape[1]), base.stride())

    def failure_flag(self, regions, out):
        """out[0] = largest LAPACK ``info`` over ``regions`` (0 = all factorisations fine), written
        on the device without a host synchronisation."""
        infos = [self.blocks[l].info for l in regions if self.blocks[l] is not None and self.blocks[l].info is not None]
        if infos:
            out.copy_(torch.stack([i.reshape(()) for i in infos]).max().to(out.dtype).reshape(1))
        return out

    def check(self, regions=None):
        """Raise ``numpy.linalg.LinAlgError`` if any factorisation met a non-positive pivot."""
        for l in (range(self.n_regions) if regions is None else regions):
            if self.blocks[l] is not None and self.blocks[l].info is not None:
                dev.raise_if_not_pd(self.blocks[l].info)
