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
#: upper bound on the streams of the pool (dist.share_one_gpu sets 1: ranks sharing one GPU)
MAX_BLOCK_STREAMS = 8


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
    want = max(1, min(want, n_blocks, MAX_BLOCK_STREAMS))
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
        self.batch = None                # _FittedBatch this block's factor lives in (or None)
        self.batch_index = -1

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


def _layer_array(views, group):
    """The 2-D array (from the start of its storage) that the region views ``views[l]``, l in group, are row
    ranges of; None if they are not slices of one row-major array."""
    v0 = views[group[0]]
    st = v0.untyped_storage().data_ptr()
    rows = 0
    for l in group:
        v = views[l]
        if v.untyped_storage().data_ptr() != st or v.stride() != v0.stride() or v.stride(1) != 1 \
                or v.stride(0) != v.shape[1] or v.storage_offset() % v.stride(0) != 0:
            return None
        rows = max(rows, v.storage_offset() // v.stride(0) + v.shape[0])
    return torch.as_strided(v0, (int(rows), int(v0.shape[1])), v0.stride(), 0)


def _same_rows(arrays, group):
    """Whether region l starts at the same row (counted from the start of the storage) in every one of ``arrays``
    (lists of region views), for every l of ``group``: a view of y at a non-zero storage offset, say, does not."""
    def row_of(v):
        return v.storage_offset() // max(1, v.stride(0))
    return all(row_of(a[l]) == row_of(arrays[0][l]) for a in arrays[1:] for l in group)


class _FittedBatch(object):
    """Equal-sized blocks fitted together: their factors share one arena, so that a prediction can
    address block i at ``base + i * stride`` (cimrgp_layer_predict)."""

    def __init__(self, regions, n, starts, karena, ws_arena, z, bias, noise):
        self.regions, self.n, self.starts = list(regions), int(n), starts
        self.karena, self.ws_arena, self.z, self.bias, self.noise = karena, ws_arena, z, bias, noise


class DensePosterior(object):
    """One resolution: a list of :class:`DenseBlock`, updated in place."""

    def __init__(self, n_regions, dy, kernel, noise_region_specific=True, bias_region_specific=True):
        self.n_regions = int(n_regions)
        self.dy = int(dy)
        self.kernel = kernel
        self.noise_region_specific = noise_region_specific
        self.bias_region_specific = bias_region_specific
        self.blocks = [None] * self.n_regions
        self.batches = []                # _FittedBatch records of the last sweep (equal-sized blocks fitted together)

    def needs_whole_layer(self):
        """Whether the layer's fit reads statistics over ALL its regions (a shared bias, or a shared noise taken from
        the targets): such a layer needs the whole latent function on every rank (MRGP._fit)."""
        return not (self.bias_region_specific and (self.noise_region_specific or self.kernel.noise is not None))

    def update_scale_given_axis(self, y_mean, x, f_bar, train_out, owned=None, keep_factors=True):
        """``y_mean``, ``x``, ``f_bar``, ``train_out``: lists indexed by region of device
        views (the reference passes lists indexed by region too, Posteriors.py:35).
        ``owned``: iterable of the region ids this process computes (default all)."""
        regions = range(self.n_regions) if owned is None else owned
        shared_bias = shared_noise = None
        if self.needs_whole_layer():
            # layer-wide statistics over the concatenation of all regions (regions are
            # contiguous slices of one array: region 0's base with the total length)
            y_all, f_all = self._whole_layer(y_mean), self._whole_layer(f_bar)
            stats = dev.block_stats(y_all, f_all)
            if not self.bias_region_specific:
                shared_bias = stats[:self.dy]
            if not self.noise_region_specific and self.kernel.noise is None:
                shared_noise = dev.noise_from_stats(stats, self.dy, NOISE_FRACTION, NOISE_FLOOR * self.kernel.sf)
        regions = list(regions)
        self.batches = []
        if not regions:
            return
        # equal-sized small blocks: one batch per size (a uniform index set has at most two sizes per
        # layer, the last region taking the remainder, IndexSetGenerator.py:51-65)
        by_size = {}
        for l in regions:
            by_size.setdefault(int(x[l].shape[0]), []).append(l)
        single = []
        for n_l, group in by_size.items():
            # the batched call addresses a block by ONE row offset into the layer's arrays: every array must be a
            # row-major 2-D array of which the region views are slices, at the same rows in all four
            sliced = all(_layer_array(v, group) is not None for v in (y_mean, x, f_bar, train_out)) \
                and _same_rows((y_mean, x, f_bar, train_out), group)
            if len(group) >= 2 and n_l <= BATCH_MAX_N and sliced:
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
        """Fit ``group`` (regions of equal size) with ONE C call per sub-batch (cimrgp_layer_fit): statistics,
        residual rows, Gram matrices, the batched factorisation with the residual rows carried, the batched
        backward solve and the training-point prediction are each launched once for all the blocks.  The
        matrices live in one arena (batch x n x ld); a group whose arena would not fit the free memory is cut
        into sub-batches (and the arena of a sub-batch is dropped at once when the factors are not kept)."""
        k = self.kernel
        n = int(x[group[0]].shape[0])
        q = self.dy
        device, dtype = y_mean[group[0]].device, y_mean[group[0]].dtype
        ld = dev.padded_ld(n)
        ws_bytes = max((dev.potrf_workspace_bytes(n, dtype) + 15) // 16 * 16, 16)
        esz = torch.empty((), dtype=dtype).element_size()
        per_block = n * ld * esz + ws_bytes + 6 * q * ld * esz
        free_bytes = torch.cuda.mem_get_info(device)[0]
        cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        budget = (free_bytes + cached) * (0.8 if keep_factors else 0.4)
        per_call = int(max(1, min(len(group), budget // per_block)))
        # the layer's arrays: every region view is a slice of them (regions are contiguous ranges,
        # Inputs.py:57-60), so a block is a row offset into them
        y_all, x_all = _layer_array(y_mean, group), _layer_array(x, group)
        f_all, t_all = _layer_array(f_bar, group), _layer_array(train_out, group)
        def row_of(views, l):
            return views[l].storage_offset() // views[l].stride(0)
        for c0 in range(0, len(group), per_call):
            sub = group[c0:c0 + per_call]
            nb = len(sub)
            karena = torch.empty((nb, n, ld), dtype=dtype, device=device)
            ws_arena = torch.empty((nb, ws_bytes), dtype=torch.uint8, device=device)
            info = torch.zeros(nb, dtype=torch.int32, device=device)
            bias = torch.empty((nb, q), dtype=dtype, device=device)
            noise = torch.empty(nb, dtype=dtype, device=device)
            z = torch.empty((nb, n, q), dtype=dtype, device=device)
            alpha = torch.empty((nb, n, q), dtype=dtype, device=device)
            rows_y = [row_of(y_mean, l) for l in sub]
            if any(row_of(v, l) != r for v in (x, f_bar, train_out) for l, r in zip(sub, rows_y)):
                raise ValueError('the region views of a layer must be the same row ranges of x, y, f_bar and train_out')
            starts = torch.tensor(rows_y, dtype=torch.int64).to(device, non_blocking=True)
            dev.layer_fit(x_all, y_all, f_all, t_all, starts, n, k.l, k.sf, -1.0 if k.noise is None else float(k.noise),
                          NOISE_FRACTION, NOISE_FLOOR * k.sf, shared_bias, shared_noise, karena, ws_arena, info, bias, noise,
                          z, alpha)
            batch = _FittedBatch(sub, n, starts, karena, ws_arena, z, bias, noise) if keep_factors else None
            if batch is not None:
                self.batches.append(batch)
            for i, l in enumerate(sub):
                blk = DenseBlock(x[l], k)
                blk.info = info[i:i + 1]
                blk.alpha = alpha[i]
                blk.bias = bias[i]
                blk.noise = noise[i:i + 1]
                if keep_factors:
                    blk.lbuf, blk.ws, blk.z = karena[i], ws_arena[i], z[i]
                    blk.batch, blk.batch_index = batch, i
                self.blocks[l] = blk

    def predict_layer(self, x_all, xs, test_bounds, owned, mean, var, add_noise, fan_factory):
        """Accumulate the layer's predictive mean and variance at the test points: test block l =
        rows test_bounds[l] of ``xs``, served by training block l (MRGP.py:782-803).  Blocks that were fitted
        together and have equally many test points go through ONE batched call per sub-batch
        (cimrgp_layer_predict); the others one by one on the stream pool."""
        done = set()
        for bt in self.batches:
            by_ns = {}
            for i, l in enumerate(bt.regions):
                if l in owned:
                    a, b = (int(v) for v in test_bounds[l])
                    by_ns.setdefault(b - a, []).append((i, a))
            for ns, items in by_ns.items():
                idx = [i for i, _ in items]
                contiguous = idx == list(range(idx[0], idx[0] + len(idx)))
                if ns <= 0 or len(items) < 2 or not contiguous:
                    continue
                ldw = dev.padded_ld(bt.n)
                esz = xs.element_size()
                free_bytes = torch.cuda.mem_get_info(xs.device)[0] + torch.cuda.memory_reserved(xs.device) \
                    - torch.cuda.memory_allocated(xs.device)
                per_call = int(max(1, min(len(items), (0.5 * free_bytes) // max(1, ns * ldw * esz))))
                for c0 in range(0, len(items), per_call):
                    part = items[c0:c0 + per_call]
                    i0, nb = part[0][0], len(part)
                    t_starts = torch.tensor([a for _, a in part], dtype=torch.int64).to(xs.device, non_blocking=True)
                    dev.layer_predict(x_all, bt.starts[i0:i0 + nb], bt.n, xs, t_starts, ns, self.kernel.l, self.kernel.sf,
                                      bt.karena[i0:i0 + nb], bt.ws_arena[i0:i0 + nb], bt.z[i0:i0 + nb], bt.bias[i0:i0 + nb],
                                      bt.noise[i0:i0 + nb] if add_noise else None, mean, var)
                done.update(bt.regions[i] for i, _ in items)
        rest = [l for l in owned if l not in done]
        if not rest:
            return
        fan = fan_factory(len(rest), max(self.blocks[l].n for l in rest))
        for l in rest:
            a, b = (int(v) for v in test_bounds[l])
            with torch.cuda.stream(fan.stream()):
                self.blocks[l].predict(xs[a:b], mean[a:b], var[a:b], add_noise=add_noise)
        fan.join()

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
