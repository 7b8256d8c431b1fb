"""Input holder, per-block view and the learned input warp (reference Inputs.py:8-60).

Regions are contiguous sample ranges, so a block's inputs are a slice -- on the device a pointer
offset, never a gather kernel.  With ``learn_inputs=True`` the inputs are replaced by a regular
grid z = linspace(min x, max x, N) and an exact RBF GP x -> z is kept for warping test inputs
(the only place the reference itself calls a dense GP); above 3000 samples the reference fits
that GP on a random subsample -- one point per block of a 2-layer uniform index set, topped up to
3000 -- which is reproduced with the same random draws, on the HIP-backed ``GP_RBF``.
"""
import random

import numpy as np

from .IndexSetGenerator import IndexSetUniform

#: sample count up to which the warp GP is fitted on everything (Inputs.py:20)
WARP_FULL_FIT_MAX = 3000


def _grid_like(values):
    """Regular grid over the range of ``values`` with as many points (Inputs.py:12,16)."""
    return np.linspace(np.min(values), np.max(values), values.shape[0])[:, None]


def _divider_for(n_samps, rate):
    """Number of blocks the subsample draws one point from: rate x n, with the rate scaled down
    by 10 per decade of n above 10^4 (Inputs.py:62-71)."""
    decades = 0 if n_samps < 10000 else (1 if n_samps < 100000 else (2 if n_samps < 1000000 else 3))
    return int(np.floor(rate * 10.0 ** (-decades) * n_samps))


def _subsample_ids(n_samps):
    """Indices of the reference's warp subsample (Inputs.py:23-42), same draw order: the block
    count from ``random.uniform``, one ``np.random.permutation`` per block, one for the top-up."""
    blocks = IndexSetUniform(sample_length=n_samps, resolution=1,
                             divider=_divider_for(n_samps, random.uniform(.1, .2))).index_set[-1]
    picked = [int(np.random.permutation(np.asarray(blk))[0]) for blk in blocks]
    missing = WARP_FULL_FIT_MAX - len(picked)
    if missing > 0:
        pool = np.delete(np.arange(n_samps), picked)
        picked = list(np.random.permutation(pool)[:missing]) + picked
    return np.unique(picked)


def space_filling_order(x, bits=16):
    """Permutation that sorts (N x d) inputs along a space-filling curve, so that the
    CONTIGUOUS index ranges of an ``IndexSetUniform`` are spatially compact regions.

    The reference partitions by sample order only (``x[T_jl, :]``, Inputs.py:57-60): whatever
    order the caller's samples arrive in defines the regions.  For 1-D inputs sorting gives
    intervals; for d = 2 this helper gives the Hilbert-curve order (any contiguous range of a
    Hilbert order is a connected, compact patch; aligned power-of-two ranges are squares),
    for d > 2 the Morton (Z-order) interleave.  Train and test points must be ordered with the
    same call; predictions come back in the sorted order (``out[perm] = pred`` undoes it).
    """
    x = np.asarray(x, dtype=np.float64)
    if x.ndim != 2:
        raise ValueError('inputs must be (N x d)')
    n, d = x.shape
    if d == 1:
        return np.argsort(x[:, 0], kind='stable')
    lo, hi = x.min(axis=0), x.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    side = 1 << int(bits)
    cells = np.minimum(((x - lo) / span * side).astype(np.int64), side - 1)
    if d == 2:
        # Hilbert index of cell (cx, cy), iterative quadrant rotation from the top bit down
        cx, cy = cells[:, 0].copy(), cells[:, 1].copy()
        key = np.zeros(n, dtype=np.int64)
        s = side >> 1
        while s > 0:
            rx = ((cx & s) > 0).astype(np.int64)
            ry = ((cy & s) > 0).astype(np.int64)
            key += s * s * ((3 * rx) ^ ry)
            flip = (ry == 0) & (rx == 1)
            cx = np.where(flip, side - 1 - cx, cx)
            cy = np.where(flip, side - 1 - cy, cy)
            swap = ry == 0
            cx, cy = np.where(swap, cy, cx), np.where(swap, cx, cy)
            s >>= 1
        return np.argsort(key, kind='stable')
    per_dim = max(1, min(int(bits), 62 // d))
    cells >>= (int(bits) - per_dim)
    key = np.zeros(n, dtype=np.int64)
    for b in range(per_dim):
        for k in range(d):
            key |= ((cells[:, k] >> b) & 1) << (b * d + (d - 1 - k))
    return np.argsort(key, kind='stable')


class Inputs(object):
    def __init__(self, x, index_set, learn_inputs=False, full_x=None, input_model=None, model_factory=None):
        """``x``: (N x d) NumPy array, already normalised by the caller (MRGP.py:69).
        ``model_factory``: callable returning a fresh ``RegressionMethod``; default: the HIP
        ``GP_RBF`` with its hyper-parameter optimisation, the counterpart of the reference's
        ``GP_RBF()``."""
        self.learn_inputs = learn_inputs
        self.index_set = index_set
        self.input_model = input_model
        self.x = x
        if learn_inputs is not True:
            return
        source = x if full_x is None else full_x
        target = _grid_like(source)
        if input_model is None:
            self.input_model = self._fit_warp(source, target, model_factory)
        self.z = target                      # the grid of full_x when that was given (Inputs.py:52)
        self.x = _grid_like(x)

    @staticmethod
    def _fit_warp(source, target, model_factory):
        if model_factory is None:
            from .RegressionInput import GP_RBF
            model_factory = lambda: GP_RBF(optimize=True)
        n_samps = source.shape[0]
        if n_samps <= WARP_FULL_FIT_MAX:
            model = model_factory()
            model.fit([source, target])
            return model
        models = []
        for _ in range(1):                   # the reference's n_repeats = 1: a one-member ensemble
            ids = _subsample_ids(n_samps)
            model = model_factory()
            model.fit([source[ids, :], target[ids, :]])
            models.append(model)
        return models

    def get_inputs(self, resolution, region):
        a, b = self.index_set.bounds[resolution][region]
        return self.x[int(a):int(b), :]

    def warp(self, test_x, full_x=None):
        """Map test inputs through the learned warp (MRGP.py:770-778): the ensemble mean when
        several models were fitted, the stored grid when ``full_x`` was given."""
        if not self.learn_inputs:
            return test_x
        if full_x is not None:
            return self.z
        if isinstance(self.input_model, list):
            return np.mean([m.predict(test_x) for m in self.input_model], axis=0)
        return self.input_model.predict(test_x)

    @staticmethod
    def _get_best_divider(n_samps, rate=0.2):
        return _divider_for(n_samps, rate)
