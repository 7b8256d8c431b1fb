"""Basis intervals of the Laplacian eigenfunctions: the fixed data-range rule and the adaptive
update of ciMRGP (reference BasisInterval.py:6-134).

``BasisInterval(use_prior=True, opt_interval_factor=(1., 1.2))`` and
``max_input_range_by_factor_of`` are the reference's.  ``learn`` is called by the model once per
layer and sweep; where the reference rebuilds an (n x m) basis matrix on the host for every
probe of the scalar minimiser, here a probe is one launch of ``cimrgp_basis_moments`` with the
candidate interval (the kernel regenerates the basis in registers) and a few m-sized host sums.
"""
import warnings

import numpy as np
import torch
from scipy import optimize

from . import device as dev


class BasisInterval(object):

    def __init__(self, use_prior=True, opt_interval_factor=(1., 1.2)):
        self.basis_interval = None
        self.use_prior = use_prior
        self.opt_interval_factor = opt_interval_factor

    def max_input_range_by_factor_of(self, inputs, factor):
        return factor * np.max(np.abs(inputs), axis=0)

    def learn(self, model, layer, region, targets):
        """New interval (length dx) of block (layer, region): per input dimension p, the bounded
        scalar minimiser of the reference (``scipy.optimize.fminbound``) on [max|x_p| f0,
        min(n_basis, max|x_p| f0 f1)] applied to ``_objective``, the other dimensions held at
        their current intervals (BasisInterval.py:64-92).  ``targets``: the (n x dy) device tensor
        the block was fitted to in this sweep."""
        j, l = layer, region
        a, b = (int(v) for v in model.index_set_obj.bounds[j][l])
        st = model.stats_obj[j]
        x_dev = model._x_dev[a:b]
        fbar = model._latent[j][0][a:b]
        zero = torch.zeros(model.dy, dtype=targets.dtype, device=targets.device)
        # 2 f_bar - y, from the residual kernel applied twice: f_bar - (y - f_bar)
        folded = dev.residual(fbar, dev.residual(targets, fbar, zero), zero)
        eau = st.scale_axis_mean[l]
        pack = dict(x=x_dev, folded=folded, eau=eau, bias=np.asarray(st.bias_of(l), dtype=np.float64), tau=st.noise_of(l),
                    quad=2.0 * np.sum(eau * eau, axis=0) + st.scale_axis_central_moment2[l],
                    ard_moment=(model.shared_stats.ard_mean if hasattr(model, 'shared_stats') else st.ard_mean[l])
                    * st.scale_moment2[l],
                    spectral=model.spectral_density_obj[j], m=model.n_basis, zeros=np.zeros_like(eau))
        current = np.asarray(model.train_basis_intervals[j][l], dtype=np.float64)
        x_host = model._x_host[a:b]
        orders = np.arange(1, model.n_basis + 1, dtype=np.float64)
        out = np.zeros(model.dx)
        for p in range(model.dx):
            others = [k for k in range(model.dx) if k != p]
            lam_others = np.sum((np.pi * orders[:, None] / (2.0 * current[None, others])) ** 2, axis=1)
            low = np.max(np.abs(x_host[:, p])) * self.opt_interval_factor[0]
            high = min(model.n_basis, low * self.opt_interval_factor[1])
            if high < low:
                warnings.warn("Number of basis functions is less than the input range; the result might be "
                              "suboptimal.  Try an interval factor below 1 or normalised inputs.")
                high = low * self.opt_interval_factor[1]
            out[p] = optimize.fminbound(self._objective, low, high, args=(p, current, lam_others, orders, pack), full_output=0)
        return out

    def _objective(self, candidate, p, current, lam_others, orders, pack):
        """Minus the interval-dependent part of the bound (BasisInterval.py:94-134): with
        psi = Phi under the candidate interval,
            -1/2 E[tau] sum_i [ (2|E[au]_i|^2 + c2_i) sum psi_i^2 + 2 E[au]_i . sum psi_i (2(b + f_bar) - y) ]
            -1/2 sum_i [ log S_i - 1/2 E[alpha_i] E[a_i^2] / S_i ],   S_i = S(sqrt(lambda_i)) under the candidate."""
        interval = current.copy()
        interval[p] = candidate
        mom = dev.basis_moments(pack['x'], interval, pack['m'], pack['folded'], None, None, pack['zeros'])
        cross = np.einsum('ci,ic->i', pack['eau'], 2.0 * np.outer(mom.colsum, pack['bias']) + mom.proj)
        ll = -0.5 * pack['tau'] * np.sum(pack['quad'] * mom.colsum2 + 2.0 * cross)
        if self.use_prior is False:
            return -ll
        lam = (np.pi * orders / (2.0 * candidate)) ** 2 + lam_others
        spec = np.ones_like(lam) if pack['spectral'] is None else np.asarray(pack['spectral'].spectral(np.sqrt(lam)))
        prior = -0.5 * np.sum(np.log(spec) - 0.5 * pack['ard_moment'] / spec)
        return -(ll + prior)
