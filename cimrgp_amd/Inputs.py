"""Input holder / per-block gather and the learned input warp (reference Inputs.py:8-60).

Regions are contiguous sample ranges, so ``get_inputs`` is a slice -- on the device a pointer
offset, never a gather kernel.  With ``learn_inputs=True`` the inputs are warped onto a regular
grid z = linspace(min x, max x, N) by an exact RBF GP x -> z (Inputs.py:11-22), which is the
only caller of the dense GP inside the reference; above 3000 samples the reference fits that GP
on a 3000-point subsample (Inputs.py:23-48) -- reproduced here, with the HIP-backed ``GP_RBF``.
"""
import random

import numpy as np

from .IndexSetGenerator import IndexSetUniform


class Inputs(object):
    def __init__(self, x, index_set, learn_inputs=False, full_x=None, input_model=None, model_factory=None):
        """``x``: (N x d) NumPy array (already normalised by the caller, MRGP.py:69).
        ``model_factory``: callable returning a fresh ``RegressionMethod`` (default: the HIP
        ``GP_RBF`` with the optimisation step, as the reference's ``GP_RBF()``)."""
        self.learn_inputs = learn_inputs
        self.index_set = index_set
        if self.learn_inputs is True:
            if model_factory is None:
                from .RegressionInput import GP_RBF
                model_factory = lambda: GP_RBF(optimize=True)
            z = np.atleast_2d(np.linspace(start=np.min(x), stop=np.max(x), num=x.shape[0])).T
            if full_x is None:
                train_data = [x, z]
            else:
                z_full = np.atleast_2d(np.linspace(start=np.min(full_x), stop=np.max(full_x),
                                                   num=full_x.shape[0])).T
                train_data = [full_x, z_full]
            if input_model is None:
                if train_data[0].shape[0] < 3001:
                    self.input_model = model_factory()
                    self.input_model.fit(train_data)
                else:
                    n_samps = train_data[0].shape[0]
                    n_repeats = 1
                    min_length = 3000
                    n_divide = self._get_best_divider(n_samps, rate=random.uniform(.1, .2))
                    regions = IndexSetUniform(sample_length=n_samps, resolution=1, divider=n_divide).index_set[-1]
                    ids_all = np.arange(n_samps)
                    self.input_model = []
                    for _ in range(n_repeats):
                        ids_l = [int(np.random.permutation(np.asarray(r))[0]) for r in regions]
                        if min_length > len(ids_l):
                            rem_ids = np.delete(ids_all, ids_l)
                            ids_rep = list(np.random.permutation(rem_ids)[0:min_length - len(ids_l)]) + ids_l
                        else:
                            ids_rep = ids_l
                        ids = np.sort(np.unique(ids_rep))
                        model = model_factory()
                        model.fit([train_data[0][ids, :], train_data[1][ids, :]])
                        self.input_model.append(model)
            else:
                self.input_model = input_model
            self.z = train_data[1]
            self.x = z
        else:
            self.input_model = input_model
            self.x = x

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
        if n_samps < 10000:
            factor = 1 * rate
        elif n_samps < 100000:
            factor = 1e-1 * rate
        elif n_samps < 1000000:
            factor = 1e-2 * rate
        else:
            factor = 1e-3 * rate
        return int(np.floor(factor * n_samps))
