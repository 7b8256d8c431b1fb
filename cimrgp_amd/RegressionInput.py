"""Exact-GP regression plugin: the drop-in boundary of the dense path.

``RegressionMethod`` mirrors the reference's plugin base class
(RegressionInput.py:10-52): ``fit(train_data=[X, Y]) -> bool`` and
``predict(test_data) -> ndarray`` with column-wise z-scoring of inputs and
labels (population std) done by the base class.  ``GP_RBF`` replaces the
GPy-backed subclass (RegressionInput.py:55-67) with the HIP kernels: isotropic
RBF (GPy defaults l = 1, variance = 1 on the z-scored inputs), Gaussian noise
``labels.var() * 0.01`` on the z-scored labels; hyper-parameters are FIXED
(``model.optimize()``, RegressionInput.py:63, is listed as next in SURVEY 8f).
"""
import abc

import numpy as np
import torch

from . import device as dev
from .KernelClass import RBFKernel
from .Posteriors import DenseBlock, NOISE_FRACTION


class RegressionMethod(object):
    __metaclass__ = abc.ABCMeta

    def __init__(self):
        self.preprocess = True

    def _preprocess(self, data, train):
        """Zero-mean, unit-variance normalisation by default."""
        if train:
            inputs, labels = data
            self.data_mean = inputs.mean(axis=0)
            self.data_std = inputs.std(axis=0)
            self.labels_mean = labels.mean(axis=0)
            self.labels_std = labels.std(axis=0)
            return ((inputs - self.data_mean) / self.data_std,
                    (labels - self.labels_mean) / self.labels_std)
        return (data - self.data_mean) / self.data_std

    def _reverse_trans_labels(self, labels):
        return labels * self.labels_std + self.labels_mean

    def fit(self, train_data):
        if self.preprocess:
            train_data = self._preprocess(train_data, True)
        return self._fit(train_data)

    def predict(self, test_data):
        if self.preprocess:
            test_data = self._preprocess(test_data, False)
        labels = self._predict(test_data)
        if self.preprocess:
            labels = self._reverse_trans_labels(labels)
        return labels

    @abc.abstractmethod
    def _fit(self, train_data):
        """Fit the model. Return True if successful"""
        return True

    @abc.abstractmethod
    def _predict(self, test_data):
        """Predict on test data"""
        return None


class GP_RBF(RegressionMethod):
    name = 'GP_RBF'

    def __init__(self, lengthscale=1., variance=1., dtype='f64', device=None):
        super(GP_RBF, self).__init__()
        self.kernel = RBFKernel(l=lengthscale, sf=variance)
        self.dtype = dev.as_torch_dtype(dtype)
        self.device = device
        self.block = None

    def _fit(self, train_data):
        inputs, labels = train_data
        device = dev.require_gpu(self.device)
        inputs = np.atleast_2d(np.asarray(inputs, dtype=np.float64))
        labels = np.atleast_2d(np.asarray(labels, dtype=np.float64))
        # the noise of the plugin is a property of the (z-scored) labels as a whole
        self.kernel.noise = float(labels.var()) * NOISE_FRACTION
        x = dev.to_device(inputs, self.dtype, device)
        y = dev.to_device(labels, self.dtype, device)
        self.block = DenseBlock(x, self.kernel)
        zero_bias = torch.zeros(y.shape[1], dtype=self.dtype, device=device)
        sink = torch.zeros_like(y)
        self.block.fit(y, None, sink, shared_bias=zero_bias)
        dev.raise_if_not_pd(self.block.info)
        return True

    def _predict(self, test_data):
        return self._predict_mean_var(test_data, want_var=False)[0]

    def _predict_mean_var(self, test_data, want_var):
        blk = self.block
        xs = dev.to_device(np.atleast_2d(np.asarray(test_data, dtype=np.float64)), self.dtype, blk.x.device)
        q = blk.alpha.shape[1]
        mean = torch.zeros((xs.shape[0], q), dtype=self.dtype, device=xs.device)
        var = torch.zeros(xs.shape[0], dtype=self.dtype, device=xs.device) if want_var else None
        blk.predict(xs, mean, var)
        return (mean.double().cpu().numpy(), None if var is None else var.double().cpu().numpy())

    def predict_with_variance(self, test_data):
        """Mean (un-z-scored) and latent predictive variance (in z-scored label units
        times labels_std^2 per column is left to the caller; returned as is)."""
        if self.preprocess:
            test_data = self._preprocess(test_data, False)
        mean, var = self._predict_mean_var(test_data, want_var=True)
        if self.preprocess:
            mean = self._reverse_trans_labels(mean)
        return mean, var
