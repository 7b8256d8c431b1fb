"""Exact-GP regression plugin: the drop-in boundary of the dense path.

``RegressionMethod`` mirrors the reference's plugin base class
(RegressionInput.py:10-52): ``fit(train_data=[X, Y]) -> bool`` and
``predict(test_data) -> ndarray`` with column-wise z-scoring of inputs and
labels (population std) done by the base class.  ``GP_RBF`` replaces the
GPy-backed subclass (RegressionInput.py:55-67) with the HIP kernels: isotropic
RBF (GPy defaults l = 1, variance = 1 on the z-scored inputs), Gaussian noise
``labels.var() * 0.01`` on the z-scored labels.  ``optimize=True`` reproduces the
``model.optimize()`` step (RegressionInput.py:63): L-BFGS-B (SciPy, the optimiser GPy's
default ``'lbfgsb'`` wraps) on the log marginal likelihood over (variance, length-scale,
noise), all constrained positive through a log transform, starting from those defaults;
objective and gradient are evaluated on the GPU (one Cholesky, K^-1 = L^-T L^-1 through the
MFMA kernels, one fused reduction for the gradient).  Default is fixed hyper-parameters.
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

    def __init__(self, lengthscale=1., variance=1., dtype='f64', device=None, optimize=True, max_iters=1000, ARD=False):
        """``ARD=True``: one length-scale per input dimension (GPy's ``RBF(ARD=True)``, which the
        reference's comparison script fits, scripts/tests/GPRBF_vs_ciMRGP_vs_fiMRGP.py:118; the plugin
        itself is ``ARD=False``, RegressionInput.py:60).  Inputs are divided by their length-scales
        and every kernel then runs with unit length-scale; ``self.lengthscales`` holds the vector.
        ``optimize=True`` (default) is the reference's behaviour: its ``_fit`` always calls
        ``model.optimize()`` (RegressionInput.py:63).  ``optimize=False`` keeps the starting
        hyper-parameters (GPy's defaults l = 1, variance = 1, noise = 1 % of the label variance):
        the opt-in fast path, and the definition of the fixed-parameter parity target."""
        super(GP_RBF, self).__init__()
        self._initial = (float(lengthscale), float(variance))
        self.ARD = bool(ARD)
        self.lengthscales = None             # ARD: (d,) vector after fit
        self.kernel = RBFKernel(l=lengthscale, sf=variance)
        self.dtype = dev.as_torch_dtype(dtype)
        self.device = device
        self.block = None
        self.optimize = optimize
        self.max_iters = max_iters
        self.optimizer_result = None

    # ---- log marginal likelihood and its gradient, on the GPU ----------------------------
    def log_marginal_likelihood(self, x, y, ell, sf, noise, want_grad=True):
        """LML of targets y (device, n x q) under K = sf E(ell) + noise I, and its gradient
        w.r.t. (log sf, log ell, log noise).  Raises LinAlgError if K is not PD."""
        n, q = y.shape
        kbuf = dev.rbf_gram(x, ell, sf, noise, lower_only=True)
        ws, info = dev.potrf(kbuf, n)
        alpha = y.clone()
        dev.potrs(kbuf, n, ws, alpha)
        dev.raise_if_not_pd(info)
        half_logdet = float(dev.logdet_half(kbuf, n).item())
        fit = float((y * alpha).sum().item())
        lml = -0.5 * fit - q * half_logdet - 0.5 * n * q * np.log(2 * np.pi)
        if not want_grad:
            return lml, None
        # K^-1 = U U^T with U = L^-T: the identity carried through the row-wise solve, then a SYRK
        u = dev.alloc_matrix(n, n, x.dtype, x.device)
        u.zero_()
        u[:n, :n].fill_diagonal_(1.0)
        dev.trsm_rows(kbuf, n, ws, u, n)
        kinv = dev.alloc_matrix(n, n, x.dtype, x.device)
        kinv.zero_()
        dev.syrk_lower(kinv, u, n, n)            # lower(kinv) = -K^-1
        kinv.neg_()
        grad = dev.lml_grad(x, kinv, n, alpha, ell, sf, noise).cpu().numpy()
        return lml, grad

    def log_marginal_likelihood_ard(self, x, y, ells, sf, noise):
        """ARD twin of :meth:`log_marginal_likelihood`: ``ells`` (d,) length-scales; gradient w.r.t.
        (log sf, log l_1 .. log l_d, log noise)."""
        scale = torch.as_tensor(1.0 / np.asarray(ells, dtype=np.float64), dtype=x.dtype, device=x.device)
        xs = (x * scale).contiguous()
        n, q = y.shape
        kbuf = dev.rbf_gram(xs, 1.0, sf, noise, lower_only=True)
        ws, info = dev.potrf(kbuf, n)
        alpha = y.clone()
        dev.potrs(kbuf, n, ws, alpha)
        dev.raise_if_not_pd(info)
        half_logdet = float(dev.logdet_half(kbuf, n).item())
        lml = -0.5 * float((y * alpha).sum().item()) - q * half_logdet - 0.5 * n * q * np.log(2 * np.pi)
        u = dev.alloc_matrix(n, n, x.dtype, x.device)
        u.zero_()
        u[:n, :n].fill_diagonal_(1.0)
        dev.trsm_rows(kbuf, n, ws, u, n)
        kinv = dev.alloc_matrix(n, n, x.dtype, x.device)
        kinv.zero_()
        dev.syrk_lower(kinv, u, n, n)
        kinv.neg_()
        return lml, dev.lml_grad_ard(xs, kinv, n, alpha, sf, noise).cpu().numpy()

    def _optimize_ard(self, x, y, noise0):
        from scipy.optimize import minimize
        d = x.shape[1]
        theta0 = np.log([self.kernel.sf] + [self.kernel.l] * d + [noise0])

        def objective(theta):
            sf, ells, noise = np.exp(theta[0]), np.exp(theta[1:1 + d]), np.exp(theta[-1])
            try:
                lml, grad = self.log_marginal_likelihood_ard(x, y, ells, sf, noise)
            except np.linalg.LinAlgError:
                return 1e100, np.zeros(d + 2)
            return -lml, -grad

        res = minimize(objective, theta0, jac=True, method='L-BFGS-B', options=dict(maxiter=self.max_iters))
        self.optimizer_result = res
        self.lengthscales = np.exp(res.x[1:1 + d])
        self.kernel = RBFKernel(l=1.0, sf=float(np.exp(res.x[0])), noise=float(np.exp(res.x[-1])))

    def _optimize(self, x, y, noise0):
        from scipy.optimize import minimize
        theta0 = np.log([self.kernel.sf, self.kernel.l, noise0])

        def objective(theta):
            sf, ell, noise = np.exp(theta)
            try:
                lml, grad = self.log_marginal_likelihood(x, y, ell, sf, noise)
            except np.linalg.LinAlgError:
                return 1e100, np.zeros(3)
            return -lml, -grad

        res = minimize(objective, theta0, jac=True, method='L-BFGS-B', options=dict(maxiter=self.max_iters))
        self.optimizer_result = res
        sf, ell, noise = np.exp(res.x)
        self.kernel = RBFKernel(l=float(ell), sf=float(sf), noise=float(noise))

    def _fit(self, train_data):
        inputs, labels = train_data
        device = dev.require_gpu(self.device)
        inputs = np.atleast_2d(np.asarray(inputs, dtype=np.float64))
        labels = np.atleast_2d(np.asarray(labels, dtype=np.float64))
        # every fit starts from the constructor's values (a re-fit does not start from the
        # previous optimum); the noise of the plugin is a property of the (z-scored) labels as a whole
        self.kernel = RBFKernel(l=self._initial[0], sf=self._initial[1])
        self.kernel.noise = float(labels.var()) * NOISE_FRACTION
        x = dev.to_device(inputs, self.dtype, device)
        y = dev.to_device(labels, self.dtype, device)
        if self.ARD:
            if self.optimize:
                self._optimize_ard(x, y, self.kernel.noise)
            else:
                self.lengthscales = np.full(x.shape[1], self.kernel.l)
                self.kernel = RBFKernel(l=1.0, sf=self.kernel.sf, noise=self.kernel.noise)
            self._scale = torch.as_tensor(1.0 / self.lengthscales, dtype=self.dtype, device=device)
            x = (x * self._scale).contiguous()           # unit length-scale from here on
        elif self.optimize:
            self._optimize(x, y, self.kernel.noise)
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
        if self.ARD:
            xs = (xs * self._scale).contiguous()
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
