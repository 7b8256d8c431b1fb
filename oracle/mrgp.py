"""Oracle (test infrastructure): the multiresolution residual chain over dense
exact-GP blocks, and the ``GP_RBF`` plugin semantics, in FP64 NumPy.

Structure follows the reference (citations into /root/reference):
  * blocks (j, l) = contiguous index ranges of ``IndexSetUniform``
    (IndexSetGenerator.py:51-65), inputs gathered per block (Inputs.py:57-60);
  * layer j+1 is fitted on  y - f_bar,  f_bar = sum over coarser layers of the
    per-region predictions at the training points, scattered back by index
    set (Stats.py:126-157; targets = raw observations per region,
    LatentOutputs.py:11-18, minus f_bar, Posteriors.py:68);
  * per-region bias and noise (``bias_region_specific`` /
    ``noise_region_specific``, MRGP.py:27-28);
  * prediction = sum over layers of the concatenated per-region predictions
    (MRGP.py:782-803); variance = sum over layers (MRGP.py:902-905);
  * test points are split by an index set built on N* with the same
    resolution/divider; test block (j,l) is served by training block (j,l)
    (MRGP.py:757-803, scripts/tests/ciMRGP_vs_fiMRGP.py:62).
The per-block arithmetic is the dense exact GP of ``oracle.dense`` (PARITY
UNPINNED there, see that module).
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .dense import block_fit, block_predict
from .structure import zscore_fit, zscore_apply

NOISE_FRACTION = 0.01     # RegressionInput.py:62  labels.var()*0.01
NOISE_FLOOR = 1e-8        # times sf2; keeps a constant residual block PD


@dataclass
class DenseLayerSpec:
    """Fixed RBF hyper-parameters of one resolution (per-layer kernel objects,
    MRGP.py:71-79).  ``noise=None`` -> 0.01 * var(block targets)."""
    ell: float = 1.0
    sf2: float = 1.0
    noise: Optional[float] = None


def _noise_from_targets(r, spec):
    """Fixed noise of the kernel object, else the plugin's rule applied to the
    block: 0.01 x the pooled population variance of the targets about their
    per-column block means (== ``labels.var()*0.01`` on z-scored labels,
    RegressionInput.py:62)."""
    if spec.noise is not None:
        return float(spec.noise)
    pooled = float(np.mean((r - r.mean(axis=0)) ** 2))
    return max(NOISE_FRACTION * pooled, NOISE_FLOOR * spec.sf2)


def mrgp_fit(x, y, bounds, specs, bias_region_specific=True,
             noise_region_specific=True):
    """One coarse-to-fine sweep.  x (N x d) already normalised, y (N x q).

    Returns a list over layers of lists over regions of dicts
    (a, b, bias, noise, alpha, L) plus the final f_bar (N x q).
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    f_bar = np.zeros_like(y)
    model = []
    for j, layer in enumerate(bounds):
        spec = specs[j]
        resid = y - f_bar
        shared_bias = resid.mean(axis=0)
        shared_noise = _noise_from_targets(resid, spec)
        mu_layer = np.zeros_like(y)
        blocks = []
        for (a, b) in layer:
            a, b = int(a), int(b)
            r = resid[a:b]
            bias = r.mean(axis=0) if bias_region_specific else shared_bias
            rc = r - bias
            noise = _noise_from_targets(r, spec) if noise_region_specific else shared_noise
            fit = block_fit(x[a:b], rc, spec.ell, spec.sf2, noise)
            # K_noiseless alpha = (K + noise I) alpha - noise alpha = rc - noise alpha
            mu_layer[a:b] = rc - noise * fit['alpha'] + bias
            blocks.append(dict(a=a, b=b, bias=bias, noise=noise,
                               alpha=fit['alpha'], L=fit['L']))
        f_bar = f_bar + mu_layer
        model.append(blocks)
    return model, f_bar


def mrgp_predict(x, model, specs, xs, test_bounds, want_var=True,
                 include_noise=True):
    """Sum over layers of concatenated per-region predictions at ``xs``
    (N* x d, normalised with the TRAIN statistics)."""
    x = np.asarray(x, dtype=np.float64)
    xs = np.asarray(xs, dtype=np.float64)
    n_layers = len(test_bounds)
    if n_layers > len(model):
        raise ValueError('resolution in the test index set must be smaller or equal '
                         'to that in the train set.')
    q = model[0][0]['alpha'].shape[1]
    mean = np.zeros((xs.shape[0], q))
    var = np.zeros(xs.shape[0]) if want_var else None
    for j in range(n_layers):
        spec = specs[j]
        if len(test_bounds[j]) != len(model[j]):
            raise ValueError('number of regions in the training must be the same as test.')
        for l, (ta, tb) in enumerate(test_bounds[j]):
            blk = model[j][l]
            ta, tb = int(ta), int(tb)
            m, v = block_predict(x[blk['a']:blk['b']], blk, xs[ta:tb], spec.ell,
                                 spec.sf2, want_var)
            mean[ta:tb] += m + blk['bias']
            if want_var:
                var[ta:tb] += v
                if include_noise and j == n_layers - 1:
                    var[ta:tb] += blk['noise']
    return mean, var


def gp_rbf_fit(inputs, labels, ell=1.0, sf2=1.0):
    """``GP_RBF().fit([inputs, labels])`` with FIXED hyper-parameters.

    RegressionInput.py:16-34,58-62: z-score inputs and labels (population std),
    isotropic RBF with GPy defaults l=1, sf2=1, Gaussian noise initialised to
    ``labels.var()*0.01`` on the z-scored labels.  ``model.optimize()``
    (RegressionInput.py:63) is NOT reproduced.
    """
    stats = zscore_fit(inputs, labels)
    xz, yz = zscore_apply(stats, inputs=inputs, labels=labels)
    noise = float(yz.var()) * NOISE_FRACTION
    fit = block_fit(xz, yz, ell, sf2, noise)
    return dict(stats=stats, xz=xz, fit=fit, ell=ell, sf2=sf2, noise=noise)


def gp_rbf_predict(state, test_inputs, want_var=False):
    """``GP_RBF().predict(test)``: z-score test inputs with the train stats,
    predictive mean, un-z-score the labels (RegressionInput.py:36-42,66-67)."""
    xs = zscore_apply(state['stats'], inputs=test_inputs)
    mean, var = block_predict(state['xz'], state['fit'], xs, state['ell'],
                              state['sf2'], want_var)
    mean = zscore_apply(state['stats'], inverse_labels=mean)
    return (mean, var) if want_var else mean


def gp_lml_and_grad(x, y, ell, sf2, noise):
    """Log marginal likelihood of an exact GP with K = sf2 E + noise I (q outputs sharing K)
    and its gradient w.r.t. (log sf2, log ell, log noise) -- Rasmussen & Williams eq. 5.8/5.9.
    This is the objective GPy's ``model.optimize()`` (RegressionInput.py:63) climbs."""
    import scipy.linalg as sla
    from .dense import rbf_gram
    n, q = y.shape
    kf = rbf_gram(x, None, ell, sf2, 0.0)
    k = kf + noise * np.eye(n)
    chol = sla.cholesky(k, lower=True)
    alpha = sla.cho_solve((chol, True), y)
    lml = -0.5 * np.sum(y * alpha) - q * np.sum(np.log(np.diag(chol))) - 0.5 * n * q * np.log(2 * np.pi)
    kinv = sla.cho_solve((chol, True), np.eye(n))
    g = alpha @ alpha.T - q * kinv
    d2 = -2.0 * ell * ell * np.log(np.maximum(kf / sf2, 1e-300))
    grad = np.array([0.5 * np.sum(g * kf), 0.5 * np.sum(g * kf * d2 / (ell * ell)), 0.5 * noise * np.trace(g)])
    return lml, grad


def gp_rbf_optimize(inputs, labels, max_iters=1000):
    """``GP_RBF().fit`` INCLUDING ``model.optimize()``: L-BFGS-B on -LML over
    (log sf2, log ell, log noise) from GPy's defaults (1, 1, 0.01 var)."""
    from scipy.optimize import minimize
    stats = zscore_fit(inputs, labels)
    xz, yz = zscore_apply(stats, inputs=inputs, labels=labels)
    theta0 = np.log([1.0, 1.0, float(yz.var()) * NOISE_FRACTION])

    def objective(theta):
        sf2, ell, noise = np.exp(theta)
        try:
            lml, grad = gp_lml_and_grad(xz, yz, ell, sf2, noise)
        except np.linalg.LinAlgError:
            return 1e100, np.zeros(3)
        return -lml, -grad

    res = minimize(objective, theta0, jac=True, method='L-BFGS-B', options=dict(maxiter=max_iters))
    sf2, ell, noise = np.exp(res.x)
    fit = block_fit(xz, yz, ell, sf2, noise)
    return dict(stats=stats, xz=xz, fit=fit, ell=ell, sf2=sf2, noise=noise, result=res)


def gp_lml_and_grad_ard(x, y, ells, sf2, noise):
    """ARD twin of :func:`gp_lml_and_grad` (GPy ``RBF(ARD=True)``, the reference's comparison
    script scripts/tests/GPRBF_vs_ciMRGP_vs_fiMRGP.py:118): one length-scale per input dimension;
    gradient w.r.t. (log sf2, log l_1 .. log l_d, log noise)."""
    import scipy.linalg as sla
    from .dense import rbf_gram
    ells = np.asarray(ells, dtype=np.float64)
    n, q = y.shape
    xs = x / ells
    kf = rbf_gram(xs, None, 1.0, sf2, 0.0)
    chol = sla.cholesky(kf + noise * np.eye(n), lower=True)
    alpha = sla.cho_solve((chol, True), y)
    lml = -0.5 * np.sum(y * alpha) - q * np.sum(np.log(np.diag(chol))) - 0.5 * n * q * np.log(2 * np.pi)
    g = alpha @ alpha.T - q * sla.cho_solve((chol, True), np.eye(n))
    grad = [0.5 * np.sum(g * kf)]
    for k in range(x.shape[1]):
        dk = xs[:, k][:, None] - xs[:, k][None, :]
        grad.append(0.5 * np.sum(g * kf * dk * dk))
    grad.append(0.5 * noise * np.trace(g))
    return lml, np.array(grad)


def gp_rbf_optimize_ard(inputs, labels, max_iters=1000):
    """``GPRegression(RBF(ARD=True))`` + ``optimize()`` on z-scored data, L-BFGS-B from GPy's defaults."""
    from scipy.optimize import minimize
    stats = zscore_fit(inputs, labels)
    xz, yz = zscore_apply(stats, inputs=inputs, labels=labels)
    d = xz.shape[1]
    theta0 = np.log([1.0] + [1.0] * d + [float(yz.var()) * NOISE_FRACTION])

    def objective(theta):
        try:
            lml, grad = gp_lml_and_grad_ard(xz, yz, np.exp(theta[1:1 + d]), np.exp(theta[0]), np.exp(theta[-1]))
        except np.linalg.LinAlgError:
            return 1e100, np.zeros(d + 2)
        return -lml, -grad

    res = minimize(objective, theta0, jac=True, method='L-BFGS-B', options=dict(maxiter=max_iters))
    sf2, ells, noise = np.exp(res.x[0]), np.exp(res.x[1:1 + d]), np.exp(res.x[-1])
    fit = block_fit(xz / ells, yz, 1.0, sf2, noise)
    return dict(stats=stats, xz=xz / ells, fit=fit, ell=1.0, ells=ells, sf2=sf2, noise=noise, result=res)


def gp_rbf_predict_ard(state, test_inputs, want_var=False):
    xs = zscore_apply(state['stats'], inputs=test_inputs) / state['ells']
    mean, var = block_predict(state['xz'], state['fit'], xs, 1.0, state['sf2'], want_var)
    mean = zscore_apply(state['stats'], inverse_labels=mean)
    return (mean, var) if want_var else mean
