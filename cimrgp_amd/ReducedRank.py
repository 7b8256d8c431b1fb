"""The reference's reduced-rank multiresolution model (fiMRGP and ciMRGP) with its
N-dependent arithmetic on the GPU  --  SURVEY.md 8f rank 2.

Every block (layer j, region l) expands its targets in ``n_basis`` Laplacian eigenfunctions
(KernelClass.py:9-37) weighted by a Matern spectral prior and runs the mean-field sweep of
MRGP.py:574-731.  What scales with the number of samples is only ever

  * the sums  Phi^T r, colsum Phi, colsum Phi^2, sum r, sum |r|^2, sum f_var  over a block,
  * applying  bias + Phi E[au]^T  and  bias_var + Phi^2 c2  to n points,

and those are HIP kernels (csrc/reduced.hip) behind ``cimrgp_basis_moments`` and
``cimrgp_basis_apply``.  Neither reads Phi from memory: a row of it is regenerated in registers
from x by a sine recurrence, so a sweep streams only x, y and the latent function.
``cimrgp_laplace_basis`` materialises Phi for callers that want ``model.phi_x``.  The factor updates themselves touch
m x q numbers per block and stay on the host, in the classes below, which keep the reference's
names and attribute layout (lists over regions) so that code written against
``model.posterior_obj[j]`` / ``model.stats_obj[j]`` keeps working:

  IndependentPrior / IndependentPosterior / IndependentStats     Priors.py:137, Posteriors.py:215, Stats.py:164
  SharedPrior / Prior, SharedPosterior / Posterior, SharedStats / Stats
                                                                 Priors.py:8,57  Posteriors.py:455,9  Stats.py:369,7

Methods that took ``phi_x`` and ``y_mean`` in the reference take ``moments`` here: the list over
regions of ``device.BlockMoments`` reduced on the GPU.  Supported: region-specific and shared
noise / bias, fixed and adaptive basis intervals (``BasisInterval``), SNR-initialised noise,
input warping, the lower bound of ``fit(n_iter, tol)``; priors are the non-informative ones.
"""
import weakref

import numpy as np
import torch
from scipy.optimize import fsolve
from scipy.special import gammaln, psi

from . import device as dev
from . import dist
from .BasisInterval import BasisInterval
from .Inputs import Inputs
from .KernelClass import LaplacianEigenpairs
from .MRGP import MultiResolutionGaussianProcess

EPSILON = 1e-45         # Priors.py:5
#: 'minpack' = the reference's solver for the soft permutation (parity to rounding);
#: 'sinkhorn' = converged log-domain Sinkhorn iteration, exact, ~1e-8 away.
OMEGA_SOLVER = 'minpack'


# =================================================================================================
#  Bingham axes: normaliser by the first-order saddle point, PD repair
# =================================================================================================
def bingham_normaliser(kappa):
    """log C and rho = d log C / d kappa for a batch of Bingham concentration vectors
    ``kappa`` (..., p); first-order saddle-point approximation of Kume & Wood (2005) as used
    by the reference (computeRealBinghamConstant.py:12-154, CommonDensities.py:73-79).

    With Lambda = -kappa shifted so min(Lambda) = 0.1, t solves  1/2 sum 1/(Lambda - t) = 1
    on [0.1 - p, -0.4];  K1 is convex and increasing there and non-negative at the right end,
    so Newton started at -0.4 descends monotonically onto the root."""
    kappa = np.asarray(kappa, dtype=np.float64)
    shape = kappa.shape
    lam = -kappa.reshape(-1, shape[-1])
    p = shape[-1]
    shift = 0.1 - lam.min(axis=1, keepdims=True)
    lam = lam + shift
    t = np.full((lam.shape[0], 1), 0.1 - 0.5)
    for _ in range(100):
        u = 1.0 / (lam - t)
        f = 0.5 * u.sum(axis=1, keepdims=True) - 1.0
        step = f / (0.5 * (u * u).sum(axis=1, keepdims=True))
        t_new = np.maximum(t - step, 0.1 - p)
        if np.all(np.abs(t_new - t) <= 4e-16 * np.abs(t)):
            t = t_new
            break
        t = t_new
    gap = lam - t
    u = 1.0 / gap
    k2 = 0.5 * (u ** 2).sum(axis=1, keepdims=True)
    k3 = (u ** 3).sum(axis=1, keepdims=True)
    log_c = 0.5 * (np.log(2.0) + (p - 1) * np.log(np.pi) - np.log(k2) - np.log(gap).sum(axis=1, keepdims=True)) - t + shift
    w = 0.5 * u ** 2 / k2                                  # dt/dLambda_k along K1 == 1
    rho = 0.5 * (-u ** 3 + k3 * w) / k2 + 0.5 * (u - u.sum(axis=1, keepdims=True) * w) + w
    return log_c.reshape(shape[:-1]), rho.reshape(shape)


def isPD(mat):
    """Cholesky as the test of positive definiteness (SanityCheck.py:59-65)."""
    try:
        np.linalg.cholesky(mat)
        return True
    except np.linalg.LinAlgError:
        return False


def _pd_mask(mats):
    """Which matrices of a stack (B x p x p) pass Cholesky: the unblocked recursion LAPACK runs,
    vectorised over the stack (a pivot <= 0 or NaN fails, as in dpotrf2)."""
    batch, p = mats.shape[0], mats.shape[-1]
    low = np.zeros_like(mats)
    ok = np.ones(batch, dtype=bool)
    with np.errstate(invalid='ignore', divide='ignore'):
        for j in range(p):
            pivot = mats[:, j, j] - np.sum(low[:, j, :j] ** 2, axis=1)
            ok &= pivot > 0                                   # NaN compares false
            root = np.sqrt(np.where(pivot > 0, pivot, 1.0))
            low[:, j, j] = root
            for i in range(j + 1, p):
                low[:, i, j] = (mats[:, i, j] - np.sum(low[:, i, :j] * low[:, j, :j], axis=1)) / root
    return ok


def nearestPD(mats):
    """Nearest symmetric positive-definite matrix for every member of a stack (B x p x p), or
    for a single matrix: Higham (1988) via D'Errico's nearestSPD as varied in
    SanityCheck.py:17-57 -- polar factor from the SVD, then diagonal nudges of growing size,
    -min_eig * k^2 + spacing(|A|_F), until Cholesky passes."""
    single = mats.ndim == 2
    stack = mats[None] if single else mats
    sym = 0.5 * (stack + np.swapaxes(stack, 1, 2))
    _, sing, vt = np.linalg.svd(sym)
    polar = np.einsum('bki,bk,bkj->bij', vt, sing, vt)
    cand = 0.5 * (sym + polar)
    cand = 0.5 * (cand + np.swapaxes(cand, 1, 2))
    eps = np.spacing(np.sqrt(np.sum(stack * stack, axis=(1, 2))))
    eye = np.eye(stack.shape[-1])
    bad = np.flatnonzero(~_pd_mask(cand))
    k = 1
    while bad.size:
        low = np.min(np.real(np.linalg.eigvals(cand[bad])), axis=1)
        cand[bad] += (-low * k ** 2 + eps[bad])[:, None, None] * eye
        bad = bad[~_pd_mask(cand[bad])]
        k += 1
    return cand[0] if single else cand


def fit_axes(candidates):
    """Posteriors.py:276-290 for a stack of candidate parameter matrices (B x p x p): repair the
    ones that fail Cholesky, eigen-decompose (descending), normaliser and rho from the raw
    eigenvalues, kappa clamped at zero afterwards.  Returns (b, kappa, axes, rho, log_const)."""
    used = np.array(candidates, dtype=np.float64, copy=True)
    bad = ~_pd_mask(used)
    if np.any(bad):
        used[bad] = nearestPD(used[bad])
    vals, vecs = np.linalg.eig(used)
    vals, vecs = np.real(vals), np.real(vecs)
    order = np.argsort(vals, axis=1)[:, ::-1]
    raw = np.take_along_axis(vals, order, axis=1)
    axes = np.take_along_axis(vecs, order[:, None, :], axis=2)
    log_c, rho = bingham_normaliser(raw)
    return used, np.where(raw < 0, 0.0, raw), axes, rho, log_c


class _AxisFactors(object):
    """m Bingham factors over the unit sphere in R^dy: parameter matrices and what the
    reference caches of them (Priors.py:10-16)."""

    def __init__(self, n_basis, dy):
        self.axis_bingham_b = np.zeros((n_basis, dy, dy))
        self.axis_bingham_kappa = np.zeros((n_basis, dy))
        self.axis_bingham_axes = np.tile(np.eye(dy), (n_basis, 1, 1))
        log_c, rho = bingham_normaliser(np.zeros((n_basis, dy)))
        self.axis_bingham_rho = rho
        self.axis_bingham_log_const = log_c

    def set_axes(self, candidates, fitted=None):
        """Install the factors fitted to ``candidates`` (m x dy x dy); ``fitted`` lets a caller
        that batched several regions through ``fit_axes`` hand the slice over."""
        (self.axis_bingham_b, self.axis_bingham_kappa, self.axis_bingham_axes, self.axis_bingham_rho,
         self.axis_bingham_log_const) = fit_axes(candidates) if fitted is None else fitted

    def cov(self):
        """E[u u^T] per factor = sum_d rho_d v_d v_d^T (Stats.py:249-257)."""
        return np.einsum('iad,id,ibd->iab', self.axis_bingham_axes, self.axis_bingham_rho, self.axis_bingham_axes)


# =================================================================================================
#  priors
# =================================================================================================
class SharedPrior(_AxisFactors):
    """Priors.py:8-54: the axis and ARD priors shared by all regions (ciMRGP)."""

    def __init__(self, n_basis, dy, prior_influence=1.0):
        _AxisFactors.__init__(self, n_basis, dy)
        self.n_basis, self.dy = n_basis, dy
        self.ard_gamma_shape = EPSILON * np.ones(n_basis)
        self.ard_gamma_scale = self.ard_gamma_shape / prior_influence


class Prior(object):
    """Priors.py:57-134: per-region scale, noise and bias priors."""

    def __init__(self, n_basis, dy, n_regions, spectral_density, noise_var=None, noise_region_specific=True,
                 bias_region_specific=True):
        self.n_basis, self.dy, self.n_regions = n_basis, dy, n_regions
        self.noise_region_specific = noise_region_specific
        self.bias_region_specific = bias_region_specific
        self.scale_precision = [1.0 / np.asarray(s) for s in spectral_density]
        noise_var = 1.0 if noise_var is None else noise_var
        # a shared factor is one scalar / vector, a region-specific one a list (Priors.py:84-134)
        if noise_region_specific:
            self.noise_gamma_shape = [EPSILON] * n_regions
            self.noise_gamma_scale = [(EPSILON + 1.0) * noise_var] * n_regions
        else:
            self.noise_gamma_shape = EPSILON
            self.noise_gamma_scale = (EPSILON + 1.0) * noise_var
        if bias_region_specific:
            self.bias_normal_mean = [np.zeros(dy) for _ in range(n_regions)]
            self.bias_normal_precision = [EPSILON] * n_regions
        else:
            self.bias_normal_mean = np.zeros(dy)
            self.bias_normal_precision = EPSILON


class IndependentPrior(Prior):
    """Priors.py:137-279: the same plus one axis/ARD prior set per region (fiMRGP)."""

    def __init__(self, n_basis, dy, n_regions, spectral_density, prior_influence=1.0, noise_var=None,
                 noise_region_specific=True, bias_region_specific=True):
        Prior.__init__(self, n_basis, dy, n_regions, spectral_density, noise_var, noise_region_specific, bias_region_specific)
        self.axis = [_AxisFactors(n_basis, dy) for _ in range(n_regions)]
        self.ard_gamma_shape = [EPSILON * np.ones(n_basis) for _ in range(n_regions)]
        self.ard_gamma_scale = [s / prior_influence for s in self.ard_gamma_shape]

    @property
    def axis_bingham_b(self):
        return [a.axis_bingham_b for a in self.axis]

    @property
    def axis_bingham_log_const(self):
        return [a.axis_bingham_log_const for a in self.axis]


# =================================================================================================
#  posteriors
# =================================================================================================
def _of(value, region, regional):
    """A region's member of a factor that is either a list over regions or shared."""
    return value[region] if regional else value


class Posterior(object):
    """Per-region factors q(a|u), q(bias|tau), q(tau) of one layer (Posteriors.py:9-211)."""

    def __init__(self, prior):
        self.n_basis, self.dy, self.n_regions = prior.n_basis, prior.dy, prior.n_regions
        self.noise_region_specific = prior.noise_region_specific
        self.bias_region_specific = prior.bias_region_specific
        self.scale_precision = [p.copy() for p in prior.scale_precision]
        self.scale_mean_zeta = [np.zeros(self.n_basis) for _ in range(self.n_regions)]
        self.scale_mean_y_tilde = [np.zeros((self.dy, self.n_basis)) for _ in range(self.n_regions)]
        if self.noise_region_specific:
            self.noise_gamma_shape = list(prior.noise_gamma_shape)
            self.noise_gamma_scale = list(prior.noise_gamma_scale)
        else:
            self.noise_gamma_shape, self.noise_gamma_scale = prior.noise_gamma_shape, prior.noise_gamma_scale
        if self.bias_region_specific:
            self.bias_normal_mean = [b.copy() for b in prior.bias_normal_mean]
            self.bias_normal_precision = list(prior.bias_normal_precision)
        else:
            self.bias_normal_mean, self.bias_normal_precision = prior.bias_normal_mean.copy(), prior.bias_normal_precision

    def _ard_mean(self, stats, shared_stats, region):
        return shared_stats.ard_mean

    def update_scale_given_axis(self, moments, prior, stats, shared_stats=None, spectral_density=None, regions=None):
        """Posteriors.py:33-73 / 298-342.  y_tilde[:, i] is the projection on basis i of the
        residual without i's own term:  Phi^T r + colsum(Phi^2) E[au],  r = r0 - bias."""
        for l in (range(self.n_regions) if regions is None else regions):
            mom = moments[l]
            tau = _of(stats.noise_mean, l, self.noise_region_specific)
            bias = _of(stats.bias_mean, l, self.bias_region_specific)
            self.scale_precision[l] = self._ard_mean(stats, shared_stats, l) / spectral_density[l] + tau * mom.colsum2
            self.scale_mean_zeta[l] = tau / self.scale_precision[l]
            self.scale_mean_y_tilde[l] = (mom.proj - np.outer(mom.colsum, bias)).T \
                + stats.scale_axis_mean[l] * mom.colsum2[None, :]

    def axis_evidence(self, stats, regions=None):
        """sum over regions of 1/2 E[tau] zeta y_tilde y_tilde^T, per basis (m x dy x dy)
        (Posteriors.py:478-488); a single region gives the fiMRGP term (Posteriors.py:268-275)."""
        total = np.zeros((self.n_basis, self.dy, self.dy))
        for l in (range(self.n_regions) if regions is None else regions):
            yt = self.scale_mean_y_tilde[l]
            tau = _of(stats.noise_mean, l, self.noise_region_specific)
            total += (0.5 * tau * self.scale_mean_zeta[l])[:, None, None] * np.einsum('ai,bi->iab', yt, yt)
        return total

    def update_bias_given_noise(self, moments, prior, stats, regions=None, n_samps=None, reduce=None):
        """Posteriors.py:75-110: ``moments`` taken with the *updated* E[au].  A shared bias pools
        the residual sums of all regions (``reduce`` adds the other ranks' share, ``n_samps`` is
        the size of every region of the layer)."""
        regions = range(self.n_regions) if regions is None else regions
        if self.bias_region_specific:
            for l in regions:
                self.bias_normal_precision[l] = prior.bias_normal_precision[l] + moments[l].n
                self.bias_normal_mean[l] = (prior.bias_normal_mean[l] * prior.bias_normal_precision[l]
                                            + moments[l].resid_sum) / self.bias_normal_precision[l]
            return
        pooled = np.zeros(self.dy)
        for l in regions:
            pooled = pooled + moments[l].resid_sum
        if reduce is not None:
            pooled = reduce(pooled)
        total_n = sum(moments[l].n for l in regions) if n_samps is None else sum(n_samps)
        self.bias_normal_precision = prior.bias_normal_precision + total_n
        self.bias_normal_mean = (prior.bias_normal_mean * prior.bias_normal_precision + pooled) / self.bias_normal_precision

    def _y_var_times_n(self):
        """Whether the target variance enters the noise update multiplied by the block size:
        not in the all-regional variant, yes in the other three (Posteriors.py:135,158,176,199)."""
        return not (self.noise_region_specific and self.bias_region_specific)

    def update_noise(self, moments, y_var, prior, posterior, stats, regions=None, n_samps=None, reduce=None):
        """Posteriors.py:113-211, the four region-specific / shared combinations.  The residual of
        the mean term carries no bias; the bias enters as precision * |mean|^2 (``term4``)."""
        regions = list(range(self.n_regions) if regions is None else regions)
        times_n = self._y_var_times_n()

        def bias_terms(l):
            p0 = _of(prior.bias_normal_precision, l, self.bias_region_specific)
            w0 = _of(prior.bias_normal_mean, l, self.bias_region_specific)
            p1 = _of(posterior.bias_normal_precision, l, self.bias_region_specific)
            w1 = _of(posterior.bias_normal_mean, l, self.bias_region_specific)
            return p0 * float(np.dot(w0, w0)) - p1 * float(np.dot(w1, w1))

        def data_terms(l):
            mom = moments[l]
            var_au = float(np.dot(mom.colsum2, stats.scale_axis_central_moment2[l]))
            return mom.resid_sq + mom.fvar_sum + var_au + (y_var[l] * mom.n if times_n else y_var[l])

        if self.noise_region_specific:
            for l in regions:
                self.noise_gamma_shape[l] = prior.noise_gamma_shape[l] + 0.5 * self.dy * moments[l].n
                self.noise_gamma_scale[l] = prior.noise_gamma_scale[l] + 0.5 * (bias_terms(l) + data_terms(l))
            return
        pooled = sum(data_terms(l) for l in regions)
        if self.bias_region_specific:
            pooled += sum(bias_terms(l) for l in regions)
        if reduce is not None:
            pooled = float(reduce(np.array([pooled]))[0])
        if not self.bias_region_specific:
            pooled += bias_terms(0)
        total_n = sum(moments[l].n for l in regions) if n_samps is None else sum(n_samps)
        self.noise_gamma_shape = prior.noise_gamma_shape + 0.5 * self.dy * total_n
        self.noise_gamma_scale = prior.noise_gamma_scale + 0.5 * pooled


class IndependentPosterior(Posterior):
    """Posteriors.py:215-452: every region also owns its axes and ARD weights."""

    def _y_var_times_n(self):
        """Posteriors.py:402,422,440,463: multiplied by n except with regional noise + shared bias."""
        return not (self.noise_region_specific and not self.bias_region_specific)

    def __init__(self, prior):
        Posterior.__init__(self, prior)
        self.axis = [_AxisFactors(self.n_basis, self.dy) for _ in range(self.n_regions)]
        self.ard_gamma_shape = [s.copy() for s in prior.ard_gamma_shape]
        self.ard_gamma_scale = [s.copy() for s in prior.ard_gamma_scale]

    def _ard_mean(self, stats, shared_stats, region):
        return stats.ard_mean[region]

    def update_axis(self, prior, posterior, stats, regions=None):
        """Posteriors.py:257-290."""
        regions = list(range(self.n_regions) if regions is None else regions)
        if not regions:
            return
        prior_b = prior.axis_bingham_b
        cands = np.concatenate([np.tensordot(stats.omega[l], prior_b[l], axes=1) + self.axis_evidence(stats, regions=[l])
                                for l in regions])
        fitted = fit_axes(cands)                 # one batched pass over all regions of the layer
        m = self.n_basis
        for k, l in enumerate(regions):
            self.axis[l].set_axes(None, fitted=tuple(part[k * m:(k + 1) * m] for part in fitted))

    def update_ard(self, prior, stats, spectral_density, regions=None):
        """Posteriors.py:293-300 (the 0.5 * n_regions shape increment is the reference's)."""
        for l in (range(self.n_regions) if regions is None else regions):
            self.ard_gamma_shape[l] = stats.omega[l] @ prior.ard_gamma_shape[l] + 0.5 * stats.n_regions
            self.ard_gamma_scale[l] = stats.omega[l] @ prior.ard_gamma_scale[l] \
                + 0.5 * stats.scale_moment2[l] / spectral_density[l]

    # reference attribute layout (lists over regions)
    axis_bingham_b = property(lambda self: [a.axis_bingham_b for a in self.axis])
    axis_bingham_kappa = property(lambda self: [a.axis_bingham_kappa for a in self.axis])
    axis_bingham_rho = property(lambda self: [a.axis_bingham_rho for a in self.axis])
    axis_bingham_axes = property(lambda self: [a.axis_bingham_axes for a in self.axis])
    axis_bingham_log_const = property(lambda self: [a.axis_bingham_log_const for a in self.axis])


class SharedPosterior(_AxisFactors):
    """Posteriors.py:455-541: the axes and ARD weights all regions and layers share."""

    def __init__(self, prior):
        _AxisFactors.__init__(self, prior.n_basis, prior.dy)
        self.n_basis, self.dy = prior.n_basis, prior.dy
        for name in ('axis_bingham_b', 'axis_bingham_kappa', 'axis_bingham_rho', 'axis_bingham_axes',
                     'axis_bingham_log_const', 'ard_gamma_shape', 'ard_gamma_scale'):
            setattr(self, name, np.array(getattr(prior, name), copy=True))

    def snapshot(self):
        """What the next layer uses as its prior (``deepcopy`` in MRGP.py:578-584)."""
        return SharedPosterior(self)

    def update_axis(self, prior, evidence, shared_stats):
        """Posteriors.py:470-500; ``evidence`` = Posterior.axis_evidence summed over all regions
        (and over ranks)."""
        self.set_axes(np.tensordot(shared_stats.omega, prior.axis_bingham_b, axes=1) + evidence)

    def update_ard(self, prior, moment_over_spectral, n_regions, shared_stats):
        """Posteriors.py:503-512; ``moment_over_spectral`` = sum_l E[a^2]_l / S_l (length m)."""
        self.ard_gamma_shape = shared_stats.omega @ prior.ard_gamma_shape + 0.5 * n_regions
        self.ard_gamma_scale = shared_stats.omega @ prior.ard_gamma_scale + 0.5 * moment_over_spectral


# =================================================================================================
#  statistics (expectations under the posteriors)
# =================================================================================================
class Stats(object):
    """Stats.py:7-157."""

    def __init__(self, posterior):
        qd = posterior
        self.n_basis, self.dy, self.n_regions = qd.n_basis, qd.dy, qd.n_regions
        self.noise_region_specific = qd.noise_region_specific
        self.bias_region_specific = qd.bias_region_specific
        self.scale_axis_mean = [np.zeros((self.dy, self.n_basis)) for _ in range(self.n_regions)]
        self.scale_moment2 = [np.zeros(self.n_basis) for _ in range(self.n_regions)]
        self.scale_axis_central_moment2 = [np.zeros(self.n_basis) for _ in range(self.n_regions)]
        self.noise_mean = [0.0] * self.n_regions
        self.noise_log_mean = [0.0] * self.n_regions
        self.bias_mean = [None] * self.n_regions
        self.bias_var = [0.0] * self.n_regions
        self.update_noise(qd)
        self.update_bias(qd)

    # a region's view of factors that may be shared
    def noise_of(self, region):
        return _of(self.noise_mean, region, self.noise_region_specific)

    def noise_log_of(self, region):
        return _of(self.noise_log_mean, region, self.noise_region_specific)

    def bias_of(self, region):
        return _of(self.bias_mean, region, self.bias_region_specific)

    def bias_var_of(self, region):
        return _of(self.bias_var, region, self.bias_region_specific)

    def _axis_cov(self, stats, region):
        return stats.axis_cov

    def update_scale(self, posterior, stats, regions=None):
        """Stats.py:66-100 / 264-297: E[a u], E[a^2], E|a u - E[a u]|^2 per basis."""
        for l in (range(self.n_regions) if regions is None else regions):
            cov = self._axis_cov(stats, l)
            zeta = posterior.scale_mean_zeta[l]
            yt = posterior.scale_mean_y_tilde[l]
            inv_prec = 1.0 / posterior.scale_precision[l]
            cov_y = np.einsum('iab,bi->ai', cov, yt)
            self.scale_axis_mean[l] = zeta[None, :] * cov_y
            quad = np.einsum('ai,ai->i', yt, cov_y)
            quad_sq = np.einsum('ai,ai->i', cov_y, cov_y)             # y^T C C y, C symmetric
            self.scale_moment2[l] = inv_prec + zeta ** 2 * quad
            self.scale_axis_central_moment2[l] = inv_prec + zeta ** 2 * (quad - quad_sq)

    def update_noise(self, posterior, regions=None):
        """Stats.py:102-112."""
        if not self.noise_region_specific:
            self.noise_mean = posterior.noise_gamma_shape / posterior.noise_gamma_scale
            self.noise_log_mean = psi(posterior.noise_gamma_shape) - np.log(posterior.noise_gamma_scale)
            return
        for l in (range(self.n_regions) if regions is None else regions):
            self.noise_mean[l] = posterior.noise_gamma_shape[l] / posterior.noise_gamma_scale[l]
            self.noise_log_mean[l] = psi(posterior.noise_gamma_shape[l]) - np.log(posterior.noise_gamma_scale[l])

    def update_bias(self, posterior, regions=None):
        """Stats.py:114-124."""
        if not self.bias_region_specific:
            self.bias_mean = posterior.bias_normal_mean
            self.bias_var = 1.0 / posterior.bias_normal_precision
            return
        for l in (range(self.n_regions) if regions is None else regions):
            self.bias_mean[l] = posterior.bias_normal_mean[l]
            self.bias_var[l] = 1.0 / posterior.bias_normal_precision[l]


class IndependentStats(Stats):
    """Stats.py:164-348."""

    def __init__(self, posterior):
        Stats.__init__(self, posterior)
        qd = posterior
        self.axis_cov = [np.zeros((self.n_basis, self.dy, self.dy)) for _ in range(self.n_regions)]
        self.ard_mean = [None] * self.n_regions
        self.ard_log_mean = [None] * self.n_regions
        self.update_ard(qd)
        self.omega = [np.ones((self.n_basis, self.n_basis)) / self.n_basis for _ in range(self.n_regions)]

    def _axis_cov(self, stats, region):
        return stats.axis_cov[region]

    def update_axis(self, posterior, regions=None):
        for l in (range(self.n_regions) if regions is None else regions):
            self.axis_cov[l] = posterior.axis[l].cov()

    def update_ard(self, posterior, regions=None):
        for l in (range(self.n_regions) if regions is None else regions):
            self.ard_mean[l] = posterior.ard_gamma_shape[l] / posterior.ard_gamma_scale[l]
            self.ard_log_mean[l] = psi(posterior.ard_gamma_shape[l]) - np.log(posterior.ard_gamma_scale[l])


class SharedStats(object):
    """Stats.py:369-462."""

    def __init__(self, posterior):
        self.n_basis, self.dy = posterior.n_basis, posterior.dy
        self.axis_cov = np.zeros((self.n_basis, self.dy, self.dy))
        self.update_ard(posterior)
        self.omega = np.ones((self.n_basis, self.n_basis)) / self.n_basis

    def update_axis(self, posterior):
        self.axis_cov = posterior.cov()

    def update_ard(self, posterior):
        self.ard_mean = posterior.ard_gamma_shape / posterior.ard_gamma_scale
        self.ard_log_mean = psi(posterior.ard_gamma_shape) - np.log(posterior.ard_gamma_scale)

    def update_omega(self, prior, stats=None):
        """Soft permutation aligning basis i of this layer with factor k of the previous one
        (Stats.py:405-462): omega_ik ~ exp(E log p_k(axis_i, alpha_i)) scaled to unit row and
        column sums.  The scalings come from MINPACK's hybrid solver started at zero on the
        interleaved row/column residuals -- the reference's solver, start and ordering, so
        the loosely converged answer coincides with its."""
        m = self.n_basis
        log_w = np.einsum('iab,kba->ik', self.axis_cov, prior.axis_bingham_b) - prior.axis_bingham_log_const[None, :] \
            + (prior.ard_gamma_shape * np.log(prior.ard_gamma_scale) - gammaln(prior.ard_gamma_shape))[None, :] \
            + np.outer(self.ard_log_mean, prior.ard_gamma_shape - 1.0) - np.outer(self.ard_mean, prior.ard_gamma_scale)

        def lse(mat, axis):
            top = np.max(mat, axis=axis, keepdims=True)
            return np.log(np.sum(np.exp(mat - top), axis=axis)) + np.squeeze(top, axis=axis)

        def residuals(ln_eta):
            ln_a, ln_b = ln_eta[:m], ln_eta[m:]
            out = np.empty(2 * m)
            out[0::2] = ln_a + lse(log_w + ln_b[None, :], 1)
            out[1::2] = ln_b + lse(log_w + ln_a[:, None], 0)
            return out

        if OMEGA_SOLVER == 'sinkhorn':
            # log-domain Sinkhorn on the same problem, run to convergence.  Exact
            # where MINPACK stops at xtol = 1.5e-8, so the two agree to ~1e-8, not to rounding.
            ln_a, ln_b = np.zeros(m), np.zeros(m)
            for _ in range(10000):
                ln_a = -lse(log_w + ln_b[None, :], 1)
                ln_b_new = -lse(log_w + ln_a[:, None], 0)
                done = np.max(np.abs(ln_b_new - ln_b)) < 1e-14
                ln_b = ln_b_new
                if done:
                    break
            self.omega = np.exp(ln_a[:, None] + ln_b[None, :] + log_w)
            return
        ln_eta = fsolve(residuals, np.zeros(2 * m))
        self.omega = np.exp(ln_eta[:m, None] + ln_eta[None, m:] + log_w)


# =================================================================================================
#  the model
# =================================================================================================
class _LayerStatsView(object):
    """Adds the device-resident latent function of a layer to its host statistics, in the
    reference's per-region list form (Stats.py:56-62)."""

    def __init__(self, model, layer):
        self._m, self._j = weakref.proxy(model), layer     # no model <-> view cycle

    def latent(self):
        f, v = self._m._latent[self._j]
        bounds = self._m.index_set_obj.bounds[self._j]
        return ([f[int(a):int(b)].double().cpu().numpy() for a, b in bounds],
                [v[int(a):int(b)].double().cpu().numpy() for a, b in bounds])


class ReducedRankMRGP(MultiResolutionGaussianProcess):
    """``MultiResolutionGaussianProcess`` when a basis-function object is given: the reference's
    constructor keywords (MRGP.py:15-34) and public methods, GPU-backed."""

    def __init__(self, train_xy, n_basis=None, index_set_obj=None, basis_function_obj=None, spectral_density_obj=None,
                 basis_interval_obj=None, interval_factor=1, adaptive_inputs=False, standard_normalized_inputs=True,
                 axis_resolution_specific=False, ard_resolution_specific=False, noise_region_specific=True,
                 bias_region_specific=True, noninformative_initialization=True, snr_ratio=None, full_x=None,
                 input_model=None, forced_independence=False, verbose=False, dtype='f64', device=None,
                 process_group=None, keep_factors=True):
        self.verbose = verbose
        self.forced_independence = bool(forced_independence)
        if self.forced_independence:
            self.axis_resolution_specific = self.ard_resolution_specific = True
        else:
            if axis_resolution_specific or ard_resolution_specific:
                raise TypeError("not yet supported")                     # MRGP.py:50-51
            self.axis_resolution_specific = self.ard_resolution_specific = False
        for flag in (noise_region_specific, bias_region_specific):
            if flag is not True and flag is not False:
                raise TypeError('region_specific can be either True or False.')       # Priors.py:102,134
        if noninformative_initialization is not True:
            raise ValueError('not yet implemented...')                   # Priors.py:26
        if index_set_obj is None or n_basis is None:
            raise ValueError('index_set_obj and n_basis are required')
        if int(n_basis) > 64:
            raise ValueError('n_basis must be at most 64 on this path')
        if self.forced_independence:
            basis_interval_obj = None                                    # MRGP.py:110-111
        self.adaptive_basis_intervals = basis_interval_obj is not None               # MRGP.py:113-129
        self.adaptive_inputs = adaptive_inputs
        self.standard_normalized_inputs = standard_normalized_inputs
        self.noise_region_specific = noise_region_specific
        self.bias_region_specific = bias_region_specific

        self.n_layers = index_set_obj.get_n_resolutions() + 1
        self.index_set_obj = index_set_obj
        self.n_basis = int(n_basis)
        x_train = np.asarray(train_xy[0], dtype=np.float64)
        y_train = np.asarray(train_xy[1], dtype=np.float64)
        self.observations = y_train
        self.dy = y_train.shape[1]
        if self.dy < 2:
            raise ValueError('output dimension must be greater than 1')
        if self.dy > 8:
            raise ValueError('output dimension must be at most 8 on this path')
        if x_train.shape[0] != index_set_obj.sample_length:
            raise ValueError('index set was built for a different number of samples')
        x_train, self.full_x, self.mean_x_train, self.std_x_train = self._normalize_inputs(x_train, full_x)

        self.spectral_density_obj = self._per_layer(spectral_density_obj, 'spectral_density_obj')
        self.basis_function_obj = self._per_layer(basis_function_obj, 'basis_function_obj')
        for b in self.basis_function_obj:
            if not isinstance(b, LaplacianEigenpairs):
                raise TypeError('basis_function_obj must be LaplacianEigenpairs')
        self.interval_factor = self._per_layer(interval_factor, 'interval_factor')
        self.basis_interval_obj = [BasisInterval() for _ in range(self.n_layers)] if basis_interval_obj is None \
            else self._per_layer(basis_interval_obj, 'basis_interval_obj')
        self.use_prior = [s is not None for s in self.spectral_density_obj]

        self.device = dev.require_gpu(device)
        self.dtype = dev.as_torch_dtype(dtype)
        self.group = process_group
        self.rank, self.world_size = dist.world(process_group)

        self.input_obj = Inputs(x=x_train, index_set=index_set_obj, learn_inputs=self.adaptive_inputs,
                                full_x=self.full_x, input_model=input_model)
        x_host = np.asarray(self.input_obj.x, dtype=np.float64)
        self.dx = x_host.shape[1]
        if self.dx > 8:
            raise ValueError('input dimension must be at most 8 on this path')
        self._x_dev = dev.to_device(x_host, self.dtype, self.device)
        self._y = dev.to_device(y_train, self.dtype, self.device)
        bounds = index_set_obj.bounds
        self.n_regions = [len(layer) for layer in bounds]
        self.n_samps = [[int(b - a) for a, b in layer] for layer in bounds]
        self.owner = [dist.assign_blocks(self.n_samps[j], self.world_size) for j in range(self.n_layers)]

        # basis intervals, eigenvalues, prior spectral weights (MRGP.py:136-176,297-335) and Phi on the GPU
        self.train_basis_intervals, self.lambda_, self.spectral_density_prior = [], [], []
        self._x_host = x_host
        for j in range(self.n_layers):
            iv_j, lam_j, spec_j = [], [], []
            for l, (a, b) in enumerate(bounds[j]):
                a, b = int(a), int(b)
                interval = self.basis_interval_obj[j].max_input_range_by_factor_of(x_host[a:b], self.interval_factor[j])
                lam, spec = self._eigenvalues_and_prior(j, interval)
                iv_j.append(interval)
                lam_j.append(lam)
                spec_j.append(spec)
            self.train_basis_intervals.append(iv_j)
            self.lambda_.append(lam_j)
            self.spectral_density_prior.append(spec_j)

        sf = [1.0 if s is None else s.sf for s in self.spectral_density_obj]
        influence = float(np.mean(sf))
        noise_var0 = None if snr_ratio is None else self._compute_initial_noise_var_from_snr(y_train, snr_ratio)
        self.prior_obj, self.posterior_obj, self.stats_obj = [], [], []
        for j in range(self.n_layers):
            nv = noise_var0 if j == 0 else None
            if self.forced_independence:
                pr = IndependentPrior(self.n_basis, self.dy, self.n_regions[j], self.spectral_density_prior[j], influence, nv,
                                      noise_region_specific, bias_region_specific)
                po = IndependentPosterior(pr)
                st = IndependentStats(po)
            else:
                pr = Prior(self.n_basis, self.dy, self.n_regions[j], self.spectral_density_prior[j], nv,
                           noise_region_specific, bias_region_specific)
                po = Posterior(pr)
                st = Stats(po)
            self.prior_obj.append(pr)
            self.posterior_obj.append(po)
            self.stats_obj.append(st)
        if not self.forced_independence:
            self.shared_prior = SharedPrior(self.n_basis, self.dy, influence)
            self.shared_posterior = SharedPosterior(self.shared_prior)
            self.shared_stats = SharedStats(self.shared_posterior)

        n0 = y_train.shape[0]
        zero_f = torch.zeros((n0, self.dy), dtype=self.dtype, device=self.device)
        zero_v = torch.zeros(n0, dtype=self.dtype, device=self.device)
        self._latent = [(zero_f, zero_v) for _ in range(self.n_layers)]
        self.y_var = [[0.0] * self.n_regions[j] for j in range(self.n_layers)]
        self._targets = [dict() for _ in range(self.n_layers)]
        self._fitted = False
        self.lower_bound = []
        self.lower_bound_layer = [[] for _ in range(self.n_layers)]

    def _eigenvalues_and_prior(self, j, interval):
        """lambda_i = sum_k (pi i / 2 L_k)^2 and the prior weight S(sqrt(lambda_i)) (MRGP.py:297-357)."""
        orders = np.arange(1, self.n_basis + 1, dtype=np.float64)
        lam = np.sum((np.pi * orders[:, None] / (2.0 * np.asarray(interval, dtype=np.float64)[None, :])) ** 2, axis=1)
        if self.spectral_density_obj[j] is None:
            return lam, np.ones(self.n_basis)
        return lam, np.asarray(self.spectral_density_obj[j].spectral(np.sqrt(lam)), dtype=np.float64)

    def _per_layer(self, obj, name):
        if isinstance(obj, list):
            if len(obj) != self.n_layers:
                raise ValueError(name + ' must be a list of the same length as the number of resolutions + 1')
            return obj
        return [obj] * self.n_layers

    # ------------------------------------------------------------------------------- fitting
    def fit(self, n_iter=1, tol=1e-3, min_iter=10):
        """MRGP.py:367-412.  With ``tol`` ciMRGP records its lower bound after every sweep and
        stops once, past ``min_iter`` sweeps, layer 0's bound moves by less than ``tol``;
        fiMRGP runs ``n_iter`` sweeps either way (as in the reference)."""
        n_iter = int(n_iter)
        min_iter = min(min_iter, n_iter)
        for it in range(1, n_iter + 1):
            if self.forced_independence:
                self._independent_fit()
                continue
            self._fit()
            if tol is None:
                continue
            total, per_layer = self._compute_lower_bound()
            self.lower_bound.append(total)
            for j in range(self.n_layers):
                self.lower_bound_layer[j].append(per_layer[j])
            if it > min_iter:
                delta0 = self.lower_bound_layer[0][-1] - self.lower_bound_layer[0][-2]
                if self.verbose:
                    print("\nTotal ELBO: %.4f ... dELBO %.4f" % (self.lower_bound[-1], self.lower_bound[-1] - self.lower_bound[-2]))
                if abs(delta0) < abs(tol):
                    break
        self._fitted = True

    def _compute_lower_bound(self):
        """MRGP.py:414-571, term by term as the reference codes it: the data term is the bare sum
        of squared errors and variances plus the log-normaliser; the axis term multiplies its two
        matrices elementwise before the trace; for j > 0 the "previous" shared posterior is the
        current one (``_fit_tol`` aliases it, MRGP.py:379).  One more pass of the moments kernel
        per block supplies the squared residual under the final statistics."""
        sp, ss = self.shared_posterior, self.shared_stats
        per_layer = []
        diag_cov = np.einsum('iaa->ia', ss.axis_cov)

        def gamma_term(shape, scale, log_mean, mean):
            return shape * np.log(scale) - gammaln(shape) + (shape - 1) * log_mean - scale * mean

        for j in range(self.n_layers):
            po, pr, st = self.posterior_obj[j], self.prior_obj[j], self.stats_obj[j]
            prev = self.shared_prior if j == 0 else sp
            owned = self._owned(j)
            mom = self._moments(j, owned, self._targets[j])
            local = 0.0
            for l in owned:
                mm, bias = mom[l], st.bias_of(l)
                sq = mm.resid_sq - 2.0 * float(np.dot(bias, mm.resid_sum)) + mm.n * float(np.dot(bias, bias))
                local += sq + mm.fvar_sum + float(np.dot(mm.colsum2, st.scale_axis_central_moment2[l]))
                local += np.sum(0.5 * ss.ard_log_mean / self.spectral_density_prior[j][l]
                                - 0.5 * ss.ard_mean * st.scale_moment2[l] / self.spectral_density_prior[j][l])
                local -= np.sum(0.5 * np.log(po.scale_precision[l]) - 0.5)
            total = float(self._sum_over_ranks(np.array([local]))[0])
            for l in range(self.n_regions[j]):
                n_l = self.n_samps[j][l]
                tau, log_tau = st.noise_of(l), st.noise_log_of(l)
                total += st.bias_var_of(l) + self.y_var[j][l] * n_l + 0.5 * self.dy * (log_tau - np.log(2 * np.pi)) * n_l
                w = st.bias_of(l)
                p0 = _of(pr.bias_normal_precision, l, self.bias_region_specific)
                w0 = _of(pr.bias_normal_mean, l, self.bias_region_specific)
                p1 = _of(po.bias_normal_precision, l, self.bias_region_specific)
                spread = 1.0 / (p1 * tau) + np.dot(w, w) - 2 * np.dot(w, w0) + np.dot(w0, w0)
                total += 0.5 * self.dy * (np.log(p0) + log_tau - np.log(2 * np.pi)) + 0.5 * p0 * tau * spread
                total -= 0.5 * self.dy * (np.log(p1) + log_tau - np.log(2 * np.pi)) - 0.5
                total += gamma_term(_of(pr.noise_gamma_shape, l, self.noise_region_specific),
                                    _of(pr.noise_gamma_scale, l, self.noise_region_specific), log_tau, tau)
                total -= gamma_term(_of(po.noise_gamma_shape, l, self.noise_region_specific),
                                    _of(po.noise_gamma_scale, l, self.noise_region_specific), log_tau, tau)
            prev_diag = np.einsum('kaa->ka', prev.axis_bingham_b)
            total += np.sum(ss.omega * (-prev.axis_bingham_log_const[None, :] + diag_cov @ prev_diag.T))
            total -= np.sum(-sp.axis_bingham_log_const + np.sum(diag_cov * np.einsum('iaa->ia', sp.axis_bingham_b), axis=1))
            total += np.sum(ss.omega * gamma_term(prev.ard_gamma_shape[None, :], prev.ard_gamma_scale[None, :],
                                                  ss.ard_log_mean[:, None], ss.ard_mean[:, None]))
            total -= np.sum(gamma_term(sp.ard_gamma_shape, sp.ard_gamma_scale, ss.ard_log_mean, ss.ard_mean))
            per_layer.append(float(total))
        return float(np.sum(per_layer)), per_layer

    @property
    def phi_x(self):
        """Phi per block (device tensors), materialised on demand -- the sweep itself never
        stores it (the reference keeps ``phi_x[j][l]`` resident, MRGP.py:160-176)."""
        return [[dev.laplace_basis(self._x_dev[int(a):int(b)], self.train_basis_intervals[j][l], self.n_basis)
                 for l, (a, b) in enumerate(self.index_set_obj.bounds[j])] for j in range(self.n_layers)]

    def _block_views(self, j, l):
        a, b = (int(v) for v in self.index_set_obj.bounds[j][l])
        f, v = self._latent[j]
        return a, b, f[a:b], v[a:b]

    def _moments(self, j, owned, targets):
        out = {}
        for l in owned:
            a, b, fbar, fvar = self._block_views(j, l)
            out[l] = dev.basis_moments(self._x_dev[a:b], self.train_basis_intervals[j][l], self.n_basis, targets[l], fbar,
                                       fvar, self.stats_obj[j].scale_axis_mean[l])
        return out

    def _update_latent_functions(self, j, owned):
        """Stats.py:316-348: the next layer sees the sum of all coarser layers' means and
        variances at the training points; built incrementally, one collective per layer."""
        f, v = self._latent[j]
        fused = torch.zeros((f.shape[0], self.dy + 1), dtype=self.dtype, device=self.device)
        delta_f = torch.zeros_like(f)
        delta_v = torch.zeros_like(v)
        st = self.stats_obj[j]
        for l in owned:
            a, b = (int(t) for t in self.index_set_obj.bounds[j][l])
            dev.basis_apply(self._x_dev[a:b], self.train_basis_intervals[j][l], self.n_basis, st.scale_axis_mean[l],
                            st.bias_of(l), st.scale_axis_central_moment2[l], st.bias_var_of(l), mean=delta_f[a:b],
                            var=delta_v[a:b], accumulate=False)
        if self.world_size > 1:
            fused[:, :self.dy] = delta_f
            fused[:, self.dy] = delta_v
            dist.allreduce_sum_(fused, self.group)
            delta_f, delta_v = fused[:, :self.dy].contiguous(), fused[:, self.dy].contiguous()
        self._latent[j + 1] = (f + delta_f, v + delta_v)

    def _sync_regions(self, j):
        """Multi-GPU: every rank needs every region's host statistics (a few kB per region);
        shared noise / bias factors are already identical everywhere."""
        if self.world_size == 1:
            return
        po, st = self.posterior_obj[j], self.stats_obj[j]
        m, q = self.n_basis, self.dy
        width = q * m + 3 * m + q + 5
        buf = torch.zeros((self.n_regions[j], width), dtype=torch.float64, device=self.device)
        for l in self._owned(j):
            row = np.concatenate([st.scale_axis_mean[l].ravel(), st.scale_moment2[l], st.scale_axis_central_moment2[l],
                                  po.scale_precision[l], np.asarray(st.bias_of(l)).ravel(),
                                  [st.bias_var_of(l), st.noise_of(l), st.noise_log_of(l),
                                   _of(po.noise_gamma_shape, l, self.noise_region_specific),
                                   _of(po.noise_gamma_scale, l, self.noise_region_specific)]])
            buf[l] = torch.as_tensor(row).to(self.device)
        dist.allreduce_sum_(buf, self.group)
        host = buf.cpu().numpy()
        for l in range(self.n_regions[j]):
            if self.owner[j][l] == self.rank:
                continue
            row = host[l]
            st.scale_axis_mean[l] = row[:q * m].reshape(q, m)
            st.scale_moment2[l] = row[q * m:q * m + m]
            st.scale_axis_central_moment2[l] = row[q * m + m:q * m + 2 * m]
            po.scale_precision[l] = row[q * m + 2 * m:q * m + 3 * m]
            if self.bias_region_specific:
                st.bias_mean[l] = row[q * m + 3 * m:q * m + 3 * m + q]
                st.bias_var[l] = row[-5]
                po.bias_normal_mean[l] = st.bias_mean[l]
                po.bias_normal_precision[l] = 1.0 / row[-5]
            if self.noise_region_specific:
                st.noise_mean[l], st.noise_log_mean[l] = row[-4], row[-3]
                po.noise_gamma_shape[l], po.noise_gamma_scale[l] = row[-2], row[-1]

    def _independent_fit(self):
        """One sweep of fiMRGP (MRGP.py:663-731); targets are the observations of the region
        (LatentOutputs.py:11-18)."""
        for j in range(self.n_layers):
            owned = self._owned(j)
            pr, po, st = self.prior_obj[j], self.posterior_obj[j], self.stats_obj[j]
            targets = {l: self._y[int(self.index_set_obj.bounds[j][l][0]):int(self.index_set_obj.bounds[j][l][1])]
                       for l in owned}
            y_var = [0.0] * self.n_regions[j]

            def local_phase():
                mom = self._moments(j, owned, targets)
                po.update_scale_given_axis(mom, pr, st, spectral_density=self.spectral_density_prior[j], regions=owned)
                po.update_axis(pr, po, st, regions=owned)
                st.update_axis(po, regions=owned)
                st.update_scale(po, st, regions=owned)
                po.update_ard(pr, st, self.spectral_density_prior[j], regions=owned)
                st.update_ard(po, regions=owned)
                return self._moments(j, owned, targets)                # residuals under the new E[au]
            mom = self._together(local_phase)
            po.update_bias_given_noise(mom, pr, st, regions=owned, n_samps=self.n_samps[j], reduce=self._sum_over_ranks)
            po.update_noise(mom, y_var, pr, po, st, regions=owned, n_samps=self.n_samps[j], reduce=self._sum_over_ranks)
            st.update_bias(po, regions=owned)
            st.update_noise(po, regions=owned)
            self.y_var[j] = y_var
            self._sync_regions(j)
            if j + 1 < self.n_layers:
                self._update_latent_functions(j, owned)

    def _fit(self):
        """One sweep of ciMRGP (MRGP.py:574-661).  For j > 0 the targets are the layer's own
        current reconstruction  Phi E[au]^T + bias + f_bar  with variance 1 / E[tau]
        (LatentOutputs.py:20-49)."""
        for j in range(self.n_layers):
            owned = self._owned(j)
            pr, po, st = self.prior_obj[j], self.posterior_obj[j], self.stats_obj[j]
            targets = {}
            y_var = [0.0] * self.n_regions[j]
            for l in range(self.n_regions[j]):
                if j > 0:
                    y_var[l] = 1.0 / st.noise_of(l)
            for l in owned:
                a, b, fbar, _ = self._block_views(j, l)
                if j == 0:
                    targets[l] = self._y[a:b]
                else:
                    t = fbar.clone()
                    dev.basis_apply(self._x_dev[a:b], self.train_basis_intervals[j][l], self.n_basis, st.scale_axis_mean[l],
                                    st.bias_of(l), mean=t, accumulate=True)
                    targets[l] = t
            self._targets[j] = targets
            previous = self.shared_prior if j == 0 else self.shared_posterior.snapshot()

            def local_phase():
                mom = self._moments(j, owned, targets)
                po.update_scale_given_axis(mom, pr, st, shared_stats=self.shared_stats,
                                           spectral_density=self.spectral_density_prior[j], regions=owned)
                return po.axis_evidence(st, regions=owned)
            evidence = self._sum_over_ranks(self._together(local_phase))
            self.shared_posterior.update_axis(previous, evidence, self.shared_stats)
            self.shared_stats.update_axis(self.shared_posterior)
            st.update_scale(po, self.shared_stats, regions=owned)
            over_spec = np.zeros(self.n_basis)
            for l in owned:
                over_spec += st.scale_moment2[l] / self.spectral_density_prior[j][l]
            self.shared_posterior.update_ard(previous, self._sum_over_ranks(over_spec), self.n_regions[j], self.shared_stats)
            self.shared_stats.update_ard(self.shared_posterior)
            self.shared_stats.update_omega(previous, self.shared_stats)
            mom = self._together(lambda: self._moments(j, owned, targets))
            po.update_bias_given_noise(mom, pr, st, regions=owned, n_samps=self.n_samps[j], reduce=self._sum_over_ranks)
            po.update_noise(mom, y_var, pr, po, st, regions=owned, n_samps=self.n_samps[j], reduce=self._sum_over_ranks)
            st.update_bias(po, regions=owned)
            st.update_noise(po, regions=owned)
            self.y_var[j] = y_var
            self._sync_regions(j)
            if self.adaptive_basis_intervals:                            # MRGP.py:626-636
                self._learn_basis_intervals(j, owned, targets)
            if j + 1 < self.n_layers:
                self._update_latent_functions(j, owned)

    def _learn_basis_intervals(self, j, owned, targets):
        """New intervals for the layer's regions (BasisInterval.learn), then eigenvalues and
        prior spectral weights rebuilt from them; Phi itself is never stored."""
        d = self.dx
        fresh = np.zeros((self.n_regions[j], d))
        for l in owned:
            fresh[l] = self.basis_interval_obj[j].learn(self, j, l, targets[l])
        fresh = self._sum_over_ranks(fresh)
        for l in range(self.n_regions[j]):
            self.train_basis_intervals[j][l] = fresh[l].copy()
            self.lambda_[j][l], self.spectral_density_prior[j][l] = self._eigenvalues_and_prior(j, fresh[l])

    def _together(self, local_phase):
        """Run a rank-local phase (GPU sums + the regions' own updates).  With several ranks an
        exception on one of them is raised on EVERY rank (``dist.raise_together``) instead of
        leaving the others blocked in the next collective."""
        if self.world_size == 1:
            return local_phase()
        out, err = None, None
        try:
            out = local_phase()
        except Exception as exc:            # noqa: BLE001 -- re-raised on every rank just below
            err = exc
        dist.raise_together(err, self.group, self.device)
        return out

    def _sum_over_ranks(self, host_array):
        if self.world_size == 1:
            return host_array
        t = torch.as_tensor(np.ascontiguousarray(host_array, dtype=np.float64)).to(self.device)
        dist.allreduce_sum_(t, self.group)
        return t.cpu().numpy()

    # ---------------------------------------------------------------------------- prediction
    def _predict(self, test_x, index_set, want_var, include_noise=True):
        if not self._fitted:
            raise RuntimeError('call fit() before predicting')
        xs = self._prepare_test(test_x)
        ns = xs.shape[0]
        mean = torch.zeros((ns, self.dy), dtype=self.dtype, device=self.device)
        total = torch.zeros(ns, dtype=self.dtype, device=self.device) if want_var else None
        if index_set is None:
            # every prediction is taken from resolution 0 (MRGP.py:733-762, 832-860)
            if self.owner[0][0] == self.rank:
                st = self.stats_obj[0]
                dev.basis_apply(xs, self.train_basis_intervals[0][0], self.n_basis, st.scale_axis_mean[0], st.bias_of(0),
                                st.scale_axis_central_moment2[0], st.bias_var_of(0), mean=mean, var=total, accumulate=False)
        else:
            n_layers = index_set.get_n_resolutions() + 1
            coarser = torch.zeros(ns, dtype=self.dtype, device=self.device) if want_var else None
            for j in range(n_layers):
                st = self.stats_obj[j]
                own = torch.zeros(ns, dtype=self.dtype, device=self.device) if want_var else None
                for l in self._owned(j):
                    a, b = (int(v) for v in index_set.bounds[j][l])
                    dev.basis_apply(xs[a:b], self.train_basis_intervals[j][l], self.n_basis, st.scale_axis_mean[l],
                                    st.bias_of(l), st.scale_axis_central_moment2[l], st.bias_var_of(l), mean=mean[a:b],
                                    var=own[a:b] if want_var else None, accumulate=True)
                    if want_var:
                        # MRGP.py:905-937: own term + n_l / E[tau] + the coarser layers' variance at the
                        # region's FIRST test point (``latent_f_var[l][0]``)
                        total[a:b] += own[a:b] + (b - a) / st.noise_of(l)
                        if j > 0:
                            total[a:b] += coarser[a]
                if want_var:
                    if self.world_size > 1:
                        dist.allreduce_sum_(own, self.group)
                    coarser = coarser + own
        if self.world_size > 1:
            dist.allreduce_sum_(mean, self.group)
            if want_var:
                dist.allreduce_sum_(total, self.group)
        return mean.double().cpu().numpy(), (total.double().cpu().numpy() if want_var else None)

    def get_basis_contributions(self):
        """MRGP.py:973-982."""
        return [[self.stats_obj[j].scale_moment2[l] / np.sum(self.stats_obj[j].scale_moment2[l])
                 for l in range(self.n_regions[j])] for j in range(self.n_layers)]

    def latent_functions(self, layer):
        """(list of means, list of variances) per region of ``layer`` at the training points
        (the reference's ``stats_obj[j].latent_f_mean / latent_f_var``)."""
        return _LayerStatsView(self, layer).latent()
