"""Oracle (test infrastructure): the reference's own reduced-rank block model.

This is the variational Hilbert-space GP the reference actually runs (SURVEY.md
section 0 and 8f rank 2): Laplacian eigenfunctions, a Matern spectral density,
Bingham-distributed output axes, Gamma ARD / noise and Normal bias factors,
swept layer by layer.  Restated in plain numpy from the published update
equations as the reference implements them; every function cites the reference
lines it follows.  PINNED by ``tests/golden/reference_model_*.npz`` -- outputs
of the reference itself run in this container (``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU leg may
import this module; the product path (``cimrgp_amd``) never does.
"""
import numpy as np
from scipy.optimize import brentq, fsolve
from scipy.special import gammaln, logsumexp, psi

TINY = 1e-45            # Priors.py:5 -- shape of every "non-informative" Gamma / precision


# ----------------------------------------------------------------------------------------------
# basis functions and prior spectral weights
# ----------------------------------------------------------------------------------------------
def laplace_basis(x, interval, n_basis):
    """Phi (n x m) and eigenvalues (m) on [-L, L]^d (KernelClass.py:22-37 assembled as
    MRGP.py:337-357: product over input dimensions, sum of per-dimension eigenvalues)."""
    x = np.asarray(x, dtype=np.float64)
    half = np.asarray(interval, dtype=np.float64)[None, :]
    phi = np.empty((x.shape[0], n_basis))
    lam = np.empty(n_basis)
    for i in range(n_basis):
        order = i + 1
        per_dim = np.sin(np.pi * order * (x + half) / (2 * half)) / np.sqrt(half)
        phi[:, i] = np.prod(per_dim, axis=1)
        lam[i] = np.sum((np.pi * order / (2 * half)) ** 2)
    return phi, lam


def matern_spectral(s, nu, ell, sf):
    """Matern spectral density, log-domain form of KernelClass.py:73-86."""
    log_arg = np.log(2 * nu) - 2 * np.log(ell)
    return np.exp(np.log(sf) + 0.5 * np.log(2 * np.pi) + nu * log_arg + gammaln(nu + 0.5) - gammaln(nu)
                  - (nu + 0.5) * np.log(np.exp(log_arg) + np.asarray(s) ** 2))


def basis_interval(x, factor=1.0):
    """BasisInterval.py:15-16."""
    return factor * np.max(np.abs(x), axis=0)


# ----------------------------------------------------------------------------------------------
# Bingham normaliser (first-order saddle point) and the PD repair
# ----------------------------------------------------------------------------------------------
def bingham_saddle(kappa):
    """log C(kappa) and d log C / d kappa by the first-order saddle-point approximation of
    Kume & Wood (2005), as computeRealBinghamConstant.py:12-154 does it: eigenvalues negated
    and shifted so the smallest is 0.1, K1(t) = 1 solved by Brent's method on
    [0.1 - p, 0.1 - 0.5], log C from their eq. (15), gradient by implicit differentiation."""
    kappa = np.asarray(kappa, dtype=np.float64)
    p = kappa.shape[-1]
    lam = -kappa
    shift = 0.1 - np.min(lam)
    lam = lam + shift
    t_hat = brentq(lambda t: 0.5 * np.sum(1.0 / (lam - t)) - 1.0, 0.1 - p, 0.1 - 0.5)
    u = 1.0 / (lam - t_hat)
    k2 = 0.5 * np.sum(u ** 2)
    k3 = np.sum(u ** 3)
    log_c = 0.5 * (np.log(2) + (p - 1) * np.log(np.pi) - np.log(k2) - np.sum(np.log(lam - t_hat))) - t_hat + shift
    dt = (0.5 * u ** 2) / k2                         # d t_hat / d lam_k at constant K1
    dlogk2 = (-u ** 3 + k3 * dt) / k2
    dsumlog = u - np.sum(u) * dt
    grad = 0.5 * dlogk2 + 0.5 * dsumlog + dt          # sign already flipped back to kappa
    return log_c, grad


def is_pd(mat):
    """SanityCheck.py:59-65."""
    try:
        np.linalg.cholesky(mat)
        return True
    except np.linalg.LinAlgError:
        return False


def nearest_pd(mat):
    """Higham's nearest symmetric PSD matrix plus the diagonal nudging loop
    (SanityCheck.py:17-57)."""
    sym = (mat + mat.T) / 2
    _, s, vt = np.linalg.svd(sym)
    polar = vt.T @ np.diag(s) @ vt
    out = (sym + polar) / 2
    out = (out + out.T) / 2
    if is_pd(out):
        return out
    spacing = np.spacing(np.linalg.norm(mat))
    eye = np.eye(mat.shape[0])
    k = 1
    while not is_pd(out):
        mineig = np.min(np.real(np.linalg.eigvals(out)))
        out += eye * (-mineig * k ** 2 + spacing)
        k += 1
    return out


def bingham_from_matrix(b):
    """CommonDensities.py:73-79 + the clean-up of Posteriors.py:283-290: eigen-decompose,
    sort descending, normaliser and rho from the *unclamped* eigenvalues, then clamp kappa."""
    vals, vecs = np.linalg.eig(b)
    order = vals.argsort()[::-1]
    kappa = vals[order]
    axes = vecs[:, order]
    log_c, rho = bingham_saddle(np.real(kappa))
    kappa = np.real(kappa).copy()
    kappa[kappa < 0] = 0.0
    return dict(kappa=kappa, axes=np.real(axes), rho=np.real(rho), log_const=float(np.real(log_c)))


def axis_update(b_candidate):
    """Repair (Posteriors.py:278-282) then fit the Bingham; returns (b_used, bingham)."""
    b_used = b_candidate if is_pd(b_candidate) else nearest_pd(b_candidate)
    return b_used, bingham_from_matrix(b_used)


def axis_cov(bing):
    """E[u u^T] = sum_d rho_d v_d v_d^T  (Stats.py:249-257, 385-392)."""
    return (bing['axes'] * bing['rho'][None, :]) @ bing['axes'].T


def soft_permutation(prev, cov, ard_log_mean, ard_mean):
    """Stats.py:405-462: omega[i, k] proportional to exp(E log p(axis_i, alpha_i | factor k of the
    previous layer)), scaled to unit row and column sums.  The reference finds the scalings
    with MINPACK's hybrid solver from a zero start; so does this, on the same residuals in
    the same order, so the (loosely converged) answer is the same."""
    m = cov.shape[0]
    log_w = np.empty((m, m))
    for i in range(m):
        for k in range(m):
            log_w[i, k] = np.trace(cov[i] @ prev['b'][k]) - prev['log_const'][k] \
                + prev['ard_shape'][k] * np.log(prev['ard_scale'][k]) - gammaln(prev['ard_shape'][k]) \
                + (prev['ard_shape'][k] - 1) * ard_log_mean[i] - prev['ard_scale'][k] * ard_mean[i]

    def residuals(ln_eta):
        ln_a, ln_b = ln_eta[:m], ln_eta[m:]
        rows = ln_a + logsumexp(ln_b[None, :] + log_w, axis=1)
        cols = ln_b + logsumexp(ln_a[:, None] + log_w, axis=0)
        return np.stack([rows, cols], axis=1).ravel()

    ln_eta = fsolve(residuals, np.zeros(2 * m))
    return np.exp(ln_eta[:m, None] + ln_eta[None, m:] + log_w)


# ----------------------------------------------------------------------------------------------
# per-block pieces shared by both model flavours
# ----------------------------------------------------------------------------------------------
def y_tilde(y, phi, eau, bias, fbar):
    """Posteriors.py:326-342: for every basis i the projection of the residual that leaves
    out i's own contribution.  Restated as Phi^T r + colsum(Phi^2) * E[au]."""
    resid = y - fbar - bias - phi @ eau.T
    return (phi.T @ resid).T + eau * np.sum(phi * phi, axis=0)[None, :]


def scale_stats(zeta, precision, ytil, cov):
    """Stats.py:264-297 for one block: E[a u], E[a^2], and the central second moment."""
    m = zeta.shape[0]
    eau = np.empty_like(ytil)
    mom2 = np.empty(m)
    cen2 = np.empty(m)
    for i in range(m):
        yt = ytil[:, i]
        eau[:, i] = zeta[i] * (cov[i] @ yt)
        mom2[i] = 1.0 / precision[i] + zeta[i] ** 2 * (yt @ cov[i] @ yt)
        cen2[i] = 1.0 / precision[i] + zeta[i] ** 2 * (yt @ (cov[i] - cov[i] @ cov[i]) @ yt)
    return eau, mom2, cen2


class _Block:
    """State of one (layer, region) block: everything the reference spreads over its
    Prior / Posterior / Stats objects for that region."""

    def __init__(self, x, y_rows, n_basis, dy, spectral, prior_influence, factor, noise_var=1.0):
        self.x = x
        self.rows = y_rows
        self.interval = basis_interval(x, factor)
        self.phi, self.lam = laplace_basis(x, self.interval, n_basis)
        self.spec = spectral(np.sqrt(self.lam))                    # MRGP.py:297-303
        n = x.shape[0]
        self.n = n
        # priors (Priors.py:180-279) and their initial statistics (Stats.py:180-238)
        self.ard_shape0 = TINY * np.ones(n_basis)
        self.ard_scale0 = self.ard_shape0 / prior_influence
        self.noise_shape0 = TINY
        self.noise_scale0 = (TINY + 1.0) * noise_var                   # Priors.py:262-267
        self.bias_prec0 = TINY
        self.ard_mean = self.ard_shape0 / self.ard_scale0
        self.ard_log_mean = psi(self.ard_shape0) - np.log(self.ard_scale0)
        self.noise_shape, self.noise_scale = self.noise_shape0, self.noise_scale0
        self.noise_mean = self.noise_shape0 / self.noise_scale0
        self.noise_log_mean = psi(self.noise_shape0) - np.log(self.noise_scale0)
        self.y_var = 0.0
        self.bias_mean = np.zeros(dy)
        self.bias_prec = self.bias_prec0
        self.bias_var = 1.0 / self.bias_prec0
        self.eau = np.zeros((dy, n_basis))
        self.mom2 = np.zeros(n_basis)
        self.cen2 = np.zeros(n_basis)
        self.fbar = np.zeros((n, dy))
        self.fvar = np.zeros(n)
        self.precision = 1.0 / self.spec
        self.zeta = np.zeros(n_basis)
        self.ytil = np.zeros((dy, n_basis))

    def scale_given_axis(self, y, ard_mean):
        """Posteriors.py:298-324."""
        self.precision = ard_mean / self.spec + self.noise_mean * np.sum(self.phi * self.phi, axis=0)
        self.zeta = self.noise_mean / self.precision
        self.ytil = y_tilde(y, self.phi, self.eau, self.bias_mean, self.fbar)

    def residual_sums(self, y):
        """The block sums the bias and noise updates share (Posteriors.py:345-372, 396-452):
        sum and squared norm of  y - Phi E[au]^T - f_bar  (no bias), sum f_var, sum Phi^2 c2."""
        resid = y - self.phi @ self.eau.T - self.fbar
        return np.sum(resid, axis=0), np.sum(resid * resid), np.sum(self.fvar), np.sum(self.phi ** 2 * self.cen2)

    def rebuild_basis(self, interval, spectral):
        """MRGP.py:305-335 after an interval update."""
        self.interval = np.asarray(interval, dtype=np.float64)
        self.phi, self.lam = laplace_basis(self.x, self.interval, self.phi.shape[1])
        self.spec = spectral(np.sqrt(self.lam))

    def contribution(self, phi=None):
        """bias + Phi E[au]^T and its variance (Stats.py:316-348)."""
        phi = self.phi if phi is None else phi
        return self.bias_mean + phi @ self.eau.T, self.bias_var + (phi ** 2) @ self.cen2


class ReducedRankModel:
    """fiMRGP (``forced_independence=True``) and ciMRGP (False) of the reference with its
    flag variants: region-specific or shared noise and bias, fixed or adaptive basis intervals,
    SNR-initialised noise, the lower bound of ``fit(n_iter, tol)``; non-informative priors,
    no input warping."""

    def __init__(self, x, y, bounds, n_basis, nu=1.0, ell=1.0, sf=1.0, forced_independence=True,
                 interval_factor=1.0, snr_ratio=None, noise_region_specific=True, bias_region_specific=True,
                 adaptive_basis_intervals=False, opt_interval_factor=(1.0, 1.2)):
        x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        self.mean_x = np.mean(x, 0)
        self.std_x = np.std(x, 0)
        self.std_x[self.std_x == 0] = 1
        self.xn = (x - self.mean_x) / self.std_x                      # MRGP.py:278-295
        self.bounds = bounds
        self.m = n_basis
        self.dy = self.y.shape[1]
        self.fi = forced_independence
        self.n_layers = len(bounds)
        self.noise_regional = noise_region_specific
        self.bias_regional = bias_region_specific
        self.adaptive = adaptive_basis_intervals and not forced_independence      # MRGP.py:110-111
        self.opt_factor = opt_interval_factor
        self.lower_bound, self.lower_bound_layer = [], [[] for _ in bounds]
        spectral = lambda s: matern_spectral(s, nu, ell, sf)
        self.spectral = spectral
        noise_var0 = 1.0
        if snr_ratio is not None:                                      # MRGP.py:966-971, layer 0 only (:196-203)
            n0 = self.y.shape[0]
            noise_var0 = (np.linalg.norm(self.y) ** 2 / n0 - np.dot(self.y.mean(0), self.y.mean(0))) / snr_ratio
        self.blocks = [[_Block(self.xn[a:b], slice(int(a), int(b)), n_basis, self.dy, spectral, sf, interval_factor,
                               noise_var0 if j == 0 else 1.0)
                        for a, b in layer] for j, layer in enumerate(bounds)]
        zero_bing = bingham_from_matrix(np.zeros((self.dy, self.dy)))
        if self.fi:
            for layer in self.blocks:
                for blk in layer:
                    blk.b = np.zeros((n_basis, self.dy, self.dy))
                    blk.cov = np.zeros((n_basis, self.dy, self.dy))
        else:
            self.sh = dict(b=np.zeros((n_basis, self.dy, self.dy)),
                           log_const=np.full(n_basis, zero_bing['log_const']),
                           ard_shape=TINY * np.ones(n_basis), ard_scale=TINY * np.ones(n_basis) / sf)
            self.sh_prior = {k: v.copy() for k, v in self.sh.items()}
            self.sh_cov = np.zeros((n_basis, self.dy, self.dy))
            self.sh_ard_mean = self.sh['ard_shape'] / self.sh['ard_scale']
            self.sh_ard_log_mean = psi(self.sh['ard_shape']) - np.log(self.sh['ard_scale'])
            self.omega = np.ones((n_basis, n_basis)) / n_basis

    # -- one sweep over the layers ----------------------------------------------------------
    def _latent_for(self, layer_index):
        """Stats.py:316-348: sum of all coarser layers' contributions at the training points."""
        n0 = self.y.shape[0]
        mean = np.zeros((n0, self.dy))
        var = np.zeros(n0)
        for jp in range(layer_index):
            parts = [blk.contribution() for blk in self.blocks[jp]]
            mean += np.concatenate([p[0] for p in parts])
            var += np.concatenate([p[1] for p in parts])
        for blk in self.blocks[layer_index]:
            blk.fbar = mean[blk.rows]
            blk.fvar = var[blk.rows]

    def sweep_independent(self):
        """MRGP.py:663-731 with forced independence: every block owns its axes and ARD."""
        m = self.m
        for j, layer in enumerate(self.blocks):
            n_regions = len(layer)
            for blk in layer:
                blk.y = self.y[blk.rows]                                # LatentOutputs.py:11-18
                blk.scale_given_axis(blk.y, blk.ard_mean)
            for blk in layer:                                           # Posteriors.py:262-290
                for i in range(m):
                    cand = 0.5 * blk.noise_mean * blk.zeta[i] * np.outer(blk.ytil[:, i], blk.ytil[:, i])
                    blk.b[i], bing = axis_update(cand)                  # prior B' is zero
                    blk.cov[i] = axis_cov(bing)
            for blk in layer:
                blk.eau, blk.mom2, blk.cen2 = scale_stats(blk.zeta, blk.precision, blk.ytil, blk.cov)
            for blk in layer:                                           # Posteriors.py:293-300
                blk.ard_shape = np.full(m, np.sum(blk.ard_shape0 / m)) + 0.5 * n_regions
                blk.ard_scale = np.full(m, np.sum(blk.ard_scale0 / m)) + 0.5 * blk.mom2 / blk.spec
                blk.ard_mean = blk.ard_shape / blk.ard_scale
            for blk in layer:
                blk.y_var = 0.0
            self._bias_and_noise(layer)
            if j + 1 < self.n_layers:
                self._latent_for(j + 1)

    def _bias_and_noise(self, layer):
        """Bias (Posteriors.py:75-110 / 345-372) then noise (:113-211 / 375-452) of one layer for
        the four region-specific / shared combinations.  The noise residual carries no bias --
        the bias enters as  - precision |mean|^2.  Whether the target variance is multiplied by
        the block size differs between the variants and between the two posterior classes
        exactly as in the reference (only ciMRGP has a non-zero target variance)."""
        sums = [blk.residual_sums(blk.y) for blk in layer]
        dy = self.dy
        if self.bias_regional:
            for blk, sm in zip(layer, sums):
                blk.bias_prec = blk.bias_prec0 + blk.n
                blk.bias_mean = sm[0] / blk.bias_prec
        else:
            prec = layer[0].bias_prec0 + sum(blk.n for blk in layer)
            mean = sum(sm[0] for sm in sums) / prec
            for blk in layer:
                blk.bias_prec, blk.bias_mean = prec, mean
        both_regional = self.noise_regional and self.bias_regional
        times_n = both_regional if self.fi else not both_regional
        if self.fi and self.noise_regional and not self.bias_regional:
            times_n = False                                             # Posteriors.py:420-422
        terms = []
        for blk, sm in zip(layer, sums):
            y_var = blk.y_var * blk.n if times_n else blk.y_var
            terms.append(sm[1] + sm[2] + sm[3] + y_var)
        bias_term = [blk.bias_prec * np.dot(blk.bias_mean, blk.bias_mean) for blk in layer]
        if self.noise_regional:
            for blk, t, bt in zip(layer, terms, bias_term):
                blk.noise_shape = blk.noise_shape0 + 0.5 * dy * blk.n
                blk.noise_scale = blk.noise_scale0 + 0.5 * (0.0 - bt + t)
        else:
            shape = layer[0].noise_shape0 + sum(0.5 * dy * blk.n for blk in layer)
            if self.bias_regional:
                scale = layer[0].noise_scale0 + sum(0.5 * (0.0 - bt + t) for t, bt in zip(terms, bias_term))
            else:
                scale = layer[0].noise_scale0 + 0.5 * (0.0 - bias_term[0] + sum(terms))
            for blk in layer:
                blk.noise_shape, blk.noise_scale = shape, scale
        for blk in layer:
            blk.bias_var = 1.0 / blk.bias_prec
            blk.noise_mean = blk.noise_shape / blk.noise_scale
            blk.noise_log_mean = psi(blk.noise_shape) - np.log(blk.noise_scale)

    def sweep_shared(self):
        """MRGP.py:574-661 (ciMRGP): one Bingham axis and one ARD weight per basis function,
        shared by all regions and handed from layer to layer as the next layer's prior, with
        the soft permutation ``omega`` aligning basis i of this layer to basis k of the last."""
        m = self.m
        for j, layer in enumerate(self.blocks):
            n_regions = len(layer)
            if j == 0:
                prev = {k: v.copy() for k, v in self.sh_prior.items()}
                for blk in layer:
                    blk.y = self.y[blk.rows]
                    blk.y_var = 0.0                                      # LatentOutputs.py:6-9
            else:
                prev = {k: v.copy() for k, v in self.sh.items()}
                for blk in layer:                                        # LatentOutputs.py:20-49
                    blk.y = blk.phi @ blk.eau.T + (blk.bias_mean + blk.fbar)
                    blk.y_var = 1.0 / blk.noise_mean
            for blk in layer:
                blk.scale_given_axis(blk.y, self.sh_ard_mean)
            for i in range(m):                                           # Posteriors.py:470-500
                cand = np.tensordot(self.omega[i], prev['b'], axes=1)
                for blk in layer:
                    cand = cand + 0.5 * blk.noise_mean * blk.zeta[i] * np.outer(blk.ytil[:, i], blk.ytil[:, i])
                self.sh['b'][i], bing = axis_update(cand)
                self.sh['log_const'][i] = bing['log_const']
                self.sh_cov[i] = axis_cov(bing)
            for blk in layer:
                blk.eau, blk.mom2, blk.cen2 = scale_stats(blk.zeta, blk.precision, blk.ytil, self.sh_cov)
            for i in range(m):                                           # Posteriors.py:503-512
                self.sh['ard_shape'][i] = np.sum(self.omega[i] * prev['ard_shape']) + 0.5 * n_regions
                self.sh['ard_scale'][i] = np.sum(self.omega[i] * prev['ard_scale']) \
                    + 0.5 * sum(blk.mom2[i] / blk.spec[i] for blk in layer)
            self.sh_ard_mean = self.sh['ard_shape'] / self.sh['ard_scale']
            self.sh_ard_log_mean = psi(self.sh['ard_shape']) - np.log(self.sh['ard_scale'])
            self.omega = soft_permutation(prev, self.sh_cov, self.sh_ard_log_mean, self.sh_ard_mean)
            self._bias_and_noise(layer)
            if self.adaptive:                                            # MRGP.py:626-636
                for blk in layer:
                    blk.rebuild_basis(self._learn_interval(blk), self.spectral)
            if j + 1 < self.n_layers:
                self._latent_for(j + 1)

    def _learn_interval(self, blk):
        """BasisInterval.py:18-54,64-92: per input dimension a bounded scalar minimisation
        (scipy's ``fminbound``, as the reference) of ``_interval_objective`` between the data
        range and 1.2x it (capped at n_basis), the other dimensions held at their old values."""
        from scipy.optimize import fminbound
        d = blk.x.shape[1]
        m = self.m
        orders = np.arange(1, m + 1)
        per_dim = [np.sin(np.pi * orders[None, :] * (blk.x[:, [k]] + blk.interval[k]) / (2 * blk.interval[k]))
                   / np.sqrt(blk.interval[k]) for k in range(d)]
        lam_dim = [(np.pi * orders / (2 * blk.interval[k])) ** 2 for k in range(d)]
        out = np.zeros(d)
        for p in range(d):
            if d > 1:
                others = [k for k in range(d) if k != p]
                penalty_phi = np.prod([per_dim[k] for k in others], axis=0)
                penalty_lam = np.sum([lam_dim[k] for k in others], axis=0)
            else:
                penalty_phi = np.ones((blk.n, m))                       # MRGP.py:341-343: untouched when dx == 1
                penalty_lam = np.zeros(m)
            low = np.max(np.abs(blk.x[:, p])) * self.opt_factor[0]
            high = min(m, low * self.opt_factor[1])
            if high < low:
                high = low * self.opt_factor[1]
            out[p] = fminbound(self._interval_objective, low, high, args=(blk, p, penalty_phi, penalty_lam), full_output=0)
        return out

    def _interval_objective(self, interval, blk, p, penalty_phi, penalty_lam):
        """BasisInterval.py:94-134, negated expected log-likelihood terms that depend on the
        interval of dimension p plus the spectral prior term."""
        m = self.m
        orders = np.arange(1, m + 1)
        xp = blk.x[:, [p]]
        phi = np.sin(np.pi * orders[None, :] * (xp + interval) / (2 * interval)) / np.sqrt(interval)
        lam = (np.pi * orders / (2 * interval)) ** 2
        spec = self.spectral(np.sqrt(lam + penalty_lam))
        psi_ = penalty_phi * phi
        sq = np.sum(psi_ ** 2, axis=0)
        term1 = np.sum(blk.eau ** 2, axis=0) * sq
        term2 = np.einsum('nc,ci,ni->i', blk.bias_mean + blk.fbar, blk.eau, psi_)
        term3 = np.einsum('nc,ci,ni->i', blk.y, blk.eau, psi_)
        term4 = blk.cen2 * sq
        ll = -0.5 * blk.noise_mean * np.sum(2 * term1 + 4 * term2 - 2 * term3 + term4)
        prior = -0.5 * np.sum(np.log(spec) - 0.5 * (self.sh_ard_mean * blk.mom2) / spec)
        return -(ll + prior)

    # -- lower bound (MRGP.py:414-571) ----------------------------------------------------------
    def _lower_bound(self):
        """The quantity ``fit(n_iter, tol)`` monitors, term by term as the reference codes it --
        including that the data term is the bare sum of squared-error and variance terms plus
        the log-normaliser (no -E[tau]/2 factor), that the axis term multiplies the two matrices
        elementwise before the trace, and that for j > 0 the "previous" shared posterior is the
        current one (MRGP.py:379 aliases instead of copying)."""
        m = self.m
        per_layer = []
        for j, layer in enumerate(self.blocks):
            prev = self.sh_prior if j == 0 else self.sh
            data = 0.0
            const = 0.0
            for blk in layer:
                resid = blk.y - blk.phi @ blk.eau.T - blk.fbar - blk.bias_mean
                data += np.sum(resid * resid) + np.sum(blk.fvar) + np.sum(blk.phi ** 2 * blk.cen2) + blk.bias_var \
                    + blk.y_var * blk.n
                const += 0.5 * self.dy * (blk.noise_log_mean - np.log(2 * np.pi)) * blk.n
            data += const
            scale = 0.0
            for blk in layer:
                scale += np.sum(0.5 * self.sh_ard_log_mean / blk.spec - 0.5 * self.sh_ard_mean * blk.mom2 / blk.spec)
                scale -= np.sum(0.5 * np.log(blk.precision) - 0.5)
            diag_cov = np.einsum('iaa->ia', self.sh_cov)
            axis_p = np.sum(self.omega * (-prev['log_const'][None, :] + diag_cov @ np.einsum('kaa->ka', prev['b']).T))
            axis_q = np.sum(-self.sh['log_const'] + np.sum(diag_cov * np.einsum('iaa->ia', self.sh['b']), axis=1))

            def gamma_term(shape, scale_, log_mean, mean):
                return shape * np.log(scale_) - gammaln(shape) + (shape - 1) * log_mean - scale_ * mean

            ard_p = np.sum(self.omega * gamma_term(prev['ard_shape'][None, :], prev['ard_scale'][None, :],
                                                   self.sh_ard_log_mean[:, None], self.sh_ard_mean[:, None]))
            ard_q = np.sum(gamma_term(self.sh['ard_shape'], self.sh['ard_scale'], self.sh_ard_log_mean, self.sh_ard_mean))
            bias = 0.0
            noise = 0.0
            for blk in layer:
                w = blk.bias_mean
                term = 1.0 / (blk.bias_prec * blk.noise_mean) + np.dot(w, w)
                bias += 0.5 * self.dy * (np.log(blk.bias_prec0) + blk.noise_log_mean - np.log(2 * np.pi)) \
                    + 0.5 * blk.bias_prec0 * blk.noise_mean * term
                bias -= 0.5 * self.dy * (np.log(blk.bias_prec) + blk.noise_log_mean - np.log(2 * np.pi)) - 0.5
                noise += gamma_term(blk.noise_shape0, blk.noise_scale0, blk.noise_log_mean, blk.noise_mean)
                noise -= gamma_term(blk.noise_shape, blk.noise_scale, blk.noise_log_mean, blk.noise_mean)
            per_layer.append(data + scale + (axis_p - axis_q) + (ard_p - ard_q) + bias + noise)
        return float(np.sum(per_layer)), per_layer

    def fit(self, n_iter, tol=None, min_iter=10):
        """MRGP.py:367-412.  With ``tol`` the ciMRGP sweep also records the lower bound and stops
        once, after ``min_iter`` sweeps, layer 0's bound changes by less than ``tol``."""
        min_iter = min(min_iter, n_iter)
        for it in range(1, n_iter + 1):
            if self.fi:
                self.sweep_independent()
                continue
            self.sweep_shared()
            if tol is None:
                continue
            total, per_layer = self._lower_bound()
            self.lower_bound.append(total)
            for j, v in enumerate(per_layer):
                self.lower_bound_layer[j].append(v)
            if it > min_iter and abs(self.lower_bound_layer[0][-1] - self.lower_bound_layer[0][-2]) < abs(tol):
                break

    # -- prediction -------------------------------------------------------------------------
    def _test_phi(self, xs_n, test_bounds):
        return [[laplace_basis(xs_n[a:b], blk.interval, self.m)[0] for (a, b), blk in zip(tb, layer)]
                for tb, layer in zip(test_bounds, self.blocks)]

    def predict_mean(self, xs, test_bounds=None):
        """MRGP.py:733-814."""
        xs_n = (np.asarray(xs, dtype=np.float64) - self.mean_x) / self.std_x
        if test_bounds is None:
            blk = self.blocks[0][0]
            return blk.contribution(laplace_basis(xs_n, blk.interval, self.m)[0])[0]
        phis = self._test_phi(xs_n, test_bounds)
        out = np.zeros((xs_n.shape[0], self.dy))
        for layer, layer_phi in zip(self.blocks[:len(test_bounds)], phis):
            out += np.concatenate([blk.contribution(p)[0] for blk, p in zip(layer, layer_phi)])
        return out

    def predict_var(self, xs, test_bounds=None):
        """MRGP.py:832-937.  With index sets every layer adds: the coarser layers' variance
        *at the first test point of the region* (``latent_f_var[l][0]``, MRGP.py:934), its own
        Phi^2 c2, the bias variance and n_l / E[tau]."""
        xs_n = (np.asarray(xs, dtype=np.float64) - self.mean_x) / self.std_x
        if test_bounds is None:
            blk = self.blocks[0][0]
            return blk.contribution(laplace_basis(xs_n, blk.interval, self.m)[0])[1]
        phis = self._test_phi(xs_n, test_bounds)
        ns = xs_n.shape[0]
        total = np.zeros(ns)
        coarser = np.zeros(ns)
        for j, (layer, layer_phi, tb) in enumerate(zip(self.blocks, phis, test_bounds)):
            own = []
            for blk, p, (a, b) in zip(layer, layer_phi, tb):
                var_f = coarser[int(a)] if j > 0 else 0.0
                n_l = p.shape[0]
                total[int(a):int(b)] += var_f + (p ** 2) @ blk.cen2 + n_l / blk.noise_mean + blk.bias_var
                own.append(blk.contribution(p)[1])
            coarser = coarser + np.concatenate(own)
        return total
