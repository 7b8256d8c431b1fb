"""Oracle (test infrastructure): dense exact-GP arithmetic D1-D5 in FP64.

In the reference this arithmetic is behind ``GP_RBF`` (RegressionInput.py:55-67)
inside third-party GPy -- absent here, version unpinned, no reference test
holds a number for it: PARITY UNPINNED.  This is a restatement of the
published algorithm (Rasmussen & Williams 2006, Algorithm 2.1) on
``scipy.linalg``; the kernel is GPy's ``RBF`` (isotropic squared-exponential,
``ARD=False``, RegressionInput.py:60):  k(a,b) = sf2 * exp(-|a-b|^2 / (2 l^2)).
"""
import numpy as np
import scipy.linalg as sla


def rbf_gram(xa, xb=None, ell=1.0, sf2=1.0, diag_add=0.0):
    """D1: Gram / cross-Gram matrix.  ``xa`` (na x d), ``xb`` (nb x d).

    Squared distances are formed by explicit differences per input dimension
    (the same arithmetic the HIP kernel uses), not by the |a|^2+|b|^2-2ab
    expansion, so near-duplicate points do not lose digits.
    """
    xa = np.asarray(xa, dtype=np.float64)
    same = xb is None
    xb = xa if same else np.asarray(xb, dtype=np.float64)
    d2 = np.zeros((xa.shape[0], xb.shape[0]))
    for k in range(xa.shape[1]):
        diff = xa[:, k][:, None] - xb[:, k][None, :]
        d2 += diff * diff
    gram = sf2 * np.exp(d2 * (-0.5 / (ell * ell)))
    if same and diag_add != 0.0:
        gram[np.diag_indices_from(gram)] += diag_add
    return gram


def potrf_lower(k_mat):
    """D2: K = L L^T.  Returns (L, info) with LAPACK's ``info`` convention:
    0 = success, i > 0 = leading minor of order i is not positive definite
    (the convention SanityCheck.py:59-65 relies on through numpy's
    LinAlgError)."""
    chol, info = sla.lapack.dpotrf(np.asarray(k_mat, dtype=np.float64), lower=1)
    return np.tril(chol), int(info)


def block_fit(x, r, ell, sf2, noise):
    """D1+D2+D3 for one (resolution, partition) block.

    x (n x d) inputs, r (n x q) targets (already residualised / centred),
    returns dict(L, alpha, info) with alpha = (K + noise I)^-1 r.
    """
    k_mat = rbf_gram(x, None, ell, sf2, noise)
    chol, info = potrf_lower(k_mat)
    if info != 0:
        raise np.linalg.LinAlgError('Matrix is not positive definite (info=%d)' % info)
    z = sla.solve_triangular(chol, r, lower=True)
    alpha = sla.solve_triangular(chol, z, lower=True, trans='T')
    return dict(L=chol, alpha=alpha, z=z, info=info)


def block_predict(x, fit, xs, ell, sf2, want_var=True):
    """D4 (+D5): mean* = K(xs,x) alpha;  var* = sf2 - |L^-1 k*|^2  (latent f,
    no observation noise added)."""
    ks = rbf_gram(xs, x, ell, sf2)
    mean = ks @ fit['alpha']
    if not want_var:
        return mean, None
    v = sla.solve_triangular(fit['L'], ks.T, lower=True)
    var = sf2 - np.sum(v * v, axis=0)
    return mean, var
