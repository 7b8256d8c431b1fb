"""Kernel / basis objects of the path.

``RBFKernel`` is the new covariance object of the dense path; it follows the
protocol of the reference's ``MaternKernel`` (KernelClass.py:40-65: attributes
``l``, ``sf``; methods ``kernel``, ``log_kernel``, ``spectral``,
``log_spectral``, ``estimate_kernel``) so that it can be passed as
``spectral_density_obj`` (scalar or per-layer list, MRGP.py:71-79), and adds
the Gram builder that runs on the GPU.  ``MaternKernel`` and
``LaplacianEigenpairs`` are host-side restatements of the reference's
reduced-rank objects, kept for API completeness (validated against captured
reference outputs in tests/golden/kernel_objects.npz).
"""
import numpy as np
from scipy.special import kv, gammaln


class RBFKernel(object):
    """k(r) = sf * exp(-r^2 / (2 l^2));  ``sf`` is the signal VARIANCE, as the
    reference's kernels multiply by ``sf`` directly (KernelClass.py:77).
    ``noise``: fixed Gaussian noise variance of the blocks using this kernel,
    or None for the plugin's rule 0.01 * var(targets) (RegressionInput.py:62)."""
    name = 'RBF'

    def __init__(self, l=1., sf=1., noise=None):
        if not l > 0:
            raise ValueError('length-scale must be positive')
        if not sf > 0:
            raise ValueError('signal variance must be positive')
        self.l = float(l)
        self.sf = float(sf)
        self.noise = None if noise is None else float(noise)

    # ---- scalar-distance protocol (host, NumPy) ---------------------------
    def log_kernel(self, r):
        r = np.asarray(r, dtype=np.float64)
        return np.log(self.sf) - 0.5 * (r / self.l) ** 2

    def kernel(self, r):
        return np.exp(self.log_kernel(r))

    def log_spectral(self, s):
        # 1-D spectral density in the reference's convention (KernelClass.py:80-90):
        # S(s) = sf * sqrt(2 pi) * l * exp(-l^2 s^2 / 2)
        s = np.asarray(s, dtype=np.float64)
        return np.log(self.sf) + 0.5 * np.log(2 * np.pi) + np.log(self.l) - 0.5 * (self.l * s) ** 2

    def spectral(self, s):
        return np.exp(self.log_spectral(s))

    def estimate_kernel(self, phi_x1, phi_x2, lambdas):
        """Reduced-rank reconstruction sum_p S(sqrt(lambda_p)) phi_p(x) phi_p(x')."""
        weights = self.spectral(np.sqrt(np.asarray(lambdas, dtype=np.float64)))
        return np.einsum('np,np,p->n', phi_x1, phi_x2, weights)

    # ---- dense Gram builder (device, HIP) ---------------------------------
    def gram(self, x, x2=None, diag_add=0.0, lower_only=False):
        """Gram matrix on the GPU.  ``x``/``x2``: CUDA tensors (n x d).  Returns a
        torch view (n x n2) of the padded device buffer."""
        from . import device as dev
        if x2 is None:
            buf = dev.rbf_gram(x, self.l, self.sf, diag_add, lower_only)
            return buf[:x.shape[0], :x.shape[0]]
        buf = dev.rbf_cross(x, x2, self.l, self.sf)
        return buf[:x.shape[0], :x2.shape[0]]

    def K(self, x, x2=None, dtype='f64'):
        """NumPy in / NumPy out convenience around :meth:`gram` (computed on the GPU)."""
        from . import device as dev
        device = dev.require_gpu()
        tdt = dev.as_torch_dtype(dtype)
        xd = dev.to_device(np.atleast_2d(x), tdt, device)
        x2d = None if x2 is None else dev.to_device(np.atleast_2d(x2), tdt, device)
        return self.gram(xd, x2d).cpu().numpy()


class LaplacianEigenpairs(object):
    """Dirichlet-Laplacian eigenpairs on [-L, L]^d (reference KernelClass.py:6-37):
    phi(x) = prod_k L_k^-1/2 sin(pi j (x_k + L_k) / (2 L_k)),  lambda = sum_k (pi j / 2 L_k)^2."""
    name = 'Laplacian'

    def get_eigenpairs(self, x, basis_id, basis_interval=None, per_dimension=False):
        x = np.asarray(x)
        if basis_interval is None:
            basis_interval = np.max(np.abs(x), axis=0)
        basis_interval = np.asarray(basis_interval, dtype=np.float64)
        if len(basis_interval) != x.shape[1]:
            raise ValueError('Basis interval should have the same dimensionality as the input.')
        phi, lam = self._learn(x, basis_interval, basis_id)
        if per_dimension is True:
            return phi, lam
        return np.prod(phi, axis=1), np.sum(lam)

    @staticmethod
    def _learn(x, basis_interval, basis_id):
        half = basis_interval[None, :]
        phi = np.sin((np.pi * basis_id) * (x + half) / (2 * half)) / np.sqrt(half)
        lam = ((np.pi * basis_id) / (2 * basis_interval)) ** 2
        return phi, lam


class MaternKernel(object):
    """Matern covariance of a scalar distance and its spectral density, in the log
    domain (reference KernelClass.py:40-90)."""
    name = 'Matern'

    def __init__(self, nu=1, l=1, sf=1):
        self.nu = nu
        self.l = l
        self.sf = sf

    def log_kernel(self, r):
        nu, ell = self.nu, self.l
        log_arg = np.log(np.sqrt(2 * nu) * r) - np.log(ell)
        return (np.log(self.sf) + (1 - nu) * np.log(2) - gammaln(nu) + nu * log_arg
                + np.log(kv(nu, np.exp(log_arg))))

    def kernel(self, r):
        return np.exp(self.log_kernel(r))

    def log_spectral(self, s):
        nu, ell = self.nu, self.l
        log_arg = np.log(2 * nu) - 2 * np.log(ell)
        return (np.log(self.sf) + 0.5 * np.log(2 * np.pi) + nu * log_arg + gammaln(nu + 0.5) - gammaln(nu)
                - (nu + .5) * np.log(np.exp(log_arg) + np.asarray(s) ** 2))

    def spectral(self, s):
        return np.exp(self.log_spectral(s))

    def estimate_kernel(self, phi_x1, phi_x2, lambdas):
        weights = self.spectral(np.sqrt(np.asarray(lambdas, dtype=np.float64)))
        return np.einsum('np,np,p->n', phi_x1, phi_x2, weights)
