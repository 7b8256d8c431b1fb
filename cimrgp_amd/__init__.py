"""cimrgp_amd -- MI355X-native dense covariance/posterior path of ciMRGP.

Host-side mirror of the reference's module layout (``IndexSetGenerator``,
``KernelClass``, ``Posteriors``, ``RegressionInput``, ``Inputs``, ``MRGP``)
over hand-written HIP kernels reached through the C ABI of ``libcimrgp.so``
(include/cimrgp.h).  No CPU fallback: importing the package does not need a
GPU, computing does.
"""
from .IndexSetGenerator import IndexSetUniform
from .KernelClass import RBFKernel, MaternKernel, LaplacianEigenpairs
from .BasisInterval import BasisInterval
from .RegressionInput import RegressionMethod, GP_RBF
from .Inputs import space_filling_order
from .Posteriors import DensePosterior, DenseBlock
from .MRGP import MultiResolutionGaussianProcess
from . import _lib, device, dist

__all__ = ["IndexSetUniform", "RBFKernel", "MaternKernel", "LaplacianEigenpairs", "BasisInterval", "RegressionMethod",
           "GP_RBF", "DensePosterior", "DenseBlock", "MultiResolutionGaussianProcess", "space_filling_order", "device", "dist"]
