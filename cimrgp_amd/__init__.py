"""cimrgp_amd -- MI355X-native dense covariance/posterior path of ciMRGP.

Host-side mirror of the reference's module layout (``IndexSetGenerator``,
``KernelClass``, ``Posteriors``, ``RegressionInput``, ``Inputs``, ``MRGP``)
over hand-written HIP kernels reached through the C ABI of ``libcimrgp.so``
(include/cimrgp.h).  No CPU fallback: importing the package does not need a
GPU, computing does.
"""
import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The look-ahead
# factorisation drives three queues beside the caller's and independent blocks of a layer run on a
# pool of up to 8 streams: with 4 hardware queues they share queues and serialise (measured: the
# 16 x 4096 layer of BASELINE config 3 on the stream pool 40.5 ms with 4 queues, 30.0 ms with 8).
# Read by the runtime when the process first touches the GPU; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
