"""The recipe of the reference's example (scripts/tests/ciMRGP_vs_fiMRGP.py: 32 noisy samples
of a two-output function on [1, 3], learned input warp, adaptive basis intervals, 100 000 test
points) on the MI355X backend.  Only the imports differ from the reference's script; figures are
left out.

    python examples/cimrgp_vs_fimrgp.py [n_test]
"""
import sys

import numpy as np

from cimrgp_amd import BasisInterval, IndexSetUniform, LaplacianEigenpairs, MaternKernel, MultiResolutionGaussianProcess


def f(x):
    y1 = np.log(np.log(x) + abs(np.sin(x ** 2) * np.exp(np.sin(np.cos(2 * x)))))
    y2 = np.log(np.log(x) + abs(np.sin(-x ** 2 + 3 * x + 5) + np.log(1 + abs(np.cos(x ** 2)))))
    return np.hstack([y1, y2])


def generate_data(n_train=32, n_test=100000, seed=0):
    rng = np.random.default_rng(seed)
    x = np.atleast_2d(np.linspace(1, 3, n_train)).T
    y = f(x)
    y = y + 0.1 * rng.normal(0, 1 + rng.random(y.shape))
    xs = np.atleast_2d(np.linspace(1, 3, n_test)).T
    return [x, y], [xs, f(xs)]


def run(train, test, n_res, divider, n_basis, n_iter, forced_independence, verbose=False):
    index_set = IndexSetUniform(sample_length=train[0].shape[0], resolution=n_res, divider=divider)
    mrgp = MultiResolutionGaussianProcess(train_xy=train, n_basis=n_basis, index_set_obj=index_set,
                                          basis_function_obj=LaplacianEigenpairs(),
                                          spectral_density_obj=MaternKernel(nu=1, l=1, sf=1),
                                          adaptive_inputs=True, standard_normalized_inputs=True,
                                          basis_interval_obj=BasisInterval(opt_interval_factor=(1, 1.2)), interval_factor=1,
                                          axis_resolution_specific=False, ard_resolution_specific=False,
                                          noise_region_specific=True, bias_region_specific=True,
                                          noninformative_initialization=True, forced_independence=forced_independence,
                                          snr_ratio=None, verbose=verbose)
    mrgp.fit(n_iter, None)
    index_set_test = IndexSetUniform(sample_length=test[0].shape[0], resolution=n_res, divider=divider)
    mean = mrgp.get_predicted_mean(test_x=test[0], index_set_obj=index_set_test)
    var = mrgp.get_central_moment2(test_x=test[0], index_set_obj=None, number_of_regions=None)
    resid = test[1] - mean
    r2 = 1.0 - np.sum(resid ** 2, axis=0) / np.sum((test[1] - test[1].mean(axis=0)) ** 2, axis=0)
    mll = np.mean(-0.5 * np.log(2 * np.pi * var) - 0.5 * np.sum(resid ** 2, axis=1) / var)
    return dict(model='fiMRGP' if forced_independence else 'ciMRGP', n_res=n_res, r2=float(np.mean(r2)),
                mse=float(np.mean(resid ** 2)), mll=float(mll))


if __name__ == '__main__':
    n_test = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    train, test = generate_data(n_test=n_test)
    for forced in (False, True):
        for n_res in (0, 1, 2):
            print(run(train, test, n_res=n_res, divider=2, n_basis=15, n_iter=20, forced_independence=forced))
