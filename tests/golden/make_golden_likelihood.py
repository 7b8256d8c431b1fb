"""Golden values of ``get_test_likelihood`` (reference: src/MRGP.py:825-831) from the reference itself, for the two
models whose fits are already pinned by reference_model_{fi_r2,ci_r2}.npz (same data, seed and sweeps as
make_golden.py; same shim and rules: only DATA is written).  Both call forms: without an index set (resolution 0,
region 0) and with the test index set (which re-runs the residual chain on the test grid and mutates the model's
statistics: taken last).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_likelihood.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, _import_reference, toy_f          # noqa: E402


def main():
    _import_reference()
    from IndexSetGenerator import IndexSetUniform
    from KernelClass import LaplacianEigenpairs, MaternKernel
    from MRGP import MultiResolutionGaussianProcess

    blob = {}
    for tag, res, forced in [('fi_r2', 2, True), ('ci_r2', 2, False)]:
        np.random.seed(11)
        n = 512
        x = np.atleast_2d(np.linspace(1, 3, n)).T
        y = toy_f(x) + 0.1 * np.random.normal(size=(n, 2))
        model = MultiResolutionGaussianProcess(train_xy=[x, y], n_basis=30, index_set_obj=IndexSetUniform(n, res, 2),
                                               basis_function_obj=LaplacianEigenpairs(),
                                               spectral_density_obj=MaternKernel(nu=1, l=1, sf=1),
                                               adaptive_inputs=False, forced_independence=forced)
        model.fit(5, None)
        ns = 384
        xt = np.atleast_2d(np.linspace(1.01, 2.99, ns)).T
        yt = toy_f(xt) + 0.1 * np.random.default_rng(5).normal(size=(ns, 2))
        blob[tag + '_xt'] = xt
        blob[tag + '_yt'] = yt
        blob[tag + '_ll_global'] = np.float64(model.get_test_likelihood([xt, yt]))
        blob[tag + '_ll_index'] = np.float64(model.get_test_likelihood([xt, yt], IndexSetUniform(ns, res, 2)))
        print(tag, blob[tag + '_ll_global'], blob[tag + '_ll_index'], flush=True)
    np.savez_compressed(os.path.join(OUT, 'reference_likelihood.npz'), **blob)


if __name__ == '__main__':
    main()
