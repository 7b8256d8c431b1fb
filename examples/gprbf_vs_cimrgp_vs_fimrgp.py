"""The comparison of the reference's second example (scripts/tests/GPRBF_vs_ciMRGP_vs_fiMRGP.py):
an exact RBF GP with optimised hyper-parameters next to ciMRGP and fiMRGP on the same 160 noisy
samples, 100 000 test points.  Where the reference calls GPy directly
(``GPy.models.GPRegression(...); .optimize(); .predict``) this uses the drop-in ``GP_RBF`` plugin
(exact GP on the HIP path: Gram, Cholesky, log marginal likelihood and its gradient on the GPU,
L-BFGS-B on the host).

    python examples/gprbf_vs_cimrgp_vs_fimrgp.py [n_test]
"""
import importlib.util
import os
import sys

import numpy as np

from cimrgp_amd import GP_RBF

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("cimrgp_example", os.path.join(_here, "cimrgp_vs_fimrgp.py"))
base = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(base)


def run_gp_rbf(train, test):
    model = GP_RBF(optimize=True)
    model.fit(train)
    mean, var = model.predict_with_variance(test[0])
    resid = test[1] - mean
    r2 = 1.0 - np.sum(resid ** 2, axis=0) / np.sum((test[1] - test[1].mean(axis=0)) ** 2, axis=0)
    # the plugin z-scores the labels: its latent variance is in z-scored units -> back to the outputs' scale
    var = np.asarray(var).reshape(-1) * float(np.mean(model.labels_std ** 2))
    mll = np.mean(-0.5 * np.log(2 * np.pi * var) - 0.5 * np.sum(resid ** 2, axis=1) / var)
    return dict(model='GP_RBF', r2=float(np.mean(r2)), mse=float(np.mean(resid ** 2)), mll=float(mll),
                lengthscale=float(model.kernel.l), variance=float(model.kernel.sf))


if __name__ == '__main__':
    n_test = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    train, test = base.generate_data(n_train=160, n_test=n_test)          # GPRBF_vs_...py:22: 5 * 32 samples
    print(run_gp_rbf(train, test))
    for forced in (False, True):
        print(base.run(train, test, n_res=2, divider=2, n_basis=15, n_iter=20, forced_independence=forced))
