#!/usr/bin/env python3
"""Independent cross-check fixtures for the dense oracle (D1-D5), written in the build container:

    python tests/golden/make_sklearn_golden.py     # -> tests/golden/sklearn_gp.npz

The reference's exact GP is GPy's ``GPRegression(RBF, ARD=False)`` (RegressionInput.py:58-67); GPy is
absent, so the oracle (oracle/dense.py) restates the published algorithm and is PARITY UNPINNED by
the reference.  scikit-learn 1.7.2 (in this image) implements the same exact GP (Rasmussen &
Williams Alg. 2.1) independently: ``ConstantKernel(sf2) * RBF(ell) + WhiteKernel(noise)`` with
``optimizer=None`` is  K = sf2 exp(-|a-b|^2 / 2 ell^2) + noise I.  Its predictive mean and
variance are stored here and the oracle is held to them (tests/test_oracle_golden.py).  This does
not lift the "unpinned" status (sklearn is not the reference) -- it shows that two independent
implementations of the algorithm agree to rounding."""
import os

import numpy as np
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    cases = []
    for n in (64, 257, 512):
        for d in (1, 2):
            rng = np.random.default_rng(1000 * d + n)
            x = rng.uniform(-1.7, 1.7, size=(n, d))
            x = x[np.argsort(x[:, 0])]
            y = np.stack([np.sin(3 * x[:, 0]) + 0.3 * x[:, -1], np.cos(2 * x[:, 0] * x[:, -1])], axis=1)
            y += 0.05 * rng.normal(size=y.shape)
            xs = rng.uniform(-1.9, 1.9, size=(max(8, n // 4), d))
            ell, sf2, noise = (0.6, 1.3, 0.02) if d == 1 else (0.9, 0.8, 0.05)
            kern = ConstantKernel(sf2, "fixed") * RBF(ell, "fixed") + WhiteKernel(noise, "fixed")
            gp = GaussianProcessRegressor(kernel=kern, optimizer=None, alpha=0.0, normalize_y=False).fit(x, y)
            mean, std = gp.predict(xs, return_std=True)
            # sklearn's predictive variance is that of y* (the WhiteKernel's diagonal is part of
            # kernel.diag): the latent variance of R&W eq. 2.26 is that minus the noise level
            var_latent = std[:, 0] ** 2 - noise
            tag = "n%d_d%d" % (n, d)
            cases.append(tag)
            out[tag + "_x"], out[tag + "_y"], out[tag + "_xs"] = x, y, xs
            out[tag + "_hyp"] = np.array([ell, sf2, noise])
            out[tag + "_mean"], out[tag + "_var"] = mean, var_latent
            out[tag + "_lml"] = np.array(gp.log_marginal_likelihood_value_)
    out["cases"] = np.array(cases)
    import sklearn
    out["sklearn_version"] = np.array(sklearn.__version__)
    np.savez_compressed(os.path.join(HERE, "sklearn_gp.npz"), **out)
    print("wrote", os.path.join(HERE, "sklearn_gp.npz"), cases)


if __name__ == "__main__":
    main()
