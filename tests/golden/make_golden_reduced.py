"""More golden fixtures for the reduced-rank model, from the reference itself (same shim and
rules as make_golden.py: only DATA is written).  Cases: adaptive basis intervals (1-D and 2-D
inputs), 2-D inputs without them, the lower bound of ``fit(n_iter, tol)``, and the shared
noise / bias flag variants.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_reduced.py [tag ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, _import_reference, toy_f          # noqa: E402


def toy_2d(x):
    return np.stack([np.sin(3 * x[:, 0]) + x[:, 1], np.cos(2 * x[:, 1]) * x[:, 0]], axis=1)


CASES = {
    # tag: (inputs, resolution, n_basis, forced, basis_interval, n_iter, tol, extra ctor kwargs)
    'ci_r2_bi': ('1d', 2, 30, False, True, 5, None, {}),
    'ci_r1_bi_2d': ('2d', 1, 12, False, True, 4, None, {}),
    'fi_r1_2d': ('2d', 1, 12, True, False, 4, None, {}),
    'ci_r1_2d': ('2d', 1, 12, False, False, 4, None, {}),
    'ci_r2_elbo': ('1d', 2, 30, False, False, 6, 1e-12, {}),
    'ci_r2_shared_nb': ('1d', 2, 30, False, False, 4, None, dict(noise_region_specific=False, bias_region_specific=False)),
    'ci_r2_shared_n': ('1d', 2, 30, False, False, 4, None, dict(noise_region_specific=False)),
    'ci_r2_shared_b': ('1d', 2, 30, False, False, 4, None, dict(bias_region_specific=False)),
    'fi_r2_shared_nb': ('1d', 2, 30, True, False, 4, None, dict(noise_region_specific=False, bias_region_specific=False)),
    'fi_r2_snr': ('1d', 2, 30, True, False, 4, None, dict(snr_ratio=10.0, interval_factor=1.2)),
}


def main():
    _import_reference()
    from BasisInterval import BasisInterval
    from IndexSetGenerator import IndexSetUniform
    from KernelClass import LaplacianEigenpairs, MaternKernel
    from MRGP import MultiResolutionGaussianProcess

    tags = sys.argv[1:] or list(CASES)
    for tag in tags:
        kind, res, n_basis, forced, use_bi, n_iter, tol, extra = CASES[tag]
        np.random.seed(17)
        if kind == '1d':
            n, ns = 512, 384
            x = np.atleast_2d(np.linspace(1, 3, n)).T
            y = toy_f(x) + 0.1 * np.random.normal(size=(n, 2))
            xt = np.atleast_2d(np.linspace(1.01, 2.99, ns)).T
        else:
            n, ns = 400, 200
            x = np.random.uniform(-1, 1, size=(n, 2))
            x = x[np.argsort(x[:, 0])]
            y = toy_2d(x) + 0.05 * np.random.normal(size=(n, 2))
            xt = np.random.uniform(-1, 1, size=(ns, 2))
            xt = xt[np.argsort(xt[:, 0])]
        idx = IndexSetUniform(n, res, 2)
        model = MultiResolutionGaussianProcess(train_xy=[x, y], n_basis=n_basis, index_set_obj=idx,
                                               basis_function_obj=LaplacianEigenpairs(),
                                               spectral_density_obj=MaternKernel(nu=1, l=1, sf=1),
                                               basis_interval_obj=BasisInterval(opt_interval_factor=(1, 1.2)) if use_bi else None,
                                               adaptive_inputs=False, forced_independence=forced, **extra)
        if tol is None:
            model.fit(n_iter, None)
        else:
            model.fit(n_iter, tol, min_iter=n_iter)
        idx_t = IndexSetUniform(ns, res, 2)
        blob = dict(x=x, y=y, xt=xt, n_basis=np.int64(n_basis), resolution=np.int64(res), forced_independence=np.bool_(forced),
                    n_iter=np.int64(n_iter), adaptive_basis_intervals=np.bool_(use_bi))
        for k, v in extra.items():
            blob['kw_' + k] = np.asarray(v)
        if tol is not None:
            blob['lower_bound'] = np.asarray(model.lower_bound, dtype=np.float64)
            blob['lower_bound_layer'] = np.asarray(model.lower_bound_layer, dtype=np.float64)
        blob['pred_mean_global'] = model.get_predicted_mean(xt)
        blob['pred_var_global'] = model.get_central_moment2(xt)
        blob['pred_mean_index'] = model.get_predicted_mean(xt, idx_t)
        stats = model.stats_obj
        for j in range(model.n_layers):
            noise_specific = model.noise_region_specific
            bias_specific = model.bias_region_specific
            if not noise_specific:
                blob['noise_mean_%d' % j] = np.float64(stats[j].noise_mean)
            if not bias_specific:
                blob['bias_mean_%d' % j] = np.asarray(stats[j].bias_mean)
            for l in range(model.n_regions[j]):
                key = '_%d_%d' % (j, l)
                blob['scale_axis_mean' + key] = stats[j].scale_axis_mean[l]
                blob['scale_moment2' + key] = stats[j].scale_moment2[l]
                if bias_specific:
                    blob['bias_mean' + key] = np.asarray(stats[j].bias_mean[l])
                if noise_specific:
                    blob['noise_mean' + key] = np.float64(stats[j].noise_mean[l])
                blob['interval' + key] = np.asarray(model.train_basis_intervals[j][l])
                blob['latent_f_mean' + key] = stats[j].latent_f_mean[l]
        if not forced:
            blob['shared_ard_mean'] = model.shared_stats.ard_mean
            blob['shared_omega'] = model.shared_stats.omega
            blob['shared_axis_cov'] = model.shared_stats.axis_cov
        blob['pred_var_index'] = model.get_central_moment2(xt, idx_t)     # mutates stats: last
        np.savez_compressed(os.path.join(OUT, 'reference_model_%s.npz' % tag), **blob)
        print('wrote', tag, flush=True)


if __name__ == '__main__':
    main()
