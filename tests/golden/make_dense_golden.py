"""Dense-path fixtures D1-D6 from the build's own FP64 oracle (no reference
arithmetic exists at this boundary: GPy is absent, PARITY UNPINNED).

    python tests/golden/make_dense_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20240607)
    blob = {}
    for n, d in [(64, 1), (257, 2), (512, 1)]:
        tag = 'n%d_d%d' % (n, d)
        x = rng.uniform(-1.7, 1.7, size=(n, d))
        x = x[np.argsort(x[:, 0])]
        y = np.stack([np.sin(3 * x[:, 0]) + 0.3 * x[:, -1], np.cos(2 * x[:, 0] * x[:, -1])], axis=1)
        y += 0.05 * rng.normal(size=y.shape)
        xs = rng.uniform(-1.8, 1.8, size=(37, d))
        ell, sf2, noise = 0.4, 1.3, 0.02
        fit = oracle.block_fit(x, y, ell, sf2, noise)
        mean, var = oracle.block_predict(x, fit, xs, ell, sf2, True)
        blob.update({tag + '_x': x, tag + '_y': y, tag + '_xs': xs,
                     tag + '_hyp': np.array([ell, sf2, noise]),
                     tag + '_gram_row0': oracle.rbf_gram(x, None, ell, sf2, noise)[0],
                     tag + '_gram_trace': np.float64(np.trace(oracle.rbf_gram(x, None, ell, sf2, noise))),
                     tag + '_Ldiag': np.diag(fit['L']).copy(), tag + '_Llast': fit['L'][-1].copy(),
                     tag + '_alpha': fit['alpha'], tag + '_z': fit['z'],
                     tag + '_mean': mean, tag + '_var': var})
    # multiresolution chain, config-1 shape: N=512, d=1, q=2, 3 layers, divider 2
    n = 512
    x = np.linspace(1, 3, n)[:, None]
    y = np.hstack([np.sin(3.0 * x) + 0.3 * np.cos(11.0 * x * x),
                   np.cos(2.0 * x) * np.exp(-0.3 * x) + 0.2 * np.sin(17.0 * x)]) + 0.1 * rng.normal(size=(n, 2))
    xt = np.linspace(1.01, 2.99, 384)[:, None]
    xn, _, mu, sd = oracle.normalize_inputs(x)
    xtn = (xt - mu) / sd
    bounds = oracle.index_bounds_uniform(n, 2, 2)
    tbounds = oracle.index_bounds_uniform(384, 2, 2)
    specs = [oracle.DenseLayerSpec(1.0 / 2 ** j, 1.0, None) for j in range(3)]
    model, f_bar = oracle.mrgp_fit(xn, y, bounds, specs)
    mean, var = oracle.mrgp_predict(xn, model, specs, xtn, tbounds, True, True)
    blob.update(chain_x=x, chain_y=y, chain_xt=xt, chain_fbar=f_bar, chain_mean=mean, chain_var=var,
                chain_noise=np.array([[b['noise'] for b in layer] + [np.nan] * (4 - len(layer)) for layer in model]))
    st = oracle.gp_rbf_fit(x, y)
    blob.update(plugin_mean=oracle.gp_rbf_predict(st, xt))
    np.savez_compressed(os.path.join(OUT, 'dense_oracle.npz'), **blob)
    print('wrote dense_oracle.npz')


if __name__ == '__main__':
    main()
