"""End-to-end parity of the host API (GP_RBF plugin, multiresolution model)
with the oracle, and size-independent properties at BASELINE.json's N = 8192."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ca():
    import cimrgp_amd
    cimrgp_amd.device.require_gpu()
    return cimrgp_amd


def _relerr(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / (np.max(np.abs(b)) + 1e-300))


def test_gp_rbf_plugin_matches_oracle(ca, golden_dir):
    g = np.load(os.path.join(golden_dir, "dense_oracle.npz"))
    x, y, xt = g["chain_x"], g["chain_y"], g["chain_xt"]
    model = ca.GP_RBF()
    assert model.fit([x, y]) is True
    pred = model.predict(xt)
    assert pred.shape == (xt.shape[0], 2)
    assert _relerr(pred, g["plugin_mean"]) < 1e-5            # north_star bar
    assert _relerr(pred, g["plugin_mean"]) < 1e-8            # what f64 actually delivers
    st = oracle.gp_rbf_fit(x, y)
    m, v = oracle.gp_rbf_predict(st, xt, want_var=True)
    m2, v2 = model.predict_with_variance(xt)
    assert _relerr(m2, m) < 1e-8
    assert float(np.max(np.abs(v2 - v))) < 1e-7


@pytest.mark.parametrize("res", [2, 3])
def test_mrgp_chain_matches_oracle_config1(ca, golden_dir, res):
    """BASELINE config 1: 1-D, N = 512, 3 (and 4) resolutions, divider 2."""
    g = np.load(os.path.join(golden_dir, "dense_oracle.npz"))
    x, y, xt = g["chain_x"], g["chain_y"], g["chain_xt"]
    n, ns = x.shape[0], xt.shape[0]
    kernels = [ca.RBFKernel(l=1.0 / 2 ** j, sf=1.0) for j in range(res + 1)]
    idx = ca.IndexSetUniform(n, res, 2)
    model = ca.MultiResolutionGaussianProcess([x, y], n_basis=30, index_set_obj=idx,
                                              basis_function_obj=None, spectral_density_obj=kernels,
                                              forced_independence=True)
    model.fit(5, None)
    idx_t = ca.IndexSetUniform(ns, res, 2)
    mean = model.get_predicted_mean(xt, idx_t)
    var = model.get_central_moment2(xt, idx_t)
    xn, _, mu, sd = oracle.normalize_inputs(x)
    specs = [oracle.DenseLayerSpec(1.0 / 2 ** j, 1.0, None) for j in range(res + 1)]
    omodel, f_bar = oracle.mrgp_fit(xn, y, oracle.index_bounds_uniform(n, res, 2), specs)
    omean, ovar = oracle.mrgp_predict(xn, omodel, specs, (xt - mu) / sd, oracle.index_bounds_uniform(ns, res, 2))
    assert mean.shape == (ns, 2) and var.shape == (ns,)
    assert _relerr(mean, omean) < 1e-5 and _relerr(var, ovar) < 1e-5      # north_star bar
    assert _relerr(mean, omean) < 1e-7 and _relerr(var, ovar) < 1e-6
    assert _relerr(model._f_bar_final.cpu().numpy(), f_bar) < 1e-7
    if res == 2:
        assert _relerr(mean, g["chain_mean"]) < 1e-7
        assert _relerr(var, g["chain_var"]) < 1e-6
        noise = np.array([[1.0 / v for v in model.get_stats[j].noise_mean] + [np.nan] * (4 - 2 ** j) for j in range(3)])
        np.testing.assert_allclose(noise, g["chain_noise"], rtol=1e-8)
    # index_set_obj=None: resolution 0 only (MRGP.py:726-755)
    m0 = model.get_predicted_mean(xt)
    blk = omodel[0][0]
    o0, _ = oracle.block_predict(xn, blk, (xt - mu) / sd, specs[0].ell, specs[0].sf2, False)
    assert _relerr(m0, o0 + blk["bias"]) < 1e-7
    ll = model.get_test_likelihood([xt, omean], idx_t)
    assert np.isfinite(ll)
    # latent function seen by layer 1 = layer-0 prediction at the training points (Stats.py:126-157)
    lat = model.get_stats[1].latent_f_mean
    assert len(lat) == 2 and lat[0].shape == (n // 2, 2)


def test_mrgp_2d_shared_bias_noise_and_errors(ca):
    rng = np.random.default_rng(8)
    n, ns = 600, 150
    x = rng.uniform(-2, 2, size=(n, 2))
    x = x[np.argsort(x[:, 0])]
    y = np.stack([np.sin(2 * x[:, 0]) * x[:, 1], np.cos(x[:, 0] + x[:, 1])], axis=1) + 0.05 * rng.normal(size=(n, 2))
    xt = rng.uniform(-2, 2, size=(ns, 2))
    xt = xt[np.argsort(xt[:, 0])]
    kernels = [ca.RBFKernel(l=1.0, sf=1.0), ca.RBFKernel(l=0.5, sf=0.5, noise=0.003)]
    idx = ca.IndexSetUniform(n, 1, 3)
    model = ca.MultiResolutionGaussianProcess([x, y], index_set_obj=idx, spectral_density_obj=kernels,
                                              bias_region_specific=False, noise_region_specific=False)
    model.fit()
    idx_t = ca.IndexSetUniform(ns, 1, 3)
    mean, var = model.get_predicted_mean_and_var(xt, idx_t)
    xn, _, mu, sd = oracle.normalize_inputs(x)
    specs = [oracle.DenseLayerSpec(1.0, 1.0, None), oracle.DenseLayerSpec(0.5, 0.5, 0.003)]
    omodel, _ = oracle.mrgp_fit(xn, y, oracle.index_bounds_uniform(n, 1, 3), specs, False, False)
    omean, ovar = oracle.mrgp_predict(xn, omodel, specs, (xt - mu) / sd, oracle.index_bounds_uniform(ns, 1, 3))
    assert _relerr(mean, omean) < 1e-7 and _relerr(var, ovar) < 1e-6
    with pytest.raises(ValueError):       # MRGP.py:758-760
        model.get_predicted_mean(xt, ca.IndexSetUniform(ns, 2, 3))
    with pytest.raises(ValueError):       # MRGP.py:762-764
        model.get_predicted_mean(xt, ca.IndexSetUniform(ns, 1, 2))
    # a non-PD block surfaces as LinAlgError, the reference's "not PD" convention (SanityCheck.py:59-65)
    xd = np.repeat(np.linspace(0, 1, 40)[:, None], 2, axis=0)
    yd = np.hstack([xd, xd])
    bad = ca.MultiResolutionGaussianProcess([xd, yd], index_set_obj=ca.IndexSetUniform(80, 0, 2),
                                            spectral_density_obj=ca.RBFKernel(l=1.0, sf=1.0, noise=0.0))
    # noise=0.0 is falsy-but-fixed: duplicated inputs make K singular
    with pytest.raises(np.linalg.LinAlgError):
        bad.fit()


def test_full_size_properties_n8192(ca):
    """BASELINE config 2 size (N = 8192, single block): properties that do not need an
    N^3 CPU reference -- K v = L (L^T v) (checksum of the factorisation), K alpha = r,
    and train-point prediction identity K_noiseless alpha = r - noise alpha."""
    dev = ca.device
    rng = np.random.default_rng(1234)
    n, q = 8192, 2
    x = np.sort(rng.uniform(-np.sqrt(3), np.sqrt(3), size=(n, 1)), axis=0)
    y = np.hstack([np.sin(3 * x + k) + 0.5 * np.sin(17 * x * x) for k in range(q)]) + 0.1 * rng.normal(size=(n, q))
    ell, sf2, noise = 0.1, 1.0, 0.01
    xd = dev.to_device(x, torch.float64, "cuda")
    kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=False)
    kfull = kbuf[:n, :n].clone()
    ws, info = dev.potrf(kbuf, n)
    assert int(info.item()) == 0
    lmat = torch.tril(kbuf[:n, :n])
    v = torch.from_numpy(rng.normal(size=(n, 3))).cuda()
    lhs = lmat @ (lmat.t() @ v)
    rhs = kfull @ v
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) < 1e-11
    alpha = dev.to_device(y, torch.float64, "cuda")
    dev.potrs(kbuf, n, ws, alpha)
    resid = kfull @ alpha - torch.from_numpy(y).cuda()
    assert float(resid.abs().max()) < 1e-8
    # spot-check rows of L against LAPACK on the leading 2048 x 2048 minor (CPU: seconds)
    lref, _ = oracle.potrf_lower(oracle.rbf_gram(x[:2048], None, ell, sf2, noise))
    assert _relerr(lmat[:2048, :2048].cpu().numpy(), lref) < 1e-9
    # predictive mean at the training inputs equals r - noise * alpha
    m = dev.predict_mean(xd, alpha, xd[:512], ell, sf2, None)
    want = (torch.from_numpy(y).cuda() - noise * alpha)[:512]
    assert float((m - want).abs().max()) < 1e-7
