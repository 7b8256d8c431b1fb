"""The oracle's structural restatements against outputs captured from the
reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import oracle


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("n,r,d", [(512, 3, 2), (160, 7, 2), (100000, 5, 2), (513, 2, 3), (37, 0, 2)])
def test_index_bounds_match_reference(golden_dir, n, r, d):
    g = _load(golden_dir, "structure_index_sets.npz")
    bounds = oracle.index_bounds_uniform(n, r, d)
    assert len(bounds) == r + 1
    for m, b in enumerate(bounds):
        np.testing.assert_array_equal(b, g["uniform_%d_%d_%d_layer%d" % (n, r, d, m)])


def test_index_set_materialisation_is_reference_type():
    idx = oracle.index_set_from_bounds(oracle.index_bounds_uniform(10, 1, 3))
    assert idx == [[list(range(10))], [[0, 1, 2], [3, 4, 5], [6, 7, 8, 9]]]


def test_resolution_too_large_raises():
    with pytest.raises(ValueError):
        oracle.index_bounds_uniform(5, 3, 2)


def test_normalize_inputs_matches_reference(golden_dir):
    g = _load(golden_dir, "structure_normalise.npz")
    for tag, x, fx in [("a", g["x1"], None), ("b", g["x2"], None), ("c", g["x1"], g["fullx"])]:
        xn, fxn, mu, sd = oracle.normalize_inputs(x, fx)
        np.testing.assert_array_equal(xn, g["norm_%s_x" % tag])
        np.testing.assert_array_equal(mu, g["norm_%s_mean" % tag])
        np.testing.assert_array_equal(sd, g["norm_%s_std" % tag])
        if fx is not None:
            np.testing.assert_array_equal(fxn, g["norm_%s_full" % tag])
    # zero-variance column is left unscaled (std -> 1)
    assert g["norm_b_std"][1] == 1.0


def test_gather_is_a_slice(golden_dir):
    g = _load(golden_dir, "structure_normalise.npz")
    b = oracle.index_bounds_uniform(97, 2, 2)
    a0, a1 = b[2][3]
    np.testing.assert_array_equal(g["x1"][a0:a1], g["gather_2_3"])
    a0, a1 = b[1][0]
    np.testing.assert_array_equal(g["x1"][a0:a1], g["gather_1_0"])


def test_plugin_zscore_matches_reference(golden_dir):
    g = _load(golden_dir, "structure_normalise.npz")
    st = oracle.zscore_fit(g["x1"], g["plug_y"])
    xz, yz = oracle.zscore_apply(st, inputs=g["x1"], labels=g["plug_y"])
    np.testing.assert_array_equal(xz, g["plug_xz"])
    np.testing.assert_array_equal(yz, g["plug_yz"])
    np.testing.assert_array_equal(oracle.zscore_apply(st, inputs=g["plug_xt"]), g["plug_xtz"])
    np.testing.assert_array_equal(oracle.zscore_apply(st, inverse_labels=yz[:11] * 0.5 + 0.25), g["plug_back"])
    assert float(yz.var()) * oracle.mrgp.NOISE_FRACTION == float(g["plug_noise_init"])


@pytest.mark.parametrize("tag", ["fi_r2", "fi_r3", "ci_r2"])
def test_residual_scatter_and_layer_sum_match_reference(golden_dir, tag):
    """Stats.update_latent_functions and the sum over layers, fed with the
    reference's own per-block predictions."""
    g = _load(golden_dir, "reference_model_%s.npz" % tag)
    res = int(g["resolution"])
    n = g["x"].shape[0]
    bounds = oracle.index_bounds_uniform(n, res, 2)
    train = [[g["train_pred_%d_%d" % (j, l)] for l in range(len(bounds[j]))] for j in range(res + 1)]
    trainv = [[g["train_predvar_%d_%d" % (j, l)] for l in range(len(bounds[j]))] for j in range(res + 1)]
    for j in range(1, res + 1):
        lat = oracle.latent_from_coarser(train, bounds, j)
        latv = oracle.latent_from_coarser(trainv, bounds, j)
        for l in range(len(bounds[j])):
            np.testing.assert_allclose(lat[l], g["latent_f_mean_%d_%d" % (j, l)], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(latv[l], g["latent_f_var_%d_%d" % (j, l)], rtol=1e-12, atol=1e-13)
    ns = g["xt"].shape[0]
    tb = oracle.index_bounds_uniform(ns, res, 2)
    test = [[g["test_pred_%d_%d" % (j, l)] for l in range(len(tb[j]))] for j in range(res + 1)]
    np.testing.assert_allclose(oracle.sum_over_layers(test), g["pred_mean_index"], rtol=1e-11, atol=1e-12)
    # index_set_obj=None: layer 0 / region 0 only (MRGP.py:726-755)
    assert g["pred_mean_global"].shape == (ns, 2)


def test_dense_oracle_is_self_consistent(golden_dir):
    """K alpha = r, variance in [0, sf2], fixtures reproduce."""
    g = _load(golden_dir, "dense_oracle.npz")
    for tag in ["n64_d1", "n257_d2", "n512_d1"]:
        x, y, xs = g[tag + "_x"], g[tag + "_y"], g[tag + "_xs"]
        ell, sf2, noise = g[tag + "_hyp"]
        fit = oracle.block_fit(x, y, ell, sf2, noise)
        k = oracle.rbf_gram(x, None, ell, sf2, noise)
        np.testing.assert_allclose(k @ fit["alpha"], y, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(fit["L"] @ fit["L"].T, k, rtol=1e-12, atol=1e-12)
        mean, var = oracle.block_predict(x, fit, xs, ell, sf2, True)
        np.testing.assert_allclose(mean, g[tag + "_mean"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(var, g[tag + "_var"], rtol=1e-8, atol=1e-12)
        assert np.all(var > -1e-10) and np.all(var <= sf2 + 1e-12)


def test_non_pd_reports_lapack_info():
    x = np.array([[0.0], [0.0], [1.0]])         # duplicated point, no noise
    _, info = oracle.potrf_lower(oracle.rbf_gram(x, None, 1.0, 1.0, 0.0))
    assert info == 2
    with pytest.raises(np.linalg.LinAlgError):
        oracle.block_fit(x, np.zeros((3, 2)), 1.0, 1.0, 0.0)


def test_dense_oracle_agrees_with_scikit_learn(golden_dir):
    """Independent cross-check of the PARITY-UNPINNED dense oracle (GPy, the reference's backend at
    RegressionInput.py:58-67, is absent): scikit-learn's exact GP with fixed hyper-parameters on the
    same arrays (fixtures written by tests/golden/make_sklearn_golden.py, n in {64, 257, 512},
    d in {1, 2}, two outputs sharing one kernel).  Two independent implementations of Rasmussen &
    Williams Alg. 2.1 agree to rounding; the reference's own numbers stay unavailable."""
    g = np.load(os.path.join(golden_dir, "sklearn_gp.npz"))
    for tag in [str(t) for t in g["cases"]]:
        x, y, xs = g[tag + "_x"], g[tag + "_y"], g[tag + "_xs"]
        ell, sf2, noise = g[tag + "_hyp"]
        fit = oracle.block_fit(x, y, ell, sf2, noise)
        mean, var = oracle.block_predict(x, fit, xs, ell, sf2, True)
        np.testing.assert_allclose(mean, g[tag + "_mean"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(var, g[tag + "_var"], rtol=1e-6, atol=1e-9)
        # log marginal likelihood: sklearn sums the outputs' likelihoods, as oracle.gp_lml_and_grad does
        lml, _ = oracle.gp_lml_and_grad(x, y, ell, sf2, noise)
        np.testing.assert_allclose(lml, float(g[tag + "_lml"]), rtol=1e-9)
