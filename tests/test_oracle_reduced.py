"""The reduced-rank oracle (oracle/reduced.py) against the reference's own fitted models
(tests/golden/reference_model_*.npz, written by tests/golden/make_golden.py from the reference
run in this container).  This is what pins parity for the 8f-rank-2 path.  CPU only."""
import os

import numpy as np
import pytest

from oracle import index_bounds_uniform
from oracle.reduced import (ReducedRankModel, bingham_saddle, laplace_basis, matern_spectral, nearest_pd, is_pd)

RTOL = 1e-11        # same algorithm, different summation order: rounding only


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _fit(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "reference_model_%s.npz" % tag))
    res = int(z["resolution"])
    bounds = index_bounds_uniform(z["x"].shape[0], res, 2)
    model = ReducedRankModel(z["x"], z["y"], bounds, int(z["n_basis"]),
                             forced_independence=bool(z["forced_independence"]))
    model.fit(5)
    return z, res, model


@pytest.mark.parametrize("tag", ["fi_r2", "fi_r3", "ci_r2"])
def test_fitted_state_matches_reference(golden_dir, tag):
    z, res, model = _fit(golden_dir, tag)
    for j, layer in enumerate(model.blocks):
        for l, blk in enumerate(layer):
            key = "_%d_%d" % (j, l)
            np.testing.assert_allclose(blk.interval, z["interval" + key], rtol=0, atol=0)
            assert _rel(blk.eau, z["scale_axis_mean" + key]) < RTOL
            assert _rel(blk.bias_mean, z["bias_mean" + key]) < RTOL
            assert _rel(blk.fbar, z["latent_f_mean" + key]) < RTOL
            assert _rel(blk.fvar, z["latent_f_var" + key]) < RTOL
            mean, var = blk.contribution()
            assert _rel(mean, z["train_pred" + key]) < RTOL
            assert _rel(var, z["train_predvar" + key]) < RTOL


@pytest.mark.parametrize("tag", ["fi_r2", "fi_r3", "ci_r2"])
def test_predictions_match_reference(golden_dir, tag):
    z, res, model = _fit(golden_dir, tag)
    tb = index_bounds_uniform(z["xt"].shape[0], res, 2)
    assert _rel(model.predict_mean(z["xt"]), z["pred_mean_global"]) < RTOL
    assert _rel(model.predict_var(z["xt"]), z["pred_var_global"]) < RTOL
    assert _rel(model.predict_mean(z["xt"], tb), z["pred_mean_index"]) < RTOL
    assert _rel(model.predict_var(z["xt"], tb), z["pred_var_index"]) < RTOL


@pytest.mark.parametrize("tag", ["fi_r1_2d", "ci_r1_2d", "fi_r2_snr", "ci_r2_shared_nb", "ci_r2_shared_n", "ci_r2_shared_b",
                                 "fi_r2_shared_nb", "ci_r2_bi", "ci_r1_bi_2d"])
def test_flag_variants_match_reference(golden_dir, tag):
    """2-D inputs, SNR-initialised noise, shared noise / bias, adaptive basis intervals."""
    z = np.load(os.path.join(golden_dir, "reference_model_%s.npz" % tag))
    res = int(z["resolution"])
    kw = {}
    for name in ("noise_region_specific", "bias_region_specific"):
        if "kw_" + name in z.files:
            kw[name] = bool(z["kw_" + name])
    for name in ("snr_ratio", "interval_factor"):
        if "kw_" + name in z.files:
            kw[name] = float(z["kw_" + name])
    kw["adaptive_basis_intervals"] = bool(z["adaptive_basis_intervals"])
    model = ReducedRankModel(z["x"], z["y"], index_bounds_uniform(z["x"].shape[0], res, 2), int(z["n_basis"]),
                             forced_independence=bool(z["forced_independence"]), **kw)
    model.fit(int(z["n_iter"]))
    tol = 1e-9 if "_bi" in tag else RTOL
    for j, layer in enumerate(model.blocks):
        for l, blk in enumerate(layer):
            key = "_%d_%d" % (j, l)
            assert _rel(blk.interval, z["interval" + key]) < tol
            assert _rel(blk.eau, z["scale_axis_mean" + key]) < tol
            assert _rel(blk.mom2, z["scale_moment2" + key]) < tol
    if not model.fi:
        assert _rel(model.omega, z["shared_omega"]) < 1e-7
        assert _rel(model.sh_ard_mean, z["shared_ard_mean"]) < tol
    tb = index_bounds_uniform(z["xt"].shape[0], res, 2)
    assert _rel(model.predict_mean(z["xt"]), z["pred_mean_global"]) < tol
    assert _rel(model.predict_var(z["xt"]), z["pred_var_global"]) < tol
    assert _rel(model.predict_mean(z["xt"], tb), z["pred_mean_index"]) < tol
    assert _rel(model.predict_var(z["xt"], tb), z["pred_var_index"]) < tol


def test_lower_bound_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "reference_model_ci_r2_elbo.npz"))
    model = ReducedRankModel(z["x"], z["y"], index_bounds_uniform(512, 2, 2), 30, forced_independence=False)
    n_iter = int(z["n_iter"])
    model.fit(n_iter, 1e-12, min_iter=n_iter)
    got = np.array(model.lower_bound_layer)
    assert np.max(np.abs(got - z["lower_bound_layer"]) / np.abs(z["lower_bound_layer"])) < 1e-10
    assert np.max(np.abs(np.array(model.lower_bound) - z["lower_bound"]) / np.abs(z["lower_bound"])) < 1e-10


def test_kernel_objects_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "kernel_objects.npz"))
    phi, lam = laplace_basis(g["lap_x2"], np.array([2.0, 1.7]), 7)
    for bid in (1, 2, 7):
        assert _rel(phi[:, bid - 1], g["lap2_f_%d" % bid]) < 1e-13
        assert _rel(lam[bid - 1], g["lap2_l_%d" % bid]) < 1e-13
    phi, lam = laplace_basis(g["lap_x1"], np.max(np.abs(g["lap_x1"]), axis=0), 7)   # default interval
    for bid in (1, 2, 7):
        assert _rel(phi[:, bid - 1], g["lap1_f_%d" % bid]) < 1e-13
    for nu in (0.5, 1.0, 1.5, 2.5):
        tag = str(nu).replace('.', 'p')
        assert _rel(matern_spectral(g["mat_s"], nu, 0.7, 1.3), g["mat_s_" + tag]) < 1e-13


def test_bingham_saddle_properties():
    # uniform case: all eigenvalues equal -> rho = 1/p each (trace of E[uu^T] is 1)
    for p in (2, 3, 5):
        log_c, rho = bingham_saddle(np.zeros(p))
        np.testing.assert_allclose(rho, np.full(p, 1.0 / p), rtol=1e-9)
        # shifting all eigenvalues by c adds c to log C and leaves rho alone
        log_c2, rho2 = bingham_saddle(np.full(p, 3.5))
        np.testing.assert_allclose(log_c2 - log_c, 3.5, rtol=1e-9)
        np.testing.assert_allclose(rho2, rho, rtol=1e-9)
    # rho is the gradient of log C: finite differences
    kappa = np.array([4.0, 1.0, 0.2])
    _, rho = bingham_saddle(kappa)
    for d in range(3):
        e = np.zeros(3)
        e[d] = 1e-5
        fd = (bingham_saddle(kappa + e)[0] - bingham_saddle(kappa - e)[0]) / 2e-5
        assert abs(fd - rho[d]) < 1e-6
    assert abs(np.sum(rho) - 1.0) < 1e-9
    # strong concentration: E[uu^T] collapses onto the leading axis
    _, rho = bingham_saddle(np.array([500.0, 0.0]))
    assert rho[0] > 0.99


def test_nearest_pd_repairs_rank_one():
    v = np.array([1.0, 2.0])
    mat = np.outer(v, v) * 1e-3
    assert not is_pd(mat - 1e-12 * np.eye(2))
    fixed = nearest_pd(mat - 1e-12 * np.eye(2))
    assert is_pd(fixed)
    assert np.max(np.abs(fixed - mat)) < 1e-10


def test_matern_spectral_is_positive_and_decreasing():
    s = np.linspace(0.0, 20.0, 50)
    w = matern_spectral(s, 1.0, 1.0, 1.0)
    assert np.all(w > 0) and np.all(np.diff(w) < 0)
