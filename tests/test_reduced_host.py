"""Host logic of the reduced-rank model (cimrgp_amd/ReducedRank.py) on CPU.

The three GPU entry points are replaced by NumPy stand-ins *inside this test only* (the product
has no CPU path), so that the factor updates, the sweep order and the prediction bookkeeping
can be checked against the reference's fitted models without a GPU.  The GPU tests
(test_gpu_reduced.py) run the same comparisons through the HIP kernels."""
import os

import numpy as np
import pytest
import torch

from cimrgp_amd import BasisInterval, IndexSetUniform, LaplacianEigenpairs, MaternKernel
from cimrgp_amd import device as dev
from cimrgp_amd import ReducedRank as rr
from cimrgp_amd.MRGP import MultiResolutionGaussianProcess

RTOL = 1e-9


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture
def numpy_device(monkeypatch):
    def phi_of(x, interval, n_basis):
        xn = x.double().numpy()
        half = np.asarray(interval, dtype=np.float64)[None, :]
        cols = [np.prod(np.sin(np.pi * (i + 1) * (xn + half) / (2 * half)) / np.sqrt(half), axis=1) for i in range(n_basis)]
        return np.stack(cols, axis=1)

    def laplace_basis(x, interval, n_basis, out=None):
        return torch.as_tensor(phi_of(x, interval, n_basis)).to(x.dtype)

    def basis_moments(x, interval, n_basis, y, fbar, fvar, eau):
        p = phi_of(x, interval, n_basis)
        r0 = y.double().numpy() - (0 if fbar is None else fbar.double().numpy()) - p @ np.asarray(eau).T
        m, q = p.shape[1], r0.shape[1]
        rec = np.concatenate([(p.T @ r0).ravel(), p.sum(0), (p * p).sum(0), r0.sum(0), [np.sum(r0 * r0)],
                              [0.0 if fvar is None else float(fvar.double().sum())]])
        return dev.BlockMoments(rec, m, q, p.shape[0])

    def basis_apply(x, interval, n_basis, eau, bias=None, c2=None, bias_var=0.0, mean=None, var=None, accumulate=False):
        p = phi_of(x, interval, n_basis)
        if mean is not None:
            mu = p @ np.asarray(eau).T + (0 if bias is None else np.asarray(bias))
            t = torch.as_tensor(mu).to(mean.dtype)
            mean.copy_(mean + t if accumulate else t)
        if var is not None:
            v = bias_var + (p * p) @ (np.zeros(p.shape[1]) if c2 is None else np.asarray(c2))
            t = torch.as_tensor(v).to(var.dtype)
            var.copy_(var + t if accumulate else t)

    def residual(y, fbar, bias, out=None):
        return y - fbar - bias

    monkeypatch.setattr(dev, "residual", residual)
    monkeypatch.setattr(dev, "require_gpu", lambda device=None: torch.device("cpu"))
    monkeypatch.setattr(dev, "laplace_basis", laplace_basis)
    monkeypatch.setattr(dev, "basis_moments", basis_moments)
    monkeypatch.setattr(dev, "basis_apply", basis_apply)


def _kwargs_of(z):
    """Constructor keywords recorded in a golden file (tests/golden/make_golden_reduced.py)."""
    kw = {}
    for name in ("noise_region_specific", "bias_region_specific"):
        if "kw_" + name in z:
            kw[name] = bool(z["kw_" + name])
    for name in ("snr_ratio", "interval_factor"):
        if "kw_" + name in z:
            kw[name] = float(z["kw_" + name])
    if "adaptive_basis_intervals" in z and bool(z["adaptive_basis_intervals"]):
        kw["basis_interval_obj"] = BasisInterval(opt_interval_factor=(1, 1.2))
    return kw


def _build(z, **kw):
    n = z["x"].shape[0]
    idx = IndexSetUniform(n, int(z["resolution"]), 2)
    return MultiResolutionGaussianProcess(train_xy=[z["x"], z["y"]], n_basis=int(z["n_basis"]), index_set_obj=idx,
                                          basis_function_obj=LaplacianEigenpairs(),
                                          spectral_density_obj=MaternKernel(nu=1, l=1, sf=1), adaptive_inputs=False,
                                          forced_independence=bool(z["forced_independence"]), **kw)


@pytest.mark.parametrize("tag", ["fi_r2", "fi_r3", "ci_r2"])
def test_host_model_matches_reference(golden_dir, numpy_device, tag):
    z = np.load(os.path.join(golden_dir, "reference_model_%s.npz" % tag))
    model = _build(z)
    assert isinstance(model, rr.ReducedRankMRGP) and isinstance(model, MultiResolutionGaussianProcess)
    model.fit(5, None)
    for j in range(model.n_layers):
        f_mean, f_var = model.latent_functions(j)
        for l in range(model.n_regions[j]):
            key = "_%d_%d" % (j, l)
            st = model.stats_obj[j]
            np.testing.assert_array_equal(model.train_basis_intervals[j][l], z["interval" + key])
            assert _rel(st.scale_axis_mean[l], z["scale_axis_mean" + key]) < RTOL
            assert _rel(st.bias_mean[l], z["bias_mean" + key]) < RTOL
            assert _rel(f_mean[l], z["latent_f_mean" + key]) < RTOL
            assert _rel(f_var[l], z["latent_f_var" + key]) < RTOL
    idx_t = IndexSetUniform(z["xt"].shape[0], int(z["resolution"]), 2)
    assert _rel(model.get_predicted_mean(z["xt"]), z["pred_mean_global"]) < RTOL
    assert _rel(model.get_central_moment2(z["xt"]), z["pred_var_global"]) < RTOL
    assert _rel(model.get_predicted_mean(z["xt"], idx_t), z["pred_mean_index"]) < RTOL
    assert _rel(model.get_central_moment2(z["xt"], idx_t), z["pred_var_index"]) < RTOL
    contrib = model.get_basis_contributions()
    assert abs(np.sum(contrib[0][0]) - 1.0) < 1e-12


@pytest.mark.parametrize("tag", ["fi_r1_2d", "ci_r1_2d", "fi_r2_snr", "ci_r2_shared_nb", "ci_r2_shared_n", "ci_r2_shared_b",
                                 "fi_r2_shared_nb", "ci_r2_bi", "ci_r1_bi_2d"])
def test_host_model_flag_variants_match_reference(golden_dir, numpy_device, tag):
    """2-D inputs, SNR-initialised noise, shared noise / bias, adaptive basis intervals."""
    z = dict(np.load(os.path.join(golden_dir, "reference_model_%s.npz" % tag)))
    model = _build(z, **_kwargs_of(z))
    model.fit(int(z["n_iter"]), None)
    tol = 1e-8 if "_bi" in tag else RTOL            # the interval minimiser stops at xtol = 1e-5
    for j in range(model.n_layers):
        st = model.stats_obj[j]
        for l in range(model.n_regions[j]):
            key = "_%d_%d" % (j, l)
            assert _rel(model.train_basis_intervals[j][l], z["interval" + key]) < tol
            assert _rel(st.scale_axis_mean[l], z["scale_axis_mean" + key]) < tol
            assert _rel(st.scale_moment2[l], z["scale_moment2" + key]) < tol
            if model.bias_region_specific:
                assert _rel(st.bias_mean[l], z["bias_mean" + key]) < tol
            if model.noise_region_specific:
                assert abs(st.noise_mean[l] - z["noise_mean" + key]) < tol * abs(z["noise_mean" + key])
        if not model.noise_region_specific:
            assert abs(st.noise_mean - z["noise_mean_%d" % j]) < tol * abs(z["noise_mean_%d" % j])
        if not model.bias_region_specific:
            assert _rel(st.bias_mean, z["bias_mean_%d" % j]) < tol
    if not model.forced_independence:
        assert _rel(model.shared_stats.omega, z["shared_omega"]) < 1e-7
        assert _rel(model.shared_stats.ard_mean, z["shared_ard_mean"]) < tol
    idx_t = IndexSetUniform(z["xt"].shape[0], int(z["resolution"]), 2)
    assert _rel(model.get_predicted_mean(z["xt"]), z["pred_mean_global"]) < tol
    assert _rel(model.get_central_moment2(z["xt"]), z["pred_var_global"]) < tol
    assert _rel(model.get_predicted_mean(z["xt"], idx_t), z["pred_mean_index"]) < tol
    assert _rel(model.get_central_moment2(z["xt"], idx_t), z["pred_var_index"]) < tol


def test_host_lower_bound_matches_reference(golden_dir, numpy_device):
    """fit(n_iter, tol): the recorded bound per sweep and per layer (MRGP.py:374-398,414-571)."""
    z = dict(np.load(os.path.join(golden_dir, "reference_model_ci_r2_elbo.npz")))
    model = _build(z)
    n_iter = int(z["n_iter"])
    model.fit(n_iter, 1e-12, min_iter=n_iter)
    assert len(model.lower_bound) == n_iter
    got = np.array(model.lower_bound_layer)
    assert got.shape == z["lower_bound_layer"].shape
    assert np.max(np.abs(got - z["lower_bound_layer"]) / np.abs(z["lower_bound_layer"])) < 1e-9
    assert np.max(np.abs(np.array(model.lower_bound) - z["lower_bound"]) / np.abs(z["lower_bound"])) < 1e-9
    # early stop: a loose tolerance ends the loop right after min_iter
    model2 = _build(z)
    model2.fit(n_iter, 1e60, min_iter=2)
    assert len(model2.lower_bound) == 3


def test_bingham_normaliser_matches_brent(golden_dir):
    """The vectorised Newton solve against scipy's Brent on the same equation."""
    from scipy.optimize import brentq
    rng = np.random.default_rng(3)
    for p in (2, 3, 6):
        kappa = np.sort(rng.gamma(1.0, 5.0, size=(40, p)), axis=1)[:, ::-1]
        log_c, rho = rr.bingham_normaliser(kappa)
        assert np.allclose(rho.sum(axis=1), 1.0, rtol=0, atol=1e-10)
        for row in range(0, 40, 7):
            lam = -kappa[row]
            shift = 0.1 - lam.min()
            lam = lam + shift
            t = brentq(lambda s: 0.5 * np.sum(1.0 / (lam - s)) - 1.0, 0.1 - p, -0.4, xtol=1e-15, rtol=1e-15)
            k2 = 0.5 * np.sum((lam - t) ** -2.0)
            ref = 0.5 * (np.log(2) + (p - 1) * np.log(np.pi) - np.log(k2) - np.sum(np.log(lam - t))) - t + shift
            assert abs(log_c[row] - ref) < 1e-10 * max(1.0, abs(ref))


def test_unsupported_configurations_raise(numpy_device, golden_dir):
    z = np.load(os.path.join(golden_dir, "reference_model_fi_r2.npz"))
    with pytest.raises(TypeError):
        _build(z, noise_region_specific=None)
    with pytest.raises(TypeError):
        _build(dict(z, forced_independence=np.bool_(False)), axis_resolution_specific=True)
    with pytest.raises(ValueError):
        MultiResolutionGaussianProcess(train_xy=[z["x"], z["y"][:, :1]], n_basis=5,
                                       index_set_obj=IndexSetUniform(z["x"].shape[0], 1, 2),
                                       basis_function_obj=LaplacianEigenpairs(), spectral_density_obj=MaternKernel())


def test_nearest_pd_and_axis_factors():
    ax = rr._AxisFactors(4, 2)
    v = np.array([[1.0, 0.5], [0.0, 2.0], [3.0, 3.0], [1e-30, 0.0]])
    cand = np.einsum('ia,ib->iab', v, v)
    ax.set_axes(cand)
    cov = ax.cov()
    for i in range(4):
        assert abs(np.trace(cov[i]) - 1.0) < 1e-9             # E|u|^2 = 1 on the sphere
        assert rr.isPD(ax.axis_bingham_b[i])
    # a strongly concentrated factor points along its evidence vector
    lead = ax.axis_bingham_axes[2][:, 0]
    assert abs(abs(lead @ (v[2] / np.linalg.norm(v[2]))) - 1.0) < 1e-9


def test_sinkhorn_omega_agrees_with_minpack(monkeypatch):
    """the converged Sinkhorn scaling is doubly stochastic to rounding and sits
    within MINPACK's stopping tolerance of the reference's answer."""
    rng = np.random.default_rng(11)
    m, q = 12, 3
    prior = rr.SharedPrior(m, q, 1.0)
    v = rng.normal(size=(m, q))
    prior.set_axes(np.einsum('ia,ib->iab', v, v) + 0.1 * np.eye(q))
    prior.ard_gamma_shape = rng.uniform(0.5, 2.0, size=m)
    prior.ard_gamma_scale = rng.uniform(0.5, 2.0, size=m)
    post = rr.SharedPosterior(prior)
    w = rng.normal(size=(m, q))
    post.set_axes(np.einsum('ia,ib->iab', w, w) + 0.2 * np.eye(q))
    post.ard_gamma_shape = rng.uniform(1.0, 3.0, size=m)
    post.ard_gamma_scale = rng.uniform(1.0, 3.0, size=m)
    stats = rr.SharedStats(post)
    stats.update_axis(post)
    stats.update_omega(prior)
    ref = stats.omega.copy()
    monkeypatch.setattr(rr, "OMEGA_SOLVER", "sinkhorn")
    stats.update_omega(prior)
    assert np.max(np.abs(stats.omega.sum(0) - 1.0)) < 1e-12 and np.max(np.abs(stats.omega.sum(1) - 1.0)) < 1e-12
    assert np.max(np.abs(ref.sum(0) - 1.0)) < 1e-6
    assert np.max(np.abs(stats.omega - ref)) < 1e-6 * np.max(ref)


def test_random_region_index_sets(numpy_device):
    """IndexSetUniform(n_regions=[...]): random contiguous regions (IndexSetGenerator.py:67-92),
    test regions given by ``number_of_regions``; against the pinned oracle on the same bounds."""
    from oracle.reduced import ReducedRankModel
    rng = np.random.default_rng(2)
    n, ns = 300, 120
    x = np.sort(rng.uniform(1, 3, size=(n, 1)), axis=0)
    y = np.hstack([np.sin(3 * x), np.cos(2 * x)]) + 0.05 * rng.normal(size=(n, 2))
    xs = np.sort(rng.uniform(1, 3, size=(ns, 1)), axis=0)
    np.random.seed(5)
    idx = IndexSetUniform(n, 1, None, n_regions=[1, 3])
    idx_t = IndexSetUniform(ns, 1, None, n_regions=[1, 3])
    assert [len(b) for b in idx.bounds] == [1, 3] and idx.bounds[1][0][0] == 0 and idx.bounds[1][-1][1] == n
    model = MultiResolutionGaussianProcess(train_xy=[x, y], n_basis=10, index_set_obj=idx,
                                           basis_function_obj=LaplacianEigenpairs(),
                                           spectral_density_obj=MaternKernel(nu=1, l=1, sf=1), forced_independence=True)
    model.fit(3, None)
    got = model.get_predicted_mean(xs, idx_t, number_of_regions=[1, 3])
    omodel = ReducedRankModel(x, y, idx.bounds, 10, forced_independence=True)
    omodel.fit(3)
    assert _rel(got, omodel.predict_mean(xs, idx_t.bounds)) < 1e-9
    with pytest.raises(ValueError):
        model.get_predicted_mean(xs, idx_t, number_of_regions=[1, 2])
