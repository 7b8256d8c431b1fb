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
    model = ca.GP_RBF(optimize=False)
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


def test_two_panel_far_updates_ragged_n9001(ca):
    """Just past the pairing threshold with a ragged last panel (9001 = 35 panels + 41 columns)."""
    dev = ca.device
    rng = np.random.default_rng(7)
    n = 9001
    x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
    xd = dev.to_device(x, torch.float64, "cuda")
    kbuf = dev.rbf_gram(xd, 0.1, 1.0, 0.01, lower_only=False)
    kfull = kbuf[:n, :n].clone()
    _, info = dev.potrf(kbuf, n)
    assert int(info.item()) == 0
    lmat = torch.tril(kbuf[:n, :n])
    v = torch.from_numpy(rng.normal(size=(n, 2))).cuda()
    rhs = kfull @ v
    assert float((lmat @ (lmat.t() @ v) - rhs).abs().max() / rhs.abs().max()) < 1e-11


def test_two_panel_far_updates_n10240(ca):
    """N = 10240: large enough for the factorisation to update the far part of the trailing matrix
    once per two panels (K = 512); checksum K v = L (L^T v), LAPACK on the leading minor, and the
    carried-rows variant against the plain one."""
    dev = ca.device
    rng = np.random.default_rng(99)
    n = 10240
    x = np.sort(rng.uniform(-2.0, 2.0, size=(n, 1)), axis=0)
    ell, sf2, noise = 0.1, 1.0, 0.01
    xd = dev.to_device(x, torch.float64, "cuda")
    kbuf = dev.rbf_gram(xd, ell, sf2, noise, lower_only=False)
    kfull = kbuf[:n, :n].clone()
    ws, info = dev.potrf(kbuf, n)
    assert int(info.item()) == 0
    lmat = torch.tril(kbuf[:n, :n])
    v = torch.from_numpy(rng.normal(size=(n, 3))).cuda()
    rhs = kfull @ v
    assert float((lmat @ (lmat.t() @ v) - rhs).abs().max() / rhs.abs().max()) < 1e-11
    lref, _ = oracle.potrf_lower(oracle.rbf_gram(x[:1536], None, ell, sf2, noise))
    assert _relerr(lmat[:1536, :1536].cpu().numpy(), lref) < 1e-9
    # the last rows see every pair: compare the bottom-right corner with a second, rows-carrying run
    k2 = dev.rbf_gram(xd, ell, sf2, noise, lower_only=True)
    w = dev.alloc_matrix(64, n, torch.float64, "cuda")
    w[:64, :n] = torch.from_numpy(rng.normal(size=(64, n))).cuda()
    w0 = w[:64, :n].clone()
    _, info2 = dev.potrf_rows(k2, n, w, 64)
    assert int(info2.item()) == 0
    assert float((torch.tril(k2[:n, :n]) - lmat).abs().max()) < 1e-12
    back = w[:64, :n] @ lmat.t()                    # (B L^-T) L^T = B
    assert float((back - w0).abs().max() / w0.abs().max()) < 1e-9


def _two_rank_worker(rank, world, port, out_dir, shared=False):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as td
    import cimrgp_amd as ca
    # two ranks share the one GPU of the box: gloo carries the (device) tensors; on a multi-GPU
    # node the same code runs with backend "nccl" (RCCL) and one GPU per rank
    torch.cuda.set_device(0)
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rng = np.random.default_rng(21)
    n, ns, res = 640, 200, 3
    x = np.sort(rng.uniform(-2, 2, size=(n, 1)), axis=0)
    y = np.hstack([np.sin(3 * x), np.cos(5 * x) * x]) + 0.1 * rng.normal(size=(n, 2))
    xs = np.sort(rng.uniform(-2, 2, size=(ns, 1)), axis=0)
    kernels = [ca.RBFKernel(l=1.0 / 2 ** j, sf=1.0) for j in range(res + 1)]
    model = ca.MultiResolutionGaussianProcess([x, y], index_set_obj=ca.IndexSetUniform(n, res, 2),
                                              spectral_density_obj=kernels, bias_region_specific=not shared,
                                              noise_region_specific=not shared)
    assert (model.rank, model.world_size) == (rank, world)
    owned = [len(model._owned(j)) for j in range(res + 1)]
    model.fit()
    mean, var = model.get_predicted_mean_and_var(xs, ca.IndexSetUniform(ns, res, 2))
    # the fine layers' equal-sized blocks must have gone through the batched fit on every rank (a view
    # into a shared buffer at an odd offset used to disable it on the local layers)
    batched = [len(p.batches) for p in model.posterior_obj]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mean=mean, var=var, owned=np.array(owned),
             f_bar=model._f_bar_final.cpu().numpy(), x=x, y=y, xs=xs, local_from=model._local_from, batched=np.array(batched))
    td.destroy_process_group()


@pytest.mark.parametrize("shared", [False, True])
def test_two_rank_sharded_model_on_gpu(ca, tmp_path, shared):
    """N > 1 path end to end on real kernels: blocks sharded over 2 ranks, per-layer residual
    all-reduce, one fused [mean | var] reduce; every rank must hold the single-process result.
    shared: bias and noise shared by the regions of a layer (bias_region_specific = noise_region_specific = False,
    MRGP.py:27-28) -- statistics over the WHOLE latent function, which nested ownership leaves valid on a rank's own
    ranges only: every layer must then be exchanged (round 5)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), shared), nprocs=2, join=True)
    g0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    g1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    x, y, xs = g0["x"], g0["y"], g0["xs"]
    n, ns, res = x.shape[0], xs.shape[0], 3
    xn, _, mu, sd = oracle.normalize_inputs(x)
    specs = [oracle.DenseLayerSpec(1.0 / 2 ** j, 1.0, None) for j in range(res + 1)]
    omodel, f_bar = oracle.mrgp_fit(xn, y, oracle.index_bounds_uniform(n, res, 2), specs, not shared, not shared)
    omean, ovar = oracle.mrgp_predict(xn, omodel, specs, (xs - mu) / sd, oracle.index_bounds_uniform(ns, res, 2))
    for g in (g0, g1):
        assert _relerr(g["mean"], omean) < 1e-7
        assert _relerr(g["var"], ovar) < 1e-6
        assert _relerr(g["f_bar"], f_bar) < 1e-7
        # nested ownership: layers 1.. are local (2 ranks: the anchor is layer 1) unless the statistics are shared
        assert int(g["local_from"]) == (4 if shared else 1)
        assert g["batched"].tolist()[2:] == [1, 1]          # 2 and 4 equal blocks per rank: one batched fit each
    # layer 0 has one block (rank 0), finer layers are split evenly
    assert g0["owned"].tolist() == [1, 1, 2, 4] and g1["owned"].tolist() == [0, 1, 2, 4]


@pytest.mark.parametrize("n,d", [(64, 1), (300, 2), (777, 1)])
def test_log_marginal_likelihood_and_gradient(ca, n, d):
    """The objective of `model.optimize()` (RegressionInput.py:63) and its gradient on the GPU."""
    rng = np.random.default_rng(n)
    x = rng.uniform(-2, 2, size=(n, d))
    y = np.stack([np.sin(2 * x[:, 0]) + x[:, -1], np.cos(x[:, 0] * x[:, -1])], axis=1) + 0.1 * rng.normal(size=(n, 2))
    model = ca.GP_RBF(optimize=False)
    xd = ca.device.to_device(x, torch.float64, "cuda")
    yd = ca.device.to_device(y, torch.float64, "cuda")
    for ell, sf, noise in [(1.0, 1.0, 0.01), (0.6, 1.7, 0.1)]:
        lml, grad = model.log_marginal_likelihood(xd, yd, ell, sf, noise)
        olml, ograd = oracle.gp_lml_and_grad(x, y, ell, sf, noise)
        assert abs(lml - olml) < 1e-8 * abs(olml)
        np.testing.assert_allclose(grad, ograd, rtol=1e-7, atol=1e-7 * np.max(np.abs(ograd)))


def test_gp_rbf_optimize_matches_oracle(ca):
    rng = np.random.default_rng(5)
    n = 250
    x = np.sort(rng.uniform(0, 6, size=(n, 1)), axis=0)
    y = np.hstack([np.sin(2 * x), np.cos(3 * x) + 0.3 * x]) + 0.15 * rng.normal(size=(n, 2))
    xt = np.linspace(0.1, 5.9, 40)[:, None]
    model = ca.GP_RBF(optimize=True)
    assert model.fit([x, y]) is True
    ref = oracle.gp_rbf_optimize(x, y)
    # same optimiser, same start, same objective: the optima agree closely
    got = np.array([model.kernel.l, model.kernel.sf, model.kernel.noise])
    want = np.array([ref["ell"], ref["sf2"], ref["noise"]])
    np.testing.assert_allclose(got, want, rtol=1e-3)
    assert model.optimizer_result.success
    # the optimum beats the fixed defaults on the marginal likelihood, and predictions agree
    pred = model.predict(xt)
    opred = oracle.gp_rbf_predict(ref, xt)
    assert _relerr(pred, opred) < 1e-4
    fixed = ca.GP_RBF(optimize=False)
    fixed.fit([x, y])
    assert float(np.mean((pred - np.hstack([np.sin(2 * xt), np.cos(3 * xt) + 0.3 * xt])) ** 2)) < \
        float(np.mean((fixed.predict(xt) - np.hstack([np.sin(2 * xt), np.cos(3 * xt) + 0.3 * xt])) ** 2)) * 1.5


def test_adaptive_inputs_warp(ca):
    """`adaptive_inputs=True` (both reference scripts use it): inputs are warped onto a regular
    grid by an exact GP x -> z (Inputs.py:11-22) and test inputs go through the same model
    (MRGP.py:770-778).  Checked against the oracle fed with the oracle's own warp."""
    rng = np.random.default_rng(12)
    n, ns, res = 200, 80, 1
    x = np.sort(rng.uniform(1, 3, size=(n, 1)) ** 2, axis=0)          # unevenly spaced inputs
    y = np.hstack([np.sin(2 * x), np.cos(x)]) + 0.05 * rng.normal(size=(n, 2))
    xs = np.sort(rng.uniform(1.5, 8.5, size=(ns, 1)), axis=0)
    kernels = [ca.RBFKernel(l=1.0, sf=1.0), ca.RBFKernel(l=0.5, sf=1.0)]
    model = ca.MultiResolutionGaussianProcess([x, y], index_set_obj=ca.IndexSetUniform(n, res, 2),
                                              spectral_density_obj=kernels, adaptive_inputs=True)
    assert model.dx == 1 and isinstance(model.input_obj.input_model, ca.GP_RBF)
    model.fit()
    mean = model.get_predicted_mean(xs, ca.IndexSetUniform(ns, res, 2))
    # oracle: same pipeline on the CPU
    xn, _, mu, sd = oracle.normalize_inputs(x)
    z = np.linspace(xn.min(), xn.max(), n)[:, None]
    warp = oracle.gp_rbf_optimize(xn, z)
    np.testing.assert_allclose(model.input_obj.x, z)
    zs = oracle.gp_rbf_predict(warp, (xs - mu) / sd)
    np.testing.assert_allclose(model.input_obj.warp((xs - mu) / sd), zs, rtol=1e-3, atol=1e-3)
    specs = [oracle.DenseLayerSpec(1.0, 1.0, None), oracle.DenseLayerSpec(0.5, 1.0, None)]
    omodel, _ = oracle.mrgp_fit(z, y, oracle.index_bounds_uniform(n, res, 2), specs)
    omean, _ = oracle.mrgp_predict(z, omodel, specs, model.input_obj.warp((xs - mu) / sd),
                                   oracle.index_bounds_uniform(ns, res, 2), want_var=False)
    assert _relerr(mean, omean) < 1e-6


def test_ard_lml_gradient_and_optimised_fit(ca):
    """``GP_RBF(ARD=True)``: one length-scale per input dimension, as the reference's comparison
    script fits with GPy (scripts/tests/GPRBF_vs_ciMRGP_vs_fiMRGP.py:118).  LML and its d + 2
    gradient entries against the oracle; the optimised fit and its predictions against the oracle's
    L-BFGS-B optimum."""
    rng = np.random.default_rng(31)
    n = 220
    x = rng.uniform(-2, 2, size=(n, 3))
    # the second input matters little, the third not at all: ARD must give it a long length-scale
    y = np.stack([np.sin(2.5 * x[:, 0]) + 0.2 * x[:, 1], np.cos(1.5 * x[:, 0]) * (1 + 0.1 * x[:, 1])], axis=1)
    y += 0.05 * rng.normal(size=y.shape)
    xt = rng.uniform(-1.8, 1.8, size=(50, 3))
    model = ca.GP_RBF(ARD=True)
    xd = ca.device.to_device(x, torch.float64, "cuda")
    yd = ca.device.to_device(y, torch.float64, "cuda")
    for ells, sf, noise in [((1.0, 1.0, 1.0), 1.0, 0.01), ((0.5, 1.7, 3.0), 1.4, 0.1)]:
        lml, grad = model.log_marginal_likelihood_ard(xd, yd, ells, sf, noise)
        olml, ograd = oracle.gp_lml_and_grad_ard(x, y, ells, sf, noise)
        assert grad.shape == (5,)
        assert abs(lml - olml) < 1e-8 * abs(olml)
        np.testing.assert_allclose(grad, ograd, rtol=1e-7, atol=1e-7 * np.max(np.abs(ograd)))
    assert model.fit([x, y]) is True
    ref = oracle.gp_rbf_optimize_ard(x, y)
    assert model.optimizer_result.success
    np.testing.assert_allclose(model.lengthscales[0], ref["ells"][0], rtol=2e-3)
    assert model.lengthscales[2] > 5 * model.lengthscales[0]            # the irrelevant input is switched off
    pred = model.predict(xt)
    opred = oracle.gp_rbf_predict_ard(ref, xt)
    assert _relerr(pred, opred) < 1e-3
    # isotropic start without optimisation = the plain kernel
    fixed_ard = ca.GP_RBF(ARD=True, optimize=False)
    fixed_iso = ca.GP_RBF(optimize=False)
    fixed_ard.fit([x, y])
    fixed_iso.fit([x, y])
    assert _relerr(fixed_ard.predict(xt), fixed_iso.predict(xt)) < 1e-12
