"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference).  The reference is
imported unmodified with the in-process shim recorded in SURVEY.md 8c:
``scipy.misc.logsumexp`` was removed upstream and ``GPy``/``gpflow`` are not
installed, so a forwarding attribute and two empty stub modules are planted
before import; with ``adaptive_inputs=False`` GPy is never touched.  Only
DATA (inputs and the reference's outputs) is written; no reference source.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py
"""
import os
import sys
import types
import warnings

import numpy as np

warnings.filterwarnings("ignore")
REF = os.environ.get("CIMRGP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    import scipy.misc
    import scipy.special
    scipy.misc.logsumexp = scipy.special.logsumexp
    sys.modules.setdefault('GPy', types.ModuleType('GPy'))
    sys.modules.setdefault('gpflow', types.ModuleType('gpflow'))
    sys.path.insert(0, os.path.join(REF, 'src'))
    sys.dont_write_bytecode = True


def _bounds_of(index_set):
    """(start, stop) per region; asserts each region is a contiguous range."""
    out = []
    for layer in index_set:
        b = np.zeros((len(layer), 2), dtype=np.int64)
        for l, idx in enumerate(layer):
            assert idx == list(range(idx[0], idx[-1] + 1))
            b[l] = (idx[0], idx[-1] + 1)
        out.append(b)
    return out


def toy_f(x):
    """Seeded synthetic two-output target on x in [1, 3] (own generator)."""
    return np.hstack([np.sin(3.0 * x) + 0.3 * np.cos(11.0 * x * x),
                      np.cos(2.0 * x) * np.exp(-0.3 * x) + 0.2 * np.sin(17.0 * x)])


def main():
    _import_reference()
    from IndexSetGenerator import IndexSetUniform
    from KernelClass import LaplacianEigenpairs, MaternKernel
    from MRGP import MultiResolutionGaussianProcess
    from RegressionInput import GP_RBF
    from Inputs import Inputs

    # ---- a1: index sets ------------------------------------------------
    blob = {}
    for (n, r, d) in [(512, 3, 2), (160, 7, 2), (100000, 5, 2), (513, 2, 3), (37, 0, 2)]:
        idx = IndexSetUniform(n, r, d)
        for m, b in enumerate(_bounds_of(idx.index_set)):
            blob['uniform_%d_%d_%d_layer%d' % (n, r, d, m)] = b
        blob['uniform_%d_%d_%d_divider' % (n, r, d)] = np.int64(idx.divider)
    np.random.seed(7)
    idx = IndexSetUniform(400, 2, None, n_regions=[1, 3, 5])
    for m, b in enumerate(_bounds_of(idx.index_set)):
        blob['random_400_layer%d' % m] = b
    blob['random_400_seed'] = np.int64(7)
    np.savez_compressed(os.path.join(OUT, 'structure_index_sets.npz'), **blob)

    # ---- a3: input normalisation; a2: gather; a15 pre/post-processing ---
    rng = np.random.RandomState(3)
    x1 = rng.uniform(-2, 5, size=(97, 1))
    x2 = rng.normal(size=(64, 2)) * np.array([3.0, 0.0]) + np.array([1.0, 4.0])  # zero-std column
    fullx = rng.uniform(-3, 6, size=(130, 1))
    fake = types.SimpleNamespace(standard_normalized_inputs=True)
    blob = dict(x1=x1, x2=x2, fullx=fullx)
    for name, (xt, fx) in dict(a=(x1, None), b=(x2, None), c=(x1, fullx)).items():
        xn, fxn, mu, sd = MultiResolutionGaussianProcess._normalize_inputs(fake, xt.copy(), None if fx is None else fx.copy())
        blob['norm_%s_x' % name] = xn
        blob['norm_%s_mean' % name] = mu
        blob['norm_%s_std' % name] = sd
        if fxn is not None:
            blob['norm_%s_full' % name] = fxn
    idx = IndexSetUniform(97, 2, 2)
    inp = Inputs(x=x1, index_set=idx, learn_inputs=False, full_x=None, input_model=None)
    blob['gather_2_3'] = inp.get_inputs(2, 3)
    blob['gather_1_0'] = inp.get_inputs(1, 0)
    y1 = np.hstack([np.sin(x1), x1 ** 2]) + 0.05 * rng.normal(size=(97, 2))
    xt = rng.uniform(-2, 5, size=(11, 1))
    plug = GP_RBF()
    xz, yz = plug._preprocess([x1, y1], True)
    blob.update(plug_y=y1, plug_xt=xt, plug_xz=xz, plug_yz=yz,
                plug_xtz=plug._preprocess(xt, False),
                plug_back=plug._reverse_trans_labels(yz[:11] * 0.5 + 0.25),
                plug_noise_init=np.float64(yz.var() * 0.01))
    np.savez_compressed(os.path.join(OUT, 'structure_normalise.npz'), **blob)

    # ---- a4/a5: kernel objects -------------------------------------------
    blob = {}
    xs1 = rng.uniform(-1.5, 1.5, size=(33, 1))
    xs2 = rng.uniform(-1.5, 1.5, size=(29, 2))
    lap = LaplacianEigenpairs()
    blob.update(lap_x1=xs1, lap_x2=xs2)
    for bid in (1, 2, 7):
        f, lam = lap.get_eigenpairs(xs1, bid)
        blob['lap1_f_%d' % bid] = f
        blob['lap1_l_%d' % bid] = np.float64(lam)
        f, lam = lap.get_eigenpairs(xs2, bid, basis_interval=np.array([2.0, 1.7]))
        blob['lap2_f_%d' % bid] = f
        blob['lap2_l_%d' % bid] = np.float64(lam)
        f, lam = lap.get_eigenpairs(xs2, bid, basis_interval=np.array([2.0, 1.7]), per_dimension=True)
        blob['lap2pd_f_%d' % bid] = f
        blob['lap2pd_l_%d' % bid] = lam
    r = np.linspace(0.05, 4.0, 40)
    s = np.linspace(0.0, 9.0, 40)
    blob.update(mat_r=r, mat_s=s)
    for nu in (0.5, 1.0, 1.5, 2.5):
        mk = MaternKernel(nu=nu, l=0.7, sf=1.3)
        tag = str(nu).replace('.', 'p')
        blob['mat_k_' + tag] = mk.kernel(r)
        blob['mat_lk_' + tag] = mk.log_kernel(r)
        blob['mat_s_' + tag] = mk.spectral(s)
        blob['mat_ls_' + tag] = mk.log_spectral(s)
    phi_a = rng.normal(size=(21, 6))
    phi_b = rng.normal(size=(21, 6))
    lam = np.sort(rng.uniform(0.1, 30.0, size=6))
    blob.update(est_phi_a=phi_a, est_phi_b=phi_b, est_lambda=lam,
                est_out=MaternKernel(nu=1.5, l=0.7, sf=1.3).estimate_kernel(phi_a, phi_b, lam))
    np.savez_compressed(os.path.join(OUT, 'kernel_objects.npz'), **blob)

    # ---- a11/a14 + config-1 recipe: full model runs ------------------------
    for tag, res, forced in [('fi_r2', 2, True), ('fi_r3', 3, True), ('ci_r2', 2, False)]:
        np.random.seed(11)
        n = 512
        x = np.atleast_2d(np.linspace(1, 3, n)).T
        y = toy_f(x) + 0.1 * np.random.normal(size=(n, 2))
        n_basis = 30
        idx = IndexSetUniform(n, res, 2)
        model = MultiResolutionGaussianProcess(train_xy=[x, y], n_basis=n_basis, index_set_obj=idx,
                                               basis_function_obj=LaplacianEigenpairs(),
                                               spectral_density_obj=MaternKernel(nu=1, l=1, sf=1),
                                               adaptive_inputs=False, forced_independence=forced)
        model.fit(5, None)
        ns = 384
        xt = np.atleast_2d(np.linspace(1.01, 2.99, ns)).T
        idx_t = IndexSetUniform(ns, res, 2)
        blob = dict(x=x, y=y, xt=xt, n_basis=np.int64(n_basis), resolution=np.int64(res),
                    forced_independence=np.bool_(forced))
        blob['pred_mean_global'] = model.get_predicted_mean(xt)
        blob['pred_var_global'] = model.get_central_moment2(xt)
        blob['pred_mean_index'] = model.get_predicted_mean(xt, idx_t)
        # per-block train-point predictions and the reference's residual scatter
        stats = model.stats_obj
        for j in range(model.n_layers):
            for l in range(model.n_regions[j]):
                p = stats[j].bias_mean[l] + model.phi_x[j][l] @ stats[j].scale_axis_mean[l].T
                v = stats[j].bias_var[l] + (model.phi_x[j][l] ** 2) @ stats[j].scale_axis_central_moment2[l]
                blob['train_pred_%d_%d' % (j, l)] = p
                blob['train_predvar_%d_%d' % (j, l)] = v
                blob['latent_f_mean_%d_%d' % (j, l)] = stats[j].latent_f_mean[l]
                blob['latent_f_var_%d_%d' % (j, l)] = np.asarray(stats[j].latent_f_var[l]).reshape(-1)
                blob['scale_axis_mean_%d_%d' % (j, l)] = stats[j].scale_axis_mean[l]
                blob['bias_mean_%d_%d' % (j, l)] = np.asarray(stats[j].bias_mean[l])
                blob['interval_%d_%d' % (j, l)] = np.asarray(model.train_basis_intervals[j][l])
        # per-block test predictions (sum-over-layers pin)
        xtn = (xt - model.mean_x_train) / model.std_x_train
        for j in range(model.n_layers):
            for l in range(model.n_regions[j]):
                xb = xtn[idx_t.index_set[j][l], :]
                phi = np.stack([LaplacianEigenpairs().get_eigenpairs(xb, i + 1, model.train_basis_intervals[j][l])[0]
                                for i in range(n_basis)], axis=1)
                blob['test_pred_%d_%d' % (j, l)] = stats[j].bias_mean[l] + phi @ stats[j].scale_axis_mean[l].T
        # variance through the index-set path mutates stats: do it last
        blob['pred_var_index'] = model.get_central_moment2(xt, idx_t)
        np.savez_compressed(os.path.join(OUT, 'reference_model_%s.npz' % tag), **blob)
        print('wrote', tag)


if __name__ == '__main__':
    main()
