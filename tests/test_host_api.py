"""Host-side mirror of the reference interface, against captured reference
outputs.  CPU only (no compute calls into the HIP library)."""
import os

import numpy as np
import pytest

import cimrgp_amd as ca


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("n,r,d", [(512, 3, 2), (160, 7, 2), (100000, 5, 2), (513, 2, 3), (37, 0, 2)])
def test_index_set_uniform_matches_reference(golden_dir, n, r, d):
    g = _load(golden_dir, "structure_index_sets.npz")
    idx = ca.IndexSetUniform(n, r, d)
    assert idx.get_n_resolutions() == r
    assert idx.divider == int(g["uniform_%d_%d_%d_divider" % (n, r, d)])
    for m in range(r + 1):
        ref = g["uniform_%d_%d_%d_layer%d" % (n, r, d, m)]
        np.testing.assert_array_equal(idx.bounds[m], ref)
        regs = idx.get_index_set(m)
        assert len(regs) == len(ref)
        for l in (0, len(regs) - 1):
            assert list(regs[l]) == list(range(ref[l][0], ref[l][1]))
    # regions work as NumPy fancy indices like the reference's lists
    x = np.arange(n)[:, None]
    np.testing.assert_array_equal(x[idx.index_set[r][0], :], x[:len(idx.index_set[r][0])])


def test_index_set_random_regions_match_reference(golden_dir):
    g = _load(golden_dir, "structure_index_sets.npz")
    np.random.seed(int(g["random_400_seed"]))
    idx = ca.IndexSetUniform(400, 2, None, n_regions=[1, 3, 5])
    for m in range(3):
        np.testing.assert_array_equal(idx.bounds[m], g["random_400_layer%d" % m])


def test_index_set_errors():
    with pytest.raises(ValueError):
        ca.IndexSetUniform(5, 3, 2)


def test_laplacian_eigenpairs_match_reference(golden_dir):
    g = _load(golden_dir, "kernel_objects.npz")
    lap = ca.LaplacianEigenpairs()
    for bid in (1, 2, 7):
        f, lam = lap.get_eigenpairs(g["lap_x1"], bid)
        np.testing.assert_allclose(f, g["lap1_f_%d" % bid], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(lam, g["lap1_l_%d" % bid], rtol=1e-14)
        f, lam = lap.get_eigenpairs(g["lap_x2"], bid, basis_interval=np.array([2.0, 1.7]))
        np.testing.assert_allclose(f, g["lap2_f_%d" % bid], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(lam, g["lap2_l_%d" % bid], rtol=1e-14)
        f, lam = lap.get_eigenpairs(g["lap_x2"], bid, basis_interval=np.array([2.0, 1.7]), per_dimension=True)
        np.testing.assert_allclose(f, g["lap2pd_f_%d" % bid], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(lam, g["lap2pd_l_%d" % bid], rtol=1e-14)
    with pytest.raises(ValueError):
        lap.get_eigenpairs(g["lap_x2"], 1, basis_interval=np.array([1.0]))


def test_matern_kernel_matches_reference(golden_dir):
    g = _load(golden_dir, "kernel_objects.npz")
    for nu in (0.5, 1.0, 1.5, 2.5):
        mk = ca.MaternKernel(nu=nu, l=0.7, sf=1.3)
        tag = str(nu).replace(".", "p")
        np.testing.assert_allclose(mk.kernel(g["mat_r"]), g["mat_k_" + tag], rtol=1e-12)
        np.testing.assert_allclose(mk.log_kernel(g["mat_r"]), g["mat_lk_" + tag], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(mk.spectral(g["mat_s"]), g["mat_s_" + tag], rtol=1e-12)
        np.testing.assert_allclose(mk.log_spectral(g["mat_s"]), g["mat_ls_" + tag], rtol=1e-12, atol=1e-13)
    mk = ca.MaternKernel(nu=1.5, l=0.7, sf=1.3)
    np.testing.assert_allclose(mk.estimate_kernel(g["est_phi_a"], g["est_phi_b"], g["est_lambda"]), g["est_out"],
                               rtol=1e-12)


def test_rbf_kernel_protocol():
    k = ca.RBFKernel(l=0.5, sf=2.0)
    assert (k.l, k.sf, k.name) == (0.5, 2.0, "RBF")
    r = np.linspace(0, 3, 7)
    np.testing.assert_allclose(k.kernel(r), 2.0 * np.exp(-r ** 2 / (2 * 0.25)))
    np.testing.assert_allclose(np.exp(k.log_kernel(r)), k.kernel(r))
    # spectral density integrates back to k(0): (1/2pi) int S(s) ds = sf
    s = np.linspace(-60, 60, 200001)
    np.testing.assert_allclose(np.trapezoid(k.spectral(s), s) / (2 * np.pi), 2.0, rtol=1e-9)
    np.testing.assert_allclose(np.exp(k.log_spectral(s[::5000])), k.spectral(s[::5000]))
    # the RBF is the nu -> infinity limit of the Matern; the reference's Matern density carries
    # sqrt(2 pi) where the textbook one has 2 sqrt(pi) (KernelClass.py:86), hence the sqrt(2)
    big = ca.MaternKernel(nu=400.0, l=0.5, sf=2.0)
    pts = np.array([0.0, 1.0, 2.0])
    np.testing.assert_allclose(np.sqrt(2.0) * big.spectral(pts), k.spectral(pts), rtol=2e-2)
    with pytest.raises(ValueError):
        ca.RBFKernel(l=0.0)


def test_regression_method_preprocessing_matches_reference(golden_dir):
    g = _load(golden_dir, "structure_normalise.npz")
    plug = ca.GP_RBF(optimize=False)
    xz, yz = plug._preprocess([g["x1"], g["plug_y"]], True)
    np.testing.assert_array_equal(xz, g["plug_xz"])
    np.testing.assert_array_equal(yz, g["plug_yz"])
    np.testing.assert_array_equal(plug._preprocess(g["plug_xt"], False), g["plug_xtz"])
    np.testing.assert_array_equal(plug._reverse_trans_labels(yz[:11] * 0.5 + 0.25), g["plug_back"])


def test_model_constructor_errors_come_before_any_gpu_use():
    x = np.linspace(0, 1, 32)[:, None]
    idx = ca.IndexSetUniform(32, 1, 2)
    with pytest.raises(ValueError):      # MRGP.py:65-66
        ca.MultiResolutionGaussianProcess([x, x], index_set_obj=idx)
    y = np.hstack([x, x])
    with pytest.raises(TypeError):       # MRGP.py:50
        ca.MultiResolutionGaussianProcess([x, y], index_set_obj=idx, axis_resolution_specific=True)
    with pytest.raises(ValueError):      # MRGP.py:74-76
        ca.MultiResolutionGaussianProcess([x, y], index_set_obj=idx, spectral_density_obj=[ca.RBFKernel()])
    with pytest.raises(TypeError):
        ca.MultiResolutionGaussianProcess([x, y], index_set_obj=idx, spectral_density_obj=ca.MaternKernel())


def test_block_assignment_is_deterministic_and_balanced():
    from cimrgp_amd.dist import assign_blocks
    own = assign_blocks([4096] * 16, 8)
    assert sorted(np.bincount(own, minlength=8)) == [2] * 8
    own = assign_blocks([65536], 8)
    assert own.tolist() == [0]
    own = assign_blocks([100, 100, 300, 100], 2)
    assert own[2] != own[0] or own[2] != own[1]
    np.testing.assert_array_equal(assign_blocks([5, 7, 7, 3], 3), assign_blocks([5, 7, 7, 3], 3))


def test_root_block_policy_index_sets():
    """BASELINE config 4: the hierarchy starts at a layer whose blocks fit one device
    (``first_divider_power``); default 0 is the reference's index set, bit for bit."""
    import oracle
    ref = ca.IndexSetUniform(1000, 3, 2)
    same = ca.IndexSetUniform(1000, 3, 2, first_divider_power=0)
    assert all((a == b).all() for a, b in zip(ref.bounds, same.bounds))
    idx = ca.IndexSetUniform(262144, 4, 2, first_divider_power=3)
    assert idx.n_regions_per_layer() == [8, 16, 32, 64, 128]
    assert [int(b[0, 1] - b[0, 0]) for b in idx.bounds] == [32768, 16384, 8192, 4096, 2048]
    for got, want in zip(idx.bounds, oracle.index_bounds_uniform(262144, 4, 2, 3)):
        np.testing.assert_array_equal(got, want)
    ragged = ca.IndexSetUniform(1003, 1, 2, first_divider_power=2)        # remainder to the last region
    assert ragged.n_regions_per_layer() == [4, 8] and int(ragged.bounds[1][-1, 1]) == 1003
    assert ca.IndexSetUniform(100, 0, 2, first_divider_power=2).n_regions_per_layer() == [4]
    with pytest.raises(ValueError):
        ca.IndexSetUniform(100, 1, 2, first_divider_power=-1)
    with pytest.raises(ValueError):
        ca.IndexSetUniform(100, 3, 2, first_divider_power=5)              # finer than one sample per region


def test_space_filling_order():
    grid = np.array([(i, j) for i in range(16) for j in range(16)], dtype=float)
    perm = ca.space_filling_order(grid, bits=4)
    assert sorted(perm.tolist()) == list(range(256))
    steps = np.abs(np.diff(grid[perm], axis=0)).sum(axis=1)
    assert (steps == 1).all()                     # a Hilbert curve moves to an edge neighbour every step
    x1 = np.array([[3.0], [1.0], [2.0]])
    assert ca.space_filling_order(x1).tolist() == [1, 2, 0]
    rng = np.random.default_rng(0)
    x3 = rng.uniform(size=(500, 3))
    p3 = ca.space_filling_order(x3)
    assert sorted(p3.tolist()) == list(range(500))
    # Morton order in 3-D: the first eighth of the curve stays inside one octant
    first = x3[p3[:40]]
    assert (np.ptp(first, axis=0) <= 0.55).all()


def test_workloads_are_seeded_and_ordered():
    import workloads
    x, y = workloads.make_block(256)
    x2, y2 = workloads.make_block(256)
    np.testing.assert_array_equal(x, x2)
    np.testing.assert_array_equal(y, y2)
    assert (np.diff(x[:, 0]) >= 0).all()
    xa, ya, xsa = workloads.make_chain_2d(1024, order=ca.space_filling_order)
    assert xa.shape == (1024, 2) and ya.shape == (1024, 2) and xsa.shape == (256, 2)
    # train and test points follow ONE curve: test block l lies where train block l lies
    for tr, te in zip(np.split(xa, 8), np.split(xsa, 8)):
        assert np.linalg.norm(tr.mean(axis=0) - te.mean(axis=0)) < 0.8


def test_hilbert_order_makes_compact_regions():
    """Contiguous index ranges of Hilbert-sorted 2-D points are compact patches (what an RBF block
    needs); sorting by one coordinate gives strips as long as the whole domain."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, size=(8192, 2))
    xs = x[ca.space_filling_order(x)]
    for k in (8, 128):
        extent = np.mean([np.max(np.ptp(b, axis=0)) for b in np.split(xs, k)])
        assert extent < 1.6 * 2.0 / np.sqrt(k)      # longest side close to the ideal square's 2/sqrt(k)
    strips = np.mean([np.max(np.ptp(b, axis=0)) for b in np.split(x[np.argsort(x[:, 0])], 128)])
    assert strips > 1.9


def test_batched_fit_needs_the_same_row_ranges_in_every_layer_array():
    """Posteriors._fit_batched addresses a block by ONE row offset into the layer's arrays (cimrgp_layer_fit): region
    views of x, y, f_bar and train_out must be the same row ranges of their arrays, or the layer takes the
    block-by-block path.  A y that is itself a view at a non-zero storage offset is the case that used to raise."""
    import torch
    from cimrgp_amd.Posteriors import _layer_array, _same_rows
    n, q = 12, 2
    bounds = [(0, 6), (6, 12)]
    x_all = torch.zeros((n, 1), dtype=torch.float64)
    y_all = torch.zeros((n, q), dtype=torch.float64)
    big = torch.zeros((n + 3, q), dtype=torch.float64)
    y_off = big[3:]                                           # same shape, storage offset 3 rows
    views = lambda t: [t[a:b] for a, b in bounds]
    group = [0, 1]
    assert all(_layer_array(v, group) is not None for v in (views(x_all), views(y_all), views(y_off)))
    assert _same_rows((views(y_all), views(x_all), views(y_all)), group)
    assert not _same_rows((views(y_off), views(x_all), views(y_all)), group)
