"""Seeded synthetic inputs of BASELINE.json's configurations, shared by
bench.py, tools/ and tests/ so that the GPU path and the CPU oracle are always fed the
same arrays.  NumPy only; nothing here touches the GPU or the oracle.

  config 2   1-D, one N = 8192 block                       make_block
  config 3   1-D, N = 65536, IndexSetUniform(N, 4, 2)      make_chain_1d   (31 blocks)
  config 4   2-D, N = 262144, 5 resolutions, root policy   make_chain_2d   (8..128 regions)
  config 5   1-D, one n = 16384 block, f32 vs f64          make_block + noise / length-scale grid
"""
import numpy as np

SQRT3 = float(np.sqrt(3.0))


def targets_1d(x, q, rng, noise_sd=0.1):
    """y_k = sin(3x + k) + 0.5 sin(17 x^2) + noise_sd N(0, 1)   (SURVEY.md 8d cfg 2/3)."""
    y = np.hstack([np.sin(3 * x + k) + 0.5 * np.sin(17 * x * x) for k in range(q)])
    return y + noise_sd * rng.normal(size=y.shape)


def make_block(n, q=2, seed=1234):
    """Config 2 / 5: x ~ sorted U(-sqrt3, sqrt3)."""
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(-SQRT3, SQRT3, size=(n, 1)), axis=0)
    return x, targets_1d(x, q, rng)


def block_test_points(ns):
    return np.linspace(-1.7, 1.7, ns)[:, None]


def make_chain_1d(n, q=2, seed=1234, ns=None):
    """Config 3: sorted 1-D inputs, N/4 sorted test points."""
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(-SQRT3, SQRT3, size=(n, 1)), axis=0)
    y = targets_1d(x, q, rng)
    ns = n // 4 if ns is None else ns
    xs = np.sort(rng.uniform(-1.7, 1.7, size=(ns, 1)), axis=0)
    return x, y, xs


def targets_2d(x, q, rng, noise_sd=0.1):
    """Smooth trend + medium + fine structure in both coordinates (every layer has work)."""
    a, b = x[:, :1], x[:, 1:2]
    y = np.hstack([np.sin(2 * a + k) * np.cos(1.5 * b) + 0.4 * np.sin(7 * a * b + k) + 0.2 * np.cos(19 * (a - b))
                   for k in range(q)])
    return y + noise_sd * rng.normal(size=y.shape)


def make_chain_2d(n, q=2, seed=1234, ns=None, order=None):
    """Config 4: x ~ U(-sqrt3, sqrt3)^2 sorted along a space-filling curve so that contiguous
    index blocks are compact patches (the reference partitions by sample order only,
    Inputs.py:57-60).  ``order``: callable (N x 2) -> permutation; the caller passes
    ``cimrgp_amd.space_filling_order`` (kept out of this module so it stays NumPy-only)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-SQRT3, SQRT3, size=(n, 2))
    ns = n // 4 if ns is None else ns
    xs = rng.uniform(-1.7, 1.7, size=(ns, 2))
    if order is not None:
        # one curve for both sets: cells are laid out over the training inputs' bounding box
        both = np.vstack([x, xs])
        perm = order(both)
        is_train = perm < n
        x = both[perm[is_train]]
        xs = both[perm[~is_train]]
    y = targets_2d(x, q, rng)
    return x, y, xs


def chain_length_scales(n_layers, d, ell0=1.0):
    """Per-layer length-scales: halved per layer in 1-D (SURVEY.md 8d cfg 3: l_j = l_0 2^-j);
    in 2-D a region's area halves per layer, so its linear extent shrinks by sqrt 2."""
    step = 0.5 if d == 1 else 0.5 ** 0.5
    return [ell0 * step ** j for j in range(n_layers)]
