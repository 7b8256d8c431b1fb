"""Oracle (test infrastructure): structural pieces of the path.

Each function restates one reference routine; citations are into
/root/reference.  Pinned by ``tests/golden/structure_*.npz`` (captured from
the reference itself, see ``tests/golden/make_golden.py``).
"""
import numpy as np


def index_bounds_uniform(sample_length, resolution, divider, first_divider_power=0):
    """Block boundaries of ``IndexSetUniform`` with a uniform divider.

    Follows IndexSetGenerator.py:51-65: layer m has ``divider**m`` contiguous
    blocks of ``floor(N / divider**m)`` samples; the remainder goes to the
    last block.  Returns ``bounds[m]`` = int64 array of shape (n_regions, 2)
    holding [start, stop) per region instead of materialised index lists.
    ``resolution == 0`` forces ``divider = 0`` (IndexSetGenerator.py:18-20),
    which still yields one region because ``0**0 == 1``.
    ``first_divider_power`` p > 0 (not in the reference; root-block policy of
    BASELINE config 4): layer m has ``divider**(m + p)`` blocks, i.e. the
    hierarchy starts at a layer whose blocks fit one device.
    """
    sample_length = int(sample_length)
    resolution = int(resolution)
    first_divider_power = int(first_divider_power)
    divider = 0 if (resolution == 0 and first_divider_power == 0) else int(divider)
    bounds = []
    for m in range(resolution + 1):
        n_regions = int(np.power(divider, m + first_divider_power))
        per_region = sample_length // n_regions
        if per_region < 1:
            raise ValueError('*** Chosen resolution is too large! ***')
        starts = np.arange(n_regions, dtype=np.int64) * per_region
        stops = starts + per_region
        stops[-1] = sample_length
        bounds.append(np.stack([starts, stops], axis=1))
    return bounds


def index_set_from_bounds(bounds):
    """Materialise ``index_set[m][l]`` as Python lists (the reference's type,
    IndexSetGenerator.py:61-64) from [start, stop) bounds."""
    return [[list(range(int(a), int(b))) for a, b in layer] for layer in bounds]


def normalize_inputs(x_train, full_x=None):
    """z-score the inputs once, globally (MRGP.py:278-295).

    Statistics come from ``full_x`` when given, else from ``x_train``;
    population std (ddof=0); a zero std is replaced by 1.
    Returns (x_train_n, full_x_n, mean, std).
    """
    ref = x_train if full_x is None else full_x
    std = np.std(ref, 0)
    std[std == 0] = 1
    mean = np.mean(ref, 0)
    x_train_n = (x_train - mean) / std
    full_x_n = None if full_x is None else (full_x - mean) / std
    return x_train_n, full_x_n, mean, std


def zscore_fit(inputs, labels):
    """Train-side pre-processing of the ``RegressionMethod`` plugin
    (RegressionInput.py:16-24): column-wise population mean/std of inputs and
    labels, no zero-std guard (the reference has none)."""
    stats = dict(data_mean=inputs.mean(axis=0), data_std=inputs.std(axis=0),
                 labels_mean=labels.mean(axis=0), labels_std=labels.std(axis=0))
    return stats


def zscore_apply(stats, inputs=None, labels=None, inverse_labels=None):
    """Apply (RegressionInput.py:24,26) or invert (RegressionInput.py:28-29)
    the plugin's z-scoring."""
    out = []
    if inputs is not None:
        out.append((inputs - stats['data_mean']) / stats['data_std'])
    if labels is not None:
        out.append((labels - stats['labels_mean']) / stats['labels_std'])
    if inverse_labels is not None:
        out.append(inverse_labels * stats['labels_std'] + stats['labels_mean'])
    return out[0] if len(out) == 1 else tuple(out)


def concat_regions(per_region):
    """Concatenate per-region arrays of one layer in region order
    (MRGP.py:802, Stats.py:152-153)."""
    return np.concatenate(per_region)


def latent_from_coarser(per_layer_region, bounds, resolution):
    """Residual chain (Stats.py:126-157): the latent function seen by layer
    ``resolution`` is the sum over all coarser layers j' < resolution of the
    concatenated per-region predictions at the training points, then sliced
    per region of layer ``resolution`` by its index set.

    ``per_layer_region[j'][l]`` = prediction of block (j', l) at its own
    training points ((n_l x q) for the mean, (n_l,) for the variance).
    Returns the list over regions of layer ``resolution``.
    """
    total = None
    for jp in range(resolution):
        layer = concat_regions(per_layer_region[jp])
        total = layer if total is None else total + layer
    return [total[int(a):int(b)] for a, b in bounds[resolution]]


def sum_over_layers(per_layer_region):
    """Final prediction = sum over layers of the concatenated per-region
    predictions (MRGP.py:802-803; variance MRGP.py:902-905)."""
    return sum(concat_regions(layer) for layer in per_layer_region)
