"""Model orchestrator of the dense path: coarse-to-fine residual chain over
exact RBF GP blocks, sum-over-resolutions prediction.

API mirror of the reference's ``MultiResolutionGaussianProcess`` (MRGP.py:15-34
constructor keywords, ``fit`` :367, ``get_predicted_mean`` :805,
``get_central_moment2`` :816, ``get_test_likelihood`` :825, properties
``get_posterior`` / ``get_stats`` :359-365).  What changes is the per-block
arithmetic: where the reference expands each block in a sine basis and runs
mean-field updates, this class gives each block an exact RBF GP whose Gram
build, Cholesky and solves are hand-written HIP kernels (Posteriors.DenseBlock).
The structure is the reference's:
  * inputs z-scored once, globally (MRGP.py:278-295);
  * block (j, l) = contiguous index range of the index set (IndexSetGenerator.py:51-65);
  * layer j is fitted on  observations - f_bar,  f_bar = sum of the coarser
    layers' predictions at the training points (Stats.py:126-157,
    LatentOutputs.py:11-18, Posteriors.py:68);
  * prediction = sum over layers of the concatenated per-region predictions,
    test block (j, l) served by training block (j, l) (MRGP.py:757-803);
    variance likewise (MRGP.py:902-905).
With fixed hyper-parameters one sweep is exact, so ``fit`` runs a single sweep
whatever ``n_iter`` is (further sweeps would reproduce it bit for bit).
"""
import weakref

import numpy as np
import torch

from . import device as dev
from . import dist
from .Inputs import Inputs
from .KernelClass import RBFKernel
from .Posteriors import DensePosterior, _Fanout


class DenseStats(object):
    """Per-layer read-only view of what the reference keeps in ``Stats``
    (Stats.py:7-64): latent function per region, bias and noise per region."""

    def __init__(self, model, layer):
        # a proxy, not a reference: model -> stats_obj -> model would be a cycle, and a cycle keeps
        # tens of GB of Cholesky factors alive until the garbage collector happens to run
        self._m = weakref.proxy(model)
        self._j = layer
        self.n_regions = model.n_regions[layer]
        self.dy = model.dy

    @property
    def latent_f_mean(self):
        fb = self._m._f_bar_layers[self._j]
        return [fb[int(a):int(b)].double().cpu().numpy() for a, b in self._m.index_set_obj.bounds[self._j]]

    @property
    def bias_mean(self):
        return [None if b is None else b.bias.double().cpu().numpy() for b in self._m.posterior_obj[self._j].blocks]

    @property
    def noise_mean(self):
        """E[tau]: noise PRECISION, the reference's convention (Stats.py:30-35)."""
        return [None if b is None else 1.0 / float(b.noise.item()) for b in self._m.posterior_obj[self._j].blocks]


class MultiResolutionGaussianProcess(object):
    def __new__(cls, *args, **kwargs):
        """A basis-function object selects the reference's own reduced-rank blocks
        (``ReducedRank.ReducedRankMRGP``); without one the blocks are
        exact RBF GPs (this class)."""
        if cls is MultiResolutionGaussianProcess:
            basis = kwargs.get('basis_function_obj', args[3] if len(args) > 3 else None)
            if basis is not None:
                from .ReducedRank import ReducedRankMRGP
                return object.__new__(ReducedRankMRGP)
        return object.__new__(cls)

    def __init__(self, train_xy,
                 n_basis=None,
                 index_set_obj=None,
                 basis_function_obj=None,
                 spectral_density_obj=None,
                 basis_interval_obj=None,
                 interval_factor=1,
                 adaptive_inputs=False,
                 standard_normalized_inputs=True,
                 axis_resolution_specific=False,
                 ard_resolution_specific=False,
                 noise_region_specific=True,
                 bias_region_specific=True,
                 noninformative_initialization=True,
                 snr_ratio=None,
                 full_x=None,
                 input_model=None,
                 forced_independence=False,
                 verbose=False,
                 dtype='f64',
                 device=None,
                 process_group=None,
                 keep_factors=True):
        self.verbose = verbose
        self.forced_independence = forced_independence
        if forced_independence is not True and (axis_resolution_specific or ard_resolution_specific):
            raise TypeError("not yet supported")
        if index_set_obj is None:
            raise ValueError('index_set_obj is required')
        self.adaptive_inputs = adaptive_inputs
        self.standard_normalized_inputs = standard_normalized_inputs
        self.noise_region_specific = noise_region_specific
        self.bias_region_specific = bias_region_specific
        self.keep_factors = keep_factors

        self.n_layers = index_set_obj.get_n_resolutions() + 1
        self.index_set_obj = index_set_obj
        self.n_basis = n_basis
        x_train = np.asarray(train_xy[0], dtype=np.float64)
        y_train = np.asarray(train_xy[1], dtype=np.float64)
        self.observations = y_train
        self.dy = y_train.shape[1]
        if self.dy < 2:
            raise ValueError('output dimension must be greater than 1')
        if x_train.shape[0] != index_set_obj.sample_length:
            raise ValueError('index set was built for a different number of samples')

        x_train, self.full_x, self.mean_x_train, self.std_x_train = self._normalize_inputs(x_train, full_x)

        if spectral_density_obj is None:
            spectral_density_obj = RBFKernel()
        if isinstance(spectral_density_obj, list):
            if len(spectral_density_obj) != self.n_layers:
                raise ValueError('spectral_density_obj must be a list of the same length as the number of '
                                 'resolutions + 1')
            self.spectral_density_obj = spectral_density_obj
        else:
            self.spectral_density_obj = [spectral_density_obj] * self.n_layers
        for k in self.spectral_density_obj:
            if not isinstance(k, RBFKernel):
                # a spectral density without a basis-function object: neither path applies
                raise TypeError('spectral_density_obj must be an RBFKernel when basis_function_obj is None')
        if snr_ratio is not None:
            # reference: initial noise variance of layer 0 from an SNR (MRGP.py:196-199,966-971)
            self.spectral_density_obj = list(self.spectral_density_obj)
            k0 = self.spectral_density_obj[0]
            self.spectral_density_obj[0] = RBFKernel(k0.l, k0.sf, self._compute_initial_noise_var_from_snr(y_train, snr_ratio))

        self.device = dev.require_gpu(device)
        self.dtype = dev.as_torch_dtype(dtype)
        self.group = process_group
        self.rank, self.world_size = dist.world(process_group)

        # learned input warp x -> regular grid (Inputs.py:8-55): an exact RBF GP fitted on the GPU
        self.input_obj = Inputs(x=x_train, index_set=index_set_obj, learn_inputs=self.adaptive_inputs,
                                full_x=self.full_x, input_model=input_model)
        self._x_dev = dev.to_device(self.input_obj.x, self.dtype, self.device)
        self.dx = self.input_obj.x.shape[1]
        self.n_regions = [len(layer) for layer in index_set_obj.bounds]
        self.n_samps = [[int(b - a) for a, b in layer] for layer in index_set_obj.bounds]
        self.x = [[self._x_dev[int(a):int(b)] for a, b in index_set_obj.bounds[j]] for j in range(self.n_layers)]
        self._y = dev.to_device(y_train, self.dtype, self.device)
        # nested ownership: layers >= self._first_local exchange nothing during the fit (dist.plan_layers)
        self.owner, self._first_local = dist.plan_layers(index_set_obj.bounds, self.world_size)

        self.posterior_obj = [DensePosterior(self.n_regions[j], self.dy, self.spectral_density_obj[j],
                                             noise_region_specific, bias_region_specific)
                              for j in range(self.n_layers)]
        self.stats_obj = [DenseStats(self, j) for j in range(self.n_layers)]
        self._f_bar_layers = [None] * self.n_layers
        self._fitted = False
        self.lower_bound = []
        self.lower_bound_layer = [[] for _ in range(self.n_layers)]

    # ------------------------------------------------------------------ helpers
    def _normalize_inputs(self, x_train, full_x):
        x = x_train if full_x is None else np.asarray(full_x, dtype=np.float64)
        if self.standard_normalized_inputs is True:
            std_x_train = np.std(x, 0)
            std_x_train[std_x_train == 0] = 1
            mean_x_train = np.mean(x, 0)
            x_train = (x_train - mean_x_train) / std_x_train
            if full_x is not None:
                full_x = (x - mean_x_train) / std_x_train
        else:
            mean_x_train = None
            std_x_train = None
        return x_train, full_x, mean_x_train, std_x_train

    def _slices(self, t, layer, bounds=None):
        bounds = self.index_set_obj.bounds if bounds is None else bounds
        return [t[int(a):int(b)] for a, b in bounds[layer]]

    def _owned(self, layer):
        return [l for l in range(self.n_regions[layer]) if self.owner[layer][l] == self.rank]

    @property
    def get_posterior(self):
        return self.posterior_obj

    @property
    def get_stats(self):
        return self.stats_obj

    # ---------------------------------------------------------------------- fit
    def fit(self, n_iter=1, tol=1e-3, min_iter=10):
        self._fit()

    def _fit(self):
        n, q = self._y.shape
        f_bar = torch.zeros_like(self._y)
        self._layer_events = [torch.cuda.Event(enable_timing=True) for _ in range(self.n_layers + 1)]
        failed = torch.zeros(self.n_layers, dtype=self.dtype, device=self.device)
        # Layers from self._first_local on are LOCAL (dist.plan_layers): a rank's blocks lie inside row ranges whose
        # coarser predictions it computed itself, so the residual chain needs no exchange there; their predictions
        # (and failure flags) are assembled on every rank by ONE all-reduce after the sweep.
        # NOT when a layer takes statistics over ALL its regions (shared bias / shared noise: bias_region_specific or
        # noise_region_specific False): those read the whole latent function, which a rank holds on its own ranges
        # only while the layers are local -- then every layer is exchanged as it is fitted (round 5).
        local_from = self._first_local if self.world_size > 1 else self.n_layers
        if any(p.needs_whole_layer() for p in self.posterior_obj[local_from:]):
            local_from = self.n_layers
        self._local_from = local_from
        n_local = max(0, self.n_layers - local_from) if self.world_size > 1 else 0
        # every local layer writes a zero-offset (N x q) array of its own (the batched fit addresses a block by ONE row
        # offset from the start of the storage: a view into a shared buffer at an odd offset would disable it);
        # the flags sit in a small tensor; both are copied into ONE flat buffer for the single collective
        local_pred = [torch.zeros((n, q), dtype=self.dtype, device=self.device) for _ in range(n_local)]
        local_flag = torch.zeros(max(n_local, 1), dtype=self.dtype, device=self.device)
        for j in range(self.n_layers):
            self._layer_events[j].record()
            self._f_bar_layers[j] = f_bar
            is_local = j >= local_from and n_local > 0
            if is_local:
                layer_pred, flag = local_pred[j - local_from], local_flag[j - local_from:j - local_from + 1]
                buf = None
            else:
                # [layer's training-point prediction (N x q) | failure flag]: ONE buffer, one collective
                buf = torch.zeros(n * q + 1, dtype=self.dtype, device=self.device)
                layer_pred, flag = buf[:n * q].view(n, q), buf[n * q:]
            owned = self._owned(j)
            self.posterior_obj[j].update_scale_given_axis(
                y_mean=self._slices(self._y, j), x=self.x[j], f_bar=self._slices(f_bar, j),
                train_out=self._slices(layer_pred, j), owned=owned, keep_factors=self.keep_factors)
            # A non-PD block must fail on EVERY rank, not only on its owner (whose exception
            # would leave the others blocked in the collective): the blocks' LAPACK-style info
            # words ride in the same all-reduce and every rank raises after it.
            self.posterior_obj[j].failure_flag(owned, out=flag)
            if not is_local:
                # residual chain (Stats.py:126-157): every rank needs the whole layer's prediction
                dist.allreduce_sum_(buf, self.group)
                failed[j:j + 1].copy_(buf[n * q:])           # read ONCE, after the sweep: the fit stays enqueue-only
            f_bar = f_bar + layer_pred                       # local layers: valid on this rank's own ranges, all it reads
        if n_local:
            tail = torch.cat([t.reshape(-1) for t in local_pred] + [local_flag[:n_local]])
            dist.allreduce_sum_(tail, self.group)
            # the latent function of every layer, now complete on every rank
            f_bar = self._f_bar_layers[local_from]
            for j in range(local_from, self.n_layers):
                k = j - local_from
                self._f_bar_layers[j] = f_bar
                f_bar = f_bar + tail[k * n * q:(k + 1) * n * q].view(n, q)
            failed[local_from:].copy_(tail[n_local * n * q:])
        self._layer_events[self.n_layers].record()
        # (a host read per layer made the device wait for the host's enqueue of the next layer: 2-3 ms on the
        # fine layers of config 4.  After a failed factorisation the later layers run on garbage -- harmless:
        # every kernel is bounded and flags its own `info` -- and the first failing layer is the one reported.)
        flags = failed.cpu().numpy()
        if np.any(flags != 0.0):
            j = int(np.flatnonzero(flags != 0.0)[0])
            self.posterior_obj[j].check(self._owned(j))      # the owner reports the leading minor
            if dev.is_watchdog(flags[j]):                    # a schedule failure on another rank: not "not PD"
                raise RuntimeError('cimrgp_potrf: schedule watchdog (a block of layer %d owned by another rank)' % j)
            raise np.linalg.LinAlgError('Matrix is not positive definite (a block of layer %d '
                                        'owned by another rank)' % j)
        self._f_bar_final = f_bar
        self._fitted = True

    def layer_fit_ms(self):
        """Device time of each layer of the last sweep (events on the fit's stream)."""
        torch.cuda.synchronize(self.device)
        ev = self._layer_events
        return [float(ev[j].elapsed_time(ev[j + 1])) for j in range(self.n_layers)]

    # ------------------------------------------------------------------ predict
    def _prepare_test(self, test_x):
        test_x = np.asarray(test_x, dtype=np.float64)
        if self.standard_normalized_inputs is True:
            test_x = (test_x - self.mean_x_train) / self.std_x_train
        if self.adaptive_inputs is True:                 # MRGP.py:770-778
            test_x = self.input_obj.warp(test_x, self.full_x)
        return dev.to_device(test_x, self.dtype, self.device)

    def _check_index_set(self, index_set, number_of_regions):
        if index_set.get_n_resolutions() > self.index_set_obj.get_n_resolutions():
            raise ValueError('resolution in the test index set must be smaller or equal to that in the '
                             'train set.')
        if number_of_regions is None:
            if getattr(self.index_set_obj, 'divider', None) != getattr(index_set, 'divider', None):
                raise ValueError('divider on the training index_set must be'
                                 ' the same as in the test index_set.')
        else:
            if (self.n_regions == number_of_regions) is False:
                raise ValueError('number of regions in the training must be the same as test.')
        for j in range(index_set.get_n_resolutions() + 1):
            if len(index_set.bounds[j]) != self.n_regions[j]:
                raise ValueError('number of regions in the training must be the same as test.')

    def _predict(self, test_x, index_set, want_var, include_noise=True):
        if not self._fitted:
            raise RuntimeError('call fit() before predicting')
        xs = self._prepare_test(test_x)
        ns = xs.shape[0]
        # fused [mean | var] buffer: one collective for the sum over resolutions
        fused = torch.zeros((self.dy + 1, ns), dtype=self.dtype, device=self.device)
        mean = torch.zeros((ns, self.dy), dtype=self.dtype, device=self.device)
        var = fused[self.dy] if want_var else None
        if index_set is None:
            # every prediction is taken from resolution 0 (MRGP.py:726-755), which presumes ONE
            # root region; with a multi-region first layer the test points need an index set
            if self.n_regions[0] != 1:
                raise ValueError('index_set_obj is required when the first layer has more than one region')
            if self.owner[0][0] == self.rank:
                self.posterior_obj[0].blocks[0].predict(xs, mean, var)
        else:
            n_layers = index_set.get_n_resolutions() + 1
            for j in range(n_layers):
                last = (j == n_layers - 1)
                owned = self._owned(j)
                if not owned:
                    continue
                # blocks of a layer write disjoint test ranges: equal-sized ones in ONE batched call, the
                # others in flight together on the stream pool; layers accumulate into the same ranges and
                # follow one another
                if want_var:
                    self.posterior_obj[j].predict_layer(self._x_dev, xs, index_set.bounds[j], set(owned), mean, var,
                                                        include_noise and last,
                                                        lambda cnt, nmax: _Fanout(self.device, cnt, nmax))
                    continue
                fan = _Fanout(self.device, len(owned), max(self.n_samps[j][l] for l in owned))
                for l in owned:
                    a, b = (int(v) for v in index_set.bounds[j][l])
                    blk = self.posterior_obj[j].blocks[l]
                    with torch.cuda.stream(fan.stream()):
                        blk.predict(xs[a:b], mean[a:b], None)
                fan.join()
        if self.world_size > 1:
            fused[:self.dy] = mean.t()
            dist.allreduce_sum_(fused, self.group)
            mean = fused[:self.dy].t()
        mean_np = mean.double().cpu().numpy()
        var_np = fused[self.dy].double().cpu().numpy() if want_var else None
        return mean_np, var_np

    def get_predicted_mean(self, test_x, index_set_obj=None, number_of_regions=None):
        if index_set_obj is not None:
            self._check_index_set(index_set_obj, number_of_regions)
        return self._predict(test_x, index_set_obj, want_var=False)[0]

    def get_central_moment2(self, test_x, index_set_obj=None, number_of_regions=None):
        if index_set_obj is not None:
            self._check_index_set(index_set_obj, number_of_regions)
        return self._predict(test_x, index_set_obj, want_var=True)[1]

    def get_predicted_mean_and_var(self, test_x, index_set_obj=None, number_of_regions=None, include_noise=True):
        """One pass for both moments (the reference needs two calls)."""
        if index_set_obj is not None:
            self._check_index_set(index_set_obj, number_of_regions)
        return self._predict(test_x, index_set_obj, want_var=True, include_noise=include_noise)

    def get_test_likelihood(self, test, index_set_obj=None, number_of_regions=None):
        test_x, test_y = test[0], test[1]
        mf, vf = self.get_predicted_mean_and_var(test_x, index_set_obj, number_of_regions)
        ll = -0.5 * np.log(2 * np.pi * vf) - 0.5 * (np.linalg.norm((test_y - mf), axis=1) ** 2) / vf
        return np.mean(ll)

    @staticmethod
    def _compute_initial_noise_var_from_snr(y, snr_ratio):
        n_samps = y.shape[0]
        y_var = (np.linalg.norm(y) ** 2) / n_samps - np.dot(np.mean(y, axis=0), np.mean(y, axis=0))
        return y_var / snr_ratio
