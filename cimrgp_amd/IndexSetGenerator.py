"""Partition generator: the (resolution, partition) work units of the path.

Mirror of the reference's ``IndexSetUniform`` (IndexSetGenerator.py:4-92): same
constructor, attributes and ``index_set[m][l]`` indexing.  Every region is a
contiguous sample range, so regions are kept as Python ``range`` objects
(O(1) memory, usable wherever the reference's ``list(range(a, b))`` is: len,
iteration, NumPy fancy indexing) plus an int64 ``bounds[m]`` array of
[start, stop) pairs, which is what the device path consumes (a block's inputs
are a pointer offset, not a gather).
"""
import numpy as np


class IndexSetUniform(object):
    def __init__(self, sample_length, resolution, divider, n_regions=None,
                 min_percentage_of_samples_per_region=None, first_divider_power=0):
        """
        :param sample_length: number of samples
        :param resolution: number of layers minus one
        :param divider: layer m is cut into ``divider**m`` equal contiguous regions,
            the remainder going to the last one
        :param n_regions: list with the number of regions per layer; regions are then
            random contiguous ranges (NumPy global RNG, as in the reference) and
            ``divider`` is ignored
        :param min_percentage_of_samples_per_region: rejection threshold for the
            random regions, default 25% of the average region size
        :param first_divider_power: ROOT-BLOCK POLICY (not in the reference, default 0 =
            the reference): layer m is cut into ``divider**(m + first_divider_power)``
            regions, so the hierarchy starts at a layer whose regions fit one device
            instead of at the single all-samples region.  The reference allows a
            multi-region layer 0 only through ``n_regions=[...]``
            (IndexSetGenerator.py:28-43) and assumes one root region when it sizes
            the latent function (``n_samps_0 = len(index_set[0][0])``, Stats.py:127);
            this package sizes it by ``sample_length``.
        """
        self.resolution = int(resolution)
        self.first_divider_power = int(first_divider_power)
        if self.first_divider_power < 0:
            raise ValueError('first_divider_power must be >= 0')
        if n_regions is None:
            self.divider = 0 if (self.resolution == 0 and self.first_divider_power == 0) else int(divider)
        self.sample_length = int(sample_length)
        self.bounds = []
        if n_regions is None:
            for m in range(self.resolution + 1):
                self.bounds.append(self._uniform_bounds(m))
        else:
            self.region_ind = []
            self.min_number_of_samples_per_region = []
            share = 0.25 if min_percentage_of_samples_per_region is None else min_percentage_of_samples_per_region
            for m in range(self.resolution + 1):
                k = n_regions[m]
                self.min_number_of_samples_per_region.append(
                    int(np.floor(np.divide(self.sample_length, k) * share)))
                bounds, cuts = self._random_bounds(m, k)
                self.bounds.append(bounds)
                self.region_ind.append(cuts)
        self.index_set = [[range(int(a), int(b)) for a, b in layer] for layer in self.bounds]

    def get_n_resolutions(self):
        return self.resolution

    def get_index_set(self, resolution):
        return self.index_set[int(resolution)]

    def n_regions_per_layer(self):
        return [len(layer) for layer in self.bounds]

    def _uniform_bounds(self, resolution):
        n_regions = int(np.power(self.divider, resolution + self.first_divider_power))
        per_region = self.sample_length // n_regions
        if per_region < 1:
            raise ValueError('*** Chosen resolution is too large! ***')
        starts = np.arange(n_regions, dtype=np.int64) * per_region
        stops = starts + per_region
        stops[-1] = self.sample_length
        return np.stack([starts, stops], axis=1)

    def _random_bounds(self, resolution, number_of_regions):
        n = self.sample_length
        if number_of_regions == 1:
            return np.array([[0, n]], dtype=np.int64), None
        smallest = self.min_number_of_samples_per_region[resolution]
        while True:
            # same draw sequence as the reference: one randint(1, n) per interior cut
            cuts = [0] + [np.random.randint(1, n) for _ in range(number_of_regions - 1)] + [n]
            cuts = np.sort(cuts)
            if not np.any(np.diff(cuts) < smallest):
                break
        bounds = np.stack([cuts[:-1], cuts[1:]], axis=1).astype(np.int64)
        return bounds, cuts
