"""Input holder / per-block gather (reference Inputs.py:8-60).

Regions are contiguous sample ranges, so ``get_inputs`` is a slice -- on the
device a pointer offset, never a gather kernel.  The learned input warp
(``learn_inputs=True``: a GP_RBF fit x -> linspace grid, Inputs.py:11-48) is
listed as next in SURVEY 8f and raises until it is built.
"""


class Inputs(object):
    def __init__(self, x, index_set, learn_inputs=False, full_x=None, input_model=None):
        if learn_inputs is True:
            raise TypeError('not yet supported')
        self.learn_inputs = learn_inputs
        self.index_set = index_set
        self.input_model = input_model
        self.x = x

    def get_inputs(self, resolution, region):
        a, b = self.index_set.bounds[resolution][region]
        return self.x[int(a):int(b), :]
