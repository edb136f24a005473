"""Drop-in for architectures/anchors.py:8-60 (`Anchors`).

The reference rebuilds the anchors in host numpy (fp64) and copies them to the GPU on every
forward (32-55 ms).  Here the fp64 shift+add runs in a HIP kernel
(`mi355_anchors_generate`) and the result is cached per (H, W, device): anchors depend on the
image shape only (anchors.py:40).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops_detect


class Anchors(nn.Module):
    def __init__(self, pyramid_levels=None, strides=None, sizes=None, ratios=None, scales=None,
                 **kwargs):
        super(Anchors, self).__init__()
        self.pyramid_levels = [3, 4, 5, 6, 7] if pyramid_levels is None else pyramid_levels
        self.strides = [2 ** x for x in self.pyramid_levels] if strides is None else strides
        self.sizes = [2 ** (x + 2) for x in self.pyramid_levels] if sizes is None else sizes
        self.ratios = np.array([0.5, 1, 2]) if ratios is None else np.array(ratios)
        self.scales = (np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)]) if scales is None
                       else np.array(scales))
        self._cache = {}
        if 'logger' in kwargs and kwargs['logger'] is not None:      # anchors.py:19-22
            kwargs['logger'].info('Anchors Sizes   : {}'.format(str(self.sizes)))
            kwargs['logger'].info('Anchors Scales  : {}'.format(str(self.scales)))
            kwargs['logger'].info('Anchors Ratios  : {}'.format(str(self.ratios)))

    @property
    def num_anchors_per_cell(self):
        return len(self.ratios) * len(self.scales)

    def forward(self, image):
        h, w = int(image.shape[2]), int(image.shape[3])
        dev = image.device if image.is_cuda else torch.device('cuda', torch.cuda.current_device())
        key = (h, w, dev.index)
        out = self._cache.get(key)
        if out is None:
            out = ops_detect.anchors_generate(h, w, self.strides, self.sizes, self.ratios,
                                              self.scales, dev)
            self._cache = {key: out}
        return out
