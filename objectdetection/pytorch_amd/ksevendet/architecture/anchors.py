"""Drop-in for ksevendet/architecture/anchors.py:8-24 (`Anchors`): the architectures/anchors.py class plus the
YAML conveniences of the ksevendet configs — `sizes_mapper` (a lambda in a string, :14-16) and `scales` given
as a string (:20-21) are evaluated as the reference does; generation itself is the same fp64 HIP kernel."""
import numpy as np

from ...architectures.anchors import Anchors as _Anchors


class Anchors(_Anchors):
    def __init__(self, pyramid_levels=None, strides=None, sizes=None, ratios=None, scales=None, **kwargs):
        levels = [3, 4, 5, 6, 7] if pyramid_levels is None else pyramid_levels
        if 'sizes_mapper' in kwargs:                       # anchors.py:14-16
            mapper = eval(kwargs['sizes_mapper'])
            sizes = [mapper(x) for x in levels]
        if isinstance(scales, str):                        # anchors.py:20-21
            scales = eval(scales)
        super(Anchors, self).__init__(pyramid_levels=pyramid_levels, strides=strides, sizes=sizes, ratios=ratios,
                                      scales=None if scales is None else np.array(scales),
                                      logger=kwargs.get('logger', None))
