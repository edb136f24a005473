"""Mirror of the parts of architectures/utils_efficientnet.py the hot path needs: block table
(:231-257), scaling rules (:62-83, :138-153), swish modules (:38-59; arithmetic in HIP).
`load_pretrained_weights` (:299-313) is a network fetch and is not provided."""
import collections
import math

import torch.nn as nn

from .utils_resnet import _PlanOnly

GlobalParams = collections.namedtuple('GlobalParams', [
    'batch_norm_momentum', 'batch_norm_epsilon', 'dropout_rate', 'num_classes', 'width_coefficient',
    'depth_coefficient', 'depth_divisor', 'min_depth', 'drop_connect_rate', 'image_size'])
BlockArgs = collections.namedtuple('BlockArgs', [
    'kernel_size', 'num_repeat', 'input_filters', 'output_filters', 'expand_ratio', 'id_skip', 'stride',
    'se_ratio'])
GlobalParams.__new__.__defaults__ = (None,) * len(GlobalParams._fields)
BlockArgs.__new__.__defaults__ = (None,) * len(BlockArgs._fields)


class MemoryEfficientSwish(_PlanOnly):
    pass


class Swish(_PlanOnly):
    pass


def round_filters(filters, global_params):
    multiplier = global_params.width_coefficient
    if not multiplier:
        return filters
    divisor = global_params.depth_divisor
    min_depth = global_params.min_depth
    filters *= multiplier
    min_depth = min_depth or divisor
    new_filters = max(min_depth, int(filters + divisor / 2) // divisor * divisor)
    if new_filters < 0.9 * filters:
        new_filters += divisor
    return int(new_filters)


def round_repeats(repeats, global_params):
    multiplier = global_params.depth_coefficient
    if not multiplier:
        return repeats
    return int(math.ceil(multiplier * repeats))


def efficientnet_params(model_name):
    return {'efficientnet-b0': (1.0, 1.0, 224, 0.2), 'efficientnet-b1': (1.0, 1.1, 240, 0.2),
            'efficientnet-b2': (1.1, 1.2, 260, 0.3), 'efficientnet-b3': (1.2, 1.4, 300, 0.3),
            'efficientnet-b4': (1.4, 1.8, 380, 0.4), 'efficientnet-b5': (1.6, 2.2, 456, 0.4),
            'efficientnet-b6': (1.8, 2.6, 528, 0.5), 'efficientnet-b7': (2.0, 3.1, 600, 0.5),
            'efficientnet-b8': (2.2, 3.6, 672, 0.5)}[model_name]


_BLOCKS = [(3, 1, 32, 16, 1, 1), (3, 2, 16, 24, 6, 2), (5, 2, 24, 40, 6, 2), (3, 3, 40, 80, 6, 2),
           (5, 3, 80, 112, 6, 1), (5, 4, 112, 192, 6, 2), (3, 1, 192, 320, 6, 1)]   # k, r, i, o, e, s


def get_model_params(model_name, override_params=None):
    w, d, s, p = efficientnet_params(model_name)
    blocks_args = [BlockArgs(kernel_size=k, num_repeat=r, input_filters=i, output_filters=o, expand_ratio=e,
                             id_skip=True, stride=[st], se_ratio=0.25) for (k, r, i, o, e, st) in _BLOCKS]
    gp = GlobalParams(batch_norm_momentum=0.99, batch_norm_epsilon=1e-3, dropout_rate=p, drop_connect_rate=0.2,
                      num_classes=1000, width_coefficient=w, depth_coefficient=d, depth_divisor=8, min_depth=None,
                      image_size=s)
    if override_params:
        gp = gp._replace(**override_params)
    return blocks_args, gp
