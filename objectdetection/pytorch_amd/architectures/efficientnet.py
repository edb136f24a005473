"""Parameter containers mirroring architectures/efficientnet.py (`MBConvBlock` :30-119,
`EfficientNet` :122-252): same attribute names and state_dict keys."""
import torch.nn as nn

from .utils_efficientnet import (MemoryEfficientSwish, get_model_params, round_filters, round_repeats)
from .utils_extra_efficientnet import Conv2dStaticSamePadding as Conv2d
from .utils_resnet import _PlanOnly


class MBConvBlock(_PlanOnly):
    def __init__(self, block_args, global_params):
        super().__init__()
        self._block_args = block_args
        self._bn_mom = 1 - global_params.batch_norm_momentum
        self._bn_eps = global_params.batch_norm_epsilon
        self.has_se = (block_args.se_ratio is not None) and (0 < block_args.se_ratio <= 1)
        self.id_skip = block_args.id_skip
        inp = block_args.input_filters
        oup = block_args.input_filters * block_args.expand_ratio
        if block_args.expand_ratio != 1:
            self._expand_conv = Conv2d(in_channels=inp, out_channels=oup, kernel_size=1, bias=False)
            self._bn0 = nn.BatchNorm2d(num_features=oup, momentum=self._bn_mom, eps=self._bn_eps)
        self._depthwise_conv = Conv2d(in_channels=oup, out_channels=oup, groups=oup,
                                      kernel_size=block_args.kernel_size, stride=block_args.stride, bias=False)
        self._bn1 = nn.BatchNorm2d(num_features=oup, momentum=self._bn_mom, eps=self._bn_eps)
        if self.has_se:
            nsq = max(1, int(block_args.input_filters * block_args.se_ratio))
            self._se_reduce = Conv2d(in_channels=oup, out_channels=nsq, kernel_size=1)
            self._se_expand = Conv2d(in_channels=nsq, out_channels=oup, kernel_size=1)
        self._project_conv = Conv2d(in_channels=oup, out_channels=block_args.output_filters, kernel_size=1, bias=False)
        self._bn2 = nn.BatchNorm2d(num_features=block_args.output_filters, momentum=self._bn_mom, eps=self._bn_eps)
        self._swish = MemoryEfficientSwish()


class EfficientNet(_PlanOnly):
    def __init__(self, blocks_args=None, global_params=None):
        super().__init__()
        self._global_params = global_params
        self._blocks_args = blocks_args
        bn_mom = 1 - global_params.batch_norm_momentum
        bn_eps = global_params.batch_norm_epsilon
        out_channels = round_filters(32, global_params)
        self._conv_stem = Conv2d(3, out_channels, kernel_size=3, stride=2, bias=False)
        self._bn0 = nn.BatchNorm2d(num_features=out_channels, momentum=bn_mom, eps=bn_eps)
        self._blocks = nn.ModuleList([])
        for block_args in blocks_args:
            block_args = block_args._replace(
                input_filters=round_filters(block_args.input_filters, global_params),
                output_filters=round_filters(block_args.output_filters, global_params),
                num_repeat=round_repeats(block_args.num_repeat, global_params))
            self._blocks.append(MBConvBlock(block_args, global_params))
            if block_args.num_repeat > 1:
                block_args = block_args._replace(input_filters=block_args.output_filters, stride=1)
            for _ in range(block_args.num_repeat - 1):
                self._blocks.append(MBConvBlock(block_args, global_params))
        # the classifier head (_conv_head, _bn1, _avg_pooling, _dropout, _fc) is deleted by the
        # EfficientDet wrapper (efficientdet.py:404-408) and never built here
        self._swish = MemoryEfficientSwish()

    @classmethod
    def from_name(cls, model_name, override_params=None):
        blocks_args, global_params = get_model_params(model_name, override_params)
        return cls(blocks_args, global_params)

    @classmethod
    def from_pretrained(cls, model_name, load_weights=True, advprop=True, num_classes=1000, in_channels=3):
        if load_weights:
            raise RuntimeError('load_weights=True downloads ImageNet weights (utils_efficientnet.py:299-313); '
                               'not available here: load a state_dict instead')
        return cls.from_name(model_name, override_params={'num_classes': num_classes})
