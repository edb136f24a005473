"""Parameter containers mirroring ksevendet/architecture/neck/bifpn.py: `SeparableConvBlock` (:28-73), `BiFPN` (:75-135)
and `BiFPNModule` (:137-285) — same constructor arguments, attribute names and state_dict keys (bifpn_modules.i.conv_up.j.
{depthwise_conv, pointwise_conv, bn}, conv_down.j, down_channel.j.{0, 1}, higher_pyramid_conv.0.{0, 1}, down_channel_two.j.{0, 1}).
The dataflow of `BiFPNModule._forward` (:303-347; plain sums, ReLU, 3x3 / stride-2 / pad-1 max-pool, nearest 2x upsample) runs
in the launch plan (engine_kseven_bifpn.KSevenBiFPNPlan).  `attention=True` is refused: the reference's
`_forward_fast_attention` reads `self.down_channel_2` (:372), an attribute `__init__` never creates (it builds
`down_channel_two`, :243), so that path cannot run there either."""
import math

import torch.nn as nn

from ....architectures.utils_resnet import _PlanOnly


class SeparableConvBlock(_PlanOnly):
    """bifpn.py:28-73: depthwise 3x3 / pad 1 (no bias) -> pointwise 1x1 (+ bias) -> BatchNorm(momentum 0.01, eps 1e-3) -> ReLU."""

    def __init__(self, in_channels, out_channels=None, norm=True, activation=True, act_func='relu', onnx_export=True):
        super(SeparableConvBlock, self).__init__()
        if out_channels is None:
            out_channels = in_channels
        if act_func != 'relu':
            raise NotImplementedError("SeparableConvBlock(act_func='swish') is not built on the MI355X path")
        self.depthwise_conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1, bias=False,
                                        groups=in_channels)
        self.pointwise_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)
        self.norm = norm
        if self.norm:
            self.bn = nn.BatchNorm2d(num_features=out_channels, momentum=0.01, eps=1e-3)
        self.activation = activation
        if self.activation:
            self.act_func = nn.ReLU()


class BiFPNModule(_PlanOnly):
    def __init__(self, backbone_features_num, in_pyramid_levels=[3, 4, 5], neck_pyramid_levels=[3, 4, 5, 6, 7],
                 out_pyramid_levels=[3, 4, 5, 6, 7], features_num=64, attention=False, act_func='relu', index=-1,
                 first_time=False, last_time=False, epsilon=1e-4, onnx_export=True, **kwargs):
        assert act_func in ['relu', 'swish'], 'Unknown act_func: %s' % act_func
        higher = len(neck_pyramid_levels) - len(in_pyramid_levels)
        assert higher <= 2, 'Higher pyramid levels num is too large.'
        assert higher >= 1, 'Higher pyramid levels num is too small.'
        super(BiFPNModule, self).__init__()
        if attention:
            raise NotImplementedError('BiFPNModule(attention=True): the reference path reads an attribute that does not '
                                      'exist (bifpn.py:372 `self.down_channel_2`); only the plain-sum form is built')
        if act_func != 'relu':
            raise NotImplementedError("BiFPNModule(act_func='swish') is not built on the MI355X path")
        self.convert_onnx = False
        self.pyramid_sizes = None
        self.backbone_features_num = backbone_features_num
        self.in_pyramid_levels = in_pyramid_levels
        self.neck_pyramid_levels = neck_pyramid_levels
        self.out_pyramid_levels = out_pyramid_levels
        self.features_num = features_num
        self.attention = attention
        self.index = index
        self.first_time = first_time
        self.last_time = last_time
        self.epsilon = epsilon
        self.higher_pyramid_levels_num = higher
        self.in_pyramid_num = len(in_pyramid_levels)
        self.neck_pyramid_num = len(neck_pyramid_levels)
        self.out_pyramid_num = len(out_pyramid_levels)
        self.out_neck_pyramid_idx = [i for i, p in enumerate(neck_pyramid_levels) if p in out_pyramid_levels]
        logger = kwargs.get('logger', None)
        if logger:
            logger.info('---- Build BiFPN Module [%d] ----' % index)
            logger.info('Features Num: %d  First: %s  Last: %s' % (features_num, first_time, last_time))
        self.conv_up = nn.ModuleList([SeparableConvBlock(features_num, activation=True, act_func=act_func)
                                      for _ in range(self.neck_pyramid_num - 1)])
        ndown = self.neck_pyramid_num - 1 if not last_time else self.out_pyramid_num - 1        # bifpn.py:205
        self.conv_down = nn.ModuleList([SeparableConvBlock(features_num, activation=True, act_func=act_func)
                                        for _ in range(ndown)])
        self.up_sample = nn.ModuleList([nn.Upsample(scale_factor=2, mode='nearest') for _ in range(self.neck_pyramid_num - 1)])
        self.down_sample = nn.MaxPool2d(3, stride=2, padding=1)
        self.act_func = nn.ReLU()
        if self.first_time:
            self.down_channel = nn.ModuleList([nn.Sequential(
                nn.Conv2d(backbone_features_num[i], features_num, 1),
                nn.BatchNorm2d(features_num, momentum=0.01, eps=1e-3)) for i in range(self.in_pyramid_num)])
            self.higher_pyramid_conv = nn.ModuleList()
            self.higher_pyramid_conv.append(nn.Sequential(
                nn.Conv2d(backbone_features_num[2], features_num, 1),
                nn.BatchNorm2d(features_num, momentum=0.01, eps=1e-3),
                nn.MaxPool2d(3, stride=2, padding=1)))
            if higher >= 2:
                self.higher_pyramid_conv.append(nn.MaxPool2d(3, stride=2, padding=1))
            self.down_channel_two = nn.ModuleList([nn.Sequential(
                nn.Conv2d(backbone_features_num[i + 1], features_num, 1),
                nn.BatchNorm2d(features_num, momentum=0.01, eps=1e-3)) for i in range(self.in_pyramid_num - 1)])
        for m in self.modules():                            # bifpn.py:271-283
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


class BiFPN(_PlanOnly):
    def __init__(self, backbone_features_num, in_pyramid_levels=[3, 4, 5], neck_pyramid_levels=[3, 4, 5, 6, 7],
                 out_pyramid_levels=[3, 4, 5, 6, 7], features_num=64, repeats=3, attention=False, act_func='relu',
                 epsilon=1e-4, onnx_export=True, **kwargs):
        assert act_func in ['relu', 'swish'], 'Unknown act_func: %s' % act_func
        super(BiFPN, self).__init__()
        self.convert_onnx = False
        self.pyramid_sizes = None
        self.panet_buttomup = False
        self.in_pyramid_num = len(in_pyramid_levels)
        self.neck_pyramid_num = len(neck_pyramid_levels)
        self.out_pyramid_num = len(out_pyramid_levels)
        self.out_neck_pyramid_idx = [i for i, p in enumerate(neck_pyramid_levels) if p in out_pyramid_levels]
        logger = kwargs.get('logger', None)
        if logger:
            logger.info('==== Build Neck Layer ====================')
            logger.info('Build Neck: BiFPN  Repeats: %d  Features Num: %d  Attention: %s' % (repeats, features_num, attention))
        self.bifpn_modules = nn.Sequential(*[BiFPNModule(
            backbone_features_num, in_pyramid_levels=in_pyramid_levels, neck_pyramid_levels=neck_pyramid_levels,
            out_pyramid_levels=out_pyramid_levels if idx == repeats - 1 else neck_pyramid_levels,
            features_num=features_num, index=idx, first_time=idx == 0, last_time=idx == repeats - 1,
            attention=attention, logger=logger) for idx in range(repeats)])

    def set_onnx_convert_info(self, pyramid_sizes):
        self.convert_onnx = True
        self.pyramid_sizes = pyramid_sizes
