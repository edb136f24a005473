"""Parameter containers mirroring ksevendet/architecture/backbone/resnet.py: `ResNetFeatures` (:482-630, the
features-only trunk returning [C3, C4, C5]) with `BasicBlock` (:116-176) / `Bottleneck` (:179-257) and the
entry points `resnet18/34/50/101/152` (:632-760).  Same attribute names / state_dict keys (conv1, bn1, act1,
conv2, bn2, [conv3, bn3], downsample.0 / .1); the arithmetic runs in the fused launch plan
(objectdetection/pytorch_amd/engine_kseven.py).  The plain variants only: deep stems, avg-down, SE / ECA
attention, drop-path / drop-block and ResNeXt widths are not built."""
import math

import torch.nn as nn

from ....architectures.utils_resnet import _PlanOnly
from .registry import register_model


class BasicBlock(_PlanOnly):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.se = None
        self.act2 = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(_PlanOnly):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=True)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.se = None
        self.act3 = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNetFeatures(_PlanOnly):
    def __init__(self, block, layers, in_chans=3, **kwargs):
        super(ResNetFeatures, self).__init__()
        unsupported = [k for k in ('stem_type', 'avg_down', 'cardinality', 'base_width', 'block_args', 'drop_rate',
                                   'drop_path_rate', 'drop_block_rate', 'output_stride') if kwargs.get(k)]
        if unsupported and not (unsupported == ['output_stride'] and kwargs['output_stride'] == 32):
            raise NotImplementedError('ResNetFeatures options %s are not built on the MI355X path' % unsupported)
        logger = kwargs.get('logger', None)
        if logger:
            logger.info('==== Build Backbone ======================')
            logger.info('Backbone   : {}'.format('resnet'))
        self.inplanes = 64
        self.expansion = block.expansion
        self.conv1 = nn.Conv2d(in_chans, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.act1 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        channels = [64, 128, 256, 512]
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.features_num = dict(('c%d' % idx, channels[i] * block.expansion) for i, idx in enumerate([2, 3, 4, 5]))
        self.out_indices = ['c3', 'c4', 'c5']
        for m in self.modules():                            # resnet.py:561-575
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def feature_channels(self):
        return [self.features_num[idx] for idx in self.out_indices]

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(                     # downsample_conv, resnet.py:260-271 (kernel 1)
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, padding=0, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def _build(block, layers, pretrained, kwargs):
    if pretrained:
        raise RuntimeError('pretrained=True fetches ImageNet weights over the network (resnet.py load_pretrained); '
                           'not available here: load a state_dict instead')
    if not kwargs.pop('features_only', False):
        raise NotImplementedError('only the features_only trunk (what KSevenDet builds, ksevendet.py:172-176) is built')
    kwargs.pop('backbone_feature_pyramid_level', None)
    kwargs.pop('num_classes', None)
    return ResNetFeatures(block, layers, **kwargs)


@register_model
def resnet18(pretrained=False, num_classes=1000, in_chans=3, **kwargs):
    return _build(BasicBlock, [2, 2, 2, 2], pretrained, dict(kwargs, in_chans=in_chans))


@register_model
def resnet34(pretrained=False, num_classes=1000, in_chans=3, **kwargs):
    return _build(BasicBlock, [3, 4, 6, 3], pretrained, dict(kwargs, in_chans=in_chans))


@register_model
def resnet50(pretrained=False, num_classes=1000, in_chans=3, **kwargs):
    return _build(Bottleneck, [3, 4, 6, 3], pretrained, dict(kwargs, in_chans=in_chans))


@register_model
def resnet101(pretrained=False, num_classes=1000, in_chans=3, **kwargs):
    return _build(Bottleneck, [3, 4, 23, 3], pretrained, dict(kwargs, in_chans=in_chans))


@register_model
def resnet152(pretrained=False, num_classes=1000, in_chans=3, **kwargs):
    return _build(Bottleneck, [3, 8, 36, 3], pretrained, dict(kwargs, in_chans=in_chans))
