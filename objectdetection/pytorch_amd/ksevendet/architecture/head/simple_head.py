"""Parameter containers mirroring ksevendet/architecture/head/simple_head.py: `Regressor` (:78-174) and
`Classifier` (:177-315) in their default form (head_type 'simple', act_type 'relu'): `layers_num` x
[3x3 conv -> PER-LEVEL BatchNorm(momentum 0.01, eps 1e-3) -> ReLU] + a 3x3 header conv; the classifier ends in a
sigmoid (:313).  `share_weights=True` (config/model/ksevendet_resnet.yaml): one tower and one header shared by the pyramid
levels; `share_weights=False` (ksevendet_resnet_bifpn.yaml; :120-127): a tower and a header per level.  State_dict keys as
the reference: conv_tower.t.i, header.t, bn_modules.level.layer (t = 0, or the level).  The arithmetic (per-level frozen-BN fold over the shared weights, or
batch-statistics passes when the BN is in train mode) runs in engine_kseven.KSevenPlan."""
import math

import torch.nn as nn

from ....architectures.utils_resnet import _PlanOnly


class _SimpleHead(_PlanOnly):
    def __init__(self, in_features_num, out_channels, what, features_num=256, layers_num=3, num_pyramid_levels=5,
                 head_type='simple', act_type='relu', share_weights=True, conv_kernel_size=3, conv_stride=1,
                 conv_padding=1, **kwargs):
        assert head_type in ['simple', 'efficient']
        assert act_type in ['relu', 'swish']
        super(_SimpleHead, self).__init__()
        if head_type != 'simple' or act_type != 'relu':
            raise NotImplementedError("only head_type='simple', act_type='relu' (config/model/ksevendet_resnet*.yaml) are "
                                      "built on the MI355X path")
        if conv_kernel_size != 3 or conv_stride != 1 or conv_padding != 1:
            raise NotImplementedError('head convs other than 3x3 / stride 1 / pad 1 are not built')
        self.convert_onnx = False
        self.layers_num = layers_num
        self.num_pyramid_levels = num_pyramid_levels
        self.head_type = head_type
        self.share_weights = share_weights
        logger = kwargs.get('logger', None)
        if logger:
            logger.info('==== Build Head Layer ====================')
            logger.info('Head Type     : %s (%s + %s)' % (what, head_type, act_type))
            logger.info('Features Num  : %d' % features_num)
            logger.info('Layers Num    : %d' % layers_num)
        ntow = 1 if share_weights else num_pyramid_levels    # simple_head.py:116-127
        self.conv_tower = nn.ModuleList([nn.ModuleList(
            [nn.Conv2d(in_features_num if i == 0 else features_num, features_num, kernel_size=3, stride=1, padding=1)
             for i in range(layers_num)]) for _ in range(ntow)])
        self.header = nn.ModuleList([nn.Conv2d(features_num, out_channels, kernel_size=3, stride=1, padding=1)
                                     for _ in range(ntow)])
        self.bn_modules = nn.ModuleList(
            [nn.ModuleList([nn.BatchNorm2d(features_num, momentum=0.01, eps=1e-3) for i in range(layers_num)])
             for j in range(num_pyramid_levels)])
        self.act_fn = nn.ReLU()
        for m in self.modules():                            # simple_head.py:131-141
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


class Regressor(_SimpleHead):
    def __init__(self, in_features_num, num_anchors=9, **kwargs):
        super(Regressor, self).__init__(in_features_num, num_anchors * 4, 'Regression', **kwargs)


class Classifier(_SimpleHead):
    def __init__(self, in_features_num, num_anchors=9, num_classes=80, **kwargs):
        super(Classifier, self).__init__(in_features_num, num_anchors * num_classes, 'Classification', **kwargs)
        self.pyramid_sizes = None
        self.num_anchors = num_anchors
        self.num_classes = num_classes
        self.header_act = nn.Sigmoid()
