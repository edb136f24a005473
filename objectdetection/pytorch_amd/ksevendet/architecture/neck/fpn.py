"""Parameter container mirroring ksevendet/architecture/neck/fpn.py:87-299 (`PANetFPN`) and :302-346
(`topdown_lateral_module`): same attribute names / state_dict keys (conv_top, topdown_lateral_modules.i.conv_lateral,
posthoc_modules.i, [panet_buttomup_conv1_modules.i, panet_buttomup_conv2_modules.i], extra_pyramid_modules.i).
Without GroupNorm (the reference's default, `group_norm=False`).  The dataflow (fpn.py:235-290) runs in the launch
plan: top-down 1x1 laterals + nearest-2x upsample-add as the gather conv's addend, 3x3 post-hoc convs, optional
PANet bottom-up path (3x3 s2 + add + 3x3), extra levels P6 = conv3x3 s2(C5), P7 = conv3x3 s2(ReLU(P6))."""
import math

import torch.nn as nn

from ....architectures.utils_resnet import _PlanOnly


class topdown_lateral_module(_PlanOnly):
    def __init__(self, dim_in_top, dim_in_lateral):
        super(topdown_lateral_module, self).__init__()
        self.dim_in_top, self.dim_in_lateral, self.dim_out = dim_in_top, dim_in_lateral, dim_in_top
        self.group_norm = False
        self.conv_lateral = nn.Conv2d(dim_in_lateral, self.dim_out, 1, 1, 0)
        nn.init.xavier_normal_(self.conv_lateral.weight)     # fpn.py:330
        nn.init.constant_(self.conv_lateral.bias, 0)


class PANetFPN(_PlanOnly):
    def __init__(self, backbone_features_num, in_pyramid_levels=[3, 4, 5], out_pyramid_levels=[3, 4, 5, 6, 7],
                 features_num=256, panet_buttomup=False, group_norm=False, **kwargs):
        super(PANetFPN, self).__init__()
        if group_norm:
            raise NotImplementedError('PANetFPN(group_norm=True) is not built on the MI355X path')
        self.convert_onnx = False
        self.pyramid_sizes = None
        self.panet_buttomup = panet_buttomup
        self.features_num = features_num
        self.in_pyramid_levels = in_pyramid_levels
        self.out_pyramid_levels = out_pyramid_levels
        self.backbone_features_num = backbone_features_num[::-1]
        logger = kwargs.get('logger', None)
        if logger:
            logger.info('==== Build Neck Layer ====================')
            logger.info('Neck : {}'.format('FPN' if not panet_buttomup else 'PANet-FPN'))
            logger.info('FPN Features : {}'.format(features_num))
        self.conv_top = nn.Conv2d(self.backbone_features_num[0], features_num, 1, 1, 0)
        self.topdown_lateral_modules = nn.ModuleList()
        self.posthoc_modules = nn.ModuleList()
        for i in range(1, len(in_pyramid_levels)):
            self.topdown_lateral_modules.append(topdown_lateral_module(features_num, self.backbone_features_num[i]))
        for i in range(len(in_pyramid_levels)):
            self.posthoc_modules.append(nn.Conv2d(features_num, features_num, 3, 1, 1))
        if panet_buttomup:
            self.panet_buttomup_conv1_modules = nn.ModuleList()
            self.panet_buttomup_conv2_modules = nn.ModuleList()
            for i in range(len(in_pyramid_levels) - 1):
                self.panet_buttomup_conv1_modules.append(nn.Conv2d(features_num, features_num, 3, 2, 1))
                self.panet_buttomup_conv2_modules.append(nn.Conv2d(features_num, features_num, 3, 1, 1))
        self.extra_levels = out_pyramid_levels[-1] > in_pyramid_levels[-1]
        if self.extra_levels:
            self.extra_pyramid_modules = nn.ModuleList()
            dim_in = self.backbone_features_num[0]
            for i in range(in_pyramid_levels[-1], out_pyramid_levels[-1]):
                self.extra_pyramid_modules.append(nn.Conv2d(dim_in, features_num, 3, 2, 1))
                dim_in = features_num
        for m in self.modules():                            # fpn.py:196-206 (runs after the lateral modules' own init)
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()

    def set_onnx_convert_info(self, pyramid_sizes):
        self.convert_onnx = True
        self.pyramid_sizes = pyramid_sizes
