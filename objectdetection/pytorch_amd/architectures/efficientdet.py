"""Drop-in for architectures/efficientdet.py: `SeparableConvBlock` (:28-66), `BiFPN` (:69-317),
`Regressor` (:320-353), `Classifier` (:356-393), `EfficientNet` wrapper (:396-433), `EfficientDet`
(:436-576), `build_efficientdet` (:588-591).

Same constructor signatures, attribute tree and state_dict keys; the modules hold parameters, the
arithmetic is the HIP launch plan `engine_effdet.EffDetPlan` (depthwise / SE / BiFPN kernels of
include/mi355det_effdet.h + the MFMA gather conv for every 1x1 convolution).

Deterministic path: BatchNorm frozen (`freeze_bn`, efficientdet.py:523-526, as train.py:273-277
does) and drop-connect OFF (the reference draws `torch.rand` per sample in train mode,
utils_efficientnet.py:85-94; stochastic regularisation is not on the parity path — SURVEY §8c iii).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops_detect
from .anchors import Anchors
from .efficientnet import EfficientNet as EffNet
from .losses import FocalLoss
from .utils import BBoxTransform, ClipBoxes
from .utils_efficientnet import MemoryEfficientSwish, Swish
from .utils_extra_efficientnet import Conv2dStaticSamePadding, MaxPool2dStaticSamePadding
from .utils_resnet import _PlanOnly


class SeparableConvBlock(_PlanOnly):
    def __init__(self, in_channels, out_channels=None, norm=True, activation=False, onnx_export=False):
        super(SeparableConvBlock, self).__init__()
        if out_channels is None:
            out_channels = in_channels
        self.depthwise_conv = Conv2dStaticSamePadding(in_channels, in_channels, kernel_size=3, stride=1,
                                                      groups=in_channels, bias=False)
        self.pointwise_conv = Conv2dStaticSamePadding(in_channels, out_channels, kernel_size=1, stride=1)
        self.norm = norm
        if self.norm:
            self.bn = nn.BatchNorm2d(num_features=out_channels, momentum=0.01, eps=1e-3)
        self.activation = activation
        if self.activation:
            self.swish = MemoryEfficientSwish() if not onnx_export else Swish()


class BiFPN(_PlanOnly):
    def __init__(self, num_channels, conv_channels, first_time=False, epsilon=1e-4, onnx_export=False,
                 attention=True):
        super(BiFPN, self).__init__()
        self.epsilon = epsilon
        for name in ('conv6_up', 'conv5_up', 'conv4_up', 'conv3_up', 'conv4_down', 'conv5_down', 'conv6_down',
                     'conv7_down'):
            setattr(self, name, SeparableConvBlock(num_channels, onnx_export=onnx_export))
        for name in ('p6_upsample', 'p5_upsample', 'p4_upsample', 'p3_upsample'):
            setattr(self, name, nn.Upsample(scale_factor=2, mode='nearest'))
        for name in ('p4_downsample', 'p5_downsample', 'p6_downsample', 'p7_downsample'):
            setattr(self, name, MaxPool2dStaticSamePadding(3, 2))
        self.swish = MemoryEfficientSwish() if not onnx_export else Swish()
        self.first_time = first_time
        if self.first_time:
            def cb(cin):
                return nn.Sequential(Conv2dStaticSamePadding(cin, num_channels, 1),
                                     nn.BatchNorm2d(num_channels, momentum=0.01, eps=1e-3))
            self.p5_down_channel = cb(conv_channels[2])
            self.p4_down_channel = cb(conv_channels[1])
            self.p3_down_channel = cb(conv_channels[0])
            self.p5_to_p6 = nn.Sequential(Conv2dStaticSamePadding(conv_channels[2], num_channels, 1),
                                          nn.BatchNorm2d(num_channels, momentum=0.01, eps=1e-3),
                                          MaxPool2dStaticSamePadding(3, 2))
            self.p6_to_p7 = nn.Sequential(MaxPool2dStaticSamePadding(3, 2))
            self.p4_down_channel_2 = cb(conv_channels[1])
            self.p5_down_channel_2 = cb(conv_channels[2])
        for name, n in (('p6_w1', 2), ('p5_w1', 2), ('p4_w1', 2), ('p3_w1', 2), ('p4_w2', 3), ('p5_w2', 3),
                        ('p6_w2', 3), ('p7_w2', 2)):
            setattr(self, name, nn.Parameter(torch.ones(n, dtype=torch.float32), requires_grad=True))
            setattr(self, name + '_relu', nn.ReLU())
        self.attention = attention


class Regressor(_PlanOnly):
    def __init__(self, in_channels, num_anchors, num_layers, onnx_export=False):
        super(Regressor, self).__init__()
        self.num_layers = num_layers
        self.conv_list = nn.ModuleList(
            [SeparableConvBlock(in_channels, in_channels, norm=False, activation=False) for _ in range(num_layers)])
        self.bn_list = nn.ModuleList(
            [nn.ModuleList([nn.BatchNorm2d(in_channels, momentum=0.01, eps=1e-3) for _ in range(num_layers)])
             for _ in range(5)])
        self.header = SeparableConvBlock(in_channels, num_anchors * 4, norm=False, activation=False)
        self.swish = MemoryEfficientSwish() if not onnx_export else Swish()


class Classifier(_PlanOnly):
    def __init__(self, in_channels, num_anchors, num_classes, num_layers, onnx_export=False):
        super(Classifier, self).__init__()
        self.num_anchors = num_anchors
        self.num_classes = num_classes
        self.num_layers = num_layers
        self.conv_list = nn.ModuleList(
            [SeparableConvBlock(in_channels, in_channels, norm=False, activation=False) for _ in range(num_layers)])
        self.bn_list = nn.ModuleList(
            [nn.ModuleList([nn.BatchNorm2d(in_channels, momentum=0.01, eps=1e-3) for _ in range(num_layers)])
             for _ in range(5)])
        self.header = SeparableConvBlock(in_channels, num_anchors * num_classes, norm=False, activation=False)
        self.swish = MemoryEfficientSwish() if not onnx_export else Swish()


class EfficientNet(_PlanOnly):
    def __init__(self, compound_coef, load_weights=False):
        super(EfficientNet, self).__init__()
        self.model = EffNet.from_pretrained('efficientnet-b%d' % compound_coef, load_weights)


class _TrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, anchors, img, annot, *params):
        plan.run_heads(img)
        losses_ = plan.run_loss(anchors, annot)
        ctx.plan, ctx.serial = plan, plan.forward_serial
        ctx.inputs = (anchors, img, annot, tuple(p._version for p in plan.params))
        return losses_[0:1].clone(), losses_[1:2].clone()

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        if ctx.serial != ctx.plan.forward_serial:
            # the reference's autograd graph keeps per-call activations; the launch plan keeps ONE set per (batch, H, W):
            # another forward on the same shape (validation call, return_head, a second micro-batch) overwrote what this
            # backward needs - the forward of THIS batch is re-run first (engine.RetinaPlan.restore_forward; raises where
            # that is not possible)
            ctx.serial = ctx.plan.restore_forward(*ctx.inputs)
        grads = ctx.plan.run_backward(g_cls, g_reg)
        return (None, None, None, None) + tuple(grads)


class EfficientDet(nn.Module):
    def __init__(self, num_classes=80, compound_coef=0, load_weights=False, **kwargs):
        super(EfficientDet, self).__init__()
        logger = kwargs.get('logger', None)
        self.compound_coef = compound_coef
        self.backbone_compound_coef = [0, 1, 2, 3, 4, 5, 6, 6]
        self.fpn_num_filters = [64, 88, 112, 160, 224, 288, 384, 384]
        self.fpn_cell_repeats = [3, 4, 5, 6, 7, 7, 8, 8]
        self.input_sizes = [512, 640, 768, 896, 1024, 1280, 1280, 1536]
        self.box_class_repeats = [3, 3, 3, 4, 4, 4, 5, 5]
        self.anchor_scale = [4., 4., 4., 4., 4., 4., 4., 5.]
        self.aspect_ratios = kwargs.get('ratios', [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)])
        self.num_scales = len(kwargs.get('scales', [2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)]))
        conv_channel_coef = {0: [40, 112, 320], 1: [40, 112, 320], 2: [48, 120, 352], 3: [48, 136, 384],
                             4: [56, 160, 448], 5: [64, 176, 512], 6: [72, 200, 576], 7: [72, 200, 576]}
        my_pyramid_levels = [3, 4, 5, 6, 7]                       # efficientdet.py:461-468
        my_strides = [2 ** x for x in my_pyramid_levels]
        my_sizes = [2 ** (x + 1) * 1.25 for x in my_pyramid_levels]
        my_ratios = np.array([1, 2])
        my_scales = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)])
        num_anchors = len(my_ratios) * len(my_scales)
        self.bifpn = nn.Sequential(*[BiFPN(self.fpn_num_filters[compound_coef], conv_channel_coef[compound_coef],
                                           True if i == 0 else False, attention=True if compound_coef < 6 else False)
                                     for i in range(self.fpn_cell_repeats[compound_coef])])
        self.num_classes = num_classes
        self.regressor = Regressor(in_channels=self.fpn_num_filters[compound_coef], num_anchors=num_anchors,
                                   num_layers=self.box_class_repeats[compound_coef])
        self.classifier = Classifier(in_channels=self.fpn_num_filters[compound_coef], num_anchors=num_anchors,
                                     num_classes=num_classes, num_layers=self.box_class_repeats[compound_coef])
        anchor_kwargs = {'logger': logger} if logger is not None else {}
        self.anchors = Anchors(pyramid_levels=my_pyramid_levels, strides=my_strides, sizes=my_sizes,
                               ratios=my_ratios, scales=my_scales, **anchor_kwargs)
        self.focalLoss = FocalLoss()
        self.regressBoxes = BBoxTransform()
        self.clipBoxes = ClipBoxes()
        self.backbone_net = EfficientNet(self.backbone_compound_coef[compound_coef], load_weights)
        self._initialize_weights(logger=logger)
        self._channels_last_params()
        self.freeze_bn()
        self._plans = {}

    def _initialize_weights(self, logger=None):                   # efficientdet.py:506-521
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _channels_last_params(self):
        """Dense convs keep KRSC memory order (only the 3x3 stem differs from OIHW); depthwise
        weights stay [C,1,k,k] contiguous (the dw kernels re-lay them tap-major each step)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m.groups == 1:
                w = m.weight.data
                m.weight.data = w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def _plan(self, img):
        from .. import engine_effdet
        if not img.is_cuda:
            raise RuntimeError('mi355det EfficientDet: input on %s; the HIP path needs a ROCm GPU tensor '
                               '(there is no CPU fallback)' % img.device)
        # frozen BN (train.py:282-286, the reference loop) folds into the convs; layers left in train mode (or made
        # SynchronizedBatchNorm2d by utils.utils.replace_w_sync_bn) get batch-statistics passes in the plan
        if getattr(self, '_bn_list', None) is None:
            self._bn_list = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        bn_sig = tuple(m.training for m in self._bn_list)
        bn_sig = bn_sig if any(bn_sig) else ()
        # drop-connect runs while the MBConv blocks are in train mode (efficientnet.py:109-113 `training=self.training`)
        gp = self.backbone_net.model._global_params
        dc = bool(getattr(gp, 'drop_connect_rate', None)) and any(b.training for b in self.backbone_net.model._blocks)
        key = (img.shape[0], img.shape[2], img.shape[3], img.device.index, dc)
        from .. import engine
        if engine.precision_of(self) != 'bf16':          # `model.precision = 'fp32'`: the fp32-accurate plan
            key = key + (engine.precision_of(self),)
        if bn_sig:
            key = key + (bn_sig,)
        from .retinanet import _cached_plan

        def build():
            self._plan_drop_connect = dc
            return engine_effdet.EffDetPlan(self, img.shape[0], img.shape[2], img.shape[3], img.device)
        return _cached_plan(self, key, build)

    def forward(self, images, annotations=None, return_loss=True, return_head=False):
        """efficientdet.py:528-576 (`return_head` is an extension mirroring RetinaNet's: returns
        [classification, regression])."""
        plan = self._plan(images)
        if return_head:
            with torch.no_grad():
                plan.run_heads(images)
            return [plan.cls.clone(), plan.reg.clone()]
        anchors = self.anchors(images)
        if return_loss:
            params = [p for p in self.parameters()]
            return _TrainStep.apply(plan, anchors, images, annotations, *params)
        with torch.no_grad():
            plan.run_heads(images)
            boxes = self.regressBoxes(anchors, plan.reg)
            boxes = self.clipBoxes(boxes, images)
            score, label = ops_detect.class_max(plan.cls[0])
            idx = ops_detect.score_filter(score, 0.05)
            if idx.numel() == 0:
                dev = images.device
                return [torch.zeros([1], device=dev), torch.zeros([1], device=dev), torch.zeros([1, 4], device=dev)]
            idx = idx.long()
            keep = ops_detect.nms(boxes[0, idx, :], score[idx], 0.5)
            sel = idx[keep]
            return [score[sel], label[sel].long(), boxes[0, sel, :]]


def build_efficientdet(architecture, num_classes=80, **kwargs):
    compound_coef = int(architecture.split('-')[1][1])
    return EfficientDet(num_classes=num_classes, compound_coef=compound_coef, **kwargs)
