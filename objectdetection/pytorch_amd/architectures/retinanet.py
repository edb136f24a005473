"""Drop-in for architectures/retinanet.py: `RetinaNet` (:181-318), `PyramidFeatures` (:23-69),
`RegressionModel` (:72-115), `ClassificationModel` (:118-178) and the factories (:371-421, :925).

Same constructor signatures, attribute tree and state_dict keys (OIHW fp32 shapes), so
`train.py:179` / `load_state_dict` / `init_weights` (utils/utils.py:204-215) / `freeze_bn` work
unchanged.  The modules only HOLD parameters; `RetinaNet.forward` runs the whole
backbone -> FPN -> heads -> anchors -> loss (or decode + NMS) chain as one pre-built HIP launch
plan (engine.RetinaPlan) and exposes it to autograd as a single node whose backward is the
hand-scheduled HIP backward pass.
"""
import math

import torch
import torch.nn as nn

from .. import engine, ops_detect
from . import losses
from .anchors import Anchors
from .utils import BBoxTransform, ClipBoxes
from .utils_resnet import BasicBlock, Bottleneck, _PlanOnly


class PyramidFeatures(_PlanOnly):
    def __init__(self, C3_size, C4_size, C5_size, feature_size=256):
        super(PyramidFeatures, self).__init__()
        self.P5_1 = nn.Conv2d(C5_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P5_upsampled = nn.Upsample(scale_factor=2, mode='nearest')
        self.P5_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P4_1 = nn.Conv2d(C4_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P4_upsampled = nn.Upsample(scale_factor=2, mode='nearest')
        self.P4_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P3_1 = nn.Conv2d(C3_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P3_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P6 = nn.Conv2d(C5_size, feature_size, kernel_size=3, stride=2, padding=1)
        self.P7_1 = nn.ReLU()
        self.P7_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=2, padding=1)


class RegressionModel(_PlanOnly):
    def __init__(self, num_features_in, num_anchors=9, feature_size=256):
        super(RegressionModel, self).__init__()
        self.convert_onnx = False
        self.fixed_size = None
        self.conv1 = nn.Conv2d(num_features_in, feature_size, kernel_size=3, padding=1)
        self.act1 = nn.ReLU()
        self.conv2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
        self.act2 = nn.ReLU()
        self.conv3 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
        self.act3 = nn.ReLU()
        self.conv4 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
        self.act4 = nn.ReLU()
        self.output = nn.Conv2d(feature_size, num_anchors * 4, kernel_size=3, padding=1)


class ClassificationModel(_PlanOnly):
    def __init__(self, num_features_in, num_anchors=9, num_classes=80, prior=0.01, feature_size=256):
        super(ClassificationModel, self).__init__()
        self.convert_onnx = False
        self.fixed_size = None
        self.num_classes = num_classes
        self.num_anchors = num_anchors
        self.conv1 = nn.Conv2d(num_features_in, feature_size, kernel_size=3, padding=1)
        self.act1 = nn.ReLU()
        self.conv2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
        self.act2 = nn.ReLU()
        self.conv3 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
        self.act3 = nn.ReLU()
        self.conv4 = nn.Conv2d(feature_size, feature_size, kernel_size=3, padding=1)
        self.act4 = nn.ReLU()
        self.output = nn.Conv2d(feature_size, num_anchors * num_classes, kernel_size=3, padding=1)
        self.output_act = nn.Sigmoid()


def _cached_plan(model, key, build):
    """Launch plans are cached per (batch, H, W, device): the reference's Resizer / collater hand over a
    different padded shape almost every batch (datasettool/dataloader.py:655-691, :827-902), and a plan costs
    seconds to build.  Least-recently-used plans are dropped beyond MI355_MAX_PLANS (default 4; each owns its
    activations: ~9 GB at R50 512^2 bs16).  A plan whose parameters were re-homed is rebuilt."""
    import os
    plans = model._plans
    plan = plans.pop(key, None)
    if plan is None or plan.stale():
        if plan is not None:                          # parameters moved: every cached plan points at dead storage
            plans.clear()
        plan = build()
    plans[key] = plan                                 # (re-)insert as most recently used
    limit = max(1, int(os.environ.get('MI355_MAX_PLANS', '4')))
    while len(plans) > limit:
        plans.pop(next(iter(plans)))
    return plan


class _TrainStep(torch.autograd.Function):
    """(img, annot, *params) -> (cls_loss [1], reg_loss [1]); backward = the HIP backward plan."""

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


class RetinaNet(nn.Module):
    _plan_class = engine.RetinaPlan

    def __init__(self, num_classes, block, layers, iou_threshold=0.5, **kwargs):
        self.inplanes = 64
        super(RetinaNet, self).__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.iou_threshold = iou_threshold          # stored, never used (retinanet.py:194)
        self.convert_onnx = False
        self.is_training = False
        if block == BasicBlock:
            fpn_sizes = [self.layer2[layers[1] - 1].conv2.out_channels,
                         self.layer3[layers[2] - 1].conv2.out_channels,
                         self.layer4[layers[3] - 1].conv2.out_channels]
        elif block == Bottleneck:
            fpn_sizes = [self.layer2[layers[1] - 1].conv3.out_channels,
                         self.layer3[layers[2] - 1].conv3.out_channels,
                         self.layer4[layers[3] - 1].conv3.out_channels]
        else:
            raise ValueError('Block type %s not understood' % block)
        self.fpn = PyramidFeatures(fpn_sizes[0], fpn_sizes[1], fpn_sizes[2])
        self.regressionModel = RegressionModel(256)
        self.classificationModel = ClassificationModel(256, num_classes=num_classes)
        self.anchors = Anchors()
        self.regressBoxes = BBoxTransform()
        self.clipBoxes = ClipBoxes()
        self.focalLoss = losses.FocalLoss()
        self._finish_init()

    def _finish_init(self):
        for m in self.modules():                    # retinanet.py:232-238
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        prior = 0.01                                # retinanet.py:240-246
        self.classificationModel.output.weight.data.fill_(0)
        self.classificationModel.output.bias.data.fill_(-math.log((1.0 - prior) / prior))
        self.regressionModel.output.weight.data.fill_(0)
        self.regressionModel.output.bias.data.fill_(0)
        self._channels_last_params()
        self.freeze_bn()
        self._plans = {}

    def _channels_last_params(self):
        """fp32 masters keep the reference's OIHW *shape* (state_dict compatible) but live in
        KRSC memory order: the layout the HIP weight-prep / wgrad kernels stream."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                w = m.weight.data
                m.weight.data = w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def freeze_bn(self):
        """retinanet.py:266-270: every BatchNorm2d in eval mode (gamma/beta stay trainable)."""
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    # ------------------------------------------------------------------ plan cache
    def _plan(self, img):
        if not img.is_cuda:
            raise RuntimeError('mi355det RetinaNet: input on %s; the HIP path needs a ROCm GPU tensor '
                               '(there is no CPU fallback)' % img.device)
        if getattr(self, '_bn_list', None) is None:          # (module objects are stable; `.training` is read live)
            self._bn_list = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        # frozen BN (train.py:273-277, the reference loop) is folded into the convs; layers left in train mode (or
        # made SynchronizedBatchNorm2d by utils.utils.replace_w_sync_bn) get batch-statistics passes in the plan
        bn_sig = tuple(m.training for m in self._bn_list)
        bn_sig = bn_sig if any(bn_sig) else ()
        w = self.conv1.weight
        if w.device != img.device:
            raise RuntimeError('model on %s, input on %s' % (w.device, img.device))
        if any(isinstance(m, nn.Conv2d) and not m.weight.permute(0, 2, 3, 1).is_contiguous()
               for m in self.modules()) if not self._plans else False:
            self._channels_last_params()
        key = (img.shape[0], img.shape[2], img.shape[3], img.device.index)
        if engine.precision_of(self) != 'bf16':          # (bf16 keys stay 4-tuples)
            key = key + (engine.precision_of(self),)
        if bn_sig:
            key = key + (bn_sig,)
        return _cached_plan(self, key, lambda: self._plan_class(self, img.shape[0], img.shape[2], img.shape[3],
                                                                img.device))

    def forward(self, img_batch, annotations=None, return_head=False, return_loss=True):
        plan = self._plan(img_batch)
        if return_head:                            # retinanet.py:289-290
            with torch.no_grad():
                plan.run_heads(img_batch)
            return [plan.cls.clone(), plan.reg.clone()]
        anchors = self.anchors(img_batch)
        if return_loss:                            # retinanet.py:294-295
            params = [p for p in self.parameters()]
            return _TrainStep.apply(plan, anchors, img_batch, annotations, *params)
        # retinanet.py:297-318 (batch size 1: the mask of image 0 is applied, :302)
        with torch.no_grad():
            plan.run_heads(img_batch)
            classification, regression = plan.cls, plan.reg
            boxes = self.regressBoxes(anchors, regression)
            boxes = self.clipBoxes(boxes, img_batch)
            score, label = ops_detect.class_max(classification[0])
            idx = ops_detect.score_filter(score, 0.05)
            if idx.numel() == 0:                   # retinanet.py:304-308 (dummy shapes kept)
                dev = img_batch.device
                return [torch.zeros([1], device=dev), torch.zeros([1], device=dev),
                        torch.zeros([1, 4], device=dev)]
            idx = idx.long()
            cand_boxes = boxes[0, idx, :]
            cand_scores = score[idx]
            keep = ops_detect.nms(cand_boxes, cand_scores, 0.5)
            sel = idx[keep]
            return [score[sel], label[sel].long(), boxes[0, sel, :]]


def retinanet(depth, num_classes=80, pretrained=False, **kwargs):
    assert depth in [18, 34, 50, 101, 152]
    return {18: retinanet18, 34: retinanet34, 50: retinanet50, 101: retinanet101,
            152: retinanet152}[depth](num_classes, pretrained=pretrained, **kwargs)


def _no_fetch(pretrained):
    if pretrained:
        raise RuntimeError('pretrained=True fetches ImageNet weights over the network '
                           '(retinanet.py:393-442); not available here: load a state_dict instead')


def retinanet18(num_classes, pretrained=False, **kwargs):
    _no_fetch(pretrained)
    return RetinaNet(num_classes, BasicBlock, [2, 2, 2, 2], **kwargs)


def retinanet34(num_classes, pretrained=False, **kwargs):
    _no_fetch(pretrained)
    return RetinaNet(num_classes, BasicBlock, [3, 4, 6, 3], **kwargs)


def retinanet50(num_classes, pretrained=False, **kwargs):
    _no_fetch(pretrained)
    return RetinaNet(num_classes, Bottleneck, [3, 4, 6, 3], **kwargs)


def retinanet101(num_classes, pretrained=False, **kwargs):
    _no_fetch(pretrained)
    return RetinaNet(num_classes, Bottleneck, [3, 4, 23, 3], **kwargs)


def retinanet152(num_classes, pretrained=False, **kwargs):
    _no_fetch(pretrained)
    return RetinaNet(num_classes, Bottleneck, [3, 8, 36, 3], **kwargs)



# ---------------------------------------------------------------------------------------------
# RetinaNetTiny (retinanet.py:447-679) and RetinaNetP45P6 (retinanet.py:682-922): three pyramid levels

class PyramidFeaturesTiny(_PlanOnly):
    def __init__(self, C2_size, C3_size, feature_size=256):
        super(PyramidFeaturesTiny, self).__init__()
        self.convert_onnx = False
        self.pyramid_sizes = None
        self.P3_1 = nn.Conv2d(C3_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P3_upsampled = nn.Upsample(scale_factor=2, mode='nearest')
        self.P3_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P2_1 = nn.Conv2d(C2_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P2_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P4 = nn.Conv2d(C3_size, feature_size, kernel_size=3, stride=2, padding=1)


class PyramidFeaturesP45P6(_PlanOnly):
    def __init__(self, C4_size, C5_size, feature_size=256):
        super(PyramidFeaturesP45P6, self).__init__()
        self.convert_onnx = False
        self.pyramid_sizes = None
        self.P5_1 = nn.Conv2d(C5_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P5_upsampled = nn.Upsample(scale_factor=2, mode='nearest')
        self.P5_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P4_1 = nn.Conv2d(C4_size, feature_size, kernel_size=1, stride=1, padding=0)
        self.P4_2 = nn.Conv2d(feature_size, feature_size, kernel_size=3, stride=1, padding=1)
        self.P6 = nn.Conv2d(C5_size, feature_size, kernel_size=5, stride=3, padding=1)


class RetinaNetTiny(RetinaNet):
    """retinanet.py:486-669: stem without max-pool, layer1..3 (BasicBlock only), pyramid levels 2..4,
    anchor sizes 2^(l+2), ratios (1, 1.5, 2).  The reference switches on `self.training` and takes
    `(img_batch, annotations)` as ONE argument when training (:578-582)."""
    _plan_class = engine.RetinaPlanTwoLevel
    _stage_names = ('layer1', 'layer2', 'layer3')
    _fpn_names = ('P3_1', 'P3_2', 'P2_1', 'P2_2', 'P4')
    _input_multiple = 16

    def __init__(self, num_classes, block, layers, iou_threshold=0.5, **kwargs):
        self.inplanes = 64
        nn.Module.__init__(self)
        self.convert_onnx = False
        self.fixed_size = None
        self.is_training = False
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.iou_threshold = iou_threshold
        if block != BasicBlock:
            raise ValueError('Block type %s not understood' % block)          # retinanet.py:510-518
        fpn_sizes = [self.layer2[layers[1] - 1].conv2.out_channels, self.layer3[layers[2] - 1].conv2.out_channels]
        self.fpn = PyramidFeaturesTiny(fpn_sizes[0], fpn_sizes[1])
        self.regressionModel = RegressionModel(256)
        self.classificationModel = ClassificationModel(256, num_classes=num_classes)
        my_pyramid_levels = [2, 3, 4]
        self.anchors = Anchors(pyramid_levels=my_pyramid_levels, sizes=[int(2 ** (x + 2)) for x in my_pyramid_levels],
                               ratios=[1, 1.5, 2], **kwargs)
        self.regressBoxes = BBoxTransform()
        self.clipBoxes = ClipBoxes()
        self.focalLoss = losses.FocalLoss()
        self._finish_init()

    def forward(self, inputs, annotations=None, return_head=False, return_loss=None):
        if return_head or return_loss is not None or annotations is not None:   # (explicit RetinaNet-style call)
            return RetinaNet.forward(self, inputs, annotations, return_head,
                                     True if return_loss is None else return_loss)
        if self.training:                          # retinanet.py:578-582
            img_batch, annotations = inputs
            return RetinaNet.forward(self, img_batch, annotations, False, True)
        return RetinaNet.forward(self, inputs, None, False, False)


class RetinaNetP45P6(RetinaNet):
    """retinanet.py:722-918: full ResNet (BasicBlock only), pyramid levels 4, 5 and P6 = conv5x5 s3 p1
    over C5 (stride 96), anchor strides (16, 32, 96), sizes (32, 64, 160), ratios (1, 1.5, 2)."""
    _plan_class = engine.RetinaPlanTwoLevel
    _stage_names = ('layer1', 'layer2', 'layer3', 'layer4')
    _fpn_names = ('P5_1', 'P5_2', 'P4_1', 'P4_2', 'P6')
    _input_multiple = 96                           # ceil(H/96) cells == conv5x5 s3 p1 over H/32 rows

    def __init__(self, num_classes, block, layers, iou_threshold=0.5, pyramid_levels=[4, 5, 6], **kwargs):
        self.inplanes = 64
        nn.Module.__init__(self)
        self.convert_onnx = False
        self.fixed_size = None
        self.is_training = False
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.iou_threshold = iou_threshold
        self.pyramid_levels = pyramid_levels
        if block != BasicBlock:
            raise ValueError('Block type %s not understood' % block)          # retinanet.py:746-756
        # (the reference indexes layer3 with layers[1] and layer4 with layers[2]: retinanet.py:747)
        fpn_sizes = [self.layer3[layers[1] - 1].conv2.out_channels, self.layer4[layers[2] - 1].conv2.out_channels]
        self.fpn = PyramidFeaturesP45P6(fpn_sizes[0], fpn_sizes[1])
        self.regressionModel = RegressionModel(256)
        self.classificationModel = ClassificationModel(256, num_classes=num_classes)
        self.anchors = Anchors(pyramid_levels=self.pyramid_levels, strides=[2 ** 4, 2 ** 5, 3 * (2 ** 5)],
                               sizes=[2 ** 5, 2 ** 6, int(2.5 * (2 ** 6))], ratios=[1, 1.5, 2], **kwargs)
        self.regressBoxes = BBoxTransform()
        self.clipBoxes = ClipBoxes()
        self.focalLoss = losses.FocalLoss()
        self._finish_init()


def retinanet_tiny(num_classes, pretrained=False, **kwargs):
    _no_fetch(pretrained)                          # (the reference ignores `pretrained` here: retinanet.py:672-679)
    return RetinaNetTiny(num_classes, BasicBlock, [2, 2, 2], **kwargs)


def retinanet_p45p6(num_classes, pretrained=False, **kwargs):
    _no_fetch(pretrained)
    return RetinaNetP45P6(num_classes, BasicBlock, [2, 2, 2, 2], **kwargs)


def build_retinanet(name, num_classes=80, pretrained=False, **kwargs):
    """retinanet.py:925-939: 'retinanet-res50' style names."""
    depth = int(''.join(ch for ch in name.lower().split('res')[-1] if ch.isdigit()))
    return retinanet(depth, num_classes=num_classes, pretrained=pretrained, **kwargs)
