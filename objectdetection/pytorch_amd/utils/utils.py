"""Drop-in for the inference / init helpers of utils/utils.py that sit on the detection path:
`postprocess` (:76-114, batched decode + per-image NMS), `invert_affine` (:16-29) and `init_weights`
(:204-215).  The image I/O helpers (`preprocess`, `display`, cv2) are host code and out of scope."""
import numpy as np
import torch
from torch import nn

from .. import ops_detect


def invert_affine(metas, preds):
    """utils/utils.py:16-29 (numpy on the host; note the reference's `metas is float` identity test)."""
    for i in range(len(preds)):
        if len(preds[i]['rois']) == 0:
            continue
        if metas is float:
            preds[i]['rois'][:, [0, 2]] = preds[i]['rois'][:, [0, 2]] / metas
            preds[i]['rois'][:, [1, 3]] = preds[i]['rois'][:, [1, 3]] / metas
        else:
            new_w, new_h, old_w, old_h, padding_w, padding_h = metas[i]
            preds[i]['rois'][:, [0, 2]] = preds[i]['rois'][:, [0, 2]] / (new_w / old_w)
            preds[i]['rois'][:, [1, 3]] = preds[i]['rois'][:, [1, 3]] / (new_h / old_h)
    return preds


def postprocess(x, anchors, regression, classification, regressBoxes, clipBoxes, threshold, iou_threshold):
    """utils/utils.py:76-114: decode + clip for the whole batch, then per image: max-class score > threshold,
    class-agnostic NMS on the max-class score, (score, class) of the kept rows.  Returns the reference's
    list of {'rois', 'class_ids', 'scores'} numpy dicts.  Kept quirk: when NO image of the batch has a
    score over the threshold the reference appends an empty dict and then, lacking a `continue`, a second
    one for the same image (:83-88, :107-112) — reproduced so that callers index the same list."""
    ops_detect.require_gpu(classification, regression)
    boxes = regressBoxes(anchors, regression)
    boxes = clipBoxes(boxes, x)
    out = []
    empty = lambda: {'rois': np.array(()), 'class_ids': np.array(()), 'scores': np.array(())}   # noqa: E731
    per_image, any_over = [], False
    for i in range(x.shape[0]):
        score, label = ops_detect.class_max(classification[i])
        idx = ops_detect.score_filter(score, threshold)
        per_image.append((score, label, idx))
        any_over = any_over or idx.numel() > 0
    for i in range(x.shape[0]):
        score, label, idx = per_image[i]
        if not any_over:
            out.append(empty())
        if idx.numel() == 0:
            out.append(empty())
            continue
        idx = idx.long()
        keep = ops_detect.nms(boxes[i, idx, :], score[idx], iou_threshold)
        sel = idx[keep]
        out.append({'rois': boxes[i, sel, :].detach().cpu().numpy(),
                    'class_ids': label[sel].long().cpu().numpy(),
                    'scores': score[sel].detach().cpu().numpy()})
    return out


def init_weights(model):
    """utils/utils.py:204-215: kaiming-normal conv weights, zero conv biases (KRSC memory order is kept)."""
    for name, module in model.named_modules():
        if isinstance(module, nn.Conv2d):
            w = torch.empty(module.weight.shape, dtype=module.weight.dtype, device=module.weight.device)
            nn.init.kaiming_normal_(w)
            module.weight.data.copy_(w)               # copy_ keeps the parameter's (channels-last) strides
            if module.bias is not None:
                module.bias.data.zero_()


def replace_w_sync_bn(m):
    """utils/utils.py:140-168: swap every nn.BatchNorm2d attribute (recursively) for SynchronizedBatchNorm2d,
    keeping its parameters and running statistics."""
    from .sync_batchnorm import SynchronizedBatchNorm2d
    for var_name in dir(m):
        target_attr = getattr(m, var_name)
        if type(target_attr) == torch.nn.BatchNorm2d:
            new = SynchronizedBatchNorm2d(target_attr.num_features, target_attr.eps, target_attr.momentum,
                                          target_attr.affine)
            new.running_mean = target_attr.running_mean
            new.running_var = target_attr.running_var
            if target_attr.affine:
                new.weight = target_attr.weight
                new.bias = target_attr.bias
            setattr(m, var_name, new.to(target_attr.running_mean.device))
    if getattr(m, '_plans', None):                     # launch plans / the cached BN list point at the old modules
        m._plans.clear()
    if hasattr(m, '_bn_list'):
        m._bn_list = None
    for var_name, children in m.named_children():
        replace_w_sync_bn(children)
