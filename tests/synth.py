"""Deterministic synthetic inputs shared by tools/make_golden.py, tests/ and bench.py.

Everything here is numpy `default_rng(seed)` based, so the same arrays come out in the
build container (where the golden vectors are generated from the reference) and on the GPU
box (where /root/reference does not exist).  Input contract follows the reference's
`collater` (datasettool/dataloader.py:655-691): images `[B,3,H,W]` fp32 NCHW, annotations
`[B,M,5]` fp32 `(x1,y1,x2,y2,label)` padded with rows of -1.
"""
import zlib

import re

import numpy as np


def synth_images(batch, height, width, seed=1):
    """Post-`Normalizer` image distribution (dataloader.py:905-924): ~N(0,1) per channel."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((batch, 3, height, width), dtype=np.float32)


def synth_annotations(batch, height, width, num_real=8, pad_to=16, num_classes=80, seed=2,
                      empty_images=()):
    """`num_real` boxes/img, centre ~U(0,W)xU(0,H), w,h ~U(0.05,0.45)*(W,H),
    clipped to the image, min side 1 px, label ~U{0..C-1}; rows of -1 up to `pad_to`.
    Images listed in `empty_images` are all-pad (the `bbox_annotation.shape[0] == 0` branch,
    losses.py:52-60)."""
    rng = np.random.default_rng(seed)
    ann = -np.ones((batch, pad_to, 5), dtype=np.float32)
    for b in range(batch):
        cx = rng.uniform(0, width, num_real)
        cy = rng.uniform(0, height, num_real)
        w = rng.uniform(0.05, 0.45, num_real) * width
        h = rng.uniform(0.05, 0.45, num_real) * height
        x1 = np.clip(cx - 0.5 * w, 0, width - 1)
        y1 = np.clip(cy - 0.5 * h, 0, height - 1)
        x2 = np.clip(cx + 0.5 * w, 0, width - 1)
        y2 = np.clip(cy + 0.5 * h, 0, height - 1)
        x2 = np.maximum(x2, x1 + 1)
        y2 = np.maximum(y2, y1 + 1)
        lab = rng.integers(0, num_classes, num_real)
        if b in empty_images:
            continue
        # interleave pad rows so the `!= -1` filter (losses.py:50) is exercised mid-array
        rows = np.sort(rng.choice(pad_to, num_real, replace=False))
        ann[b, rows, 0] = x1
        ann[b, rows, 1] = y1
        ann[b, rows, 2] = x2
        ann[b, rows, 3] = y2
        ann[b, rows, 4] = lab
    return ann


def _rng_for(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def synth_state_dict(shapes, seed=0, head_std=0.01, prior=0.01, gain=1.0):
    """Deterministic weights keyed by state_dict name (independent of module construction
    order and of torch's RNG).  `shapes`: {key: tuple}.  Conv weights ~N(0, sqrt(2/(k*k*out)))
    as retinanet.py:232-235; BN gamma ~U(0.5,1.5), beta ~N(0,0.1), running_mean ~N(0,0.1),
    running_var ~U(0.5,1.5) (non-trivial so that the frozen-BN affine is exercised,
    retinanet.py:266-270); output layers perturbed N(0, head_std) instead of the reference's
    zeros (SURVEY.md §8c caveat ii), classification prior bias -log((1-p)/p) (retinanet.py:243).
    `gain` scales the non-output conv weights: with random BN statistics the plain He init
    saturates the sigmoid after ~20 layers, so each golden case records the gain that keeps the
    head outputs in a useful range (R18: 0.87, R50: 0.78).
    """
    out = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        rng = _rng_for(key, seed)
        leaf = key.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[key] = np.zeros(shape, dtype=np.int64)
        elif leaf == 'running_mean':
            out[key] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == 'running_var':
            out[key] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 4:
            o, i, kh, kw = shape
            if key.endswith('Model.output.weight') or '.header.pointwise_conv.' in key or \
                    re.search(r'\.header\.\d+\.weight$', key):          # (KSevenDet: header.0, or one header per level)
                std = head_std
            elif i == 1 or 'depthwise' in key:      # depthwise: fan = kh*kw
                std = np.sqrt(2.0 / (kh * kw))
            else:
                std = np.sqrt(2.0 / (kh * kw * o))
            if std != head_std:
                std = std * gain
            out[key] = (std * rng.standard_normal(shape)).astype(np.float32)
        elif len(shape) == 1 and leaf == 'weight':   # BN gamma
            out[key] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1 and leaf == 'bias':
            if key in ('classificationModel.output.bias', 'classifier.header.pointwise_conv.conv.bias') or \
                    re.match(r'classifier\.header\.\d+\.bias$', key):
                out[key] = np.full(shape, -np.log((1.0 - prior) / prior), dtype=np.float32)
            else:
                out[key] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:                                        # BiFPN fusion weights (some negative: relu path)
            out[key] = rng.uniform(-0.3, 1.5, shape).astype(np.float32)
    return out


def synth_head_outputs(batch, num_anchors, num_classes, seed=3, hot_fraction=0.02):
    """Classification probabilities / regression deltas shaped like the head outputs
    (retinanet.py:285-287): most probabilities near the 0.01 prior, a few confident."""
    rng = np.random.default_rng(seed)
    logits = rng.normal(-4.6, 1.0, (batch, num_anchors, num_classes))
    hot = rng.random((batch, num_anchors, num_classes)) < hot_fraction
    logits = np.where(hot, rng.normal(1.0, 2.0, logits.shape), logits)
    cls = (1.0 / (1.0 + np.exp(-logits))).astype(np.float32)
    reg = (0.5 * rng.standard_normal((batch, num_anchors, 4))).astype(np.float32)
    return cls, reg


def disable_drop_connect(model):
    """EfficientNet's drop-connect rate lives in the backbone's global params (utils_efficientnet.py GlobalParams);
    clearing it is how the deterministic (eval-mode) goldens' path is selected on a model in train mode."""
    eff = model.backbone_net.model
    eff._global_params = eff._global_params._replace(drop_connect_rate=None)
    return model
