"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ref_cpu.py header for the rules).

CPU restatement (torch fp32, functional) of the reference's EfficientDet forward:
architectures/efficientdet.py:28-66 (SeparableConvBlock), :195-267 (BiFPN fast attention),
:320-393 (Regressor / Classifier), :396-433 (EfficientNet wrapper), :436-576 (EfficientDet);
architectures/efficientnet.py:80-115 (MBConvBlock); architectures/utils_efficientnet.py:38-94,
:231-257 (swish, block table, round_filters); architectures/utils_extra_efficientnet.py:9-96
(TF-"same" conv / max-pool).  Pinned by tests/golden/e2e_effdet_*.npz (generated from the reference).

Deterministic path only: drop-connect (utils_efficientnet.py:85-94) is OFF (eval semantics /
`drop_connect_rate=None`), BatchNorm is frozen (efficientdet.py:523-526), as SURVEY.md §8c iii says.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_cpu

BN_EPS = 1e-3                      # utils_efficientnet.py:244, efficientdet.py:53

# efficientdet.py:442-460
BACKBONE_COEF = [0, 1, 2, 3, 4, 5, 6, 6]
FPN_FILTERS = [64, 88, 112, 160, 224, 288, 384, 384]
FPN_REPEATS = [3, 4, 5, 6, 7, 7, 8, 8]
BOX_CLASS_REPEATS = [3, 3, 3, 4, 4, 4, 5, 5]
# utils_efficientnet.py:138-153 (width, depth)
EFFNET_COEF = {0: (1.0, 1.0), 1: (1.0, 1.1), 2: (1.1, 1.2), 3: (1.2, 1.4), 4: (1.4, 1.8),
               5: (1.6, 2.2), 6: (1.8, 2.6), 7: (2.0, 3.1)}
# utils_efficientnet.py:235-240: (repeat, kernel, stride, expand, in, out), se_ratio 0.25 everywhere
BLOCKS = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
          (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
ANCHOR_CFG = dict(pyramid_levels=[3, 4, 5, 6, 7], strides=[8, 16, 32, 64, 128],
                  sizes=[2 ** (x + 1) * 1.25 for x in [3, 4, 5, 6, 7]], ratios=[1, 2])   # :461-468


TRACE = None          # set to a dict to capture intermediates (debug aid: tools/debug_effdet.py)


def _tr(name, t):
    if TRACE is not None:
        TRACE[name] = t.detach()
    return t


def round_filters(filters, width):
    """utils_efficientnet.py:62-74 with depth_divisor 8."""
    filters = filters * width
    new = max(8, int(filters + 4) // 8 * 8)
    if new < 0.9 * filters:
        new += 8
    return int(new)


def block_table(compound_coef):
    """Expanded MBConv list [(kernel, stride, expand, cin, cout)] of EfficientNet-b{coef}
    (efficientnet.py:150-166)."""
    width, depth = EFFNET_COEF[BACKBONE_COEF[compound_coef]]
    out = []
    for (rep, k, s, e, ci, co) in BLOCKS:
        ci, co = round_filters(ci, width), round_filters(co, width)
        rep = int(math.ceil(depth * rep))
        out.append((k, s, e, ci, co))
        for _ in range(rep - 1):
            out.append((k, 1, e, co, co))
    return round_filters(32, width), out


def same_pad(n, k, s):
    """utils_extra_efficientnet.py:33-47: (before, after) zero padding of one axis."""
    extra = s * (math.ceil(n / s) - 1) + 1 + (k - 1) - n
    return extra // 2, extra - extra // 2


def conv_same(x, w, b=None, stride=1, groups=1):
    kh, kw = w.shape[2], w.shape[3]
    t, bt = same_pad(x.shape[2], kh, stride)
    l, r = same_pad(x.shape[3], kw, stride)
    return F.conv2d(F.pad(x, [l, r, t, bt]), w, b, stride, 0, 1, groups)


def maxpool_same(x, k=3, s=2):
    """utils_extra_efficientnet.py:78-96: ZERO padding, then max-pool (pads take part in the max)."""
    t, bt = same_pad(x.shape[2], k, s)
    l, r = same_pad(x.shape[3], k, s)
    return F.max_pool2d(F.pad(x, [l, r, t, bt]), k, s)


def swish(x):
    return x * torch.sigmoid(x)


def bn(x, p, pre):
    return F.batch_norm(x, p[pre + '.running_mean'], p[pre + '.running_var'], p[pre + '.weight'],
                        p[pre + '.bias'], False, 0.0, BN_EPS)


def drop_connect(x, rate, u):
    """utils_efficientnet.py:85-94 with the uniforms `u` [B] handed in (the reference draws them with torch.rand)."""
    keep = 1 - rate
    binary = torch.floor(keep + u.reshape(-1, 1, 1, 1).to(x.dtype))
    return x / keep * binary


def mbconv(x, p, pre, k, s, e, cin, cout, dc=None):
    """efficientnet.py:80-115; dc = (rate, uniforms [B]) for the train-mode drop-connect of the skip branch
    (:109-113), None = off (eval mode / rate 0)."""
    inp = x
    if e != 1:
        x = swish(_tr(pre + '.z0', bn(conv_same(x, p[pre + '._expand_conv.conv.weight']), p, pre + '._bn0')))
    x = swish(_tr(pre + '.z1', bn(conv_same(x, p[pre + '._depthwise_conv.conv.weight'], None, s, x.shape[1]), p, pre + '._bn1')))
    sq = F.adaptive_avg_pool2d(x, 1)
    sq = swish(F.conv2d(sq, p[pre + '._se_reduce.conv.weight'], p[pre + '._se_reduce.conv.bias']))
    sq = F.conv2d(sq, p[pre + '._se_expand.conv.weight'], p[pre + '._se_expand.conv.bias'])
    x = _tr(pre + '.u', torch.sigmoid(sq) * x)
    x = bn(conv_same(x, p[pre + '._project_conv.conv.weight']), p, pre + '._bn2')
    if s == 1 and cin == cout:
        if dc is not None and dc[0]:
            x = drop_connect(x, dc[0], dc[1])
        x = x + inp
    return x


def backbone(p, img, compound_coef, dc_u=None, dc_rate=0.2):
    """efficientdet.py:411-433: returns (p3, p4, p5) = feature_maps[1:] minus P2 (:533).  dc_u: [rows, B] uniforms,
    one row per block with a skip connection in block order (train mode: rate = dc_rate * idx / len(blocks), :421-423)."""
    _, blocks = block_table(compound_coef)
    pre = 'backbone_net.model.'
    x = swish(_tr('stem_z', bn(conv_same(img, p[pre + '_conv_stem.conv.weight'], None, 2), p, pre + '_bn0')))
    feats, last = [], None
    row = 0
    for i, (k, s, e, ci, co) in enumerate(blocks):
        dc = None
        if dc_u is not None and s == 1 and ci == co:
            rate = dc_rate * float(i) / len(blocks)
            if rate:                             # block 0: rate 0 -> no draw (efficientnet.py:110)
                dc = (rate, dc_u[row])
                row += 1
        x = _tr('block%d' % i, mbconv(x, p, pre + '_blocks.%d' % i, k, s, e, ci, co, dc))
        if s == 2:
            feats.append(last)
        elif i == len(blocks) - 1:
            feats.append(x)
        last = x
    return feats[2], feats[3], feats[4]          # feature_maps[1:] = [P2,P3,P4,P5]; caller drops P2


def sepconv(x, p, pre, norm=True):
    """efficientdet.py:56-66 (activation=False everywhere it is used)."""
    x = conv_same(x, p[pre + '.depthwise_conv.conv.weight'], None, 1, x.shape[1])
    x = F.conv2d(x, p[pre + '.pointwise_conv.conv.weight'], p[pre + '.pointwise_conv.conv.bias'])
    if norm:
        x = bn(x, p, pre + '.bn')
    return x


def fusion_weights(w, eps=1e-4):
    w = F.relu(w)
    return w / (w.sum(dim=0) + eps)


_fusion_weights = fusion_weights


def bifpn_cell(p, pre, inputs, first_time, attention=True):
    """efficientdet.py:195-267 (`_forward_fast_attention`); attention=False: `_forward` (:269-317), the plain
    sum of the same inputs (D6 / D7, :478), the fusion parameters unused."""
    def fusion_weights(w):                       # noqa: F811  (shadows the module-level helper in plain mode)
        return _fusion_weights(w) if attention else torch.ones_like(w)

    def cb(x, name):           # 1x1 conv + BN
        return bn(F.conv2d(x, p[pre + name + '.0.conv.weight'], p[pre + name + '.0.conv.bias']), p, pre + name + '.1')

    def up(x):
        return F.interpolate(x, scale_factor=2, mode='nearest')

    if first_time:
        p3, p4, p5 = inputs
        p6_in = maxpool_same(cb(p5, 'p5_to_p6'))
        p7_in = maxpool_same(p6_in)
        p3_in, p4_in, p5_in = cb(p3, 'p3_down_channel'), cb(p4, 'p4_down_channel'), cb(p5, 'p5_down_channel')
    else:
        p3_in, p4_in, p5_in, p6_in, p7_in = inputs
    w = fusion_weights(p[pre + 'p6_w1'])
    p6_up = sepconv(swish(w[0] * p6_in + w[1] * up(p7_in)), p, pre + 'conv6_up')
    w = fusion_weights(p[pre + 'p5_w1'])
    p5_up = sepconv(swish(w[0] * p5_in + w[1] * up(p6_up)), p, pre + 'conv5_up')
    w = fusion_weights(p[pre + 'p4_w1'])
    p4_up = sepconv(swish(w[0] * p4_in + w[1] * up(p5_up)), p, pre + 'conv4_up')
    w = fusion_weights(p[pre + 'p3_w1'])
    p3_out = sepconv(swish(w[0] * p3_in + w[1] * up(p4_up)), p, pre + 'conv3_up')
    if first_time:
        p4_in, p5_in = cb(p4, 'p4_down_channel_2'), cb(p5, 'p5_down_channel_2')
    w = fusion_weights(p[pre + 'p4_w2'])
    p4_out = sepconv(swish(w[0] * p4_in + w[1] * p4_up + w[2] * maxpool_same(p3_out)), p, pre + 'conv4_down')
    w = fusion_weights(p[pre + 'p5_w2'])
    p5_out = sepconv(swish(w[0] * p5_in + w[1] * p5_up + w[2] * maxpool_same(p4_out)), p, pre + 'conv5_down')
    w = fusion_weights(p[pre + 'p6_w2'])
    p6_out = sepconv(swish(w[0] * p6_in + w[1] * p6_up + w[2] * maxpool_same(p5_out)), p, pre + 'conv6_down')
    w = fusion_weights(p[pre + 'p7_w2'])
    p7_out = sepconv(swish(w[0] * p7_in + w[1] * maxpool_same(p6_out)), p, pre + 'conv7_down')
    return p3_out, p4_out, p5_out, p6_out, p7_out


def head(p, pre, feats, nlayers, last_dim):
    """efficientdet.py:335-353 / :374-393: shared separable convs, per-level BN, swish, header."""
    outs = []
    for lvl, f in enumerate(feats):
        for i in range(nlayers):
            f = sepconv(f, p, pre + '.conv_list.%d' % i, norm=False)
            f = swish(bn(f, p, pre + '.bn_list.%d.%d' % (lvl, i)))
        f = sepconv(f, p, pre + '.header', norm=False)
        outs.append(f.permute(0, 2, 3, 1).reshape(f.shape[0], -1, last_dim))
    return torch.cat(outs, dim=1)


def efficientdet_heads(p, img, compound_coef, num_classes, dc_u=None):
    """efficientdet.py:528-539. Returns (classification [B,A,C] probabilities, regression [B,A,4])."""
    feats = backbone(p, img, compound_coef, dc_u)
    for c in range(FPN_REPEATS[compound_coef]):
        feats = bifpn_cell(p, 'bifpn.%d.' % c, feats, c == 0, attention=compound_coef < 6)
        for l, f in enumerate(feats):
            _tr('cell%d_p%d' % (c, l + 3), f)
    n = BOX_CLASS_REPEATS[compound_coef]
    reg = head(p, 'regressor', feats, n, 4)
    cls = torch.sigmoid(head(p, 'classifier', feats, n, num_classes))
    return cls, reg


def efficientdet_loss(p, img, annotations, compound_coef, num_classes):
    cls, reg = efficientdet_heads(p, img, compound_coef, num_classes)
    anchors = torch.from_numpy(ref_cpu.anchors_for_image(img.shape[2], img.shape[3], **ANCHOR_CFG))
    return ref_cpu.focal_loss(cls, reg, anchors, annotations)
