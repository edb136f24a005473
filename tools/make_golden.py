#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE (/root/reference) on CPU.

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 \
        python /root/repo/tools/make_golden.py

Outputs are data only (inputs + the reference's outputs); no reference source is copied.
`torchvision` is absent here and the reference imports `torchvision.ops.nms` at module import
(architectures/retinanet.py:5), so a stand-in module is registered first.  The stand-in `nms`
is OUR oracle restatement (oracle/ref_cpu.py:nms) — the decode-tail fixtures therefore pin the
reference's wiring around NMS (retinanet.py:297-318), not torchvision's NMS arithmetic, which
stays "parity unpinned" (SURVEY.md §8c).
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from oracle import ref_cpu                               # noqa: E402
import synth                                             # noqa: E402

_tv = types.ModuleType('torchvision')
_ops = types.ModuleType('torchvision.ops')


def _nms_standin(boxes, scores, iou_threshold):
    thr = iou_threshold
    return torch.from_numpy(ref_cpu.nms(boxes.detach().numpy(), scores.detach().numpy(), thr))


_ops.nms = _nms_standin
_tv.ops = _ops
sys.modules['torchvision'] = _tv
sys.modules['torchvision.ops'] = _ops

from architectures import anchors as ref_anchors         # noqa: E402
from architectures import losses as ref_losses           # noqa: E402
from architectures import utils as ref_utils             # noqa: E402
from architectures import retinanet as ref_retinanet     # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------- anchors (a-5)
def gen_anchors():
    d = {}
    effdet = dict(pyramid_levels=[3, 4, 5, 6, 7], strides=[8, 16, 32, 64, 128],
                  sizes=[2 ** (x + 1) * 1.25 for x in [3, 4, 5, 6, 7]],
                  ratios=np.array([1, 2]),
                  scales=np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)]))
    cases = [('retina', {}, (256, 256)), ('retina', {}, (512, 512)), ('retina', {}, (640, 640)),
             ('retina', {}, (100, 130)), ('retina', {}, (33, 77)), ('retina', {}, (8, 8)),
             ('effdet', effdet, (512, 512)), ('effdet', effdet, (768, 768)),
             ('effdet', effdet, (96, 160))]
    names = []
    for kind, kw, (h, w) in cases:
        a = ref_anchors.Anchors(**kw)(torch.zeros(1, 3, h, w)).numpy()
        name = '%s_%dx%d' % (kind, h, w)
        names.append(name)
        d[name + '_shape'] = np.array(a.shape)
        d[name + '_sha256'] = np.array(sha(a))
        n = a.shape[1]
        if n <= 13000:
            d[name + '_full'] = a
        else:
            idx = np.unique(np.concatenate([np.arange(0, 64), np.arange(n - 64, n),
                                            np.random.default_rng(7).integers(0, n, 1024)]))
            d[name + '_idx'] = idx
            d[name + '_rows'] = a[0, idx]
    d['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'anchors.npz'), **d)
    print('anchors:', names)


# ---------------------------------------------------------------- loss (a-6, a-7, a-8)
def run_ref_loss(cls, reg, anchors, ann):
    cls_t = torch.from_numpy(cls).clone().requires_grad_(True)
    reg_t = torch.from_numpy(reg).clone().requires_grad_(True)
    anc_t = torch.from_numpy(anchors)
    ann_t = torch.from_numpy(ann)
    fl = ref_losses.FocalLoss()
    cl, rl = fl(cls_t, reg_t, anc_t, ann_t)
    # per-loss grads separately (so that a wrong mix of the two is caught)
    gc_cls, = torch.autograd.grad(cl, cls_t, retain_graph=True, allow_unused=True)
    gr_reg, = torch.autograd.grad(rl, reg_t, retain_graph=True, allow_unused=True) \
        if rl.requires_grad else (None,)
    B, A = cls.shape[0], cls.shape[1]
    # integer intermediates straight from the reference's own ops (losses.py:64-66,79-80)
    argmax = np.zeros((B, A), dtype=np.int32)
    state = np.full((B, A), 3, dtype=np.int8)
    ioumax = np.zeros((B, A), dtype=np.float32)
    for j in range(B):
        ba = ann_t[j]
        ba = ba[ba[:, 4] != -1]
        if ba.shape[0] == 0:
            continue
        iou = ref_losses.calc_iou(anc_t[0], ba[:, :4])
        m, am = torch.max(iou, dim=1)
        st = np.ones(A, dtype=np.int8)
        st[torch.lt(m, 0.35).numpy()] = 0
        st[torch.ge(m, 0.45).numpy()] = 2
        argmax[j] = am.numpy().astype(np.int32)
        state[j] = st
        ioumax[j] = m.numpy()
    return dict(cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(),
                grad_cls=(gc_cls.numpy() if gc_cls is not None else np.zeros_like(cls)),
                grad_reg=(gr_reg.numpy() if gr_reg is not None else np.zeros_like(reg)),
                argmax=argmax, state=state, iou_max=ioumax)


def gen_loss():
    d = {}
    names = []

    def add(name, cls, reg, anchors, ann):
        r = run_ref_loss(cls, reg, anchors, ann)
        names.append(name)
        d[name + '_cls'] = cls
        d[name + '_reg'] = reg
        d[name + '_anchors'] = anchors
        d[name + '_ann'] = ann
        for k, v in r.items():
            d[name + '_' + k] = v
        print('loss case', name, 'cls', r['cls_loss'], 'reg', r['reg_loss'],
              'pos', int((r['state'] == 2).sum()), 'ign', int((r['state'] == 1).sum()))

    # (1) 256x256 anchors, C=4, B=3: normal, all-pad image, normal
    H = W = 256
    anc = ref_cpu.anchors_for_image(H, W)
    A = anc.shape[1]
    cls, reg = synth.synth_head_outputs(3, A, 4, seed=11)
    ann = synth.synth_annotations(3, H, W, num_real=8, pad_to=16, num_classes=4, seed=12,
                                  empty_images=(1,))
    add('c4_256', cls, reg, anc, ann)

    # (2) 64x64, C=80, B=2, with clamp-edge probabilities and tiny boxes (no positives in img 1)
    H = W = 64
    anc = ref_cpu.anchors_for_image(H, W)
    A = anc.shape[1]
    cls, reg = synth.synth_head_outputs(2, A, 80, seed=21)
    cls[0, :7, :5] = np.array([0.0, 1.0, 1e-4, 1.0 - 1e-4, 5e-5], dtype=np.float32)
    ann = synth.synth_annotations(2, H, W, num_real=5, pad_to=8, num_classes=80, seed=22)
    # image 0: boxes big enough for the 32-px base anchors (jittered copies of real anchors)
    pick = np.random.default_rng(23).choice(A, 5, replace=False)
    rows = np.where(ann[0, :, 4] != -1)[0]
    ann[0, rows, :4] = anc[0, pick] + np.random.default_rng(24).uniform(-2, 2, (5, 4)).astype(np.float32)
    ann[1, :, :] = -1
    ann[1, 0] = [10.0, 10.0, 10.5, 10.4, 3]       # w,h < 1 -> clamp(min=1) path never positive
    ann[1, 3] = [30.0, 31.0, 30.2, 31.9, 79]
    add('c80_64', cls, reg, anc, ann)

    # (3) hand-made anchors: IoU exactly 0.45 / 0.35 in fp32, argmax ties, degenerate GT
    anchors = np.array([[[0, 0, 10, 10],        # vs gt0 (0,0,10,4.5): IoU 45/100 = 0.45 -> pos
                         [20, 0, 30, 10],       # vs gt1 (20,0,30,3.5): IoU 0.35 -> ignore (not <)
                         [40, 0, 50, 10],       # vs gt2/gt3 identical boxes -> tie -> lowest idx
                         [60, 0, 70, 10],       # no overlap -> neg
                         [80, 0, 90, 10],       # gt4 identical -> IoU 1
                         [100, 0, 110, 10],     # gt5 w<1 (clamp min=1 in target) but IoU>=0.45?
                         [0, 50, 16, 66],
                         [0, 50, 16, 66.000004]]], dtype=np.float32)
    ann = -np.ones((1, 10, 5), dtype=np.float32)
    ann[0, 0] = [0, 0, 10, 4.5, 1]
    ann[0, 1] = [-1, -1, -1, -1, -1]
    ann[0, 2] = [20, 0, 30, 3.5, 0]
    ann[0, 3] = [40, 0, 50, 10, 2]
    ann[0, 4] = [40, 0, 50, 10, 3]
    ann[0, 6] = [80, 0, 90, 10, 0]
    ann[0, 7] = [100, 0, 100.5, 10, 1]
    ann[0, 8] = [0, 50, 16, 58, 2]
    rng = np.random.default_rng(31)
    cls = rng.uniform(0.001, 0.999, (1, 8, 4)).astype(np.float32)
    reg = rng.standard_normal((1, 8, 4)).astype(np.float32)
    add('edges', cls, reg, anchors, ann)

    # (4) EfficientDet-style 6 anchors/cell on 96x160, C=3
    eff = dict(pyramid_levels=[3, 4, 5, 6, 7], strides=[8, 16, 32, 64, 128],
               sizes=[2 ** (x + 1) * 1.25 for x in [3, 4, 5, 6, 7]], ratios=[1, 2])
    anc = ref_cpu.anchors_for_image(96, 160, **eff)
    A = anc.shape[1]
    cls, reg = synth.synth_head_outputs(2, A, 3, seed=41)
    ann = synth.synth_annotations(2, 96, 160, num_real=6, pad_to=6, num_classes=3, seed=42)
    add('eff_96x160', cls, reg, anc, ann)

    d['names'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'loss.npz'), **d)


# ---------------------------------------------------------------- decode / clip / tail (a-9..a-11)
def gen_decode():
    d = {}
    H, W = 96, 128
    anc = ref_cpu.anchors_for_image(H, W)
    A = anc.shape[1]
    rng = np.random.default_rng(51)
    reg = rng.standard_normal((2, A, 4)).astype(np.float32)
    reg[:, :, 2:] *= 2.0
    bt = ref_utils.BBoxTransform()
    boxes = bt(torch.from_numpy(anc), torch.from_numpy(reg))
    d['anchors'] = anc
    d['reg'] = reg
    d['boxes'] = boxes.numpy().copy()
    clipped = ref_utils.ClipBoxes()(boxes.clone(), torch.zeros(2, 3, H, W))
    d['clipped'] = clipped.numpy()
    d['hw'] = np.array([H, W])
    np.savez_compressed(os.path.join(OUT, 'decode.npz'), **d)
    print('decode: boxes', d['boxes'].shape)


def gen_postprocess():
    """Batched decode tail utils/utils.py:76-114 (f-4).  `cv2` is absent and only used by the image I/O
    helpers of that module: an empty stand-in module lets it import."""
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    from utils import utils as ref_uu
    H, W, B, C = 96, 128, 3, 6
    anc = torch.from_numpy(ref_cpu.anchors_for_image(H, W))
    A = anc.shape[1]
    rng = np.random.default_rng(77)
    reg = (rng.standard_normal((B, A, 4)) * 0.5).astype(np.float32)
    cls = (rng.random((B, A, C)) ** 8).astype(np.float32) * 0.9       # few confident anchors
    cls[2] *= 0.04                                                     # image 2: nothing over 0.05
    x = torch.zeros(B, 3, H, W)
    out = ref_uu.postprocess(x, anc, torch.from_numpy(reg), torch.from_numpy(cls), ref_utils.BBoxTransform(),
                             ref_utils.ClipBoxes(), 0.05, 0.5)
    d = dict(hw=np.array([H, W]), reg=reg, cls=cls, n_out=np.array(len(out)))
    for i, o in enumerate(out):
        d['rois%d' % i], d['class_ids%d' % i], d['scores%d' % i] = o['rois'], o['class_ids'], o['scores']
    out0 = ref_uu.postprocess(x, anc, torch.from_numpy(reg), torch.from_numpy(cls * 0.01), ref_utils.BBoxTransform(),
                              ref_utils.ClipBoxes(), 0.05, 0.5)
    d['n_out_all_empty'] = np.array(len(out0))                         # the double-append quirk (:83-88)
    np.savez_compressed(os.path.join(OUT, 'postprocess.npz'), **d)
    print('postprocess: images', len(out), [o['scores'].shape for o in out], 'all-empty list length', len(out0))


def input_samples():
    """Seeded ragged batch for the input-side golden: decoded uint8 images + float64 annotation rows."""
    rng = np.random.default_rng(123)
    sizes = [(37, 53), (64, 40), (1, 1), (48, 64)]
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    imgs[0][0, 0] = (0, 255, 128)
    counts = [3, 0, 1, 5]
    annots = [np.concatenate([rng.random((m, 4)) * 60.0, rng.integers(0, 80, size=(m, 1)).astype(np.float64)], axis=1)
              for m in counts]
    return imgs, annots


def gen_input():
    """Input side (a-0 / f-2): datasettool/dataloader.py `Normalizer` (:905-924) + `collater` (:655-691) run as
    they are.  The module imports torchvision.transforms, skimage and the dataset tools at the top for its
    dataset / resampling classes; none is installed and none is touched by these two: empty stand-in modules
    let the file import."""
    for name in ('torchvision.transforms', 'torchvision.utils', 'skimage', 'skimage.io', 'skimage.transform',
                 'skimage.color', 'datasettool.ksevendata_tool', 'pycocotools', 'pycocotools.coco', 'cv2'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    sys.modules['datasettool.ksevendata_tool'].KSevenData = object
    sys.modules['pycocotools.coco'].COCO = object
    from datasettool import dataloader as ref_dl
    imgs, annots = input_samples()
    norm = ref_dl.Normalizer()
    data = []
    for im, an in zip(imgs, annots):
        s = norm({'img': im.astype(np.float32) / 255.0, 'annot': an.copy(), 'scale': 1.0})     # :92 then Normalizer
        data.append({'img': torch.from_numpy(s['img']), 'annot': torch.from_numpy(s['annot']), 'scale': s['scale']})
    out = ref_dl.collater(data)
    d = dict(img=out['img'].numpy(), annot=out['annot'].numpy())
    for i, (im, an) in enumerate(zip(imgs, annots)):
        d['u8_%d' % i], d['ann_%d' % i] = im, an
    empty = ref_dl.collater([{'img': dd['img'], 'annot': torch.zeros(0, 5, dtype=torch.float64), 'scale': 1.0} for dd in data[:2]])
    d['annot_all_empty'] = empty['annot'].numpy()
    np.savez_compressed(os.path.join(OUT, 'input.npz'), **d)
    print('input: img', d['img'].shape, d['img'].dtype, 'annot', d['annot'].shape, 'all-empty', d['annot_all_empty'].shape)


# ---------------------------------------------------------------- end to end RetinaNet (a-1..a-4)
def load_synth(model, depth, num_classes, gain):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    mine = ref_cpu.retinanet_state_shapes(depth, num_classes)
    assert list(shapes.keys()) == list(mine.keys()), 'state_dict key order differs'
    assert all(tuple(shapes[k]) == tuple(mine[k]) for k in shapes)
    sd = synth.synth_state_dict(shapes, seed=0, gain=gain)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return shapes


GRAD_KEYS = ['conv1.weight', 'bn1.weight', 'bn1.bias', 'layer1.0.conv1.weight',
             'layer2.0.downsample.0.weight', 'layer2.0.downsample.1.weight',
             'layer3.1.bn2.bias', 'layer4.1.conv2.weight', 'fpn.P5_1.weight', 'fpn.P3_2.bias',
             'fpn.P6.weight', 'fpn.P7_2.bias', 'regressionModel.conv1.weight',
             'regressionModel.output.bias', 'classificationModel.conv4.bias',
             'classificationModel.output.weight', 'classificationModel.output.bias']


def _add_fp64_run(d, model, img_t, ann_t, call, keys):
    """The SAME reference model evaluated in float64 (`model.double()`): the exact value of the arithmetic the
    fp32 reference run approximates.  Some parameter gradients are ill-conditioned sums (conv1.weight: 32 768
    random-sign terms per element), so the reference's own fp32 run is 1e-3 .. 3e-3 relative L2 away from it;
    `grad64/*` lets the fp32-mode parity test measure an implementation against the exact value with the
    reference's own fp32 noise (`grad/*` vs `grad64/*`) as the yardstick."""
    model.double()
    model.zero_grad()
    cl, rl = call(model, img_t.double(), ann_t.double())
    (cl.mean() + rl.mean()).backward()
    g64 = {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
    d['cls_loss64'], d['reg_loss64'] = cl.detach().numpy(), rl.detach().numpy()
    for k in keys:
        d['grad64/' + k + '/val'] = g64[k].reshape(-1)[d['grad/' + k + '/idx']]
    d['grad_global_norm64'] = np.array(np.sqrt(sum((g ** 2).sum() for g in g64.values())))
    model.zero_grad()


def gen_e2e(depth, H, W, B, num_classes, tag, gain):
    torch.manual_seed(0)
    model = ref_retinanet.retinanet(depth, num_classes=num_classes)
    shapes = load_synth(model, depth, num_classes, gain)
    model.train()
    model.freeze_bn()
    img = synth.synth_images(B, H, W, seed=1)
    ann = synth.synth_annotations(B, H, W, num_real=8, pad_to=16, num_classes=num_classes, seed=2)
    img_t, ann_t = torch.from_numpy(img), torch.from_numpy(ann)
    cls, reg = model(img_t, return_head=True)
    cl, rl = model(img_t, ann_t)
    loss = cl.mean() + rl.mean()                                  # train.py:312-314
    model.zero_grad()
    loss.backward()
    grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters()}
    d = dict(cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(),
             depth=np.array(depth), hw=np.array([H, W]), batch=np.array(B),
             num_classes=np.array(num_classes), gain=np.array(gain))
    A = cls.shape[1]
    idx = np.arange(0, A, 37)
    d['head_idx'] = idx
    d['cls_rows'] = cls.detach().numpy()[:, idx, :]
    d['reg_rows'] = reg.detach().numpy()[:, idx, :]
    d['cls_sum'] = np.array(cls.detach().double().sum().item())
    d['reg_abs_sum'] = np.array(reg.detach().double().abs().sum().item())
    for k in GRAD_KEYS:
        if k in grads:
            g = grads[k]
            d['grad/' + k + '/norm'] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
            flat = g.reshape(-1)
            sel = np.linspace(0, flat.size - 1, min(flat.size, 512)).astype(np.int64)
            d['grad/' + k + '/idx'] = sel
            d['grad/' + k + '/val'] = flat[sel]
    d['grad_global_norm'] = np.array(np.sqrt(sum((g.astype(np.float64) ** 2).sum()
                                                 for g in grads.values())))
    d['state_keys'] = np.array(list(shapes.keys()))
    d['state_shapes'] = np.array([','.join(map(str, shapes[k])) for k in shapes])
    _add_fp64_run(d, model, img_t, ann_t, lambda m, i, a: m(i, a), [k for k in GRAD_KEYS if k in grads])
    model.float()
    # decode tail on image 0 (B must be 1, retinanet.py:302)
    model.eval()
    with torch.no_grad():
        s, c, b = model(img_t[:1], return_loss=False)
    d['det_scores'] = s.numpy()
    d['det_classes'] = c.numpy()
    d['det_boxes'] = b.numpy()
    np.savez_compressed(os.path.join(OUT, 'e2e_%s.npz' % tag), **d)
    print('e2e', tag, 'cls', cl.item(), 'reg', rl.item(), 'dets', s.shape,
          'gnorm', d['grad_global_norm'])


def gen_e2e_trainbn(depth, H, W, B, num_classes, tag, gain):
    """RetinaNet with BatchNorm left in TRAIN mode (no freeze_bn): batch statistics per layer, running-statistics
    update, gradients through the normalisation (what `replace_w_sync_bn` + train() gives on one replica,
    utils/sync_batchnorm/batchnorm.py:65-69 -> F.batch_norm(training=True))."""
    torch.manual_seed(0)
    model = ref_retinanet.retinanet(depth, num_classes=num_classes)
    shapes = load_synth(model, depth, num_classes, gain)
    model.train()                                                 # BN in train mode: NOT frozen
    img = synth.synth_images(B, H, W, seed=1)
    ann = synth.synth_annotations(B, H, W, num_real=8, pad_to=16, num_classes=num_classes, seed=2)
    img_t, ann_t = torch.from_numpy(img), torch.from_numpy(ann)
    cl, rl = model(img_t, ann_t)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters()}
    d = dict(cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(), depth=np.array(depth), hw=np.array([H, W]),
             batch=np.array(B), num_classes=np.array(num_classes), gain=np.array(gain))
    keys = [k for k in GRAD_KEYS if k in grads] + ['layer1.0.bn1.weight', 'layer4.1.bn2.bias', 'layer3.0.downsample.1.weight']
    for k in keys:
        flat = grads[k].reshape(-1)
        sel = np.linspace(0, flat.size - 1, min(flat.size, 512)).astype(np.int64)
        d['grad/' + k + '/idx'] = sel
        d['grad/' + k + '/val'] = flat[sel]
    d['grad_global_norm'] = np.array(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))
    sd = model.state_dict()
    for k in ('bn1', 'layer1.0.bn2', 'layer2.0.downsample.1', 'layer4.1.bn2'):     # after ONE forward
        d['running/' + k + '.running_mean'] = sd[k + '.running_mean'].numpy().copy()
        d['running/' + k + '.running_var'] = sd[k + '.running_var'].numpy().copy()
        d['running/' + k + '.num_batches_tracked'] = sd[k + '.num_batches_tracked'].numpy().copy()
    # fp64 yardstick: reload the same state (the fp32 forward above moved the running statistics)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=0, gain=gain).items()})
    _add_fp64_run(d, model, img_t, ann_t, lambda m, i, a: m(i, a), keys)
    np.savez_compressed(os.path.join(OUT, 'e2e_%s.npz' % tag), **d)
    print('e2e', tag, 'cls', cl.item(), 'reg', rl.item(), 'gnorm', d['grad_global_norm'])


def gen_syncbn():
    """Statistics and normalisation of the reference's SynchronizedBatchNorm2d in its PARALLEL branch, driven on
    CPU: `_compute_mean_std` (utils/sync_batchnorm/batchnorm.py:128-145) is the reference's own method, called
    on an instance with the batch sums of :76-78; the output formula is :86 evaluated with autograd so that the
    backward through mean / inv_std (incl. the clamp of :145) is the reference's.  Channel 0 has a variance below
    eps (clamp active)."""
    from utils.sync_batchnorm.batchnorm import SynchronizedBatchNorm2d as RefSBN
    g = torch.Generator().manual_seed(7)
    N, C, H, W = 6, 16, 5, 7
    x = torch.randn(N, C, H, W, generator=g)
    x[:, 0] *= 1e-3
    x[:, 3] += 2.0
    dy = torch.randn(N, C, H, W, generator=g)
    bn = RefSBN(C, eps=1e-4, momentum=0.1)
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data = torch.randn(C, generator=g)
    bn.running_mean = torch.randn(C, generator=g) * 0.1
    bn.running_var = torch.rand(C, generator=g) + 0.5
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    xr = x.clone().requires_grad_(True)
    inp = xr.view(N, C, -1)
    sum_size = N * H * W
    input_sum = inp.sum(dim=0, keepdim=True).sum(dim=-1, keepdim=True).view(-1)          # _sum_ft, :34-36
    input_ssum = (inp ** 2).sum(dim=0, keepdim=True).sum(dim=-1, keepdim=True).view(-1)
    mean, inv_std = bn._compute_mean_std(input_sum, input_ssum, sum_size)
    out = (inp - mean.unsqueeze(0).unsqueeze(-1)) * (inv_std * bn.weight).unsqueeze(0).unsqueeze(-1) \
        + bn.bias.unsqueeze(0).unsqueeze(-1)                                             # :86
    y = out.view(N, C, H, W)
    (y * dy).sum().backward()
    d = dict(x=x.numpy(), dy=dy.numpy(), y=y.detach().numpy(), dx=xr.grad.numpy(), dweight=bn.weight.grad.numpy(),
             dbias=bn.bias.grad.numpy(), weight=bn.weight.detach().numpy(), bias=bn.bias.detach().numpy(),
             eps=np.array(bn.eps), momentum=np.array(bn.momentum), running_mean0=rm0.numpy(), running_var0=rv0.numpy(),
             running_mean1=bn.running_mean.numpy(), running_var1=bn.running_var.numpy(),
             mean=mean.detach().numpy(), inv_std=inv_std.detach().numpy())
    np.savez_compressed(os.path.join(OUT, 'syncbn.npz'), **d)
    print('syncbn: clamped channels', int((inv_std.detach() >= bn.eps ** -0.5 * 0.999).sum()))


def _load_by_path(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gen_e2e_kseven(neck, H, W, B, num_classes, tag, gain, train_bn_heads=False):
    """KSevenDet.  The reference's `ksevendet` PACKAGE cannot be imported, here or anywhere:
    ksevendet/architecture/backbone/registry.py:32 is a SyntaxError (`model_name inSelectAdaptivePool2d
    mod.default_cfgs`) and backbone/layers/helpers.py:6 needs `torch._six` (gone since torch 2.0).  The parts that
    are NEW in this family are plain-torch files and are loaded from their own paths, unmodified:
        ksevendet/architecture/neck/fpn.py        PANetFPN + topdown_lateral_module
        ksevendet/architecture/head/simple_head.py Regressor, Classifier (per-level BN over shared convs)
        ksevendet/architecture/anchors.py          Anchors (sizes_mapper / string scales)
    The ResNet-18 trunk is the reference's architectures/retinanet.py ResNet (same BasicBlock arithmetic and
    parameter names as backbone/resnet.py:116-176, :482-630; already pinned by e2e_r18_256), the loss the
    reference's FocalLoss; they are wired as ksevendet.py:197-231 does (backbone -> neck -> heads -> anchors ->
    loss) with the values of config/model/ksevendet_resnet.yaml."""
    import yaml
    root = '/root/reference/ksevendet/architecture/'
    fpn_mod = _load_by_path('ref_ks_fpn', root + 'neck/fpn.py')
    head_mod = _load_by_path('ref_ks_head', root + 'head/simple_head.py')
    anc_mod = _load_by_path('ref_ks_anchors', root + 'anchors.py')
    bifpn = neck.startswith('bifpn')
    if bifpn:
        # SURVEY 8 f-3, the BiFPN neck: ksevendet/architecture/neck/bifpn.py (BiFPN / BiFPNModule / SeparableConvBlock) with
        # the values of config/model/ksevendet_resnet_bifpn.yaml (features 128, unshared simple heads, head levels [3, 4, 5]);
        # 'bifpn2' = two repeats and all five neck levels handed to the heads (first_time / last_time modules both kinds)
        bifpn_mod = _load_by_path('ref_ks_bifpn', root + 'neck/bifpn.py')
        cfg = yaml.safe_load(open('/root/reference/config/model/ksevendet_resnet_bifpn.yaml'))
        assert cfg['variant'] == 'resnet18' and cfg['neck'] == 'bifpn' and not cfg['neck_config']['bifpn_attention']
        if neck == 'bifpn2':
            cfg['neck_config']['bifpn_repeats'] = 2
            cfg['head_feature_pyramid_levels'] = [3, 4, 5, 6, 7]
    else:
        cfg = yaml.safe_load(open('/root/reference/config/model/ksevendet_resnet.yaml'))
        assert cfg['variant'] == 'resnet18' and cfg['neck'] == 'fpn'
    head_levels = cfg.get('head_feature_pyramid_levels', cfg['neck_feature_pyramid_levels']) if bifpn else cfg['neck_feature_pyramid_levels']
    torch.manual_seed(0)

    class RefKSeven(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = ref_retinanet.retinanet18(num_classes)          # trunk only: conv1 .. layer4
            for name in ('fpn', 'regressionModel', 'classificationModel', 'anchors', 'focalLoss', 'regressBoxes', 'clipBoxes'):
                delattr(self.backbone, name)
            if bifpn:                                                       # ksevendet.py:64-86
                self.neck = bifpn_mod.BiFPN([128, 256, 512], in_pyramid_levels=cfg['backbone_feature_pyramid_levels'],
                                            neck_pyramid_levels=cfg['neck_feature_pyramid_levels'],
                                            out_pyramid_levels=head_levels, features_num=cfg['neck_config']['features_num'],
                                            repeats=cfg['neck_config']['bifpn_repeats'],
                                            attention=cfg['neck_config']['bifpn_attention'])
            else:
                self.neck = fpn_mod.PANetFPN([128, 256, 512], in_pyramid_levels=cfg['backbone_feature_pyramid_levels'],
                                             out_pyramid_levels=cfg['neck_feature_pyramid_levels'],
                                             features_num=cfg['neck_config']['features_num'],
                                             panet_buttomup=(neck == 'panet-fpn'))
            self.anchors = anc_mod.Anchors(pyramid_levels=head_levels, **cfg['anchors_config'])
            na = len(self.anchors.ratios) * len(self.anchors.scales)
            self.regressor = head_mod.Regressor(cfg['neck_config']['features_num'], num_anchors=na,
                                                num_pyramid_levels=len(head_levels), **cfg['head_config'])
            self.classifier = head_mod.Classifier(cfg['neck_config']['features_num'], num_anchors=na,
                                                  num_classes=num_classes, num_pyramid_levels=len(head_levels), **cfg['head_config'])
            self.focalLoss = ref_losses.FocalLoss()

        def heads(self, img):                                               # ksevendet.py:203-231
            b = self.backbone
            x = b.maxpool(b.relu(b.bn1(b.conv1(img))))
            x1 = b.layer1(x)
            x2 = b.layer2(x1)
            x3 = b.layer3(x2)
            x4 = b.layer4(x3)
            feats = self.neck([x2, x3, x4])
            return self.classifier(feats), self.regressor(feats)

        def forward(self, img, ann):
            classification, regression = self.heads(img)
            return self.focalLoss(classification, regression, self.anchors(img), ann)

    model = RefKSeven()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=gain)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    for m in model.modules():                                               # ksevendet.py:164, :185-189: freeze_bn
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    if train_bn_heads:                                                      # per-level head BN left in train mode
        for m in list(model.regressor.bn_modules.modules()) + list(model.classifier.bn_modules.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.train()
    img = synth.synth_images(B, H, W, seed=1)
    ann = synth.synth_annotations(B, H, W, num_real=8, pad_to=16, num_classes=num_classes, seed=2)
    img_t, ann_t = torch.from_numpy(img), torch.from_numpy(ann)
    with torch.no_grad():
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        cls, reg = model.heads(img_t)
        model.load_state_dict(state0)
    cl, rl = model(img_t, ann_t)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters()}
    d = dict(cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(), hw=np.array([H, W]), batch=np.array(B),
             num_classes=np.array(num_classes), gain=np.array(gain), neck=np.array(neck), head_levels=np.array(head_levels),
             repeats=np.array(cfg['neck_config'].get('bifpn_repeats', 0)),
             train_bn_heads=np.array(train_bn_heads))
    idx = np.arange(0, cls.shape[1], 29)
    d['head_idx'], d['cls_rows'], d['reg_rows'] = idx, cls.numpy()[:, idx, :], reg.numpy()[:, idx, :]
    d['anchors'] = model.anchors(img_t).numpy()
    keys = [k for k in grads if k.startswith(('neck.', 'regressor.', 'classifier.'))] + \
           ['backbone.conv1.weight', 'backbone.layer2.0.downsample.0.weight', 'backbone.layer4.1.bn2.bias']
    for k in keys:
        flat = grads[k].reshape(-1)
        sel = np.linspace(0, flat.size - 1, min(flat.size, 256)).astype(np.int64)
        d['grad/' + k + '/idx'] = sel
        d['grad/' + k + '/val'] = flat[sel]
    d['grad_global_norm'] = np.array(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))
    d['state_keys'] = np.array(list(shapes.keys()))
    d['state_shapes'] = np.array([','.join(map(str, shapes[k])) for k in shapes])
    model.load_state_dict(state0)
    _add_fp64_run(d, model, img_t, ann_t, lambda m, i, a: m(i, a), keys)
    np.savez_compressed(os.path.join(OUT, 'e2e_%s.npz' % tag), **d)
    print('e2e', tag, 'cls', cl.item(), 'reg', rl.item(), 'gnorm', d['grad_global_norm'], 'A', cls.shape[1])


def gen_e2e_variant(kind, H, W, B, num_classes, tag, gain):
    """RetinaNetTiny (retinanet.py:486-679) / RetinaNetP45P6 (retinanet.py:722-922): three-level variants."""
    torch.manual_seed(0)
    ctor = {'tiny': ref_retinanet.retinanet_tiny, 'p45p6': ref_retinanet.retinanet_p45p6}[kind]
    model = ctor(num_classes)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=gain)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    model.freeze_bn()
    img = synth.synth_images(B, H, W, seed=1)
    ann = synth.synth_annotations(B, H, W, num_real=8, pad_to=16, num_classes=num_classes, seed=2)
    img_t, ann_t = torch.from_numpy(img), torch.from_numpy(ann)
    if kind == 'tiny':
        cl, rl = model((img_t, ann_t))                            # retinanet.py:578-582
    else:
        cl, rl = model(img_t, ann_t)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters()}
    d = dict(cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(), kind=np.array(kind), hw=np.array([H, W]),
             batch=np.array(B), num_classes=np.array(num_classes), gain=np.array(gain))
    for k in sorted(grads)[::7] + [k for k in grads if k.startswith('fpn.')]:
        g = grads[k]
        flat = g.reshape(-1)
        sel = np.linspace(0, flat.size - 1, min(flat.size, 512)).astype(np.int64)
        d['grad/' + k + '/idx'] = sel
        d['grad/' + k + '/val'] = flat[sel]
    d['grad_global_norm'] = np.array(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))
    d['state_keys'] = np.array(list(shapes.keys()))
    d['state_shapes'] = np.array([','.join(map(str, shapes[k])) for k in shapes])
    _add_fp64_run(d, model, img_t, ann_t, (lambda m, i, a: m((i, a))) if kind == 'tiny' else (lambda m, i, a: m(i, a)),
                  [k[5:-4] for k in list(d) if k.startswith('grad/') and k.endswith('/val')])
    model.float()
    d['anchors'] = model.anchors(img_t).numpy()
    model.eval()
    with torch.no_grad():
        if kind == 'tiny':
            s, c, b = model(img_t[:1])
        else:
            s, c, b = model(img_t[:1], return_loss=False)
    d['det_scores'], d['det_classes'], d['det_boxes'] = s.numpy(), c.numpy(), b.numpy()
    np.savez_compressed(os.path.join(OUT, 'e2e_%s.npz' % tag), **d)
    print('e2e', tag, 'cls', cl.item(), 'reg', rl.item(), 'dets', s.shape, 'gnorm', d['grad_global_norm'],
          'anchors', d['anchors'].shape)


EFF_GRAD_KEYS = ['backbone_net.model._conv_stem.conv.weight', 'backbone_net.model._bn0.weight',
                 'backbone_net.model._blocks.0._depthwise_conv.conv.weight',
                 'backbone_net.model._blocks.0._se_reduce.conv.weight',
                 'backbone_net.model._blocks.0._se_expand.conv.bias',
                 'backbone_net.model._blocks.1._expand_conv.conv.weight',
                 'backbone_net.model._blocks.2._bn1.bias',
                 'backbone_net.model._blocks.5._project_conv.conv.weight',
                 'backbone_net.model._blocks.15._depthwise_conv.conv.weight',
                 'bifpn.0.p6_w1', 'bifpn.0.p4_w2', 'bifpn.1.p7_w2',
                 'bifpn.0.p5_down_channel.0.conv.weight', 'bifpn.0.p5_to_p6.1.weight',
                 'bifpn.0.conv6_up.depthwise_conv.conv.weight',
                 'bifpn.1.conv4_down.pointwise_conv.conv.bias', 'bifpn.2.conv3_up.bn.weight',
                 'regressor.conv_list.0.depthwise_conv.conv.weight',
                 'regressor.conv_list.1.pointwise_conv.conv.weight', 'regressor.bn_list.3.1.bias',
                 'regressor.header.pointwise_conv.conv.bias',
                 'classifier.conv_list.2.pointwise_conv.conv.bias', 'classifier.bn_list.0.0.weight',
                 'classifier.header.depthwise_conv.conv.weight',
                 'classifier.header.pointwise_conv.conv.weight']


def gen_e2e_effdet(coef, H, W, B, num_classes, tag, gain, head_std):
    from architectures import efficientdet as ref_eff
    torch.manual_seed(0)
    model = ref_eff.build_efficientdet('efficientdet-d%d' % coef, num_classes=num_classes)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=gain, head_std=head_std)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    # deterministic path: frozen BN + drop-connect off (SURVEY.md 8c iii) == eval mode for
    # every module whose behaviour depends on `training` (BatchNorm2d, MBConvBlock.drop_connect)
    model.eval()
    img = synth.synth_images(B, H, W, seed=1)
    ann = synth.synth_annotations(B, H, W, num_real=6, pad_to=8, num_classes=num_classes, seed=2)
    img_t, ann_t = torch.from_numpy(img), torch.from_numpy(ann)
    feats = model.bifpn(tuple(model.backbone_net(img_t)[1:]))
    reg = model.regressor(feats)
    cls = model.classifier(feats)
    cl, rl = model(img_t, ann_t)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
    d = dict(cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(), coef=np.array(coef),
             hw=np.array([H, W]), batch=np.array(B), num_classes=np.array(num_classes),
             gain=np.array(gain), head_std=np.array(head_std))
    A = cls.shape[1]
    idx = np.arange(0, A, 11)
    d['head_idx'] = idx
    d['cls_rows'] = cls.detach().numpy()[:, idx, :]
    d['reg_rows'] = reg.detach().numpy()[:, idx, :]
    for lvl, f in enumerate(feats):
        d['feat%d_sample' % lvl] = f.detach().numpy()[:, ::7, ::3, ::3]
    for k in EFF_GRAD_KEYS:
        if k in grads:
            g = grads[k]
            d['grad/' + k + '/norm'] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
            flat = g.reshape(-1)
            sel = np.linspace(0, flat.size - 1, min(flat.size, 512)).astype(np.int64)
            d['grad/' + k + '/idx'] = sel
            d['grad/' + k + '/val'] = flat[sel]
    d['grad_global_norm'] = np.array(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))
    d['state_keys'] = np.array(list(shapes.keys()))
    d['state_shapes'] = np.array([','.join(map(str, shapes[k])) for k in shapes])
    with torch.no_grad():
        s, c, b = model(img_t[:1], return_loss=False)
    d['det_scores'], d['det_classes'], d['det_boxes'] = s.numpy(), c.numpy(), b.numpy()
    # the same model in float64: the yardstick for the fp32-mode gradient bound (as for RetinaNet, _add_fp64_run)
    _add_fp64_run(d, model, img_t, ann_t, lambda m, i, a: m(i, a), [k for k in EFF_GRAD_KEYS if k in grads])
    model.float()
    np.savez_compressed(os.path.join(OUT, 'e2e_%s.npz' % tag), **d)
    print('e2e', tag, 'cls', cl.item(), 'reg', rl.item(), 'dets', s.shape, 'gnorm', d['grad_global_norm'],
          'cls range', float(cls.min()), float(cls.max()))


def gen_effdet_trainbn(coef, H, W, B, num_classes, tag, gain, head_std):
    """EfficientDet with every BatchNorm2d left in TRAIN mode (`model.train()` without `freeze_bn()`; what
    `replace_w_sync_bn(model)` + train() computes on one replica, utils/sync_batchnorm/batchnorm.py:65-69): batch
    statistics in the stem, the three BNs of every MBConv block (expand / depthwise / project), the BiFPN's
    down-channel and separable convs and the heads' per-level BNs; running statistics after ONE forward (momentum
    0.01, eps 1e-3, efficientdet.py:36, utils_efficientnet.py GlobalParams); drop-connect off."""
    from architectures import efficientdet as ref_eff
    torch.manual_seed(0)
    model = ref_eff.build_efficientdet('efficientdet-d%d' % coef, num_classes=num_classes)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}

    def load():
        sd = synth.synth_state_dict(shapes, seed=0, gain=gain, head_std=head_std)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    load()
    model.train()
    synth.disable_drop_connect(model)
    img = synth.synth_images(B, H, W, seed=1)
    ann = synth.synth_annotations(B, H, W, num_real=6, pad_to=8, num_classes=num_classes, seed=2)
    img_t, ann_t = torch.from_numpy(img), torch.from_numpy(ann)
    cl, rl = model(img_t, ann_t)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
    d = dict(cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(), coef=np.array(coef), hw=np.array([H, W]),
             batch=np.array(B), num_classes=np.array(num_classes), gain=np.array(gain), head_std=np.array(head_std))
    keys = [k for k in EFF_GRAD_KEYS if k in grads] + [
        'backbone_net.model._blocks.0._bn1.weight', 'backbone_net.model._blocks.3._bn0.bias',
        'backbone_net.model._blocks.7._bn2.weight', 'bifpn.0.p4_down_channel.1.bias',
        'regressor.bn_list.0.1.weight', 'classifier.bn_list.2.0.bias', 'classifier.bn_list.4.2.weight']
    for k in keys:
        flat = grads[k].reshape(-1)
        sel = np.linspace(0, flat.size - 1, min(flat.size, 512)).astype(np.int64)
        d['grad/' + k + '/idx'] = sel
        d['grad/' + k + '/val'] = flat[sel]
        d['grad/' + k + '/norm'] = np.array(np.sqrt((grads[k].astype(np.float64) ** 2).sum()))
    d['grad_global_norm'] = np.array(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))
    sd = model.state_dict()
    for k in ('backbone_net.model._bn0', 'backbone_net.model._blocks.0._bn1', 'backbone_net.model._blocks.4._bn0',
              'backbone_net.model._blocks.15._bn2', 'bifpn.0.p5_to_p6.1', 'bifpn.1.conv4_up.bn',
              'regressor.bn_list.4.0', 'classifier.bn_list.0.2'):
        d['running/' + k + '.running_mean'] = sd[k + '.running_mean'].numpy().copy()
        d['running/' + k + '.running_var'] = sd[k + '.running_var'].numpy().copy()
        d['running/' + k + '.num_batches_tracked'] = sd[k + '.num_batches_tracked'].numpy().copy()
    load()                                                        # (the fp32 forward moved the running statistics)
    _add_fp64_run(d, model, img_t, ann_t, lambda m, i, a: m(i, a), keys)
    np.savez_compressed(os.path.join(OUT, 'e2e_%s.npz' % tag), **d)
    print('e2e', tag, 'cls', cl.item(), 'reg', rl.item(), 'gnorm', d['grad_global_norm'])


def gen_effdet_variants():
    """State-dict keys / shapes, parameter counts and anchor configuration of every EfficientDet-D0 .. D7 the reference
    builds (efficientdet.py:442-495 tables, :588 build_efficientdet): the host-API test checks the mirrored builders
    against them (only D0 / D2 / D6 have end-to-end goldens)."""
    from architectures import efficientdet as ref_eff
    d = {}
    for coef in range(8):
        torch.manual_seed(0)
        m = ref_eff.build_efficientdet('efficientdet-d%d' % coef, num_classes=7)
        sd = m.state_dict()
        d['d%d/keys' % coef] = np.array(list(sd.keys()))
        d['d%d/shapes' % coef] = np.array([','.join(map(str, v.shape)) for v in sd.values()])
        d['d%d/nparams' % coef] = np.array(sum(p.numel() for p in m.parameters()))
        a = m.anchors
        d['d%d/anchor_levels' % coef] = np.array(a.pyramid_levels)
        d['d%d/anchor_strides' % coef] = np.array(a.strides, dtype=np.float64)
        d['d%d/anchor_sizes' % coef] = np.array(a.sizes, dtype=np.float64)
        d['d%d/anchor_ratios' % coef] = np.array(a.ratios, dtype=np.float64)
        d['d%d/anchor_scales' % coef] = np.array(a.scales, dtype=np.float64)
        d['d%d/attention' % coef] = np.array([bool(getattr(c, 'attention', True)) for c in m.bifpn])
        print('effdet variant d%d: %d entries, %.2f M parameters, %d BiFPN cells' % (coef, len(sd), float(d['d%d/nparams' % coef]) / 1e6, len(m.bifpn)))
    np.savez_compressed(os.path.join(OUT, 'effdet_variants.npz'), **d)


def gen_effdet_drop_connect(coef, H, W, B, num_classes, tag, gain, head_std):
    """EfficientDet in the state the reference trains it in (train.py:273-286: `.train()` then `freeze_bn()`): the
    MBConv blocks are in train mode, so drop-connect (utils_efficientnet.py:85-94) draws one torch.rand per image
    and skip block.  The draws are recorded so that the HIP path and the oracle can be fed the same uniforms."""
    from architectures import efficientdet as ref_eff
    torch.manual_seed(0)
    model = ref_eff.build_efficientdet('efficientdet-d%d' % coef, num_classes=num_classes)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=gain, head_std=head_std)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    model.freeze_bn()
    img_t = torch.from_numpy(synth.synth_images(B, H, W, seed=1))
    ann_t = torch.from_numpy(synth.synth_annotations(B, H, W, num_real=6, pad_to=8, num_classes=num_classes, seed=2))
    draws, real_rand = [], torch.rand

    def recording_rand(*a, **k):
        t = real_rand(*a, **k)
        draws.append(t.reshape(-1).clone())
        return t
    torch.manual_seed(4321)
    torch.rand = recording_rand
    try:
        cl, rl = model(img_t, ann_t)
    finally:
        torch.rand = real_rand
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
    u = torch.stack(draws).numpy()
    d = dict(cls_loss=cl.detach().numpy(), reg_loss=rl.detach().numpy(), coef=np.array(coef), hw=np.array([H, W]),
             batch=np.array(B), num_classes=np.array(num_classes), gain=np.array(gain), head_std=np.array(head_std),
             dc_u=u)
    for k in EFF_GRAD_KEYS:
        if k in grads:
            g = grads[k]
            d['grad/' + k + '/norm'] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
            flat = g.reshape(-1)
            sel = np.linspace(0, flat.size - 1, min(flat.size, 512)).astype(np.int64)
            d['grad/' + k + '/idx'] = sel
            d['grad/' + k + '/val'] = flat[sel]
    d['grad_global_norm'] = np.array(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))
    d['state_keys'] = np.array(list(shapes.keys()))
    d['state_shapes'] = np.array([','.join(map(str, shapes[k])) for k in shapes])
    np.savez_compressed(os.path.join(OUT, 'e2e_%s.npz' % tag), **d)
    keep = 1 - 0.2 * np.arange(16) / 16
    print('e2e', tag, 'cls', cl.item(), 'reg', rl.item(), 'draws', u.shape, 'dropped branches',
          int((u < 0.2).sum()), 'gnorm', d['grad_global_norm'])


if __name__ == '__main__':
    which = sys.argv[1:] or ['anchors', 'loss', 'decode', 'e2e', 'effdet', 'variants', 'postprocess', 'input']
    if 'anchors' in which:
        gen_anchors()
    if 'loss' in which:
        gen_loss()
    if 'decode' in which:
        gen_decode()
    if 'postprocess' in which:
        gen_postprocess()
    if 'input' in which:
        gen_input()
    if 'kseven' in which:
        gen_e2e_kseven('fpn', 128, 160, 2, 12, 'kseven_r18_fpn_128x160', 0.87)
        gen_e2e_kseven('panet-fpn', 128, 128, 2, 12, 'kseven_r18_panet_128', 0.87)
        gen_e2e_kseven('fpn', 128, 128, 4, 12, 'kseven_r18_headbn_128', 0.87, train_bn_heads=True)
    if 'kseven_bifpn' in which:
        # (gains: the BiFPN's un-normalised node sums amplify; 0.87 gives logits in the hundreds, <= 0.7 leaves the prior)
        gen_e2e_kseven('bifpn', 128, 256, 2, 12, 'kseven_r18_bifpn_128x256', 0.8)
        gen_e2e_kseven('bifpn2', 128, 128, 2, 12, 'kseven_r18_bifpn2_128', 0.76)
    if 'trainbn' in which:
        gen_e2e_trainbn(18, 128, 128, 4, 12, 'r18_trainbn_128', 0.87)
    if 'syncbn' in which:
        gen_syncbn()
    if 'variants' in which:
        gen_e2e_variant('tiny', 64, 96, 2, 12, 'tiny_64x96', 0.9)
        gen_e2e_variant('p45p6', 192, 288, 1, 12, 'p45p6_192x288', 0.87)
    if 'e2e' in which:
        gen_e2e(18, 256, 256, 2, 80, 'r18_256', 0.87)     # BASELINE.json configs[0]
        gen_e2e(50, 128, 160, 1, 20, 'r50_128x160', 0.78) # bottleneck path, ragged H != W
    if 'e2e_depths' in which:                             # the other ResNet depths of retinanet.py:399-445 (basic / bottleneck)
        gen_e2e(34, 64, 96, 1, 12, 'r34_64x96', 0.8)      # BasicBlock [3, 4, 6, 3]
        gen_e2e(101, 64, 96, 1, 12, 'r101_64x96', 0.7)    # Bottleneck [3, 4, 23, 3]
    if 'effdet' in which:
        gen_e2e_effdet(0, 128, 128, 2, 8, 'effdet_d0_128', 1.0, 0.05)      # configs[3] family
        gen_e2e_effdet(2, 128, 256, 1, 5, 'effdet_d2_128x256', 1.0, 0.05)  # configs[4] family
    if 'effdet_trainbn' in which:
        gen_effdet_trainbn(0, 128, 256, 3, 8, 'effdet_d0_trainbn_128x256', 1.0, 0.05)
    if 'effdet_variants' in which:
        gen_effdet_variants()
    if 'effdet_dc' in which:
        gen_effdet_drop_connect(0, 128, 128, 4, 8, 'effdet_d0_dropconnect_128', 1.0, 0.05)
    if 'effdet6' in which:
        gen_e2e_effdet(6, 128, 128, 1, 4, 'effdet_d6_128', 0.73, 0.05)      # plain-sum BiFPN (attention=False)
