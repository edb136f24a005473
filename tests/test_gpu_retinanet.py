"""GPU end-to-end parity of the RetinaNet launch plan (bf16 MFMA convs, fp32 loss) against
(a) golden vectors from the reference (fp32 CPU) and (b) the fp32 oracle.

The convs run in bf16 with fp32 accumulation (BASELINE.json configs[1]); the reference is fp32,
so head tensors / losses / gradients are compared at bf16-level tolerances stated below, while
everything downstream of the head tensors (assignment, loss arithmetic, decode, NMS) is checked
at fp32 / bit-exact level in test_gpu_detect.py on identical inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


def _build(tag, golden_dir, dev):
    from objectdetection.pytorch_amd.architectures import retinanet
    g = np.load(os.path.join(golden_dir, 'e2e_%s.npz' % tag))
    depth, (H, W), B, nc = int(g['depth']), g['hw'], int(g['batch']), int(g['num_classes'])
    model = retinanet.retinanet(depth, num_classes=nc)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=float(g['gain']))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)
    model.train()
    model.freeze_bn()
    img = torch.from_numpy(synth.synth_images(B, int(H), int(W), seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(B, int(H), int(W), 8, 16, nc, seed=2)).to(dev)
    return g, model, img, ann, sd


def _rel(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)


@pytest.mark.parametrize('tag', ['r18_256', 'r50_128x160'])
def test_heads_losses_grads_vs_reference_golden(golden_dir, dev, tag):
    g, model, img, ann, _ = _build(tag, golden_dir, dev)
    cls, reg = model(img, return_head=True)
    idx = g['head_idx']
    # bf16 activations through ~25 (R18) / ~60 (R50) layers: 3e-2 of the tensor's range
    assert _rel(reg.cpu().numpy()[:, idx], g['reg_rows']) < 3e-2
    assert np.abs(cls.cpu().numpy()[:, idx] - g['cls_rows']).max() < 3e-2 * g['cls_rows'].max()
    cl, rl = model(img, ann)
    assert cl.shape == (1,) and rl.shape == (1,) and cl.requires_grad
    np.testing.assert_allclose(cl.item(), g['cls_loss'].item(), rtol=3e-2)
    np.testing.assert_allclose(rl.item(), g['reg_loss'].item(), rtol=3e-2)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()                     # train.py:312-319
    grads = dict((k, p.grad) for k, p in model.named_parameters())
    assert all(v is not None for v in grads.values())
    gn = np.sqrt(sum(float((v.double() ** 2).sum()) for v in grads.values()))
    np.testing.assert_allclose(gn, float(g['grad_global_norm']), rtol=5e-2)
    worst = 0.0
    for key in g.files:
        if key.startswith('grad/') and key.endswith('/val'):
            name = key[5:-4]
            sel = torch.from_numpy(g['grad/%s/idx' % name]).to(dev)
            got = grads[name].reshape(-1)[sel].cpu().numpy()   # reshape follows logical OIHW order
            ref = g[key]
            # bf16 storage of ~60 layers of activations + gradients, ReLU-mask flips included:
            # measured 2-9 % relative L2 per tensor against BOTH the fp32 reference and a
            # bf16-emulating oracle (tools/debug_grads.py) -> direction + magnitude checks here,
            # per-kernel 2e-3 checks in test_gpu_conv.py
            cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
            ratio = float(np.linalg.norm(got) / (np.linalg.norm(ref) + 1e-30))
            worst = min(worst if worst else 1.0, cos)
            rl2 = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))
            print('  golden relL2 %-44s %.4f (cos %.4f, norm ratio %.3f)' % (name, rl2, cos, ratio))
            # relative L2 per tensor against the fp32 reference (no cosine gate): about twice the spread of the bf16
            # plan against its own bf16 twin (TWIN_* below), measured here 0.036 / 0.127 general, 0.098 / 0.263 stem
            assert rl2 <= _bound(GOLDEN_REL_L2, tag, name), (name, rl2)
            assert 0.95 < ratio < 1.05, (name, ratio)
    print('worst sampled-grad cosine', worst)


# per-tensor relative-L2 bound of the bf16 plan against its CPU twin (`emulate_bf16=True`: weights rounded after
# the frozen-BN fold, input / every stored activation / every stored activation-gradient rounded to bf16, fp32
# accumulation).  What is left between the two is accumulation order plus the ReLU-mask / max-pool-argmax flips
# that order causes; bounds = 2x the spread measured on MI355X (printed by the test).  conv1.weight is the
# ill-conditioned one: each element sums ~32 768 random-sign terms (the reference's own fp32 run is 3.3e-3 from
# its fp64 value there, 10x its other backbone tensors: tests/golden `grad64/*`), so bf16 noise shows 10x as well.
# Measured on MI355X (deterministic kernels: the same on every box), r18_256 / r50_128x160 (one image, 4 x 5 cells in layer4):
#   general        0.035 / 0.154 (layer3.3.bn1.bias)           -> 0.055 / 0.23   (1.5 x)
#   conv1.weight   0.052 / 0.194                               -> 0.08  / 0.30
#   fpn.P7_2.*     0.070 / 0.219 (1 x 2 cells at 128 x 160)    -> 0.11  / 0.33
#   fpn.P5_2.*     0.104 / (general)                           -> 0.16
TWIN_REL_L2 = {'r18_256': {'': 0.055, 'conv1.weight': 0.08, 'fpn.P7_2.': 0.11, 'fpn.P5_2.': 0.16},
               'r50_128x160': {'': 0.23, 'conv1.weight': 0.30, 'fpn.P7_2.': 0.33}}
# the same plan against the reference's fp32 goldens: ~2 x the twin spread per class
GOLDEN_REL_L2 = {'r18_256': {'': 0.07, 'conv1.weight': 0.15, 'fpn.P7_2.': 0.16, 'fpn.P5_2.': 0.20},
                 'r50_128x160': {'': 0.25, 'conv1.weight': 0.40, 'fpn.P7_2.': 0.40}}


def _bound(table, tag, name):
    """Longest matching name prefix of the tag's table ('' = every other tensor)."""
    t = table[tag]
    best = max((k for k in t if name.startswith(k)), key=len)
    return t[best]


@pytest.mark.parametrize('tag', ['r18_256', 'r50_128x160'])
def test_vs_oracle_same_bf16_rounded_weights(golden_dir, dev, tag):
    """The bf16 mode against the oracle run as its numerical twin (same rounding points, fp32 accumulate):
    heads at 2.5e-2 and losses at 1e-2 (the fp32 reference only allows 3e-2), every sampled parameter gradient by
    relative L2 (not a cosine gate: a wrong tap or a dropped term in a small layer shows up here)."""
    g, model, img, ann, sd = _build(tag, golden_dir, dev)
    depth = int(g['depth'])
    p = {k: torch.from_numpy(v).clone().requires_grad_(v.dtype == np.float32 and 'running' not in k)
         for k, v in sd.items()}
    cl_o, rl_o = ref_cpu.retinanet_loss(p, img.cpu(), ann.cpu(), depth, emulate_bf16=True)
    (cl_o.mean() + rl_o.mean()).backward()
    with torch.no_grad():
        cls_o, reg_o = ref_cpu.retinanet_heads(p, img.cpu(), depth, emulate_bf16=True)
    cls, reg = model(img, return_head=True)
    assert _rel(reg.cpu().numpy(), reg_o.numpy()) < 2.5e-2        # (3e-2 against the fp32 reference; measured 1.1-2.0e-2)
    assert _rel(cls.cpu().numpy(), cls_o.numpy()) < 2.5e-2
    cl, rl = model(img, ann)
    np.testing.assert_allclose(cl.item(), cl_o.item(), rtol=1e-2)
    np.testing.assert_allclose(rl.item(), rl_o.item(), rtol=1e-2)
    # loss on the DEVICE head tensors, oracle arithmetic: isolates the fused loss at full size
    anchors = torch.from_numpy(ref_cpu.anchors_for_image(img.shape[2], img.shape[3]))
    cl_e, rl_e = ref_cpu.focal_loss(cls.cpu(), reg.cpu(), anchors, ann.cpu())
    np.testing.assert_allclose(cl.item(), cl_e.item(), rtol=1e-4)     # north_star tolerance
    np.testing.assert_allclose(rl.item(), rl_e.item(), rtol=1e-4)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    rows = []
    for name, prm in model.named_parameters():
        got = prm.grad.detach().cpu().double().reshape(-1)
        ref = p[name].grad.double().reshape(-1)
        if float(ref.norm()) < 1e-12:
            continue
        rows.append((float((got - ref).norm() / ref.norm()), name))
    rows.sort(reverse=True)
    print('bf16 plan vs bf16 twin, %s: worst per-tensor relL2 %s' % (tag, ['%s %.3f' % (n, r) for r, n in rows[:6]]))
    for r, n in rows:
        print('  twin relL2 %-44s %.4f' % (n, r))
    for r, name in rows:
        assert r <= _bound(TWIN_REL_L2, tag, name), (name, r)


def test_decode_tail_vs_golden(golden_dir, dev):
    g, model, img, ann, _ = _build('r18_256', golden_dir, dev)
    model.eval()
    scores, classes, boxes = model(img[:1], return_loss=False)
    assert classes.dtype == torch.int64 and boxes.shape[1] == 4
    s = scores.cpu().numpy()
    assert np.all(s[:-1] >= s[1:])                           # descending score order
    assert np.all(s > 0.05)
    # detections are a discrete function of bf16-perturbed scores: compare the count loosely and
    # the top detections tightly
    n_ref = g['det_scores'].shape[0]
    assert abs(len(s) - n_ref) <= 0.15 * n_ref
    np.testing.assert_allclose(s[:20], g['det_scores'][:20], rtol=5e-2)
    # exact pipeline check on the device head tensors with the oracle's decode tail
    cls, reg = model(img[:1], return_head=True)
    anchors = torch.from_numpy(ref_cpu.anchors_for_image(256, 256))
    so, co, bo = ref_cpu.decode_tail(cls.cpu(), reg.cpu(), anchors, 256, 256)
    assert len(so) == len(s)
    np.testing.assert_array_equal(classes.cpu().numpy(), co.numpy())
    np.testing.assert_allclose(s, so.numpy(), rtol=1e-6)
    np.testing.assert_allclose(boxes.cpu().numpy(), bo.numpy(), rtol=1e-4, atol=1e-3)


def test_grad_accumulation_and_arena_reuse(golden_dir, dev):
    g, model, img, ann, _ = _build('r18_256', golden_dir, dev)
    model.zero_grad()
    cl, rl = model(img, ann)
    (cl + rl).sum().backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    cl, rl = model(img, ann)
    (cl + rl).sum().backward()                              # no zero_grad: must accumulate
    for k, p in model.named_parameters():
        ref = 2 * g1[k]
        assert float((p.grad - ref).abs().max()) <= 2e-2 * float(ref.abs().max()) + 1e-12, k
    model.zero_grad()
    cl, rl = model(img, ann)
    (cl + rl).sum().backward()
    for k, p in model.named_parameters():
        assert float((p.grad - g1[k]).abs().max()) <= 2e-2 * float(g1[k].abs().max()) + 1e-12, k


@pytest.mark.parametrize('tag', ['tiny_64x96', 'p45p6_192x288'])
def test_three_level_variants_vs_reference_golden(golden_dir, dev, tag):
    """RetinaNetTiny (retinanet.py:486-679, `forward((img, annot))` switching on self.training) and
    RetinaNetP45P6 (retinanet.py:722-922, P6 = conv5x5 stride 3): state_dict keys, anchors, losses,
    sampled parameter gradients and the decode tail against the reference's outputs."""
    from objectdetection.pytorch_amd.architectures import retinanet
    g = np.load(os.path.join(golden_dir, 'e2e_%s.npz' % tag))
    kind, (H, W), B, nc = str(g['kind']), g['hw'], int(g['batch']), int(g['num_classes'])
    model = {'tiny': retinanet.retinanet_tiny, 'p45p6': retinanet.retinanet_p45p6}[kind](nc)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert list(shapes.keys()) == list(g['state_keys'])
    sd = synth.synth_state_dict(shapes, seed=0, gain=float(g['gain']))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)
    model.train()
    model.freeze_bn()
    img = torch.from_numpy(synth.synth_images(B, int(H), int(W), seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(B, int(H), int(W), 8, 16, nc, seed=2)).to(dev)
    np.testing.assert_allclose(model.anchors(img).cpu().numpy(), g['anchors'], rtol=1e-6, atol=1e-4)
    cl, rl = model((img, ann)) if kind == 'tiny' else model(img, ann)
    np.testing.assert_allclose(cl.item(), g['cls_loss'].item(), rtol=3e-2)
    np.testing.assert_allclose(rl.item(), g['reg_loss'].item(), rtol=3e-2)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = dict((k, p.grad) for k, p in model.named_parameters())
    gn = np.sqrt(sum(float((v.double() ** 2).sum()) for v in grads.values()))
    np.testing.assert_allclose(gn, float(g['grad_global_norm']), rtol=5e-2)
    for key in g.files:
        if key.startswith('grad/') and key.endswith('/val'):
            name = key[5:-4]
            sel = torch.from_numpy(g['grad/%s/idx' % name]).to(dev)
            got, ref = grads[name].reshape(-1)[sel].cpu().numpy(), g[key]
            if np.linalg.norm(ref) < 1e-12:
                assert np.linalg.norm(got) < 1e-6
                continue
            cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
            ratio = float(np.linalg.norm(got) / (np.linalg.norm(ref) + 1e-30))
            rl2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            # per-tensor relative L2 against the fp32 reference: 1.5 x the measured worst (tiny 0.113, p45p6 0.124)
            assert rl2 <= {'tiny': 0.17, 'p45p6': 0.19}[kind] and 0.9 < ratio < 1.1, (name, rl2, ratio)
    model.eval()
    s, c, b = model(img[:1]) if kind == 'tiny' else model(img[:1], return_loss=False)
    ref_s = g['det_scores']
    # bf16 scores move detections across the NMS / 0.05 thresholds: compare the confident head of the list
    top = min(10, len(ref_s), s.numel())
    np.testing.assert_allclose(s[:top].cpu().numpy(), ref_s[:top], rtol=0.1, atol=2e-2)
    assert abs(s.numel() - len(ref_s)) <= max(3, len(ref_s) // 5)
    assert c.dtype == torch.int64 and b.shape == (s.numel(), 4)


def test_plan_cache_keeps_several_shapes(golden_dir, dev):
    """The reference's collater pads every batch to its own max shape (dataloader.py:655-691): alternating
    shapes must reuse their launch plans (LRU, MI355_MAX_PLANS) instead of rebuilding one per batch, and the
    losses of a shape must not depend on what ran in between."""
    g, model, img, ann, _ = _build('r18_256', golden_dir, dev)
    small = img[:, :, :128, :192].contiguous()
    ann_s = ann.clone()
    ann_s[..., :4] = ann_s[..., :4].clamp(max=120.0)
    with torch.no_grad():
        a1 = [t.clone() for t in model(img, ann)]
        pa = model._plans[(img.shape[0], 256, 256, dev.index)]
        b1 = [t.clone() for t in model(small, ann_s)]
        pb = model._plans[(img.shape[0], 128, 192, dev.index)]
        a2 = model(img, ann)
        b2 = model(small, ann_s)
    assert model._plans[(img.shape[0], 256, 256, dev.index)] is pa and model._plans[(img.shape[0], 128, 192, dev.index)] is pb
    assert len(model._plans) == 2
    assert torch.equal(a1[0], a2[0]) and torch.equal(a1[1], a2[1]) and torch.equal(b1[0], b2[0]) and torch.equal(b1[1], b2[1])


def test_coco_result_writer(tmp_path, monkeypatch):
    """f-4: datasettool/coco_eval.py:6-63 over the device decode path: scale correction, xywh, threshold break,
    category mapping, JSON file."""
    import json
    from objectdetection.pytorch_amd.architectures import retinanet
    from objectdetection.pytorch_amd.datasettool import coco_eval
    dev = torch.device('cuda', 0)
    model = retinanet.retinanet(18, num_classes=8)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=0.87)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)

    class DS:
        set_name = 'val_synth'
        image_ids = [11, 42, 7]
        coco = None

        def __len__(self):
            return 3

        def __getitem__(self, i):
            im = synth.synth_images(1, 128, 160, seed=30 + i)[0].transpose(1, 2, 0).copy()       # HWC, as the datasets give
            return {'img': torch.from_numpy(im), 'scale': [0.5, 1.25, 1.0][i]}

        def label_to_coco_label(self, label):
            return 100 + label

    ds = DS()
    monkeypatch.chdir(tmp_path)
    model.eval()
    s0 = model(ds[0]['img'].permute(2, 0, 1).to(dev).unsqueeze(0), return_loss=False)[0]
    thr = float(s0.median())                                  # cuts every image's sorted list somewhere inside
    got = coco_eval.evaluate_coco(ds, model, threshold=thr)
    assert model.training                                     # coco_eval.py:78
    want = []
    model.eval()
    for i in range(3):
        d = ds[i]
        s, l, b = model(d['img'].permute(2, 0, 1).to(dev).unsqueeze(0), return_loss=False)
        s, l, b = s.cpu(), l.cpu(), b.cpu() / d['scale']
        for k in range(b.shape[0]):
            if float(s[k]) < thr:
                break
            want.append({'image_id': ds.image_ids[i], 'category_id': 100 + int(l[k]), 'score': float(s[k]),
                         'bbox': [float(b[k, 0]), float(b[k, 1]), float(b[k, 2] - b[k, 0]), float(b[k, 3] - b[k, 1])]})
    assert len(want) > 0 and len(got) == len(want)
    for g, w in zip(got, want):
        assert g['image_id'] == w['image_id'] and g['category_id'] == w['category_id'] and g['score'] == w['score']
        assert g['bbox'] == w['bbox']
    on_disk = json.load(open(tmp_path / 'val_synth_bbox_results.json'))
    assert on_disk == got
