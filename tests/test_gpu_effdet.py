"""GPU end-to-end parity of the EfficientDet launch plan (D0, D2) against golden vectors generated
from the reference (fp32 CPU, BN frozen, drop-connect off) and against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu, ref_effdet
import synth

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _build(tag, golden_dir, drop_connect=False):
    from objectdetection.pytorch_amd.architectures import efficientdet
    g = np.load(os.path.join(golden_dir, 'e2e_%s.npz' % tag))
    coef, (H, W), B, nc = int(g['coef']), g['hw'], int(g['batch']), int(g['num_classes'])
    model = efficientdet.build_efficientdet('efficientdet-d%d' % coef, num_classes=nc)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=float(g['gain']), head_std=float(g['head_std']))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(DEV)
    model.train()
    model.freeze_bn()
    if not drop_connect:                  # goldens made with the reference in eval mode: no drop-connect draws
        synth.disable_drop_connect(model)
    img = torch.from_numpy(synth.synth_images(B, int(H), int(W), seed=1)).to(DEV)
    ann = torch.from_numpy(synth.synth_annotations(B, int(H), int(W), 6, 8, nc, seed=2)).to(DEV)
    return g, model, img, ann, sd, coef, nc


# measured worst relL2 (stem / first MBConv block / everything else): d0 0.133 / 0.090 / 0.083, d2 0.319 / 0.167 / 0.126,
# d6 0.255 / 0.235 / 0.064, d0 with drop-connect 0.187 / 0.102 / 0.077
_STEM, _B0 = 'backbone_net.model._conv_stem', 'backbone_net.model._blocks.0.'
BF16_REL_L2 = {'effdet_d0_128': {'': 0.125, _STEM: 0.20, _B0: 0.14},
               'effdet_d2_128x256': {'': 0.19, _STEM: 0.48, _B0: 0.25},
               'effdet_d6_128': {'': 0.10, _STEM: 0.39, _B0: 0.36},
               'effdet_d0_dropconnect_128': {'': 0.115, _STEM: 0.28, _B0: 0.16}}


def _bound(table, name):
    return table[max((k for k in table if name.startswith(k)), key=len)]


@pytest.mark.parametrize('tag', ['effdet_d0_128', 'effdet_d2_128x256', 'effdet_d6_128'])
def test_effdet_vs_reference_golden(golden_dir, tag):
    g, model, img, ann, sd, coef, nc = _build(tag, golden_dir)
    cls, reg = model(img, return_head=True)
    idx = g['head_idx']
    # bf16 storage of every activation through ~100-150 layers (MBConv + 3-5 BiFPN cells + heads):
    # measured 1-3 % of the tensor range per stage (tools/debug_effdet.py)
    # (D6: 8 cells + 5-layer heads, and at 128 x 128 every pyramid map is < 16 wide: since the head convs of all levels
    #  run in one LDS-tiled launch their swish(z) inputs are rounded to bf16 like everywhere else, 5.6 % measured)
    tol = 7e-2 if coef >= 6 else 5e-2
    assert np.abs(reg.cpu().numpy()[:, idx] - g['reg_rows']).max() < tol * np.abs(g['reg_rows']).max()
    assert np.abs(cls.cpu().numpy()[:, idx] - g['cls_rows']).max() < tol * g['cls_rows'].max()
    cl, rl = model(img, ann)
    np.testing.assert_allclose(cl.item(), g['cls_loss'].item(), rtol=3e-2)
    np.testing.assert_allclose(rl.item(), g['reg_loss'].item(), rtol=3e-2)
    # exact part: fused loss on the device heads vs the oracle's arithmetic
    anchors = torch.from_numpy(ref_cpu.anchors_for_image(img.shape[2], img.shape[3], **ref_effdet.ANCHOR_CFG))
    clo, rlo = ref_cpu.focal_loss(cls.cpu(), reg.cpu(), anchors, ann.cpu())
    np.testing.assert_allclose(cl.item(), clo.item(), rtol=1e-4)
    np.testing.assert_allclose(rl.item(), rlo.item(), rtol=1e-4)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = dict((k, p.grad) for k, p in model.named_parameters())
    # D6 / D7 run the plain-sum BiFPN (efficientdet.py:269-317): the fusion parameters feed nothing and the
    # reference leaves their .grad None (so clip_grad_norm_ / AdamW skip them, train.py:320-321)
    import re
    unused = set(k for k in grads if re.search(r'\.p\d_w[12]$', k)) if coef >= 6 else set()
    assert all((v is None) == (k in unused) for k, v in grads.items()), [k for k, v in grads.items() if v is None]
    assert coef < 6 or len(unused) > 0
    gn = np.sqrt(sum(float((v.double() ** 2).sum()) for v in grads.values() if v is not None))
    np.testing.assert_allclose(gn, float(g['grad_global_norm']), rtol=6e-2)
    worst = 1.0
    for key in g.files:
        if key.startswith('grad/') and key.endswith('/val'):
            name = key[5:-4]
            if name in unused:
                assert np.linalg.norm(g[key]) == 0.0
                continue
            sel = torch.from_numpy(g['grad/%s/idx' % name]).to(DEV)
            got = grads[name].reshape(-1)[sel].cpu().numpy()
            ref = g[key]
            if np.linalg.norm(ref) == 0.0:                 # e.g. box-head BN of a level without positives
                assert np.abs(got).max() <= 1e-6 * (gn + 1e-30), name
                continue
            if np.linalg.norm(ref) < 1e-5 * gn:
                # near-zero gradient (e.g. bifpn.0.p6_w1: two components of +-1e-5 that cancel, at a global
                # norm of O(10)): differences of large bf16-rounded sums -> absolute check against the norm
                assert np.abs(got - ref).max() <= 1e-5 * gn, (name, got, ref)
                continue
            cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
            ratio = float(np.linalg.norm(got) / (np.linalg.norm(ref) + 1e-30))
            worst = min(worst, cos)
            rl2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            # per-tensor relative L2 against the fp32 reference, 1.5 x what MI355X measures per class (the kernels are
            # deterministic); the stem and the first block sit behind every bf16-stored activation of the network
            assert rl2 <= _bound(BF16_REL_L2[tag], name), (name, rl2)
            assert 0.9 < ratio < 1.1, (name, ratio)
    print(tag, 'worst sampled-grad cosine', worst)


@pytest.mark.parametrize('case', ['ragged', 'empty'])
def test_effdet_ragged_and_empty_annotations(golden_dir, case):
    """The collater's -1 padding rows (datasettool/dataloader.py:655-691) through the EfficientDet plan: an image without a
    single box beside one with boxes, and a batch without any (losses.py:50-60: both losses 0, no gradient), against the
    oracle in the fp32-accurate mode."""
    g, model, img, ann, sd, coef, nc = _build('effdet_d0_128', golden_dir)
    model.precision = 'fp32'
    ann = ann.clone()
    if case == 'ragged':
        ann[0] = -1.0                                       # image 0: only padding rows
        ann[1, 3:] = -1.0                                   # image 1: three boxes, then padding
    else:
        ann[:] = -1.0
    cl, rl = model(img, ann)
    p = {k: torch.from_numpy(v).clone() for k, v in sd.items()}
    clo, rlo = ref_effdet.efficientdet_loss(p, img.cpu(), ann.cpu(), coef, nc)
    if case == 'empty':
        assert float(cl) == 0.0 and float(rl) == 0.0 and float(clo) == 0.0 and float(rlo) == 0.0
    else:
        np.testing.assert_allclose(cl.item(), clo.mean().item(), rtol=1e-4)
        np.testing.assert_allclose(rl.item(), rlo.mean().item(), rtol=1e-4)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    gn = np.sqrt(sum(float((q.grad.double() ** 2).sum()) for q in model.parameters() if q.grad is not None))
    assert np.isfinite(gn) and ((gn == 0.0) if case == 'empty' else (gn > 0.0))


def test_effdet_atomic_weight_gradient_mode(golden_dir, monkeypatch):
    """MI355_WGRAD=atomic (fp32 atomics for the conv weight gradients, ungrouped head convs) stays a working mode: same
    losses, gradients equal to the default slab mode up to the order of fp32 additions."""
    g, model, img, ann, sd, coef, nc = _build('effdet_d0_128', golden_dir)

    def run(m):
        m.zero_grad()
        cl, rl = m(img, ann)
        (cl.mean() + rl.mean()).backward()
        return torch.stack([cl.detach().reshape(()), rl.detach().reshape(())]), \
            torch.cat([q.grad.reshape(-1) for q in m.parameters() if q.grad is not None]).clone()
    l0, g0 = run(model)
    monkeypatch.setenv('MI355_WGRAD', 'atomic')
    _, model2, _, _, _, _, _ = _build('effdet_d0_128', golden_dir)
    l1, g1 = run(model2)
    assert torch.equal(l0, l1)
    assert float((g0 - g1).norm() / g0.norm()) < 1e-4


def test_effdet_decode_tail(golden_dir):
    g, model, img, ann, sd, coef, nc = _build('effdet_d2_128x256', golden_dir)
    model.eval()
    scores, classes, boxes = model(img[:1], return_loss=False)
    cls, reg = model(img[:1], return_head=True)
    anchors = torch.from_numpy(ref_cpu.anchors_for_image(img.shape[2], img.shape[3], **ref_effdet.ANCHOR_CFG))
    so, co, bo = ref_cpu.decode_tail(cls.cpu(), reg.cpu(), anchors, img.shape[2], img.shape[3])
    assert len(so) == len(scores)
    np.testing.assert_array_equal(classes.cpu().numpy(), co.numpy())
    np.testing.assert_allclose(scores.cpu().numpy(), so.numpy(), rtol=1e-6)
    np.testing.assert_allclose(boxes.cpu().numpy(), bo.numpy(), rtol=1e-4, atol=1e-3)
    n_ref = g['det_scores'].shape[0]
    assert abs(len(so) - n_ref) <= max(3, 0.2 * n_ref)


def test_drop_connect_vs_reference_golden(golden_dir):
    """Train mode as the reference trains (train.py:273-286): drop-connect on the MBConv skip branches
    (utils_efficientnet.py:85-94), fed with the uniforms the reference drew when the golden was made."""
    g, model, img, ann, sd, coef, nc = _build('effdet_d0_dropconnect_128', golden_dir, drop_connect=True)
    plan = model._plan(img)
    assert plan.drop_connect and plan._dc_rows == g['dc_u'].shape[0]
    plan.dc_inject = torch.from_numpy(g['dc_u']).to(DEV)
    cl, rl = model(img, ann)
    np.testing.assert_allclose(cl.item(), g['cls_loss'].item(), rtol=3e-2)
    np.testing.assert_allclose(rl.item(), g['reg_loss'].item(), rtol=3e-2)
    model.zero_grad()
    (cl.mean() + rl.mean()).backward()
    grads = dict((k, p.grad) for k, p in model.named_parameters())
    gn = np.sqrt(sum(float((v.double() ** 2).sum()) for v in grads.values()))
    np.testing.assert_allclose(gn, float(g['grad_global_norm']), rtol=6e-2)
    for key in g.files:
        if key.startswith('grad/') and key.endswith('/val'):
            name = key[5:-4]
            got = grads[name].reshape(-1)[torch.from_numpy(g['grad/%s/idx' % name]).to(DEV)].cpu().numpy()
            ref = g[key]
            if np.linalg.norm(ref) < 1e-5 * gn:
                assert np.abs(got - ref).max() <= 1e-5 * gn, name
                continue
            rl2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            assert rl2 <= _bound(BF16_REL_L2['effdet_d0_dropconnect_128'], name), (name, rl2)
    # all branches kept (u = 1 - tiny): out = branch / keep + skip; all dropped (u = 0, rate > 0): out = skip
    plan.dc_inject = torch.full_like(plan.dc_inject, 0.999999)
    kept = [x.item() for x in model(img, ann)]
    plan.dc_inject = torch.zeros_like(plan.dc_inject)
    dropped = [x.item() for x in model(img, ann)]
    assert np.isfinite(kept + dropped).all() and kept != dropped
    # the draws of the product path come from the device generator: two steps differ, eval mode has no draws
    plan.dc_inject = None
    a, b = [x.item() for x in model(img, ann)], [x.item() for x in model(img, ann)]
    assert a != b
    model.eval()
    assert not model._plan(img).drop_connect
    e1, e2 = [x.item() for x in model(img, ann)], [x.item() for x in model(img, ann)]
    assert e1 == e2
