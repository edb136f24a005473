"""Pins oracle/ref_cpu.py to golden vectors produced by the reference itself
(tools/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
import synth

EFFDET = dict(pyramid_levels=[3, 4, 5, 6, 7], strides=[8, 16, 32, 64, 128],
              sizes=[2 ** (x + 1) * 1.25 for x in [3, 4, 5, 6, 7]], ratios=[1, 2])


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_anchors_bit_exact(golden_dir):
    g = _load(golden_dir, 'anchors.npz')
    for name in g['names']:
        name = str(name)
        kind, hw = name.split('_')
        h, w = map(int, hw.split('x'))
        a = ref_cpu.anchors_for_image(h, w, **(EFFDET if kind == 'effdet' else {}))
        assert tuple(a.shape) == tuple(g[name + '_shape'])
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(g[name + '_sha256'])
        if name + '_full' in g:
            assert np.array_equal(a, g[name + '_full'])
        else:
            assert np.array_equal(a[0, g[name + '_idx']], g[name + '_rows'])


def test_first_anchor_value():
    a = ref_cpu.anchors_for_image(512, 512)
    assert a.shape == (1, 49104, 4)
    np.testing.assert_array_equal(
        a[0, 0], np.array([-18.627417, -7.3137083, 26.627417, 15.313708], dtype=np.float32))


@pytest.mark.parametrize('case', ['c4_256', 'c80_64', 'edges', 'eff_96x160'])
def test_loss_and_assignment(golden_dir, case):
    g = _load(golden_dir, 'loss.npz')
    cls = torch.from_numpy(g[case + '_cls']).requires_grad_(True)
    reg = torch.from_numpy(g[case + '_reg']).requires_grad_(True)
    anc = torch.from_numpy(g[case + '_anchors'])
    ann = torch.from_numpy(g[case + '_ann'])
    cl, rl = ref_cpu.focal_loss(cls, reg, anc, ann)
    np.testing.assert_allclose(cl.detach().numpy(), g[case + '_cls_loss'], rtol=1e-6)
    np.testing.assert_allclose(rl.detach().numpy(), g[case + '_reg_loss'], rtol=1e-6)
    gc, = torch.autograd.grad(cl, cls, retain_graph=True)
    np.testing.assert_allclose(gc.numpy(), g[case + '_grad_cls'], rtol=1e-5, atol=1e-9)
    if rl.requires_grad:
        gr, = torch.autograd.grad(rl, reg)
        np.testing.assert_allclose(gr.numpy(), g[case + '_grad_reg'], rtol=1e-5, atol=1e-9)
    for j in range(cls.shape[0]):                          # integer work: bit exact
        arg, state, ioumax = ref_cpu.assign(anc[0], ann[j])
        assert np.array_equal(state.numpy(), g[case + '_state'][j])
        assert np.array_equal(arg.numpy().astype(np.int32), g[case + '_argmax'][j])
        assert np.array_equal(ioumax.numpy(), g[case + '_iou_max'][j])


def test_edges_case_semantics(golden_dir):
    """IoU exactly 0.45 is positive, exactly 0.35 is ignore, ties go to the lowest index."""
    g = _load(golden_dir, 'loss.npz')
    st, am = g['edges_state'][0], g['edges_argmax'][0]
    assert st[0] == 2 and st[1] == 1 and st[3] == 0 and st[2] == 2
    assert am[2] == 2          # filtered rows: gt(40..) twins are rows 2 and 3 -> lowest


def test_decode_clip(golden_dir):
    g = _load(golden_dir, 'decode.npz')
    boxes = ref_cpu.bbox_transform(torch.from_numpy(g['anchors']), torch.from_numpy(g['reg']))
    np.testing.assert_array_equal(boxes.numpy(), g['boxes'])
    h, w = g['hw']
    np.testing.assert_array_equal(ref_cpu.clip_boxes(boxes, int(h), int(w)).numpy(), g['clipped'])


@pytest.mark.parametrize('tag', ['r18_256', 'r50_128x160', 'r34_64x96', 'r101_64x96'])
def test_retinanet_end_to_end(golden_dir, tag):
    g = _load(golden_dir, 'e2e_%s.npz' % tag)
    depth, (H, W), B, nc = int(g['depth']), g['hw'], int(g['batch']), int(g['num_classes'])
    shapes = ref_cpu.retinanet_state_shapes(depth, nc)
    assert list(shapes.keys()) == [str(k) for k in g['state_keys']]
    assert [','.join(map(str, shapes[k])) for k in shapes] == [str(s) for s in g['state_shapes']]
    sd = synth.synth_state_dict(shapes, seed=0, gain=float(g['gain']))
    p = {k: torch.from_numpy(v).requires_grad_(v.dtype == np.float32 and 'running' not in k)
         for k, v in sd.items()}
    img = torch.from_numpy(synth.synth_images(B, int(H), int(W), seed=1))
    ann = torch.from_numpy(synth.synth_annotations(B, int(H), int(W), 8, 16, nc, seed=2))
    cls, reg = ref_cpu.retinanet_heads(p, img, depth)
    idx = g['head_idx']
    np.testing.assert_allclose(cls.detach().numpy()[:, idx], g['cls_rows'], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(reg.detach().numpy()[:, idx], g['reg_rows'], rtol=2e-4, atol=2e-5)
    anchors = torch.from_numpy(ref_cpu.anchors_for_image(int(H), int(W)))
    cl, rl = ref_cpu.focal_loss(cls, reg, anchors, ann)
    np.testing.assert_allclose(cl.detach().numpy(), g['cls_loss'], rtol=1e-4)
    np.testing.assert_allclose(rl.detach().numpy(), g['reg_loss'], rtol=1e-4)
    (cl.mean() + rl.mean()).backward()
    for key in g.files:
        if key.startswith('grad/') and key.endswith('/val'):
            name = key[5:-4]
            got = p[name].grad.reshape(-1)[torch.from_numpy(g['grad/%s/idx' % name])].numpy()
            scale = float(g['grad/%s/norm' % name]) / np.sqrt(p[name].numel())
            np.testing.assert_allclose(got, g[key], rtol=2e-3, atol=2e-3 * scale, err_msg=name)
    # decode tail (B = 1, retinanet.py:302)
    with torch.no_grad():
        s, c, b = ref_cpu.decode_tail(cls[:1].detach(), reg[:1].detach(), anchors, int(H), int(W))
    assert s.shape == tuple(g['det_scores'].shape)
    np.testing.assert_allclose(s.numpy(), g['det_scores'], rtol=2e-4)
    assert (c.numpy() == g['det_classes']).mean() > 0.999
    np.testing.assert_allclose(b.numpy(), g['det_boxes'], rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize('tag', ['effdet_d0_128', 'effdet_d2_128x256', 'effdet_d6_128'])
def test_efficientdet_end_to_end(golden_dir, tag):
    from oracle import ref_effdet
    g = _load(golden_dir, 'e2e_%s.npz' % tag)
    coef, (H, W), B, nc = int(g['coef']), g['hw'], int(g['batch']), int(g['num_classes'])
    shapes = {str(k): tuple(int(v) for v in str(s).split(',') if v) for k, s in zip(g['state_keys'], g['state_shapes'])}
    sd = synth.synth_state_dict(shapes, seed=0, gain=float(g['gain']), head_std=float(g['head_std']))
    p = {k: torch.from_numpy(v).requires_grad_(v.dtype == np.float32 and 'running' not in k)
         for k, v in sd.items()}
    img = torch.from_numpy(synth.synth_images(B, int(H), int(W), seed=1))
    ann = torch.from_numpy(synth.synth_annotations(B, int(H), int(W), 6, 8, nc, seed=2))
    cls, reg = ref_effdet.efficientdet_heads(p, img, coef, nc)
    idx = g['head_idx']
    np.testing.assert_allclose(cls.detach().numpy()[:, idx], g['cls_rows'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(reg.detach().numpy()[:, idx], g['reg_rows'], rtol=1e-4, atol=1e-5)
    anchors = torch.from_numpy(ref_cpu.anchors_for_image(int(H), int(W), **ref_effdet.ANCHOR_CFG))
    cl, rl = ref_cpu.focal_loss(cls, reg, anchors, ann)
    np.testing.assert_allclose(cl.detach().numpy(), g['cls_loss'], rtol=1e-5)
    np.testing.assert_allclose(rl.detach().numpy(), g['reg_loss'], rtol=1e-5)
    (cl.mean() + rl.mean()).backward()
    for key in g.files:
        if key.startswith('grad/') and key.endswith('/val'):
            name = key[5:-4]
            got = p[name].grad.reshape(-1)[torch.from_numpy(g['grad/%s/idx' % name])].numpy()
            scale = float(g['grad/%s/norm' % name]) / np.sqrt(p[name].numel())
            np.testing.assert_allclose(got, g[key], rtol=2e-3, atol=2e-3 * scale, err_msg=name)
    with torch.no_grad():
        s, c, b = ref_cpu.decode_tail(cls[:1].detach(), reg[:1].detach(), anchors, int(H), int(W))
    assert s.shape == tuple(g['det_scores'].shape)
    np.testing.assert_allclose(s.numpy(), g['det_scores'], rtol=1e-4)
    np.testing.assert_allclose(b.numpy(), g['det_boxes'], rtol=1e-3, atol=1e-2)


def test_efficientdet_drop_connect(golden_dir):
    """Train-mode MBConv skip branches (efficientnet.py:109-113, utils_efficientnet.py:85-94) with the reference's
    own draws: the oracle restates `x / keep * floor(keep + u)` per image."""
    from oracle import ref_effdet
    g = _load(golden_dir, 'e2e_effdet_d0_dropconnect_128.npz')
    coef, (H, W), B, nc = int(g['coef']), g['hw'], int(g['batch']), int(g['num_classes'])
    shapes = {str(k): tuple(int(v) for v in str(s).split(',') if v) for k, s in zip(g['state_keys'], g['state_shapes'])}
    sd = synth.synth_state_dict(shapes, seed=0, gain=float(g['gain']), head_std=float(g['head_std']))
    p = {k: torch.from_numpy(v).requires_grad_(v.dtype == np.float32 and 'running' not in k) for k, v in sd.items()}
    img = torch.from_numpy(synth.synth_images(B, int(H), int(W), seed=1))
    ann = torch.from_numpy(synth.synth_annotations(B, int(H), int(W), 6, 8, nc, seed=2))
    cls, reg = ref_effdet.efficientdet_heads(p, img, coef, nc, dc_u=torch.from_numpy(g['dc_u']))
    anchors = torch.from_numpy(ref_cpu.anchors_for_image(int(H), int(W), **ref_effdet.ANCHOR_CFG))
    cl, rl = ref_cpu.focal_loss(cls, reg, anchors, ann)
    np.testing.assert_allclose(cl.detach().numpy(), g['cls_loss'], rtol=1e-5)
    np.testing.assert_allclose(rl.detach().numpy(), g['reg_loss'], rtol=1e-5)
    (cl.mean() + rl.mean()).backward()
    for key in g.files:
        if key.startswith('grad/') and key.endswith('/val'):
            name = key[5:-4]
            got = p[name].grad.reshape(-1)[torch.from_numpy(g['grad/%s/idx' % name])].numpy()
            scale = float(g['grad/%s/norm' % name]) / np.sqrt(p[name].numel())
            np.testing.assert_allclose(got, g[key], rtol=2e-3, atol=2e-3 * scale, err_msg=name)
    # without the draws the losses differ: the golden really exercises the dropped branches
    cls0, reg0 = ref_effdet.efficientdet_heads(p, img, coef, nc)
    cl0, _ = ref_cpu.focal_loss(cls0, reg0, anchors, ann)
    assert abs(cl0.item() - g['cls_loss'].item()) > 1e-4 * g['cls_loss'].item()


def test_input_side_normalizer_collater(golden_dir):
    """a-0 / f-2: the numpy restatement of load scaling + Normalizer + collater against the reference's own
    output (datasettool/dataloader.py:92, :905-924, :655-691), bit for bit."""
    from oracle import ref_input
    g = np.load(os.path.join(golden_dir, 'input.npz'))
    imgs = [g['u8_%d' % i] for i in range(4)]
    anns = [g['ann_%d' % i] for i in range(4)]
    img, ann = ref_input.collate([ref_input.normalize(ref_input.load_scale(i)) for i in imgs], anns)
    assert img.dtype == np.float32 and np.array_equal(img, g['img'])
    assert np.array_equal(ann, g['annot'])
    _, ann0 = ref_input.collate([ref_input.normalize(ref_input.load_scale(i)) for i in imgs[:2]],
                                [np.zeros((0, 5))] * 2)
    assert np.array_equal(ann0, g['annot_all_empty'])
    assert img[2, :, 1:, :].max() == 0.0 and img[2, :, 0, 1:].max() == 0.0      # the 1x1 image: zero padding
