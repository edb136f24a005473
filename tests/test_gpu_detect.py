"""GPU parity: HIP detection primitives (through the C ABI) vs golden vectors and the oracle.
Bit-exact for integer/index work (assignment state + argmax, anchors, NMS keep list);
fp32 losses/grads/boxes within 1e-4 relative (north_star tolerance)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
import synth

pytestmark = pytest.mark.gpu

EFFDET = dict(pyramid_levels=[3, 4, 5, 6, 7], strides=[8, 16, 32, 64, 128],
              sizes=[2 ** (x + 1) * 1.25 for x in [3, 4, 5, 6, 7]], ratios=[1, 2])
RTOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda', 0)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_library_loaded():
    from objectdetection.pytorch_amd import _lib
    assert _lib.lib().mi355_version() >= 1


def test_anchors_golden_bit_exact(golden_dir, dev):
    from objectdetection.pytorch_amd.architectures.anchors import Anchors
    g = _g(golden_dir, 'anchors.npz')
    for name in g['names']:
        name = str(name)
        kind, hw = name.split('_')
        h, w = map(int, hw.split('x'))
        mod = Anchors(**(EFFDET if kind == 'effdet' else {}))
        a = mod(torch.zeros(1, 3, h, w, device=dev)).cpu().numpy()
        assert tuple(a.shape) == tuple(g[name + '_shape'])
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(g[name + '_sha256']), name


@pytest.mark.parametrize('hw', [(512, 512), (640, 640), (768, 768), (517, 333), (1, 1), (1024, 1333)])
def test_anchors_vs_oracle_full_size(dev, hw):
    from objectdetection.pytorch_amd.architectures.anchors import Anchors
    a = Anchors()(torch.zeros(1, 3, hw[0], hw[1], device=dev)).cpu().numpy()
    assert np.array_equal(a, ref_cpu.anchors_for_image(*hw))


@pytest.mark.parametrize('case', ['c4_256', 'c80_64', 'edges', 'eff_96x160'])
def test_focal_loss_golden(golden_dir, dev, case):
    from objectdetection.pytorch_amd.architectures.losses import FocalLoss
    g = _g(golden_dir, 'loss.npz')
    cls = torch.from_numpy(g[case + '_cls']).to(dev).requires_grad_(True)
    reg = torch.from_numpy(g[case + '_reg']).to(dev).requires_grad_(True)
    anc = torch.from_numpy(g[case + '_anchors']).to(dev)
    ann = torch.from_numpy(g[case + '_ann']).to(dev)
    fl = FocalLoss()
    cl, rl = fl(cls, reg, anc, ann)
    argmax, state, num_pos = fl.last_assignment
    assert np.array_equal(state.cpu().numpy(), g[case + '_state'])                    # bit exact
    valid = g[case + '_state'] != 3
    assert np.array_equal(argmax.cpu().numpy()[valid], g[case + '_argmax'][valid])    # bit exact
    assert np.array_equal(num_pos.cpu().numpy(), (g[case + '_state'] == 2).sum(1))
    np.testing.assert_allclose(cl.detach().cpu().numpy(), g[case + '_cls_loss'], rtol=RTOL)
    np.testing.assert_allclose(rl.detach().cpu().numpy(), g[case + '_reg_loss'], rtol=RTOL, atol=1e-7)
    gc, = torch.autograd.grad(cl, cls, retain_graph=True)
    gr, = torch.autograd.grad(rl, reg)
    ref_gc, ref_gr = g[case + '_grad_cls'], g[case + '_grad_reg']
    np.testing.assert_allclose(gc.cpu().numpy(), ref_gc, rtol=RTOL, atol=1e-6 * np.abs(ref_gc).max())
    np.testing.assert_allclose(gr.cpu().numpy(), ref_gr, rtol=RTOL, atol=1e-6 * max(np.abs(ref_gr).max(), 1e-30))


@pytest.mark.parametrize('B,H,W,C', [(2, 512, 512, 80), (3, 200, 328, 7)])
def test_focal_loss_vs_oracle_full_size(dev, B, H, W, C):
    """BASELINE configs[1] anchor count (A = 49104, C = 80) + a ragged C (scalar path)."""
    from objectdetection.pytorch_amd.architectures.losses import FocalLoss
    anc_np = ref_cpu.anchors_for_image(H, W)
    A = anc_np.shape[1]
    cls_np, reg_np = synth.synth_head_outputs(B, A, C, seed=5)
    ann_np = synth.synth_annotations(B, H, W, 8, 16, C, seed=6, empty_images=(1,))
    cls_c = torch.from_numpy(cls_np).requires_grad_(True)
    reg_c = torch.from_numpy(reg_np).requires_grad_(True)
    cl_ref, rl_ref = ref_cpu.focal_loss(cls_c, reg_c, torch.from_numpy(anc_np), torch.from_numpy(ann_np))
    (cl_ref * 1.5 + rl_ref * 0.5).sum().backward()
    cls = torch.from_numpy(cls_np).to(dev).requires_grad_(True)
    reg = torch.from_numpy(reg_np).to(dev).requires_grad_(True)
    fl = FocalLoss()
    cl, rl = fl(cls, reg, torch.from_numpy(anc_np).to(dev), torch.from_numpy(ann_np).to(dev))
    (cl * 1.5 + rl * 0.5).sum().backward()
    np.testing.assert_allclose(cl.detach().cpu().numpy(), cl_ref.detach().numpy(), rtol=RTOL)
    np.testing.assert_allclose(rl.detach().cpu().numpy(), rl_ref.detach().numpy(), rtol=RTOL)
    argmax, state, _ = fl.last_assignment
    for j in range(B):
        a_ref, s_ref, _ = ref_cpu.assign(torch.from_numpy(anc_np[0]), torch.from_numpy(ann_np[j]))
        assert np.array_equal(state[j].cpu().numpy(), s_ref.numpy())
        if (s_ref != 3).any():
            assert np.array_equal(argmax[j].cpu().numpy(), a_ref.numpy().astype(np.int32))
    for got, ref in ((cls.grad, cls_c.grad), (reg.grad, reg_c.grad)):
        ref = ref.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL, atol=1e-6 * np.abs(ref).max())


def test_focal_loss_all_images_empty(dev):
    from objectdetection.pytorch_amd.architectures.losses import FocalLoss
    anc = torch.from_numpy(ref_cpu.anchors_for_image(64, 64)).to(dev)
    A = anc.shape[1]
    cls = torch.rand(2, A, 8, device=dev, requires_grad=True)
    reg = torch.randn(2, A, 4, device=dev, requires_grad=True)
    ann = -torch.ones(2, 1, 5, device=dev)                 # collater's all-empty batch shape
    cl, rl = FocalLoss()(cls, reg, anc, ann)
    assert float(cl) == 0.0 and float(rl) == 0.0
    (cl + rl).sum().backward()
    assert float(cls.grad.abs().sum()) == 0.0 and float(reg.grad.abs().sum()) == 0.0


def test_decode_clip_golden(golden_dir, dev):
    from objectdetection.pytorch_amd.architectures.utils import BBoxTransform, ClipBoxes
    g = _g(golden_dir, 'decode.npz')
    boxes = BBoxTransform()(torch.from_numpy(g['anchors']).to(dev), torch.from_numpy(g['reg']).to(dev))

    def check(got, ref, raw):
        # A corner is centre -/+ half the size (utils.py:46-51): its rounding error scales with those OPERANDS, not with
        # the corner itself (x1 = 2.5 may come from 400.0 - 397.5).  Bound: 1e-4 relative on the corner (north star) plus
        # 4 fp32 ulps (4.8e-7) of the box's operand scale max(|corner|, width, height) of the UNCLIPPED reference box -
        # libm expf vs the device's exp (<= 2 ulp on the size) and fused vs separate multiply-add on the centre (measured: 1.41).
        scale = np.maximum(np.abs(raw).max(-1), np.maximum(raw[..., 2] - raw[..., 0], raw[..., 3] - raw[..., 1]))[..., None]
        err = np.abs(got.astype(np.float64) - ref)
        bound = RTOL * np.abs(ref) + 4 * 2.0 ** -23 * scale
        assert (err <= bound).all(), (float((err / bound).max()), float(err.max()))
        return float((err / (2.0 ** -23 * scale)).max())
    u1 = check(boxes.cpu().numpy(), g['boxes'], g['boxes'])
    h, w = int(g['hw'][0]), int(g['hw'][1])
    ret = ClipBoxes()(boxes, torch.zeros(2, 3, h, w, device=dev))
    assert ret.data_ptr() == boxes.data_ptr()              # mutates and returns (utils.py:61-71)
    u2 = check(boxes.cpu().numpy(), g['clipped'], g['boxes'])
    print('decode: worst error %.2f / %.2f ulps of the operand scale' % (u1, u2))


def _rand_boxes(rng, n, span=200.0, size=60.0):
    xy = rng.uniform(0, span, (n, 2))
    wh = rng.uniform(1, size, (n, 2))
    return np.concatenate([xy, xy + wh], 1).astype(np.float32)


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 1000, 1023, 1024, 1025, 2048, 2049, 5000, 20000, 46706])
def test_nms_vs_oracle_bit_exact(dev, n):
    from objectdetection.pytorch_amd import ops_detect
    rng = np.random.default_rng(n)
    b = _rand_boxes(rng, n, span=40.0 * np.sqrt(n) / 4 + 50)
    s = rng.random(n).astype(np.float32)
    s[rng.integers(0, n, n // 10)] = 0.5                    # equal scores -> index order
    keep = ops_detect.nms(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), 0.5)
    assert keep.dtype == torch.int64
    assert np.array_equal(keep.cpu().numpy(), ref_cpu.nms(b, s, 0.5))


@pytest.mark.parametrize('n,span', [(8192, 300.0), (8193, 300.0), (12289, 5000.0), (20000, 100.0), (30000, 1500.0)])
def test_nms_two_phase_schedule(dev, n, span):
    """Above ops_detect.NMS_DIRECT_MAX candidates the head / filtered-tail schedule runs: sparse boxes (nothing
    suppressed: every round keeps its whole head and the tail survives), dense ones (almost everything suppressed by the
    first head) and sizes around the switch must give the one-pass result = the oracle's."""
    from objectdetection.pytorch_amd import ops_detect
    rng = np.random.default_rng(n)
    b = _rand_boxes(rng, n, span=span)
    s = rng.random(n).astype(np.float32)
    s[rng.integers(0, n, n // 10)] = 0.25
    keep = ops_detect.nms(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), 0.5)
    assert np.array_equal(keep.cpu().numpy(), ref_cpu.nms(b, s, 0.5))


def test_nms_edge_cases(dev):
    from objectdetection.pytorch_amd import ops_detect
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 5], [5, 5, 5, 5], [5, 5, 5, 5]], dtype=np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.6], dtype=np.float32)
    keep = ops_detect.nms(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), 0.5)
    assert keep.cpu().tolist() == ref_cpu.nms(b, s, 0.5).tolist() == [0, 1, 2, 3]
    e = ops_detect.nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev), 0.5)
    assert e.numel() == 0


def test_class_max_and_filter(dev):
    from objectdetection.pytorch_amd import ops_detect
    rng = np.random.default_rng(3)
    cls = rng.random((5000, 80)).astype(np.float32) * 0.06
    cls[17, 5] = cls[17, 9] = 0.9                            # tie -> first index
    sc, lab = ops_detect.class_max(torch.from_numpy(cls).to(dev))
    rs, rl = torch.from_numpy(cls).max(dim=1)
    assert np.array_equal(sc.cpu().numpy(), rs.numpy())
    assert np.array_equal(lab.cpu().numpy(), rl.numpy().astype(np.int32))
    idx = ops_detect.score_filter(sc, 0.05)
    assert np.array_equal(idx.cpu().numpy(), np.nonzero(rs.numpy() > 0.05)[0].astype(np.int32))


def test_batched_postprocess_golden(golden_dir, dev):
    """utils/utils.py:76-114 (`postprocess`, batched decode + per-image NMS) against the reference's output
    on the same head tensors, including an image without detections and the all-empty double-append quirk."""
    from objectdetection.pytorch_amd.architectures.utils import BBoxTransform, ClipBoxes
    from objectdetection.pytorch_amd.utils import utils as uu
    g = np.load(os.path.join(golden_dir, 'postprocess.npz'))
    H, W = (int(v) for v in g['hw'])
    anc = torch.from_numpy(ref_cpu.anchors_for_image(H, W)).to(dev)
    reg, cls = torch.from_numpy(g['reg']).to(dev), torch.from_numpy(g['cls']).to(dev)
    x = torch.zeros(reg.shape[0], 3, H, W, device=dev)
    out = uu.postprocess(x, anc, reg, cls, BBoxTransform(), ClipBoxes(), 0.05, 0.5)
    assert len(out) == int(g['n_out'])
    for i, o in enumerate(out):
        assert o['scores'].shape == g['scores%d' % i].shape
        np.testing.assert_array_equal(o['class_ids'], g['class_ids%d' % i])
        np.testing.assert_array_equal(o['scores'], g['scores%d' % i])
        if o['rois'].size:
            np.testing.assert_allclose(o['rois'], g['rois%d' % i], rtol=1e-5, atol=1e-4)
    out0 = uu.postprocess(x, anc, reg, cls * 0.01, BBoxTransform(), ClipBoxes(), 0.05, 0.5)
    assert len(out0) == int(g['n_out_all_empty']) and all(o['scores'].size == 0 for o in out0)
