"""GPU: no launch of a plan stores outside the buffer it was handed.

MI355_GUARD mode (engine.RetinaPlan._new_guarded): every plan buffer — activations, gradients, slabs, tables, both
gradient arenas — is carved out of an allocation with a 4 KiB pattern zone on either side; after two full training
steps (forward, fused loss, backward, parameter-gradient tables, clip + AdamW) every zone must still hold the pattern.
An out-of-bounds store of a hand-written kernel otherwise lands in a neighbouring block of torch's caching allocator
(a wrong number somewhere else) or faults only when the buffer ends a segment — the shape of the unexplained abort
round 2 recorded in the fp32 trainer test.  Odd sizes on purpose: widths that do not divide the kernels' tiles."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _load(model, dev, **kw):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)
    model.train()
    model.freeze_bn()
    return model


def _retina(depth, precision, dev):
    from objectdetection.pytorch_amd.architectures import retinanet
    m = retinanet.retinanet(depth, num_classes=80)
    m.precision = precision
    return _load(m, dev, gain=0.87)


def _effdet(coef, precision, dev):
    from objectdetection.pytorch_amd.architectures import efficientdet
    m = efficientdet.build_efficientdet('efficientdet-d%d' % coef, num_classes=80)
    m.precision = precision
    return _load(m, dev, gain=1.0, head_std=0.01)


def _kseven(neck, precision, dev):
    import test_gpu_kseven as tk
    from objectdetection.pytorch_amd.ksevendet.architecture import ksevendet
    m = ksevendet.KSevenDet(tk.kseven_cfg(neck), num_classes=80, pretrained=False)
    m.precision = precision
    return _load(m, dev, gain=0.87)


CASES = [
    ('r18-fp32-256', lambda d: _retina(18, 'fp32', d), 2, 256, 256),        # the test that once aborted
    ('r18-bf16-96x160', lambda d: _retina(18, 'bf16', d), 3, 96, 160),
    ('r50-bf16-160x224', lambda d: _retina(50, 'bf16', d), 2, 160, 224),
    ('r50-fp32-128', lambda d: _retina(50, 'fp32', d), 1, 128, 128),
    ('d0-bf16-128x256', lambda d: _effdet(0, 'bf16', d), 3, 128, 256),
    ('d0-fp32-128', lambda d: _effdet(0, 'fp32', d), 2, 128, 128),
    ('kseven-fpn-bf16', lambda d: _kseven('fpn', 'bf16', d), 2, 128, 160),
    ('kseven-bifpn-bf16', lambda d: _kseven('bifpn', 'bf16', d), 2, 128, 256),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_two_training_steps_leave_every_guard_zone_intact(monkeypatch, case):
    from objectdetection.pytorch_amd import engine
    from objectdetection.pytorch_amd.trainer import Trainer
    monkeypatch.setattr(engine, 'GUARD_BYTES', 4096)
    name, make, B, H, W = case
    dev = torch.device('cuda', 0)
    model = make(dev)
    img = torch.from_numpy(synth.synth_images(B, H, W, seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(B, H, W, 5, 8, 80, seed=2)).to(dev)
    tr = Trainer(model)
    for _ in range(2):
        losses = tr.step(img, ann)
    torch.cuda.synchronize()
    assert torch.isfinite(losses).all() and tr.step_count == 2
    plans = list(model._plans.values())
    assert plans and all(len(p._guards) > 20 for p in plans)
    bad = [b for p in plans for b in p.check_guards()]
    assert not bad, 'stores outside plan buffers (allocation site, shape, dtype, side, byte offset, bytes): %r' % (bad[:8],)
    # inference path of the same plan (decode tail on image 0)
    model.eval()
    with torch.no_grad():
        model(img[:1], return_loss=False)
    torch.cuda.synchronize()
    bad = [b for p in model._plans.values() for b in p.check_guards()]
    assert not bad, 'inference: %r' % (bad[:8],)


def test_guard_check_sees_a_store_past_the_end(monkeypatch):
    """The checker itself: one float written one element past a guarded buffer is reported at byte offset 0."""
    from objectdetection.pytorch_amd import engine
    monkeypatch.setattr(engine, 'GUARD_BYTES', 4096)
    dev = torch.device('cuda', 0)
    model = _retina(18, 'bf16', dev)
    img = torch.from_numpy(synth.synth_images(1, 64, 64, seed=1)).to(dev)
    plan = model._plan(img)
    assert plan.check_guards() == []
    raw, nbytes = plan._guards[5][0], plan._guards[5][1]
    raw[4096 + nbytes:4096 + nbytes + 4] = 0
    bad = plan.check_guards()
    assert len(bad) == 1 and bad[0][3] == 'after' and bad[0][4] == 0 and bad[0][5] == 4
