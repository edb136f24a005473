#!/usr/bin/env python3
"""the TRUE reference (imported from /root/reference) timed on CPU fp32 in the
build container.  Baseline only, never a target.

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 python /root/repo/tools/time_reference_cpu.py

(a) RetinaNet-R18 256x256 bs 2: forward + FocalLoss + backward (BASELINE.json configs[0]);
(b) FocalLoss.forward / backward alone at configs[1] sizes (B=16, A=49 104, C=80, M=8 real + 8 pad rows);
(c) Anchors.forward alone at 512x512.
Warm-up 3, time 10, median; `torch.get_num_threads()` and `os.cpu_count()` printed.  torchvision is absent: the
stand-in `nms` is never reached on these paths."""
import os
import statistics
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import synth                                             # noqa: E402

_tv, _ops = types.ModuleType('torchvision'), types.ModuleType('torchvision.ops')
_ops.nms = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('nms stand-in reached'))
_tv.ops = _ops
sys.modules['torchvision'], sys.modules['torchvision.ops'] = _tv, _ops
from architectures import anchors as ref_anchors         # noqa: E402
from architectures import losses as ref_losses           # noqa: E402
from architectures import retinanet as ref_retinanet     # noqa: E402


def med(fn, warm=3, n=10):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts), max(ts)


def main():
    print('cores: os.cpu_count() = %d, torch.get_num_threads() = %d, torch %s' % (os.cpu_count(), torch.get_num_threads(), torch.__version__))
    torch.manual_seed(0)
    model = ref_retinanet.retinanet18(80)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=0, gain=0.87).items()})
    model.train()
    model.freeze_bn()
    img = torch.from_numpy(synth.synth_images(2, 256, 256, seed=1))
    ann = torch.from_numpy(synth.synth_annotations(2, 256, 256, 8, 16, 80, seed=2))

    def step():
        model.zero_grad()
        cl, rl = model(img, ann)
        (cl.mean() + rl.mean()).backward()
    m, lo, hi = med(step)
    print('(a) RetinaNet-R18 256x256 bs2 fwd+FocalLoss+bwd: median %.3f s (min %.3f, max %.3f) = %.2f img/s' % (m, lo, hi, 2 / m))

    B, A, C = 16, 49104, 80
    cls, reg = synth.synth_head_outputs(B, A, C, seed=3)
    anchors = ref_anchors.Anchors()(torch.zeros(1, 3, 512, 512))
    ann16 = torch.from_numpy(synth.synth_annotations(B, 512, 512, 8, 16, C, seed=2))
    fl = ref_losses.FocalLoss()
    cls_t = torch.from_numpy(cls).requires_grad_(True)
    reg_t = torch.from_numpy(reg).requires_grad_(True)
    m, lo, hi = med(lambda: fl(cls_t, reg_t, anchors, ann16))
    print('(b) FocalLoss.forward B=16 A=49104 C=80: median %.3f s (min %.3f, max %.3f)' % (m, lo, hi))

    def fb():
        cls_t.grad = reg_t.grad = None
        cl, rl = fl(cls_t, reg_t, anchors, ann16)
        (cl.mean() + rl.mean()).backward()
    m2, lo, hi = med(fb)
    print('    FocalLoss forward+backward: median %.3f s (min %.3f, max %.3f) -> backward ~%.3f s' % (m2, lo, hi, m2 - m))
    anc = ref_anchors.Anchors()
    x = torch.zeros(1, 3, 512, 512)
    m, lo, hi = med(lambda: anc(x), warm=3, n=20)
    print('(c) Anchors.forward 512x512 (host numpy, every forward): median %.1f ms (min %.1f, max %.1f)' % (1e3 * m, 1e3 * lo, 1e3 * hi))


if __name__ == '__main__':
    main()
