#!/usr/bin/env python3
"""How far apart are two CORRECT fp32 evaluations of the reference's arithmetic?  (CPU, oracle only: a tool
for setting honest parity bounds, not part of the product.)

Runs the oracle (oracle/ref_cpu.py, pinned to the reference's goldens) on the e2e_r18_256 inputs three ways:
fp32 through oneDNN convs, fp32 through torch's native conv kernels, and fp64; prints per-tensor relative L2
distances of the parameter gradients.  Measured in the build container (torch 2.10 CPU):

    conv1.weight                 oneDNN-vs-native 3.1e-3 | oneDNN vs fp64 3.1e-3 | native vs fp64 1.3e-4
    layer4.*                     ~1.1e-3                 | ~1.1e-3               | ~4e-5
    regressionModel.conv1-3.w    ~5.6e-4                 | ~8e-7                 | ~5.6e-4
    classificationModel.conv1-3  ~1e-4                   | ~1e-4                 | ~6e-7

i.e. single ReLU-mask / smooth-L1-kink decisions flip with the accumulation order, and ill-conditioned sums
(conv1.weight: 32 768 random-sign terms per element) amplify rounding: 1e-3 relative L2 on every gradient tensor
is not attainable between ANY two fp32 implementations; the fp32-mode test therefore measures against the
reference's fp64 value with the reference's own fp32 distance from it as the yardstick
(tests/test_gpu_fp32_mode.py, tests/golden `grad64/*`)."""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import ref_cpu
import synth
g = np.load('/root/repo/tests/golden/e2e_r18_256.npz')
depth, (H, W), B, nc = int(g['depth']), g['hw'], int(g['batch']), int(g['num_classes'])
shapes = ref_cpu.retinanet_state_shapes(depth, nc)
sd = synth.synth_state_dict(shapes, seed=0, gain=float(g['gain']))
img = torch.from_numpy(synth.synth_images(B, int(H), int(W), seed=1))
ann = torch.from_numpy(synth.synth_annotations(B, int(H), int(W), 8, 16, nc, seed=2))
def run(dt, mkldnn=True, cl=False):
    p = {k: (torch.from_numpy(v).to(dt).requires_grad_('running' not in k) if v.dtype==np.float32 else torch.from_numpy(v)) for k, v in sd.items()}
    x = img.to(dt)
    if cl: x = x.contiguous(memory_format=torch.channels_last)
    with torch.backends.mkldnn.flags(enabled=mkldnn):
        c, r = ref_cpu.retinanet_loss(p, x, ann.to(dt), depth)
        (c.mean()+r.mean()).backward()
    return {k: v.grad.double().numpy().copy() for k, v in p.items() if getattr(v,'grad',None) is not None}
a = run(torch.float32, True)
b = run(torch.float32, False)
c = run(torch.float64, True)
rows = []
for k in a:
    n = np.linalg.norm(c[k])+1e-300
    rows.append((np.linalg.norm(a[k]-b[k])/n, np.linalg.norm(a[k]-c[k])/n, np.linalg.norm(b[k]-c[k])/n, k))
rows.sort(reverse=True)
print('relL2: mkldnn-vs-native fp32 | mkldnn fp32 vs fp64 | native fp32 vs fp64')
for r in rows[:8]: print('%.2e %.2e %.2e %s' % r)
for r in rows:
    if r[3].startswith(('regressionModel','classificationModel')) and 'weight' in r[3]: print('%.2e %.2e %.2e %s' % r)
