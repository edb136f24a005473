"""GPU time of the phases of one training step (HIP events on the launch stream, median of N steps):
forward (prep + launches), loss, backward (incl. parameter-gradient tables), optimizer.  Usage: python tools/phase_times.py [arch size batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, synth
from objectdetection.pytorch_amd.trainer import Trainer

arch, S, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ('retinanet', 512, 16)
dev = torch.device('cuda', 0)
model, sd = bench.build_model(50, 80, dev, arch)
img = torch.from_numpy(synth.synth_images(B, S, S, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, 80, seed=2)).to(dev)
tr = Trainer(model, lr=1e-4, weight_decay=5e-4, clip_norm=0.1, max_annots=ann.shape[1])
for _ in range(5):
    tr.step(img, ann)
torch.cuda.synchronize()
names = ['forward', 'loss', 'backward', 'optimizer']
acc = {n: [] for n in names}
for _ in range(20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    plan = model._plan(img)
    anchors = model.anchors(img)
    ev[0].record()
    plan.run_heads(img)
    ev[1].record()
    losses = plan.run_loss(anchors, ann)
    ev[2].record()
    plan.run_backward(tr.ones[0:1], tr.ones[1:2], want_views=False)
    ev[3].record()
    tr._update(plan, tr._exchange(plan, losses))
    ev[4].record()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        acc[n].append(ev[i].elapsed_time(ev[i + 1]))
tot = 0.0
for n in names:
    v = sorted(acc[n])[len(acc[n]) // 2]
    tot += v
    print('%-10s %.3f ms' % (n, v))
print('%-10s %.3f ms (phases are serialised by the events: the real step overlaps the tail of one with the head of the next)' % ('sum', tot))
