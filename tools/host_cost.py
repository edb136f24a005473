"""Host cost of enqueueing one training step onto an EMPTY queue (no back-pressure) vs the step's GPU time.
Usage: python tools/host_cost.py [arch size batch]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, synth
from objectdetection.pytorch_amd.trainer import Trainer

arch, S, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ('efficientdet-d0', 512, 16)
dev = torch.device('cuda', 0)
model, sd = bench.build_model(50, 80, dev, arch)
img = torch.from_numpy(synth.synth_images(B, S, S, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, 80, seed=2)).to(dev)
tr = Trainer(model, lr=1e-4, weight_decay=5e-4, clip_norm=0.1, max_annots=ann.shape[1])
for _ in range(5):
    tr.step(img, ann)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(img, ann)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(1e3 * (t1 - t0)); total.append(1e3 * (t2 - t0))
host.sort(); total.sort()
plan = model._plans[next(iter(model._plans))]
n = len(plan.fwd_steps) + len(plan.bwd_steps) + len(getattr(plan, 'prep_steps', []))
print('%s %d bs%d: %d plan launches; host enqueue of one step on an empty queue %.2f ms (median), that step end to end %.2f ms'
      % (arch, S, B, n, host[5], total[5]))
