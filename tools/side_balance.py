"""How far apart do the two compute streams finish?  Events at every join_side(): time since the previous join
at which the main stream and the side stream were done.  python tools/side_balance.py [efficientdet-d0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench, synth
from objectdetection.pytorch_amd.trainer import Trainer
dev = torch.device('cuda', 0)
arch = sys.argv[1] if len(sys.argv) > 1 else 'retinanet'
model, sd = bench.build_model(50, 80, dev, arch)
img = torch.from_numpy(synth.synth_images(16, 512, 512, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(16, 512, 512, 8, 16, 80, seed=2)).to(dev)
tr = Trainer(model, lr=1e-4, weight_decay=5e-4, clip_norm=0.1, max_annots=16)
for _ in range(5): tr.step(img, ann)
plan = model._plans[next(iter(model._plans))]
orig = plan.join_side
log = []
state = {'t0': None}
def join():
    if getattr(plan, '_side_dirty', False):
        em, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        em.record(torch.cuda.current_stream(dev)); es.record(plan._side)
        log.append((state['t0'], em, es))
    orig()
    t = torch.cuda.Event(enable_timing=True); t.record(torch.cuda.current_stream(dev)); state['t0'] = t
plan.join_side = join
for _ in range(6): tr.step(img, ann)
torch.cuda.synchronize()
per = {}
for i, (t0, em, es) in enumerate(log):
    if t0 is None: continue
    per.setdefault(i % (len(log) // 6), []).append((t0.elapsed_time(em), t0.elapsed_time(es)))
for k, v in sorted(per.items()):
    v = v[1:]
    print('join %d of the step: main done %.2f ms, side done %.2f ms after the previous join' % (k, sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v)))
