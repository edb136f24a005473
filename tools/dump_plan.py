"""Launch list of a plan, in order (name per launch) + per-name counts.  Usage: python tools/dump_plan.py arch size batch"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, synth

arch, S, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ('efficientdet-d0', 512, 16)
dev = torch.device('cuda', 0)
model, sd = bench.build_model(50, 80, dev, arch)
img = torch.from_numpy(synth.synth_images(B, S, S, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, 80, seed=2)).to(dev)
cl, rl = model(img, ann); (cl.mean() + rl.mean()).backward(); torch.cuda.synchronize()
plan = model._plans[next(iter(model._plans))]
for phase in ('prep_steps', 'fwd_steps', 'bwd_steps'):
    steps = getattr(plan, phase, [])
    cnt = collections.Counter(s[2] for s in steps)
    print('==', phase, len(steps))
    for k, v in cnt.most_common():
        print('   %4d  %s' % (v, k))
for phase in ('fwd_steps', 'bwd_steps'):
    print('==', phase, 'in order')
    print(' '.join(s[2].replace('mi355_', '') for s in getattr(plan, phase)))
