import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from objectdetection.pytorch_amd.architectures import efficientdet
coef, H, W, B, nc = 2, 128, 256, 1, 5
dev = torch.device('cuda', 0)
# churn the allocator with garbage first
junk = [torch.full((1 << 22,), float('nan'), device=dev) for _ in range(64)]; del junk
model = efficientdet.build_efficientdet('efficientdet-d%d' % coef, num_classes=nc)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = synth.synth_state_dict(shapes, seed=0, gain=1.0, head_std=0.05)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev); model.eval()
img = torch.from_numpy(synth.synth_images(B, H, W, seed=1)).to(dev)
cls1, reg1 = model(img, return_head=True)
plan = model._plans[next(iter(model._plans))]
snap = {k: v.clone() for k, v in plan.dbg.items()}
cls2, reg2 = model(img, return_head=True)
print('cls diff', (cls1 - cls2).abs().max().item(), 'nan?', torch.isnan(cls1).any().item())
first = None
for k, v in plan.dbg.items():
    d = (v.float() - snap[k].float()).abs().max().item()
    if d > 0 and first is None:
        first = k
    if d > 0: print('%-45s diff %.4g' % (k, d))
print('first differing stage:', first)
