import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_cpu, ref_effdet
import synth
from objectdetection.pytorch_amd.architectures import efficientdet
coef, H, W, B, nc = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (0, 128, 128, 2, 8)
dev = torch.device('cuda', 0)
model = efficientdet.build_efficientdet('efficientdet-d%d' % coef, num_classes=nc)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = synth.synth_state_dict(shapes, seed=0, gain=1.0, head_std=0.05)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev); model.train(); model.freeze_bn()
img = torch.from_numpy(synth.synth_images(B, H, W, seed=1))
ann = torch.from_numpy(synth.synth_annotations(B, H, W, 6, 8, nc, seed=2))
cl, rl = model(img.to(dev), ann.to(dev)); (cl + rl).sum().backward()
p = {k: torch.from_numpy(v).requires_grad_(v.dtype == np.float32 and 'running' not in k) for k, v in sd.items()}
clo, rlo = ref_effdet.efficientdet_loss(p, img, ann, coef, nc); (clo + rlo).sum().backward()
print('loss', cl.item(), rl.item(), clo.item(), rlo.item())
bad = 0
for k, q in model.named_parameters():
    a = q.grad.detach().cpu().double().reshape(-1); b = p[k].grad.double().reshape(-1)
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30)); ratio = float(a.norm() / (b.norm() + 1e-30))
    if cos < 0.97 or not (0.85 < ratio < 1.18):
        bad += 1
        print('%-60s cos %.4f ratio %.4f |ref| %.3e' % (k, cos, ratio, float(b.norm())))
print('bad', bad, 'of', len(list(model.parameters())))
