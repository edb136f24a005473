"""Stage-by-stage comparison of the EfficientDet HIP plan with the fp32 oracle (debug aid)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_effdet
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
cls, reg = model(img.to(dev), return_head=True)
plan = model._plans[next(iter(model._plans))]
ref_effdet.TRACE = {}
p = {k: torch.from_numpy(v) for k, v in sd.items()}
with torch.no_grad():
    cls_o, reg_o = ref_effdet.efficientdet_heads(p, img, coef, nc)
for name, ref in ref_effdet.TRACE.items():
    if name not in plan.dbg: continue
    got = plan.dbg[name].float().cpu().permute(0, 3, 1, 2)
    if got.shape != ref.shape: print(name, 'SHAPE', got.shape, ref.shape); continue
    err = (got - ref).abs().max().item(); mag = ref.abs().max().item()
    print('%-50s err %.4g / mag %.4g  %s' % (name, err, mag, '<<<' if err > 0.03 * mag else ''))
print('cls', (cls.cpu() - cls_o).abs().max().item(), cls_o.max().item(), 'reg', (reg.cpu() - reg_o).abs().max().item(), reg_o.abs().max().item())
