"""Per-parameter gradient comparison: HIP plan (bf16) vs fp32 oracle (CPU). Debug aid."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_cpu
import synth
from objectdetection.pytorch_amd.architectures import retinanet

depth, H, W, B, nc, gain = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
dev = torch.device('cuda', 0)
model = retinanet.retinanet(depth, num_classes=nc)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = synth.synth_state_dict(shapes, seed=0, gain=gain)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev); model.train(); model.freeze_bn()
img = torch.from_numpy(synth.synth_images(B, H, W, seed=1))
ann = torch.from_numpy(synth.synth_annotations(B, H, W, 8, 16, nc, seed=2))
cl, rl = model(img.to(dev), ann.to(dev))
(cl + rl).sum().backward()
def run(emu):
    p = {k: torch.from_numpy(v).requires_grad_(v.dtype == np.float32 and 'running' not in k) for k, v in sd.items()}
    clo, rlo = ref_cpu.retinanet_loss(p, img, ann, depth, emulate_bf16=emu)
    (clo + rlo).sum().backward()
    print('emulate_bf16', emu, 'loss hip', cl.item(), rl.item(), 'oracle', clo.item(), rlo.item())
    return p
pf, pe = run(False), run(True)
for k, q in model.named_parameters():
    a = q.grad.detach().cpu().double().reshape(-1)
    out = []
    for p in (pf, pe):
        b = p[k].grad.double().reshape(-1)
        out.append((float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))))
    flag = ' <<<' if out[1][0] < 0.999 else ''
    print('%-45s fp32: cos %.5f relL2 %.4f | bf16-emu: cos %.6f relL2 %.4f%s' % (k, out[0][0], out[0][1], out[1][0], out[1][1], flag))
