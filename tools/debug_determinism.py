"""Which parameter gradients differ between two identical backward passes?  Usage: python tools/debug_determinism.py [arch size batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, synth
arch, S, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ('efficientdet-d0', 512, 16)
dev = torch.device('cuda', 0)
model = bench.build_model(50, 80, dev, arch)[0]
if arch != 'retinanet':
    synth.disable_drop_connect(model)
img = torch.from_numpy(synth.synth_images(B, S, S, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, 80, seed=2)).to(dev)

def run():
    model.zero_grad()
    cl, rl = model(img, ann)
    (cl + rl).sum().backward()
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
a, b = run(), run()
bad = [(k, float((a[k] - b[k]).abs().max()), float(a[k].abs().max())) for k in a if not torch.equal(a[k], b[k])]
print('%d of %d gradient tensors differ' % (len(bad), len(a)))
for k, d, m in bad[:40]:
    print('  %-70s max diff %.3g of %.3g' % (k, d, m))
