"""Slab workspace read by the parameter-gradient tables of a plan and the time of those launches alone.
Usage: python tools/slab_bytes.py [depth size batch [arch]]"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, synth

depth, S, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50, 512, 16)
arch = sys.argv[4] if len(sys.argv) > 4 else 'retinanet'        # e.g. efficientdet-d0, ksevendet
dev = torch.device('cuda', 0)
model, sd = bench.build_model(depth, 80, dev, arch)
img = torch.from_numpy(synth.synth_images(B, S, S, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, 80, seed=2)).to(dev)
cl, rl = model(img, ann); (cl + rl).sum().backward(); torch.cuda.synchronize()
plan = model._plans[next(iter(model._plans))]
tot = par = 0
rows = []
for s in plan.pgrad_specs:
    slab = s.Kpad * s.T * s.Cpad
    n = max(s.nslabs, 1)
    tot += n * slab * 4; par += slab * 4
    rows.append((n * slab * 4, n, s.Kpad, s.T, s.Cpad))
rows.sort(reverse=True)
for r in rows[:25]:
    print('%8.1f MB  nslabs %4d  K %4d T %2d C %4d' % (r[0] / 1e6, r[1], r[2], r[3], r[4]))
print('slab bytes %.1f MB over %d layers; one slab each would be %.1f MB' % (tot / 1e6, len(rows), par / 1e6))
st = torch.cuda.current_stream().cuda_stream
segs = plan.backward_segments(32 << 20) if hasattr(plan, 'blocks') else [(0, 0, plan.grad_numel_padded)]
for seg in segs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan._run_param_grads(0, st, (seg[1], seg[2])); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        plan._run_param_grads(0, st, (seg[1], seg[2]))
    e1.record(); torch.cuda.synchronize()
    nb = sum(n for n, *_ in [(max(s.nslabs, 1) * s.Kpad * s.T * s.Cpad * 4,) for s in plan.pgrad_specs
                            if seg[1] <= plan.grad_offsets[id(s.conv.weight)] < seg[2]])
    print('segment [%d, %d): %.1f us, %.1f MB of slabs -> %.2f TB/s' % (seg[1], seg[2], e0.elapsed_time(e1) * 100, nb / 1e6,
                                                                     nb / (e0.elapsed_time(e1) * 1e-4) / 1e12))
