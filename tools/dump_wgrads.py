"""Weight-gradient launches of a plan: problems, K, C, taps, rows per chunk, blocks.  Usage: python tools/dump_wgrads.py [depth size batch]"""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, synth
depth, S, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50, 512, 16)
dev = torch.device('cuda', 0)
model, sd = bench.build_model(depth, 80, dev)
img = torch.from_numpy(synth.synth_images(B, S, S, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, 80, seed=2)).to(dev)
cl, rl = model(img, ann); (cl + rl).sum().backward(); torch.cuda.synchronize()
plan = model._plans[next(iter(model._plans))]
for fn, args, name in plan.bwd_steps:
    if 'wgrad' not in name:
        continue
    L = args[0]._obj
    split = [a for a in args if isinstance(a, ctypes.c_int)][-1].value
    rows = [L.prob[i].N * L.prob[i].P * L.prob[i].Q for i in range(L.nprob)]
    p = L.prob[0]
    T = p.TR * p.TS
    row3 = T == 9 and p.C % 64 == 0 and p.sh == 1 and p.W <= 64 and (p.W & (p.W - 1)) == 0
    ktiles, ctiles = -(-L.K // 128), (p.C // 64 if row3 else -(-p.C // 128))
    taps = 3 if row3 else T
    chunks = sum(-(-r // split) for r in rows)
    print('%-24s np%d K%-4d C%-4d %dx%d s%d %3dx%-3d rows %s split %5d chunks %4d tiles/chunk %3d blocks %5d' %
          (name.replace('mi355_conv_', ''), L.nprob, L.K, p.C, p.TR, p.TS, p.sh, p.P, p.Q, rows[0], split, chunks,
           ktiles * ctiles * taps, chunks * ktiles * ctiles * taps))
