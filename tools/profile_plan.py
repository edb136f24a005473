"""Per-launch timing of the conv steps of a RetinaPlan (HIP events, each launch replayed alone).
Usage: python tools/profile_plan.py [depth size batch]"""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from objectdetection.pytorch_amd import ops_conv

depth, S, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50, 512, 16)
dev = torch.device('cuda', 0)
model, sd = bench.build_model(depth, 80, dev)
import synth
img = torch.from_numpy(synth.synth_images(B, S, S, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, 80, seed=2)).to(dev)
cl, rl = model(img, ann); (cl + rl).sum().backward(); torch.cuda.synchronize()
plan = model._plans[next(iter(model._plans))]
st = torch.cuda.current_stream()
rows = []
for phase, steps in (('fwd', plan.fwd_steps), ('bwd', plan.bwd_steps)):
    for fn, args, name in steps:
        if name not in ('mi355_conv_gather', 'mi355_conv_gather_splitk', 'mi355_conv_wgrad_slabs', 'mi355_conv_wgrad_colsum', 'mi355_conv_wgrad_multi'):
            continue
        L = ctypes.cast(args[0], ctypes.POINTER(ops_conv.ConvLaunch)).contents if False else args[0]._obj
        flops = 0.0
        nbytes = 2.0 * L.K * L.T * L.prob[0].C * (1 if name.startswith('mi355_conv_gather') else 2)
        desc = []
        for i in range(L.nprob):
            p = L.prob[i]
            M = p.N * p.P * p.Q
            if name.startswith('mi355_conv_gather'):
                f = 2.0 * M * L.K * p.C * p.TR * p.TS
            else:
                f = 2.0 * M * L.K * p.C * p.TR * p.TS
            flops += f
            nbytes += 2.0 * p.N * p.H * p.W * p.C          # input / x
            if name.startswith('mi355_conv_gather'):
                nbytes += M * L.K * (4.0 if L.out_f32 else 2.0)
                if p.addend: nbytes += 2.0 * M * L.K / (4 ** p.add_shift)
                if p.mask: nbytes += 2.0 * M * L.K
            else:
                nbytes += 2.0 * M * L.K
            desc.append('%dx%dx%d' % (p.N, p.P, p.Q))
        p0 = L.prob[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s = ctypes.c_void_p(st.cuda_stream)
        fn(*args, s); torch.cuda.synchronize()
        e0.record(st)
        for _ in range(5):
            fn(*args, s)
        e1.record(st); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        rows.append((phase, 'gath_sk' if name.endswith('splitk') else name[11:17], L.nprob, p0.C, L.K, p0.TR, p0.TS, p0.sh, '+'.join(desc[:2]), flops, us, max(nbytes / 5.0e6, flops / 1.2e9)))
tot_f = sum(r[9] for r in rows); tot_t = sum(r[10] for r in rows)
print('total %.1f GFLOP (padded channel counts) in %.2f ms = %.1f TFLOP/s' % (tot_f / 1e9, tot_t / 1e3, tot_f / tot_t / 1e6))
agg = {}
for r in rows:
    key = r[:9]
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += r[9]; a[2] += r[10]; a[3] += r[11]
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print('%-3s %-7s np%d C%-4d K%-4d %dx%d s%d %-26s n=%-2d %8.1f us/launch %7.1f TF/s  %5.1f%% of time  floor %6.1f us  lost %7.1f us'
          % (key + (a[0], a[2] / a[0], a[1] / a[2] / 1e6, 100 * a[2] / tot_t, a[3] / a[0], a[2] - a[3])))
print('sum of floors (max(bytes / 5 TB/s, flops / 1.2 PF/s)): %.2f ms' % (sum(r[11] for r in rows) / 1e3))

# ---- what a launch boundary costs: the same conv launches in PLAN ORDER on one stream (every launch reads what its
# predecessor just wrote, caches as the step leaves them) against the sum of the alone times above (each launch replayed
# five times over the same operands, i.e. warm).  The difference per launch is what fusing two dependent launches into
# one can at most win on top of the bytes it stops moving.
CONV = ('mi355_conv_gather', 'mi355_conv_gather_splitk', 'mi355_conv_wgrad_slabs', 'mi355_conv_wgrad_colsum', 'mi355_conv_wgrad_multi')
s = ctypes.c_void_p(st.cuda_stream)
for phase, steps in (('fwd', plan.fwd_steps), ('bwd', plan.bwd_steps)):
    chain = [(fn, args) for fn, args, name in steps if name in CONV]
    alone = sum(r[10] for r in rows if r[0] == phase)
    for fn, args in chain:
        fn(*args, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        for fn, args in chain:
            fn(*args, s)
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print('%s: %d conv launches in plan order on one stream %.2f ms; sum of their alone times %.2f ms; %.2f us per launch boundary'
          % (phase, len(chain), us / 1e3, alone / 1e3, (us - alone) / max(len(chain), 1)))
