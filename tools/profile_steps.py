"""Per-launch time of EVERY step of a launch plan (HIP events, each launch replayed alone, back to back), grouped by
(phase, entry point, operand bytes): where a step's GPU time goes when nothing overlaps.
Usage: python tools/profile_steps.py arch size batch [iters]"""
import os, sys, ctypes, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, synth

arch, S, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ('efficientdet-d0', 512, 16)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device('cuda', 0)
model, sd = bench.build_model(50, 80, dev, arch)
img = torch.from_numpy(synth.synth_images(B, S, S, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, 80, seed=2)).to(dev)
cl, rl = model(img, ann); (cl.mean() + rl.mean()).backward(); torch.cuda.synchronize()
plan = model._plans[next(iter(model._plans))]
sizes = {t.data_ptr(): t.numel() * t.element_size() for t in plan.bufs}
st = torch.cuda.current_stream()
s = ctypes.c_void_p(st.cuda_stream)
rows = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for phase in ('fwd_steps', 'bwd_steps'):
    for idx, (fn, args, name) in enumerate(getattr(plan, phase)):
        if name.startswith(('py:', '__', 'join:')):
            continue
        if name in bench.CONV_STEPS:
            L = args[0]._obj
            nbytes = bench.conv_launch_bytes(L, name)
            p = L.prob[0]
            desc = 'np%d C%d K%d %dx%d s%d %dx%d' % (L.nprob, p.C, L.K, p.TR, p.TS, p.sh, p.P, p.Q)
        else:
            seen, nbytes = set(), 0.0
            for a in args:
                v = getattr(a, 'value', None)
                if isinstance(a, ctypes.c_void_p) and v in sizes and v not in seen:
                    seen.add(v); nbytes += sizes[v]
                elif isinstance(a, ctypes.Array) and hasattr(a[0], 'aux'):
                    for lv in a:
                        for v in (lv.x, lv.dy, lv.aux):
                            if v in sizes and v not in seen:
                                seen.add(v); nbytes += sizes[v]
            desc = ' '.join(str(a.value) for a in args if isinstance(a, ctypes.c_int))[:40]
        fn(*args, s)
        e0.record(st)
        for _ in range(iters):
            fn(*args, s)
        e1.record(st)
        torch.cuda.synchronize()
        rows.append((phase[:3], name.replace('mi355_', ''), desc, e0.elapsed_time(e1) * 1e3 / iters, nbytes))
tot = sum(r[3] for r in rows)
print('%s %d bs%d: %d launches, %.2f ms when run one after another' % (arch, S, B, len(rows), tot / 1e3))
by = collections.OrderedDict()
for ph, name, desc, us, nb in rows:
    k = (ph, name, desc)
    a = by.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += us; a[2] += nb
print('-- by entry point')
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (ph, name, desc), (n, us, nb) in by.items():
    a = agg[(ph, name)]; a[0] += n; a[1] += us; a[2] += nb
for (ph, name), (n, us, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%s %-24s n=%4d %9.1f us  %5.1f %%  %8.1f MB  %6.0f GB/s  floor@5TB/s %7.1f us' %
          (ph, name, n, us, 100 * us / tot, nb / 1e6, nb / us / 1e3 if us else 0, nb / 5e6))
TOPN = int(os.environ.get('TOPN', '60'))
print('-- by launch shape (top %d)' % TOPN)
for (ph, name, desc), (n, us, nb) in sorted(by.items(), key=lambda kv: -kv[1][1])[:TOPN]:
    print('%s %-22s %-40s n=%3d %8.1f us/launch %6.0f GB/s  %5.1f %%' % (ph, name, desc, n, us / n, nb / us / 1e3 if us else 0, 100 * us / tot))
