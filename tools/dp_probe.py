"""Rank 0 trains on the GPU, rank 1 is a CPU-only peer that joins the same collectives: tells whether the
two-stream + overlapped-exchange slowdown seen with two ranks on ONE GPU is a GPU time-slicing artifact."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
rank = int(os.environ['RANK'])
dist.init_process_group('gloo')
two_stream = os.environ.get('PROBE_TWO_STREAM', '1') == '1'
steps = 8
if rank == 0:
    import bench, synth
    from objectdetection.pytorch_amd.trainer import Trainer
    dev = torch.device('cuda', 0)
    model, sd = bench.build_model(50, 80, dev)
    img = torch.from_numpy(synth.synth_images(16, 512, 512, seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(16, 512, 512, 8, 16, 80, seed=2)).to(dev)
    tr = Trainer(model, lr=1e-4, weight_decay=5e-4, clip_norm=0.1, max_annots=16)
    plan = model._plan(img)
    segs = plan.backward_segments(tr.segment_bytes)
    dist.broadcast_object_list([[(s[1], s[2]) for s in segs]], src=0)
    if two_stream:
        orig = tr._begin
        def begin(i, a):
            p, l = orig(i, a); p.use_side = True; return p, l
        tr._begin = begin
    for _ in range(3): tr.step(img, ann)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): tr.step(img, ann)
    torch.cuda.synchronize()
    print('two_stream=%s: %.2f ms/step' % (two_stream, (time.perf_counter() - t0) / steps * 1e3), flush=True)
else:
    obj = [None]
    dist.broadcast_object_list(obj, src=0)
    bufs = [torch.zeros(hi - lo) for lo, hi in obj[0]]
    loss = torch.zeros(2)
    for _ in range(3 + steps):
        works = [dist.all_reduce(b, async_op=True) for b in bufs]
        for w in works: w.wait()
        dist.all_reduce(loss)
dist.destroy_process_group()
