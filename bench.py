#!/usr/bin/env python3
"""Headline benchmark: images/sec, RetinaNet-R50 training step, 512x512, batch 16 per GPU
(BASELINE.json `metric`, configs[1]) on N MI355X of one node.

One step = train.py:292-321 of the reference: zero_grad -> forward + FocalLoss -> backward ->
(N > 1: RCCL all-reduce of the flat gradient arena) -> clip_grad_norm_(0.1) -> AdamW(lr 1e-4,
wd 5e-4).  Inputs are synthetic (tests/synth.py: N(0,1) images, 8 boxes/img padded with -1 rows
to 16) and already resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) extended with `roofline`
(dominant kernels = the MFMA convs, algorithmic FLOPs / HIP-event time of exactly those launches)
and `cpu_baseline` (the fp32 oracle port timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
# HIP spreads a process's streams over GPU_MAX_HW_QUEUES (default 4) hardware queues; once RCCL has made
# its own streams the plan's second compute stream lands on the first one's queue and the two serialise
# (12.8 instead of 10.9 ms per step, tools/nccl_rehearse.py).  Must be set before the HIP runtime starts.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402

import synth                # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (not the 2:1-sparse figure)
PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 (exact f32) = the fp32 vector rate
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s is what a float4 copy reaches)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--arch', default='retinanet',
                    help="'retinanet' (headline), 'efficientdet-dN' (configs[3], [4]) or 'ksevendet' / 'ksevendet-panet' / 'ksevendet-bifpn' (SURVEY 8 f-3)")
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-infer', action='store_true', help='skip the `infer` sub-object (batched heads, B=1 decode)')
    ap.add_argument('--graph', type=int, default=int(os.environ.get('MI355_GRAPH', '0')),
                    help='replay the step from captured hipGraphs (default off: the step is GPU-bound and the '
                         'two-stream schedule runs faster launched eagerly than as a branched graph)')
    ap.add_argument('--infer', action='store_true',
                    help='measure inference instead (SURVEY 8d): heads at --batch, and the B=1 decode + NMS path; '
                         'prints its own JSON line (not the headline metric)')
    ap.add_argument('--precision', default='bf16', choices=('bf16', 'fp32'),
                    help="'fp32': the plan's fp32-accurate mode (the mode held to the north-star 1e-4; roofline against the "
                         "157.3 TFLOP/s f32 matrix peak)")
    ap.add_argument('--sync-bn', action='store_true',
                    help='BASELINE configs[2] as named: utils.utils.replace_w_sync_bn + BatchNorm in train mode (batch '
                         'statistics, cross-replica all-reduce of [sum | sum^2] per layer and direction); on one GPU the '
                         'collectives run on a world-size-1 RCCL group')
    ap.add_argument('--torch-optim', action='store_true',
                    help='autograd + torch clip/AdamW instead of the flat HIP optimizer')
    return ap.parse_args()


def build_model(depth, num_classes, dev, arch='retinanet', precision='bf16', sync_bn=False):
    if arch not in ('retinanet', 'ksevendet', 'ksevendet-panet', 'ksevendet-bifpn') and not arch.startswith('efficientdet-d'):
        raise SystemExit("--arch must be 'retinanet', 'ksevendet[-panet|-bifpn]' or 'efficientdet-dN', got %r" % arch)
    if arch.startswith('ksevendet'):
        # config/model/ksevendet_resnet.yaml (ResNet trunk of --depth, FPN or PANet-FPN neck, simple heads
        # with per-level BN over shared convs), frozen BN as train_ksevendet.py runs it
        import yaml
        from objectdetection.pytorch_amd.ksevendet.architecture import ksevendet
        cfg = yaml.safe_load(open(os.path.join(ROOT, 'objectdetection', 'pytorch_amd', 'config', 'model',
                                               'ksevendet_resnet_bifpn.yaml' if arch.endswith('bifpn') else 'ksevendet_resnet.yaml')))
        cfg.pop('architecture')
        cfg['variant'] = 'resnet%d' % depth
        if not arch.endswith('bifpn'):            # (ksevendet_resnet_bifpn.yaml: BiFPN neck, unshared heads on levels 3..5)
            cfg['neck'] = 'panet-fpn' if arch.endswith('panet') else 'fpn'
        model = ksevendet.KSevenDet(cfg, num_classes=num_classes, pretrained=False)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = synth.synth_state_dict(shapes, seed=0, gain=0.78 if depth >= 50 else 0.87)
    elif arch.startswith('efficientdet'):
        from objectdetection.pytorch_amd.architectures import efficientdet
        model = efficientdet.build_efficientdet(arch, num_classes=num_classes)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = synth.synth_state_dict(shapes, seed=0, gain=1.0, head_std=0.05)
    else:
        from objectdetection.pytorch_amd.architectures import retinanet
        model = retinanet.retinanet(depth, num_classes=num_classes)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        # deterministic non-degenerate weights (random frozen-BN statistics, head outputs in a useful
        # range: see tests/synth.py) - random-init weights of the named architecture
        sd = synth.synth_state_dict(shapes, seed=0, gain=0.78 if depth >= 50 else 0.87)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)
    model.train()
    if sync_bn:
        # configs[2] "DP + SyncBN": utils/utils.py:140-168 swaps the BatchNorm2d layers, which then stay in train mode
        # (batch statistics, running-stat updates); no freeze_bn
        from objectdetection.pytorch_amd.utils.utils import replace_w_sync_bn
        replace_w_sync_bn(model)
        model.train()
    else:
        model.freeze_bn()                   # train.py:273-277
    if precision == 'fp32':
        model.precision = 'fp32'
    return model, sd


def clip_grad_norm_flat_(arena, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics (train.py:320) on the flat gradient arena: one
    norm kernel + one scale kernel, no host sync."""
    total = torch.linalg.vector_norm(arena)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    arena.mul_(coef)
    return total


def conv_flops(plan):
    """Algorithmic conv FLOPs of one training step (2 x MAC, true channel counts):
    forward + weight-gradient for every conv, data-gradient for all but the stem."""
    fwd = 0
    dgrad = 0
    B = plan.B

    def out_pixels(spec, shapes):
        return sum(n * p * q for (n, p, q) in shapes)

    # walk the forward launches recorded by the plan
    for L in plan.fwd_launches:
        spec, probs = L
        for (n, p, q) in probs:
            f = 2.0 * n * p * q * spec.K * getattr(spec, 'true_ct', spec.C * spec.T)   # (space-to-depth stem: the real 3 x 49 taps)
            fwd += f
            if spec.need_dgrad:
                dgrad += f
    return fwd, fwd + dgrad + fwd     # (forward, training = fwd + dgrad + wgrad)


CONV_STEPS = ('mi355_conv_gather', 'mi355_conv_gather_splitk', 'mi355_conv_wgrad_colsum', 'mi355_conv_wgrad_slabs',
              'mi355_conv_wgrad_multi', 'mi355_conv_gather_f32', 'mi355_conv_wgrad_slabs_f32')


def conv_bytes(plan):
    """Algorithmic HBM bytes of the conv launches of one step, unfused minimum per launch: every operand
    read once, every result written once (bf16 activations, bf16 weight shadows, fp32 dW slabs)."""
    total = 0.0
    for fn, args, name in plan.fwd_steps + plan.bwd_steps:
        if name in CONV_STEPS:
            total += conv_launch_bytes(args[0]._obj, name, plan.esz)
    return total


def conv_launch_bytes(L, name, esz=2):
    """One conv launch of `conv_bytes`: every operand read once, every result written once (`esz` = bytes per stored
    activation / gradient / weight-shadow element: 2 in the bf16 mode, 4 in the fp32-accurate mode)."""
    gather = name.startswith('mi355_conv_gather')
    # weight shadows / one fp32 dW tile set; a multi-layer wgrad launch writes one per problem (= per layer)
    total = float(L.K * L.T * L.prob[0].C) * (esz if gather else 4) * (L.nprob if name.endswith('_multi') else 1)
    for i in range(L.nprob):
        p = L.prob[i]
        M = p.N * p.P * p.Q
        total += float(esz) * p.N * p.H * p.W * p.C
        if gather:
            total += M * L.K * (4.0 if L.out_f32 else float(esz))
            total += (float(esz) * M * L.K / (4 ** p.add_shift)) if p.addend else 0.0
            total += float(esz) * M * L.K if p.mask else 0.0
        else:
            total += float(esz) * M * L.K
    return total


def plan_launch_bytes(plan):
    """Algorithmic HBM bytes of ONE training step of a launch plan, launch by launch: every buffer a launch
    names (activation, gradient, parameter shadow; exact plan-buffer pointers) counted once at its full size,
    conv launches by `conv_launch_bytes`.  This is the unfused-per-launch minimum SURVEY 8(d) asks for the HBM-bound
    EfficientDet configs: a launch that re-reads its inputs moves MORE than this, one that is served by the
    caches less.  Table-driven launches (weight prep, fusion weights) name only their table: their parameter
    traffic (a few MB) is left out."""
    import ctypes
    sizes = {}
    for t in plan.bufs:
        sizes[t.data_ptr()] = t.numel() * t.element_size()
    total, per_kernel = 0.0, {}
    for steps in (plan.prep_steps, plan.fwd_steps, plan.bwd_steps):
        for fn, args, name in steps:
            b = 0.0
            if name in CONV_STEPS:
                b = conv_launch_bytes(args[0]._obj, name, plan.esz)
            else:
                seen = set()
                for a in args:
                    v = getattr(a, 'value', None)
                    if isinstance(a, ctypes.c_void_p) and v in sizes and v not in seen:
                        seen.add(v)
                        b += sizes[v]
                    elif isinstance(a, ctypes.Array) and hasattr(a[0], 'aux'):     # Mi355DwLevel table of a multi-map launch
                        for lv in a:
                            for v in (lv.x, lv.dy, lv.aux):
                                if v in sizes and v not in seen:
                                    seen.add(v)
                                    b += sizes[v]
            total += b
            per_kernel[name] = per_kernel.get(name, 0.0) + b
    return total, per_kernel


from csrc_hash import csrc_hash   # noqa: E402  (tools/csrc_hash.py)


def traffic_tag(args):
    return '%s%s_%d_b%d_%s%s' % (args.arch.replace('-', '_'), ('_r%d' % args.depth) if not args.arch.startswith('efficientdet') else '',
                               args.size, args.batch, args.precision, '_syncbn' if args.sync_bn else '')


def measured_traffic(args):
    """HBM-side bytes per step from the COMMITTED PMC passes of this very workload (tools/collect_traffic.py;
    FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950): a kept profile of the same command, not
    a measurement of this run (`traffic_source: "committed"`).  The file records the hash of the kernel sources it
    was collected on; `stale` tells whether they have changed since.  -> (dict | None)"""
    tag = traffic_tag(args)
    for rnd in ('r03', 'r02', 'r01'):
        path = os.path.join(ROOT, 'profiles', '%s_pmc_traffic_%s.json' % (rnd, tag))
        if not os.path.exists(path) and tag == 'retinanet_r50_512_b16_bf16':
            path = os.path.join(ROOT, 'profiles', '%s_pmc_traffic.json' % rnd)      # (rounds 1-2: the headline only)
        if os.path.exists(path):
            t = json.load(open(path))
            return {'file': 'profiles/' + os.path.basename(path), 'conv': t.get('conv_kernels_bytes_per_step'),
                    'step': t.get('all_kernels_bytes_per_step'), 'csrc_sha16': t.get('csrc_sha16'),
                    'stale': t.get('csrc_sha16') != csrc_hash()}
    return None


def time_conv_kernels(plan, iters, names=None, prefix=None):
    """HIP-event time of ONLY the MFMA conv launches of one step (forward gather convs,
    backward gather convs = dgrad, wgrad), replayed back-to-back on the launch stream.  `prefix`: the launches whose
    entry point starts with it instead (the BatchNorm passes of the train-mode / synchronised-BN plan, their
    all-reduce host steps included)."""
    if prefix is not None:
        steps = [s for s in plan.fwd_steps + plan.bwd_steps if s[2].startswith(prefix)]
    else:
        steps = [s for s in plan.fwd_steps + plan.bwd_steps if s[2] in (names or CONV_STEPS)]
    if not steps:
        return 0.0, 0
    st = torch.cuda.current_stream()
    plan.grad_arenas[0].zero_()
    plan._run(steps, st.cuda_stream)
    plan.join_side()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        plan._run(steps, st.cuda_stream)
        plan.join_side()          # wgrad launches run on the plan's side stream: fence them into the timing
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, len(steps)


def cpu_baseline(depth, size, sd, num_classes, arch='retinanet'):
    """fp32 oracle (oracle/ref_cpu.py, oracle/ref_effdet.py = CPU ports of the reference path) forward + loss +
    backward on a bounded sample of the same workload."""
    from oracle import ref_cpu
    nthreads = torch.get_num_threads()
    B = 2
    img = torch.from_numpy(synth.synth_images(B, size, size, seed=1))
    ann = torch.from_numpy(synth.synth_annotations(B, size, size, 8, 16, num_classes, seed=2))
    p = {k: torch.from_numpy(v).clone().requires_grad_(v.dtype == np.float32 and 'running' not in k)
         for k, v in sd.items()}
    if arch != 'retinanet':
        from oracle import ref_effdet
        coef = int(arch[-1])

    def step():
        if arch == 'retinanet':
            cl, rl = ref_cpu.retinanet_loss(p, img, ann, depth)
        else:
            cl, rl = ref_effdet.efficientdet_loss(p, img, ann, coef, num_classes)
        (cl.mean() + rl.mean()).backward()
        for t in p.values():
            t.grad = None

    step()                                   # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 8):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {'value': round(B * n / dt, 3), 'unit': 'images/sec', 'cores': nthreads, 'kind': 'port',
            'sample': '%d steps of batch %d, %s %dx%d fp32 fwd+loss+bwd (no optimizer), torch CPU oracle'
                      % (n, B, 'RetinaNet-R%d' % depth if arch == 'retinanet' else 'EfficientDet-D' + arch[-1], size, size)}


def infer_bench(args, model, name, img, dev, emit=True):
    """Inference side of SURVEY 8d: (a) batched heads (`return_head=True`), (b) the reference's B=1 decode
    tail (scores > 0.05, class-agnostic NMS 0.5: retinanet.py:297-318) including the host-visible outputs."""
    was_training = model.training
    model.eval()
    retina = args.arch == 'retinanet'

    def timed(fn, n):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    with torch.no_grad():
        one = img[:1].contiguous()
        t_dec = timed(lambda: model(one, return_loss=False), args.steps)
        out = {'metric': 'images/sec inference (%s, %dx%d)' % (name, args.size, args.size), 'unit': 'images/sec',
               'n_gpus': 1, 'higher_is_better': True, 'dtype': 'bf16', 'data': 'synthetic',
               'b1_decode_nms': {'value': round(1.0 / t_dec, 2), 'ms_per_image': round(1e3 * t_dec, 3)}}
        if retina:
            t_heads = timed(lambda: model(img, return_head=True), args.steps)
            out['heads_batched'] = {'batch': args.batch, 'value': round(args.batch / t_heads, 2),
                                    'ms_per_batch': round(1e3 * t_heads, 3)}
        out['value'] = out.get('heads_batched', out['b1_decode_nms'])['value']
    if was_training:
        model.train()
        model.freeze_bn()
    if emit:
        print(json.dumps(out))
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: this parent (which has not touched the GPU:
    importing torch does not initialise HIP) starts the N ranks through torch.distributed.run, one per GPU,
    and returns their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%s (launch one rank per GPU: python -m torch.distributed.run '
                 '--nproc-per-node %d ... bench.py --gpus %d)' % (args.gpus, os.environ.get('WORLD_SIZE', '1'),
                                                                 args.gpus, args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU (no CPU fallback in the product path)'
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(ndev, 1)        # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        backend = os.environ.get('MI355_DIST_BACKEND', 'nccl')   # 'nccl' == RCCL over xGMI on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    if world == 1 and args.sync_bn:
        # the SyncBN collectives need a process group: a world-size-1 RCCL group exercises the real RCCL stream
        # machinery on one GPU (the all-reduce of one rank is a device-side no-op copy, its launch cost is real)
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(port))
        os.environ.setdefault('MI355_BN_FORCE_ALLREDUCE', '1')
        dist.init_process_group(os.environ.get('MI355_DIST_BACKEND', 'nccl'), rank=0, world_size=1, device_id=dev)
    num_classes = 80
    model, sd = build_model(args.depth, num_classes, dev, args.arch, args.precision, args.sync_bn)
    name = ('RetinaNet-R%d' % args.depth if args.arch == 'retinanet' else
            'KSevenDet-R%d-%s' % (args.depth, 'PANetFPN' if args.arch.endswith('panet') else 'BiFPN' if args.arch.endswith('bifpn') else 'FPN') if args.arch.startswith('ksevendet')
            else 'EfficientDet-D' + args.arch[-1])
    B, S = args.batch, args.size
    # each rank owns its shard of the global batch (weak scaling: per-GPU batch fixed)
    img = torch.from_numpy(synth.synth_images(B, S, S, seed=100 + rank)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(B, S, S, 8, 16, num_classes, seed=200 + rank)).to(dev)
    if args.infer:
        return infer_bench(args, model, name, img, dev)
    loss_acc = torch.zeros(2, device=dev)
    if args.torch_optim:
        params = [p for p in model.parameters()]
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=5e-4)       # train.py:259-260

        def step():
            opt.zero_grad(set_to_none=True)                       # train.py:292
            cl, rl = model(img, ann)                              # train.py:303
            loss = cl.mean() + rl.mean()                          # train.py:312-314
            loss.backward()                                       # train.py:319
            plan = model._plans[next(iter(model._plans))]
            arena = plan.grad_arenas[plan.last_arena]
            if world > 1:
                dist.all_reduce(arena, op=dist.ReduceOp.AVG)      # DP gradient exchange over xGMI
            clip_grad_norm_flat_(arena, 0.1)                      # train.py:320
            opt.step()                                            # train.py:321
            loss_acc.add_(torch.cat([cl.detach(), rl.detach()]))
    else:
        from objectdetection.pytorch_amd.trainer import Trainer
        trainer = Trainer(model, lr=1e-4, weight_decay=5e-4, clip_norm=0.1,   # train.py:259-260, 320
                          graph=bool(args.graph), max_annots=ann.shape[1],
                          overlap=os.environ.get('MI355_DP_OVERLAP', '1') != '0')

        def step():
            loss_acc.add_(trainer.step(img, ann))                 # fwd+loss+bwd+allreduce+clip+AdamW

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # per-step HIP events on the launch stream (every step ends there: the plan joins its side stream): no sync
    # inside the timed region; the MEDIAN step time is reported beside the mean the headline value is made of
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    loop_host_ms = 1e3 * (time.perf_counter() - t0) / args.steps  # host time per step INSIDE the loop: includes the waits on a
    #                                                               full launch queue (the GPU is the bottleneck), not a host cost
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    own_ms = 1e3 * elapsed / args.steps
    final_loss = [round(float(v), 5) for v in (loss_acc / (args.warmup + args.steps)).tolist()]
    # host cost of a step proper: ONE step enqueued onto an idle stream (nothing to wait for), best of 5
    host_ms = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step()
        host_ms = min(host_ms, 1e3 * (time.perf_counter() - th))
    torch.cuda.synchronize()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    ranks_seen, per_rank_ms = 1, [round(own_ms, 3)]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                         # collective-verified rank count
        ranks_seen = int(ones.item())
        allms = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allms, torch.tensor([own_ms], device=dev, dtype=torch.float64))
        per_rank_ms = [round(float(x.item()), 3) for x in allms]
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    out = {
        'metric': 'images/sec training (%s, %dx%d, bs=%d/GPU)' % (name, S, S, B),
        'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'ms_per_step_median': round(median_ms, 3),
        'host_enqueue_ms_per_step': round(host_ms, 3), 'host_loop_ms_per_step': round(loop_host_ms, 3),
        'ranks_seen': ranks_seen, 'ms_per_step_per_rank': per_rank_ms, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.precision == 'bf16' else 'f32', 'data': 'synthetic',
        'config': {'workload': '%s, %dx%d, %s, batch=%d/GPU, fwd+FocalLoss+bwd+clip+AdamW, %s'
                               % (name, S, S, 'bf16 MFMA convs (fp32 accumulate, fp32 loss and optimizer)' if args.precision == 'bf16'
                                  else 'fp32-accurate mode: exact-f32 MFMA convs (v_mfma_f32_16x16x4_f32), fp32 everywhere', B,
                                  'SynchronizedBatchNorm2d in train mode (batch statistics + [sum | sum^2] all-reduce per layer and direction)'
                                  if args.sync_bn else 'frozen BN'),
                   'global_batch': world * B, 'parallelism': 'dp%d' % world,
                   'dp_exchange': ('per-range all-reduce overlapped with backward' if os.environ.get('MI355_DP_OVERLAP', '1') != '0' else 'one all-reduce after backward') if world > 1 else None,
                   'launch': 'hipGraph replay' if (args.graph and not args.torch_optim and not (world > 1 and os.environ.get('MI355_DP_OVERLAP', '1') != '0')) else 'eager',
                   'final_loss': final_loss},
    }
    plan = model._plans[next(iter(model._plans))]
    conv_bound = args.arch == 'retinanet' or args.arch.startswith('ksevendet')
    if rank == 0 and not args.no_roofline and not conv_bound:
        # HBM-bound configs (depthwise / SE / BiFPN passes): algorithmic bytes of the step's launches over
        # the step time, against the HBM peak
        nbytes, per_kernel = plan_launch_bytes(plan)
        top = sorted(per_kernel.items(), key=lambda kv: -kv[1])[:4]
        gbs = nbytes / (median_ms * 1e-3) / 1e9
        tr = measured_traffic(args)
        out['roofline'] = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                           'frac': round(gbs / PEAK_HBM_GBS, 4), 'traffic': tr['step'] if tr else None,
                           'traffic_source': 'committed' if tr else None,
                           'traffic_note': ('HBM-side bytes of ALL kernels of one step (PMC passes of the same command, kept in %s, '
                                            'collected on kernel sources %s; %s)' % (tr['file'], tr['csrc_sha16'],
                                            'sources CHANGED since' if tr['stale'] else 'sources unchanged since')) if tr else None,
                           'traffic_stale': tr['stale'] if tr else None,
                           'algorithmic_bytes_per_step': round(nbytes),
                           'kernel': 'whole step (%d launches): every launch\'s operands once; largest by bytes: %s'
                                     % (len(plan.prep_steps) + len(plan.fwd_steps) + len(plan.bwd_steps),
                                        ', '.join('%s %.0f MB' % (k.replace('mi355_', ''), v / 1e6) for k, v in top)),
                           'step_ms_median': round(median_ms, 3)}
    if rank == 0 and not args.no_roofline and conv_bound:
        fwd_f, train_f = conv_flops(plan)
        conv_ms, nlaunch = time_conv_kernels(plan, 5)
        ach = train_f / (conv_ms * 1e-3) / 1e12
        tr = measured_traffic(args)
        peak = PEAK_BF16_TFLOPS if args.precision == 'bf16' else PEAK_F32_TFLOPS
        out['roofline'] = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak,
                           'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': tr['conv'] if tr else None,
                           'traffic_source': 'committed' if tr else None,
                           'traffic_note': ('HBM-side bytes of the conv kernels per step (PMC passes of the same command, kept in %s, '
                                            'collected on kernel sources %s; %s; not measured in this run)' % (tr['file'], tr['csrc_sha16'],
                                            'sources CHANGED since' if tr['stale'] else 'sources unchanged since')) if tr else None,
                           'traffic_stale': tr['stale'] if tr else None,
                           'algorithmic_bytes_per_step': round(conv_bytes(plan)),
                           'kernel': 'conv_gather(+row3, split-K) + conv_wgrad kernels (%d launches/step)' % nlaunch,
                           'algorithmic_gflop_per_image': round(train_f / B / 1e9, 2),
                           'conv_ms_per_step': round(conv_ms, 3)}
        if args.sync_bn:
            bn_ms, nbn = time_conv_kernels(plan, 5, prefix=('mi355_bn_nhwc', 'py:bn_allreduce', 'mi355_rccl_'))
            nlayers = len(plan.bn_saved)
            out['sync_bn'] = {'bn_layers': nlayers, 'bn_launches_per_step': nbn, 'bn_ms_per_step': round(bn_ms, 3),
                              'bn_us_per_layer_and_direction': round(1e3 * bn_ms / max(2 * nlayers, 1), 2),
                              'collectives_per_step': sum(1 for st in plan.fwd_steps + plan.bwd_steps
                                                          if st[2].startswith(('py:', 'mi355_rccl_'))),
                              'collectives_issued_by': ('the launch-plan runner (mi355_rccl_allreduce_sum_f32 table steps)'
                                                        if any(st[2].startswith('mi355_rccl_') for st in plan.fwd_steps)
                                                        else 'torch.distributed host steps'),
                              'group': '%s, world size %d' % (dist.get_backend() if dist.is_initialized() else None, world)}
    if args.arch.startswith('ksevendet') or args.sync_bn:   # (supplementary lines: training step + roofline only)
        args.no_infer = args.no_cpu_baseline = True
    if rank == 0 and world == 1 and not args.no_infer:
        inf = infer_bench(args, model, name, img, dev, emit=False)         # also inference img/s
        out['infer'] = {k: inf[k] for k in ('b1_decode_nms', 'heads_batched') if k in inf}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(args.depth, S, sd, num_classes, args.arch)
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
