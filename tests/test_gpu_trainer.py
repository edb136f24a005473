"""GPU: the flat-arena Trainer (HIP clip + AdamW, no autograd) against the reference loop body
(train.py:292-321) run with torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on the same
drop-in model through autograd."""
import copy
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _model(dev, depth=18, nc=8):
    from objectdetection.pytorch_amd.architectures import retinanet
    model = retinanet.retinanet(depth, num_classes=nc)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=0, gain=0.87)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev)
    model.train()
    model.freeze_bn()
    return model


def test_trainer_matches_torch_loop():
    from objectdetection.pytorch_amd.trainer import Trainer
    dev = torch.device('cuda', 0)
    B, H, W, nc = 2, 128, 160, 8
    img = torch.from_numpy(synth.synth_images(B, H, W, seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(B, H, W, 6, 8, nc, seed=2)).to(dev)
    ref = _model(dev)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=5e-4)
    ref_losses = []
    for _ in range(3):
        opt.zero_grad()
        cl, rl = ref(img, ann)
        (cl.mean() + rl.mean()).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        opt.step()
        ref_losses.append([cl.item(), rl.item()])
    mine = _model(dev)
    p0 = {k: v.detach().clone() for k, v in mine.named_parameters()}
    tr = Trainer(mine)
    keys = list(mine.state_dict().keys())
    got = [tr.step(img, ann).clone() for _ in range(3)]
    assert tr.step_count == 3
    assert list(mine.state_dict().keys()) == keys            # flattening keeps the state_dict
    np.testing.assert_allclose(torch.stack(got).cpu().numpy(), np.array(ref_losses), rtol=1e-2)
    for (k, a), (_, b) in zip(mine.named_parameters(), ref.named_parameters()):
        da, db = (a - p0[k]).float(), (b - p0[k]).float()
        # 3 Adam steps of lr 1e-4: |delta| <= ~3e-4 per element; wgrad atomics reorder sums
        # Adam's first steps move every element by ~lr*sign(g): elements whose tiny gradient
        # differs in the last bits (fp32 atomics reorder the wgrad sums) may differ by O(lr);
        # the bulk must agree
        # (exact optimizer arithmetic is pinned by test_flat_adamw_matches_torch below)
        diff = (da - db).abs().detach()
        assert float(diff.mean()) <= 0.15 * float(db.abs().mean()) + 1e-7, (k, float(diff.mean()))
        assert float(db.abs().max()) > 0
    # clip semantic: the gradient arena holds the clipped gradient (norm <= 0.1)
    plan = mine._plans[next(iter(mine._plans))]
    gn = float(torch.linalg.vector_norm(plan.grad_arenas[plan.last_arena]))
    assert gn <= 0.1 * 1.001


def test_trainer_skips_zero_loss_step():
    """train.py:316-317: `if bool(loss == 0): continue` (all images without annotations)."""
    from objectdetection.pytorch_amd.trainer import Trainer
    dev = torch.device('cuda', 0)
    mine = _model(dev)
    tr = Trainer(mine)
    img = torch.from_numpy(synth.synth_images(2, 64, 64, seed=1)).to(dev)
    ann = -torch.ones(2, 1, 5, device=dev)
    before = tr.flat.clone()
    losses = tr.step(img, ann)
    assert float(losses.abs().sum()) == 0.0
    assert torch.equal(tr.flat, before) and tr.step_count == 0


def test_flat_adamw_matches_torch():
    """clip_grad_norm_(0.1) + AdamW(lr 1e-4, wd 5e-4) on identical gradients: the three HIP
    launches against torch's own implementation, 4 steps (bias correction included)."""
    import ctypes
    from objectdetection.pytorch_amd import _lib
    from objectdetection.pytorch_amd._lib import c_f32, c_i64, c_int, c_vp, ptr
    dev = torch.device('cuda', 0)
    n = 1000004
    g = torch.Generator(device='cpu').manual_seed(3)
    p0 = torch.randn(n, generator=g).to(dev)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-4, weight_decay=5e-4)
    flat, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    partials = torch.zeros(1024, device=dev)
    st = c_vp(torch.cuda.current_stream().cuda_stream)
    for it in range(4):
        grad = (torch.randn(n, generator=g) * (0.001 if it % 2 else 10.0)).to(dev)   # clipped / not
        ref.grad = grad.clone()
        total = torch.nn.utils.clip_grad_norm_([ref], 0.1)
        opt.step()
        mine = grad.clone()
        _lib.call('mi355_grad_sumsq', ptr(mine), c_i64(n), ptr(partials), st)
        _lib.call('mi355_adamw_prepare', ptr(partials), c_vp(0), c_f32(0.1), ptr(state), st)
        _lib.call('mi355_adamw_flat', ptr(flat), ptr(mine), ptr(m), ptr(v), c_i64(n), c_f32(1e-4),
                  c_f32(0.9), c_f32(0.999), c_f32(1e-8), c_f32(5e-4), ptr(state), c_int(1), st)
        norm = state.view(torch.float32)[3].item()
        np.testing.assert_allclose(norm, float(total), rtol=1e-5)
        np.testing.assert_allclose(mine.cpu().numpy(), ref.grad.cpu().numpy(), rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(flat.cpu().numpy(), ref.detach().cpu().numpy(), rtol=5e-7, atol=2e-7)   # 1-2 ulp
    assert int(state[0]) == 4


@pytest.mark.parametrize('arch', ['retinanet', 'efficientdet', 'efficientdet_dropconnect'])
def test_graph_replay_matches_eager(arch):
    """Trainer(graph=True): the step replayed from captured hipGraphs follows the eager trajectory."""
    from objectdetection.pytorch_amd.trainer import Trainer
    dev = torch.device('cuda', 0)
    B, H, W, nc = 2, 128, 128, 8

    def make():
        if arch == 'retinanet':
            return _model(dev)
        from objectdetection.pytorch_amd.architectures import efficientdet
        m = efficientdet.build_efficientdet('efficientdet-d0', num_classes=nc)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = synth.synth_state_dict(shapes, seed=0, gain=1.0, head_std=0.05)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev)
        m.train()
        m.freeze_bn()
        if arch == 'efficientdet':
            return synth.disable_drop_connect(m)
        # stochastic depth on: the uniforms are drawn outside the captured region, one draw per step from the
        # plan's own generator, so the eager and the replayed run see the same sequence after the same seed
        torch.manual_seed(7)
        return m
    batches = [(torch.from_numpy(synth.synth_images(B, H, W, seed=10 + i)).to(dev),
                torch.from_numpy(synth.synth_annotations(B, H, W, 6, 8 + 2 * (i % 2), nc, seed=20 + i)).to(dev))
               for i in range(4)]
    out = {}
    for mode in (False, True):
        tr = Trainer(make(), graph=mode, max_annots=16)
        out[mode] = torch.stack([tr.step(im, an).clone() for im, an in batches]).cpu().numpy()
        assert tr.step_count == 4
    np.testing.assert_allclose(out[True], out[False], rtol=2e-2)
    assert not np.allclose(out[True][0], out[True][3])       # different batches really flowed in


def test_deterministic_split_k_mode(monkeypatch):
    """MI355_DETERMINISTIC=1: slab split-K weight gradients are bit-reproducible run to run and agree
    with the default fp32-atomic path."""
    dev = torch.device('cuda', 0)
    img = torch.from_numpy(synth.synth_images(2, 128, 128, seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(2, 128, 128, 6, 8, 8, seed=2)).to(dev)

    def grads(model):
        model.zero_grad()
        cl, rl = model(img, ann)
        (cl + rl).sum().backward()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.dim() == 4}

    base = grads(_model(dev))
    monkeypatch.setenv('MI355_DETERMINISTIC', '1')
    m = _model(dev)
    a, b = grads(m), grads(m)
    for k in a:
        assert torch.equal(a[k], b[k]), k                    # bit-reproducible
        ref = base[k]
        assert float((a[k] - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-9, k


def _dp_worker(rank, world, port, out):
    """Two ranks (gloo; both on the one GPU of the test box) x {one exchange after backward, overlapped
    per-range exchange, hipGraph replay with one exchange}: same batches -> same parameters."""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from objectdetection.pytorch_amd.trainer import Trainer
        dev = torch.device('cuda', 0)
        img = torch.from_numpy(synth.synth_images(2, 128, 128, seed=10 + rank)).to(dev)
        ann = torch.from_numpy(synth.synth_annotations(2, 128, 128, 6, 8, 8, seed=20 + rank)).to(dev)
        res = {}
        for name, overlap, graph in (('single', False, False), ('overlap', True, False), ('graph', False, True)):
            model = _model(dev)
            tr = Trainer(model, lr=1e-3, clip_norm=0.1, overlap=overlap, graph=graph, max_annots=8,
                         segment_bytes=4 << 20)
            losses = [tr.step(img, ann).clone() for _ in range(3)]
            plan = model._plans[next(iter(model._plans))]
            res[name] = dict(flat=tr.flat.cpu().numpy(), loss=torch.stack(losses).cpu().numpy(),
                             nseg=len(plan.backward_segments(4 << 20)), steps=tr.step_count)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_overlapped_dp_exchange_two_ranks():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0['overlap']['nseg'] >= 3 and r0['single']['steps'] == 3
    for name in ('single', 'overlap', 'graph'):
        assert np.array_equal(r0[name]['flat'], r1[name]['flat']), name       # replicas stay identical
    # the ranges partition the arena and a two-rank sum is order independent: bit-identical parameters
    assert np.array_equal(r0['single']['flat'], r0['overlap']['flat'])
    # hipGraph replay + one exchange between the replays (single compute stream: another accumulation order of the
    # two-stream eager schedule is not involved, the slab sums are order-fixed): the same parameters
    assert np.array_equal(r0['single']['flat'], r0['graph']['flat'])
    assert not np.array_equal(r0['single']['loss'], r1['single']['loss'])    # different shards


def test_flat_adamw_masked_skips_like_grad_none():
    """Parameters whose .grad is None are skipped by torch's clip_grad_norm_ and AdamW (no norm contribution,
    no weight decay, no moments): the masked passes with arena skip ranges do the same."""
    from objectdetection.pytorch_amd import _lib
    from objectdetection.pytorch_amd._lib import c_f32, c_i64, c_int, c_vp, ptr
    dev = torch.device('cuda', 0)
    sizes = [1000, 7, 4099, 64, 13, 50000]
    frozen = {1, 3, 4}                                       # two adjacent -> merged range, unaligned bounds
    g = torch.Generator(device='cpu').manual_seed(5)
    n = sum(sizes)
    npad = (n + 3) // 4 * 4
    p0 = torch.randn(npad, generator=g).to(dev)
    offs = np.cumsum([0] + sizes)
    refs = [torch.nn.Parameter(p0[offs[i]:offs[i + 1]].clone()) for i in range(len(sizes))]
    opt = torch.optim.AdamW(refs, lr=1e-3, weight_decay=5e-2)
    flat, m, v = p0.clone(), torch.zeros(npad, device=dev), torch.zeros(npad, device=dev)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    partials = torch.zeros(1024, device=dev)
    ranges, cur = [], None
    for i in range(len(sizes)):
        if i in frozen:
            if cur is not None and cur[1] == offs[i]:
                cur[1] = int(offs[i + 1])
            else:
                cur = [int(offs[i]), int(offs[i + 1])]
                ranges.append(cur)
    skip = torch.tensor(ranges, dtype=torch.int64, device=dev).reshape(-1)
    st = c_vp(torch.cuda.current_stream().cuda_stream)
    for it in range(3):
        grad = (torch.randn(npad, generator=g) * (0.0005 if it == 1 else 3.0)).to(dev)
        grad[n:] = 0
        for i, r in enumerate(refs):
            r.grad = None if i in frozen else grad[offs[i]:offs[i + 1]].clone()
        total = torch.nn.utils.clip_grad_norm_(refs, 0.1)
        opt.step()
        mine = grad.clone()                                   # frozen ranges hold garbage gradients: must be ignored
        _lib.call('mi355_grad_sumsq_masked', ptr(mine), c_i64(npad), ptr(partials), ptr(skip), c_int(len(ranges)), st)
        _lib.call('mi355_adamw_prepare', ptr(partials), c_vp(0), c_f32(0.1), ptr(state), st)
        _lib.call('mi355_adamw_flat_masked', ptr(flat), ptr(mine), ptr(m), ptr(v), c_i64(npad), c_f32(1e-3),
                  c_f32(0.9), c_f32(0.999), c_f32(1e-8), c_f32(5e-2), ptr(state), c_int(1), ptr(skip),
                  c_int(len(ranges)), st)
        np.testing.assert_allclose(state.view(torch.float32)[3].item(), float(total), rtol=1e-5)
    for i, r in enumerate(refs):
        got = flat[offs[i]:offs[i + 1]]
        if i in frozen:
            assert torch.equal(got, p0[offs[i]:offs[i + 1]])          # bit-identical: never touched
            assert float(m[offs[i]:offs[i + 1]].abs().max()) == 0.0
        else:
            np.testing.assert_allclose(got.cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_trainer_leaves_frozen_parameters_alone():
    """requires_grad=False parameters (the head-only fine-tune of train.py:206-216) are skipped by torch's
    clip + AdamW; the flat trainer must neither decay nor move them, and must not count them in the norm."""
    from objectdetection.pytorch_amd.trainer import Trainer
    dev = torch.device('cuda', 0)
    img = torch.from_numpy(synth.synth_images(2, 128, 128, seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(2, 128, 128, 6, 8, 8, seed=2)).to(dev)
    mine = _model(dev)
    for k, p in mine.named_parameters():
        if not (k.startswith('regressionModel') or k.startswith('classificationModel')):
            p.requires_grad = False
    before = {k: p.detach().clone() for k, p in mine.named_parameters()}
    tr = Trainer(mine)
    for _ in range(2):
        tr.step(img, ann)
    plan = mine._plans[next(iter(mine._plans))]
    assert len(plan.skip_ranges()) == 1                       # backbone + pyramid: one merged range
    moved = 0
    for k, p in mine.named_parameters():
        if p.requires_grad:
            moved += int((p.detach() != before[k]).any())
        else:
            assert torch.equal(p.detach(), before[k]), k
    assert moved > 0
    # the clip norm is the norm of the trainable gradients only
    arena = plan.grad_arenas[plan.last_arena]
    lo, hi = plan.skip_ranges()[0]
    live = torch.cat([arena[:lo], arena[hi:]])
    assert float(torch.linalg.vector_norm(live)) <= 0.1 * 1.001


def test_backward_after_another_forward_of_the_same_shape():
    """The launch plan keeps ONE set of activations per input shape; the reference's autograd graph keeps them per call.
    A forward between a loss and its backward (validation call, `return_head`: kseven_inference_debuger.py:310-317; a
    second micro-batch) must not change the gradients: the backward re-runs the forward of its own batch first.  Where
    that is impossible (train-mode BatchNorm: the running statistics would move twice; an optimizer step in between) it
    raises instead of returning the gradients of the wrong batch."""
    dev = torch.device('cuda', 0)
    model = _model(dev)
    img = torch.from_numpy(synth.synth_images(2, 64, 64, seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(2, 64, 64, 4, 8, 8, seed=2)).to(dev)
    img2 = torch.from_numpy(synth.synth_images(2, 64, 64, seed=5)).to(dev)
    ann2 = torch.from_numpy(synth.synth_annotations(2, 64, 64, 4, 8, 8, seed=6)).to(dev)

    def grads_of(i, a):
        model.zero_grad()
        cl, rl = model(i, a)
        (cl + rl).sum().backward()
        return [p.grad.clone() for p in model.parameters()]
    ref1, ref2 = grads_of(img, ann), grads_of(img2, ann2)
    # a validation-style call between the forward and the backward
    model.zero_grad()
    cl, rl = model(img, ann)
    model(img2, return_head=True)
    (cl + rl).sum().backward()
    assert all(torch.equal(p.grad, r) for p, r in zip(model.parameters(), ref1))
    # two micro-batches forwarded, then both backwards (gradient accumulation): each backward restores its own batch
    model.zero_grad()
    c1, r1 = model(img, ann)
    c2, r2 = model(img2, ann2)
    (c1 + r1).sum().backward()
    g1 = [p.grad.clone() for p in model.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(g1, ref1))
    (c2 + r2).sum().backward()
    for p, a, b in zip(model.parameters(), ref1, ref2):
        assert torch.allclose(p.grad, a + b, rtol=1e-6, atol=1e-8)
    # another SHAPE has its own plan: nothing to restore
    other = torch.from_numpy(synth.synth_images(2, 96, 64, seed=3)).to(dev)
    model.zero_grad()
    cl, rl = model(img, ann)
    model(other, return_head=True)
    (cl + rl).sum().backward()
    assert all(torch.equal(p.grad, r) for p, r in zip(model.parameters(), ref1))
    # a parameter modified in place between the forward and the backward: raises (so does the reference's autograd)
    model.zero_grad()
    cl, rl = model(img, ann)
    model(img2, return_head=True)
    with torch.no_grad():
        model.conv1.weight.mul_(1.0)
    with pytest.raises(RuntimeError, match='another forward'):
        (cl + rl).sum().backward()
    # train-mode BatchNorm: the forward cannot be re-run silently
    model.train()
    cl, rl = model(img, ann)
    model(img2, ann2)
    with pytest.raises(RuntimeError, match='train-mode BatchNorm'):
        (cl + rl).sum().backward()


def test_reduce_lr_on_plateau_drives_the_flat_optimizer():
    """train.py:268 / :346: `ReduceLROnPlateau(optimizer, patience=3)` stepped with the epoch loss must change the
    learning rate the HIP AdamW launches use."""
    from objectdetection.pytorch_amd.trainer import Trainer
    dev = torch.device('cuda', 0)
    model = _model(dev)
    tr = Trainer(model, lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(tr.optimizer, patience=0, factor=0.1)
    img = torch.from_numpy(synth.synth_images(2, 64, 64, seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(2, 64, 64, 4, 8, 8, seed=2)).to(dev)
    before = tr.flat.clone()
    tr.step(img, ann)
    d1 = float((tr.flat - before).abs().max())
    sched.step(1.0)
    sched.step(2.0)                                           # no improvement -> lr * 0.1
    assert abs(tr.lr - 1e-4) < 1e-12
    mid = tr.flat.clone()
    tr.step(img, ann)
    d2 = float((tr.flat - mid).abs().max())
    assert d2 < 0.5 * d1                                      # Adam step size follows the scheduled lr
    sd = tr.optimizer.state_dict()
    assert sd['param_groups'][0]['lr'] == tr.lr and sd['flat_state']['exp_avg'].shape == tr.m.shape
    tr.optimizer.load_state_dict(sd)


def test_graph_mode_follows_the_lr_scheduler_and_requires_grad_flips():
    """Trainer(graph=True): the captured AdamW launch reads lr / betas / eps / weight_decay from a device vector
    that is refreshed in front of a replay, so `ReduceLROnPlateau` (train.py:268, :346) keeps driving the update, and
    a requires_grad flip (train.py:206-216) re-captures the update graph with the new skip table: the graph-mode
    trainer must follow an eager trainer of the same model bit for bit through both events."""
    from objectdetection.pytorch_amd.trainer import Trainer
    dev = torch.device('cuda', 0)
    img = torch.from_numpy(synth.synth_images(2, 64, 64, seed=1)).to(dev)
    ann = torch.from_numpy(synth.synth_annotations(2, 64, 64, 4, 8, 8, seed=2)).to(dev)
    flats = []
    for graph in (False, True):
        model = _model(dev)
        tr = Trainer(model, lr=1e-3, graph=graph, max_annots=ann.shape[1])
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(tr.optimizer, patience=0, factor=0.1)
        tr.step(img, ann)                       # (graph mode: eager warm-up + capture)
        tr.step(img, ann)                       # first replay at lr 1e-3
        sched.step(1.0)
        sched.step(2.0)                         # no improvement -> lr * 0.1
        assert abs(tr.lr - 1e-4) < 1e-12
        tr.step(img, ann)                       # replay must use 1e-4
        a = tr.flat.clone()
        tr.step(img, ann)
        d_small = float((tr.flat - a).abs().max())
        frozen = model.regressionModel.output.weight
        frozen.requires_grad_(False)
        b = frozen.detach().clone()
        tr.step(img, ann)
        assert torch.equal(frozen.detach(), b), 'a frozen parameter moved (skip table frozen at capture?)'
        torch.cuda.synchronize()
        flats.append((tr.flat.clone(), d_small))
    assert flats[1][1] < 2.5e-4                 # |dp| <= lr-ish per Adam step: 1e-3 would give ~1e-3
    assert torch.equal(flats[0][0], flats[1][0]), 'graph-mode parameters differ from the eager trainer'


@pytest.mark.parametrize('branch', ['0', '1'])
def test_graph_replay_is_bit_identical_to_eager(monkeypatch, branch):
    """hipGraph replay == the eager two-stream schedule, bit for bit (losses and every parameter after 4 steps on 4
    different batches).  branch=1 captures the BRANCHED graph (weight gradients + box head on the side stream): the
    fork / join edges come from the step table (csrc/plan_run.hip), not from host-side flags, so the capture holds the
    same dependency edges as the eager replay."""
    from objectdetection.pytorch_amd.trainer import Trainer
    monkeypatch.setenv('MI355_GRAPH_BRANCH', branch)
    dev = torch.device('cuda', 0)
    B, H, W, nc = 2, 128, 128, 8
    batches = [(torch.from_numpy(synth.synth_images(B, H, W, seed=10 + i)).to(dev),
                torch.from_numpy(synth.synth_annotations(B, H, W, 6, 8 + 2 * (i % 2), nc, seed=20 + i)).to(dev))
               for i in range(4)]
    out = {}
    for mode in (False, True):
        tr = Trainer(_model(dev, depth=50), graph=mode, max_annots=16)
        losses = torch.stack([tr.step(im, an).clone() for im, an in batches])
        torch.cuda.synchronize()
        out[mode] = (losses.cpu(), tr.flat.clone().cpu())
    assert torch.equal(out[True][0], out[False][0]), 'losses differ'
    diff = (out[True][1] != out[False][1]).sum().item()
    assert diff == 0, '%d of %d parameters differ between graph replay and eager' % (diff, out[True][1].numel())
