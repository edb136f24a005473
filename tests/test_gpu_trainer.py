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


@pytest.mark.parametrize('arch', ['retinanet', 'efficientdet'])
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
        return synth.disable_drop_connect(m)       # eager and replayed steps must draw nothing to be comparable
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
    per-range exchange} x {eager, hipGraph}: same batches -> same parameters."""
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
        for name, overlap, graph in (('single', False, False), ('overlap', True, False), ('overlap_graph', True, True)):
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
    for name in ('single', 'overlap', 'overlap_graph'):
        assert np.array_equal(r0[name]['flat'], r1[name]['flat']), name       # replicas stay identical
    # the ranges partition the arena and a two-rank sum is order independent: bit-identical parameters
    assert np.array_equal(r0['single']['flat'], r0['overlap']['flat'])
    assert np.array_equal(r0['single']['flat'], r0['overlap_graph']['flat'])
    assert not np.array_equal(r0['single']['loss'], r1['single']['loss'])    # different shards
