"""GPU: the conv kernels and the whole training step at BASELINE.json configs[1] sizes (RetinaNet-R50, 512 x 512,
16 images), where an fp32 reference of the full tensors would take minutes on the host: size-independent
properties instead — exact homogeneity under a power-of-two scale (bf16 / fp32 roundings commute with it), batch
permutation (tile scheduling must not leak between images; exact) and batch slices (another launch shape: equal
up to the output rounding), fp64 spot checks of
sampled output elements, duplicate-batch invariance of the mean loss, run-to-run bit reproducibility."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _rand_bf16(shape, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def _shadow(w):
    from objectdetection.pytorch_amd import ops_conv
    K, C, R, S = w.shape
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    w_fwd = torch.zeros((K, R * S, C), dtype=torch.bfloat16, device=DEV)
    w_dgr = torch.zeros((C, R * S, K), dtype=torch.bfloat16, device=DEV)
    ops_conv.prep_weight(wd, C, None, w_fwd, w_dgr)
    return w_fwd, w_dgr


def test_head_tower_conv_at_full_size():
    """P3 level of the head towers: 16 x 64 x 64 x 256 -> 256, 3x3 (the launch class that dominates the step)."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K = 16, 64, 64, 256, 256
    x = _rand_bf16((N, H, W, C), 1)
    w = (torch.randn((K, C, 3, 3), generator=torch.Generator().manual_seed(2)) * (2.0 / (9 * C)) ** 0.5)
    w = w.to(torch.bfloat16).float()
    w_fwd, w_dgr = _shadow(w)
    y = ops_conv.conv_fwd(x, w_fwd, K, 3, 3, 1, 1)
    # exact homogeneity, batch permutation, batch slices
    assert torch.equal(ops_conv.conv_fwd(x * 2, w_fwd, K, 3, 3, 1, 1), y * 2)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).to(DEV)
    assert torch.equal(ops_conv.conv_fwd(x[perm].contiguous(), w_fwd, K, 3, 3, 1, 1), y[perm])
    # a 2-image launch takes another tile shape (another fp32 summation order): equal up to the bf16 rounding
    part = ops_conv.conv_fwd(x[5:7].contiguous(), w_fwd, K, 3, 3, 1, 1)
    assert float((part.float() - y[5:7].float()).abs().max()) <= 2.0 ** -7 * float(y.float().abs().max())
    # fp64 spot checks (corners, edges, interior)
    xc, wc = x.float().cpu().double(), w.double()
    for n, p, q, k in ((0, 0, 0, 0), (15, 63, 63, 255), (7, 0, 31, 100), (3, 40, 63, 17), (9, 33, 20, 201)):
        acc = 0.0
        for r in range(3):
            for s in range(3):
                h, ww = p - 1 + r, q - 1 + s
                if 0 <= h < H and 0 <= ww < W:
                    acc += float((xc[n, h, ww] * wc[k, :, r, s]).sum())
        assert abs(float(y[n, p, q, k]) - acc) <= 4e-3 * max(1.0, abs(acc)), (n, p, q, k, float(y[n, p, q, k]), acc)
    # data gradient: same three properties
    dy = _rand_bf16((N, H, W, K), 4)
    dx = ops_conv.conv_dgrad(dy, w_dgr, H, W, C, 3, 3, 1, 1)
    assert torch.equal(ops_conv.conv_dgrad(dy * 2, w_dgr, H, W, C, 3, 3, 1, 1), dx * 2)
    assert torch.equal(ops_conv.conv_dgrad(dy[perm].contiguous(), w_dgr, H, W, C, 3, 3, 1, 1), dx[perm])
    part = ops_conv.conv_dgrad(dy[2:3].contiguous(), w_dgr, H, W, C, 3, 3, 1, 1)
    assert float((part.float() - dx[2:3].float()).abs().max()) <= 2.0 ** -7 * float(dx.float().abs().max())
    # weight gradient (65 536-pixel reduction per element): fp64 spot checks + invariance under a batch permutation
    G = ops_conv.conv_wgrad(x, dy, 3, 3, 1, 1)                       # [K][9][C] fp32
    Gp = ops_conv.conv_wgrad(x[perm].contiguous(), dy[perm].contiguous(), 3, 3, 1, 1)
    scale = float(G.abs().max())
    assert float((G - Gp).abs().max()) <= 2e-5 * scale                # same terms, another summation order
    dyc = dy.float().cpu().double()
    for k, c, r, s in ((0, 0, 0, 0), (255, 255, 2, 2), (10, 200, 1, 1), (128, 3, 0, 2), (77, 64, 2, 0)):
        hs, he = max(0, 1 - r), min(H, H + 1 - r)                      # output rows whose tap (r, s) is inside
        ws, we = max(0, 1 - s), min(W, W + 1 - s)
        ref = float((dyc[:, hs:he, ws:we, k] * xc[:, hs + r - 1:he + r - 1, ws + s - 1:we + s - 1, c]).sum())
        assert abs(float(G[k, r * 3 + s, c]) - ref) <= 2e-4 * scale, (k, c, r, s, float(G[k, r * 3 + s, c]), ref)


def _model(nc=80):
    import bench
    return bench.build_model(50, nc, DEV)[0]


def test_training_step_properties_at_configs1_size(monkeypatch):
    """RetinaNet-R50, 512 x 512, 16 images: a batch made of the same 8 images twice has the mean loss and the
    mean gradient of the 8-image batch (per-image loss terms, losses.py:118, :161, :168); the deterministic
    split-K mode reproduces the 152 MB gradient arena bit for bit."""
    monkeypatch.setenv('MI355_DETERMINISTIC', '1')
    img8 = torch.from_numpy(synth.synth_images(8, 512, 512, seed=1)).to(DEV)
    ann8 = torch.from_numpy(synth.synth_annotations(8, 512, 512, 8, 16, 80, seed=2)).to(DEV)
    img16, ann16 = torch.cat([img8, img8]), torch.cat([ann8, ann8])
    model = _model()

    def run(img, ann):
        model.zero_grad()
        cl, rl = model(img, ann)
        (cl + rl).sum().backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        return torch.stack([cl.detach().reshape(()), rl.detach().reshape(())]), grads.clone()

    l16, g16 = run(img16, ann16)
    l16b, g16b = run(img16, ann16)
    assert torch.equal(l16, l16b) and torch.equal(g16, g16b)           # bit-reproducible at full size
    l8, g8 = run(img8, ann8)
    # (the 8-image launches take other tile shapes than the 16-image ones: bf16 activations differ in the last bit)
    np.testing.assert_allclose(l16.cpu().numpy(), l8.cpu().numpy(), rtol=2e-4)
    assert A_positive(l16)
    cos = float((g16.double() * g8.double()).sum() / (g16.double().norm() * g8.double().norm()))
    assert cos >= 1 - 1e-4, cos                                        # (bf16 head tensors: not bit-equal)
    assert abs(float(g16.norm()) / float(g8.norm()) - 1.0) <= 2e-3


def A_positive(losses):
    return bool((losses > 0).all()) and bool(torch.isfinite(losses).all())
