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
    np.testing.assert_allclose(l16.cpu().numpy(), l8.cpu().numpy(), rtol=5e-4)      # (measured 1e-4 .. 3.5e-4)
    assert A_positive(l16)
    cos = float((g16.double() * g8.double()).sum() / (g16.double().norm() * g8.double().norm()))
    assert cos >= 1 - 1e-4, cos                                        # (bf16 head tensors: not bit-equal)
    assert abs(float(g16.norm()) / float(g8.norm()) - 1.0) <= 2e-3


def A_positive(losses):
    return bool((losses > 0).all()) and bool(torch.isfinite(losses).all())


# ---------------------------------------------------------------------------------------------------------
# The other BASELINE.json configs at their full sizes: configs[2] RetinaNet-R50 640 x 640 bs16 (its SyncBN
# part: test_gpu_bn_plan.py), configs[3] EfficientDet-D0 512 x 512 bs16, configs[4] EfficientDet-D2 768 x 768 bs8.
@pytest.mark.parametrize('arch,size,batch', [('retinanet', 640, 16), ('efficientdet-d0', 512, 16),
                                             ('efficientdet-d2', 768, 8)])
def test_training_step_properties_at_other_config_sizes(monkeypatch, arch, size, batch):
    """Run-to-run reproducibility, duplicate-batch invariance of the per-image-normalised losses (losses.py:118,
    :161, :168) and of the mean gradient, finite positive losses, at the full size of the config."""
    import bench
    monkeypatch.setenv('MI355_DETERMINISTIC', '1')
    retina = arch == 'retinanet'
    model = bench.build_model(50, 80, DEV, arch)[0]
    if not retina:
        synth.disable_drop_connect(model)                  # (its draws differ per image: not duplicate-invariant)
    half = batch // 2
    imgh = torch.from_numpy(synth.synth_images(half, size, size, seed=1)).to(DEV)
    annh = torch.from_numpy(synth.synth_annotations(half, size, size, 8, 16, 80, seed=2)).to(DEV)
    img, ann = torch.cat([imgh, imgh]), torch.cat([annh, annh])

    def run(i, a):
        model.zero_grad()
        cl, rl = model(i, a)
        (cl + rl).sum().backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        return torch.stack([cl.detach().reshape(()), rl.detach().reshape(())]), grads.clone()

    l1, g1 = run(img, ann)
    l2, g2 = run(img, ann)
    assert A_positive(l1)
    # bit-reproducible: slab split-K weight gradients; EfficientDet: depthwise dW / column sums as per-block partial rows,
    # squeeze-excite FC and fusion-weight gradients as ordered sums (no fp32 atomics anywhere in the step)
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    lh, gh = run(imgh, annh)
    # (the half-batch launches take other tile shapes / slab counts: bf16 activations differ in the last bit, and
    #  ~100 bf16 layers of EfficientDet amplify that more than RetinaNet's ~60)
    np.testing.assert_allclose(l1.cpu().numpy(), lh.cpu().numpy(), rtol=5e-4 if retina else 2e-2)
    cos = float((g1.double() * gh.double()).sum() / (g1.double().norm() * gh.double().norm()))
    assert cos >= (1 - 2e-4 if retina else 0.995), cos
    assert abs(float(g1.norm()) / float(gh.norm()) - 1.0) <= (5e-3 if retina else 5e-2)


def test_depthwise_and_bifpn_node_at_d0_full_size():
    """fp64 spot checks at EfficientDet-D0 512 x 512 bs16 sizes: the first MBConv depthwise conv (16 x 256 x 256 x 32,
    k3 s1, swish on load, efficientnet.py:87-95) and a BiFPN P3 node (16 x 64 x 64 x 64 + upsampled 32 x 32,
    efficientdet.py:213-265)."""
    from objectdetection.pytorch_amd import ops_effdet as E
    from objectdetection.pytorch_amd._lib import call, ptr, c_int, stream_of
    N, H, W, C, k = 16, 256, 256, 32, 3
    z = _rand_bf16((N, H, W, C), 1)
    g = torch.Generator().manual_seed(2)
    w = (torch.randn((C, 1, k, k), generator=g) * 0.3).to(DEV)
    scale = (torch.rand(C, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(C, generator=g) * 0.1).to(DEV)
    w_t = torch.empty((k * k, C), device=DEV)
    E.dw_prep_weight(w.contiguous(), scale, w_t)
    y = E.dwconv_fwd(z, True, w_t, shift, k, 1)
    assert torch.equal(E.dwconv_fwd(z, True, w_t, shift, k, 1), y)        # reproducible
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).to(DEV)
    assert torch.equal(E.dwconv_fwd(z[perm].contiguous(), True, w_t, shift, k, 1), y[perm])
    zc, wc = z.float().cpu().double(), (w * scale[:, None, None, None]).cpu().double()
    sw = zc * torch.sigmoid(zc)
    for n, p, q, c in ((0, 0, 0, 0), (15, 255, 255, 31), (7, 0, 100, 5), (3, 128, 255, 17), (9, 77, 33, 20)):
        acc = float(shift[c])
        for r in range(k):
            for s in range(k):
                h, ww = p - 1 + r, q - 1 + s
                if 0 <= h < H and 0 <= ww < W:
                    acc += float(sw[n, h, ww, c] * wc[c, 0, r, s])
        assert abs(float(y[n, p, q, c]) - acc) <= 1e-2 * max(1.0, abs(acc)), (n, p, q, c, float(y[n, p, q, c]), acc)
    # BiFPN node: z = w0 * a + w1 * up2(b) + w2 * c (pre-activation, swish is applied by the consumer)
    N, H, W, C = 16, 64, 64, 64
    a, b, c3 = _rand_bf16((N, H, W, C), 4), _rand_bf16((N, H // 2, W // 2, C), 5), _rand_bf16((N, H, W, C), 6)
    wd = torch.tensor([0.45, 0.2, 0.35], device=DEV)
    out = torch.empty_like(a)
    call('mi355_bifpn_node_fwd', ptr(a), ptr(b), c_int(1), ptr(c3), ptr(wd), ptr(out), c_int(N), c_int(H), c_int(W),
         c_int(C), stream_of(a))
    torch.cuda.synchronize()
    idx = torch.randint(0, N * H * W * C, (4096,), generator=torch.Generator().manual_seed(7))
    n_, rem = idx // (H * W * C), idx % (H * W * C)
    p_, rem = rem // (W * C), rem % (W * C)
    q_, c_ = rem // C, rem % C
    ac, bc, cc = a.float().cpu().double(), b.float().cpu().double(), c3.float().cpu().double()
    ref = 0.45 * ac[n_, p_, q_, c_] + float(np.float32(0.2)) * bc[n_, p_ // 2, q_ // 2, c_] + float(np.float32(0.35)) * cc[n_, p_, q_, c_]
    got = out.float().cpu().double()[n_, p_, q_, c_]
    assert float((got - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
