"""GPU parity of the EfficientDet HBM-bound kernels (through the C ABI) vs torch fp32 on the same
bf16-rounded inputs: depthwise TF-"same" conv fwd / dgrad / wgrad with swish-on-load, squeeze-excite
forward / backward, BiFPN node + fusion weights, zero-padded "same" max-pool."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_effdet as R

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _bf(t):
    return t.to(torch.bfloat16).float()


def _mk(shape, seed, scale=1.0):
    return _bf(torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.bfloat16)


def _close(got, ref, tol):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    mag = ref.abs().max().item() + 1e-12
    assert err <= tol * mag, 'max err %.4g vs magnitude %.4g (tol %.3g)' % (err, mag, tol)


@pytest.mark.parametrize('N,H,W,C,k,s,sw', [(2, 16, 20, 32, 3, 1, True), (2, 17, 21, 24, 3, 2, True),
                                            (1, 16, 16, 40, 5, 2, True), (2, 9, 7, 64, 5, 1, False),
                                            (1, 8, 8, 264 * 8, 3, 1, True), (3, 40, 70, 32, 3, 1, False),
                                            (2, 33, 35, 24, 5, 1, False), (1, 64, 64, 8, 3, 1, False),
                                            (2, 19, 50, 48, 5, 1, True), (2, 17, 21, 24, 3, 2, False),
                                            (1, 16, 16, 40, 5, 2, False), (2, 32, 70, 16, 5, 2, False),
                                            (1, 40, 132, 8, 3, 2, True)])
def test_depthwise_conv(N, H, W, C, k, s, sw):
    from objectdetection.pytorch_amd import ops_effdet as E
    z = _mk((N, C, H, W), 1).requires_grad_(True)
    w = _mk((C, 1, k, k), 2, 0.3).float().requires_grad_(True)
    scale = torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5
    shift = torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.1
    x = R.swish(z) if sw else z
    y = R.conv_same(x, w * scale[:, None, None, None], None, s, C) + shift[None, :, None, None]
    gy = _mk(tuple(y.shape), 5)
    y.backward(gy)
    w_t = torch.empty((k * k, C), device=DEV)
    E.dw_prep_weight(w.detach().to(DEV).contiguous(), scale.to(DEV), w_t)
    zh = _nhwc(z.detach())
    out = E.dwconv_fwd(zh, sw, w_t, shift.to(DEV), k, s)
    _close(out.permute(0, 3, 1, 2), y.detach(), 1e-2)
    gyh = _nhwc(gy)
    dx = E.dwconv_dgrad(gyh, w_t, tuple(zh.shape), k, s, zin=zh if sw else None)
    _close(dx.permute(0, 3, 1, 2), z.grad, 1.5e-2)
    dw_t = torch.zeros((k * k, C), device=DEV)
    cs = torch.zeros(C, device=DEV)
    E.dwconv_wgrad(zh, sw, gyh, dw_t, cs, k, s)
    # raw wgrad G (w.r.t. the scaled weight) = dL/dw / scale
    _close(dw_t.t().reshape(C, 1, k, k) * scale.to(DEV)[:, None, None, None], w.grad, 5e-3)
    _close(cs, gy.sum((0, 2, 3)), 2e-3)
    # the LDS-tiled slab form: same sums, plain stores into NaN-filled rows (every element is written), and
    # bit-reproducible launch to launch
    sl = E.dwconv_wgrad_slabs(zh, sw, gyh, k, s)
    assert torch.isfinite(sl).all()
    tot = sl.double().sum(0).float()
    _close(tot[:k * k * C].view(k * k, C).t().reshape(C, 1, k, k) * scale.to(DEV)[:, None, None, None], w.grad, 5e-3)
    _close(tot[k * k * C:], gy.sum((0, 2, 3)), 2e-3)
    assert torch.equal(sl, E.dwconv_wgrad_slabs(zh, sw, gyh, k, s))


@pytest.mark.parametrize('k,sw', [(3, True), (5, False)])
def test_depthwise_wgrad_of_a_shared_conv_over_pyramid_levels(k, sw):
    """One launch for the five pyramid levels of a shared head conv (efficientdet.py:269-359) = the sum of the
    per-level weight gradients."""
    from objectdetection.pytorch_amd import ops_effdet as E
    C, N = 64, 2
    pairs, ref_w, ref_cs = [], 0.0, 0.0
    for i, hw in enumerate((40, 20, 10, 5, 3)):
        z = _mk((N, C, hw, hw + (i % 2)), 10 + i)
        gy = _mk((N, C, hw, hw + (i % 2)), 20 + i)
        zh, gyh = _nhwc(z), _nhwc(gy)
        pairs.append((zh, gyh))
        one = E.dwconv_wgrad_slabs(zh, sw, gyh, k, 1).double().sum(0)
        ref_w, ref_cs = ref_w + one[:k * k * C], ref_cs + one[k * k * C:]
    sl = E.dwconv_wgrad_slabs_multi(pairs, sw, k, 1)
    assert torch.isfinite(sl).all()
    tot = sl.double().sum(0)
    _close(tot[:k * k * C].float(), ref_w.float().cpu(), 1e-5)
    _close(tot[k * k * C:].float(), ref_cs.float().cpu(), 1e-5)
    assert torch.equal(sl, E.dwconv_wgrad_slabs_multi(pairs, sw, k, 1))


@pytest.mark.parametrize('k,sw', [(3, True), (5, False)])
def test_depthwise_fwd_and_dgrad_over_pyramid_levels_in_one_launch(k, sw):
    """mi355_dwconv_fwd_multi / _dgrad_multi == the per-map launches, bit for bit (same kernel, level table)."""
    from objectdetection.pytorch_amd import ops_effdet as E
    C, N = 64, 2
    w_t = (torch.randn((k * k, C), generator=torch.Generator().manual_seed(1)) * 0.3).to(DEV)
    bias = (torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.1).to(DEV)
    xs = [_nhwc(_mk((N, C, hw, hw + (i % 2)), 30 + i)) for i, hw in enumerate((40, 20, 10, 5, 3))]
    outs = E.dwconv_fwd_multi(xs, sw, w_t, bias, k)
    for x, o in zip(xs, outs):
        one = E.dwconv_fwd(x, sw, w_t, bias, k, 1)
        if x.shape[2] >= 16 and x.shape[1] >= 8:          # the single-map entry takes the tile kernel there
            assert torch.equal(o, one)
        else:
            _close(o, one.float().cpu(), 1e-2)
    dys = [_nhwc(_mk((N, C, x.shape[1], x.shape[2]), 50 + i)) for i, x in enumerate(xs)]
    base = [torch.randn_like(d.float()).to(torch.bfloat16) for d in dys]
    dxs = E.dwconv_dgrad_multi(dys, w_t, k, zins=xs if sw else None, dxs=[b.clone() for b in base], accumulate=True)
    for x, dy, b, dx in zip(xs, dys, base, dxs):
        one = E.dwconv_dgrad(dy, w_t, tuple(x.shape), k, 1, zin=x if sw else None, dx=b.clone(), accumulate=True)
        if x.shape[2] >= 16 and x.shape[1] >= 8:
            assert torch.equal(dx, one)
        else:
            _close(dx, one.float().cpu(), 2e-2)


def test_maxpool_same_zero_padding():
    from objectdetection.pytorch_amd import ops_effdet as E
    for (H, W) in ((16, 16), (9, 11), (8, 6)):
        x = _bf(_mk((2, 16, H, W), H) - 1.0).requires_grad_(True)    # mostly negative: pads (0) win
        y = R.maxpool_same(x)
        gy = _mk(tuple(y.shape), 3)
        y.backward(gy)
        out, idx = E.maxpool_same_fwd(_nhwc(x.detach()))
        assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), y.detach())
        gin = torch.zeros((2, H, W, 16), dtype=torch.bfloat16, device=DEV)
        E.maxpool_same_bwd(_nhwc(gy), idx, gin, False)
        _close(gin.permute(0, 3, 1, 2), x.grad, 1e-2)


@pytest.mark.parametrize('N,H,W,C,k,s', [(2, 16, 40, 32, 3, 1), (2, 9, 70, 24, 5, 1), (1, 40, 132, 16, 3, 2), (2, 32, 70, 48, 5, 2)])
def test_depthwise_forward_with_se_pooling(N, H, W, C, k, s):
    """mi355_dwconv_fwd_sepool: the same output as mi355_dwconv_fwd, bit for bit, and per-tile partial means of
    swish(out) that add up to mi355_se_pool's."""
    from objectdetection.pytorch_amd import ops_effdet as E
    from objectdetection.pytorch_amd._lib import call, ptr, c_int, stream_of
    x = _nhwc(_mk((N, C, H, W), 1))
    w_t = (torch.randn((k * k, C), generator=torch.Generator().manual_seed(1)) * 0.3).to(DEV)
    bias = (torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.1).to(DEV)
    ref = E.dwconv_fwd(x, True, w_t, bias, k, s)
    pt, P = E.same_pad(H, k, s)
    pl, Q = E.same_pad(W, k, s)
    ns = ctypes.c_int()
    call('mi355_dwconv_fwd_sepool_slabs', c_int(H), c_int(W), c_int(k), c_int(s), c_int(P), c_int(Q), ctypes.byref(ns))
    assert ns.value > 0
    out = torch.full_like(ref, float('nan'))
    part = torch.full((N, ns.value, C), float('nan'), device=DEV)
    call('mi355_dwconv_fwd_sepool', ptr(x), c_int(1), ptr(w_t), ptr(bias), ptr(out), ptr(part), c_int(N), c_int(H), c_int(W),
         c_int(C), c_int(k), c_int(s), c_int(pt), c_int(pl), c_int(P), c_int(Q), stream_of(x))
    assert torch.equal(out, ref)
    pooled = R.swish(ref.float()).mean((1, 2))
    _close(part.sum(1), pooled.cpu(), 2e-3)
    ns2 = ctypes.c_int()
    call('mi355_se_slabs', c_int(N), c_int(P * Q), ctypes.byref(ns2))
    part2 = torch.empty((N, ns2.value, C), device=DEV)
    call('mi355_se_pool', ptr(ref), c_int(1), ptr(part2), c_int(N), c_int(P * Q), c_int(C), stream_of(x))
    _close(part.sum(1), part2.sum(1).cpu(), 1e-5)


@pytest.mark.parametrize('N,H,W,C,k,acc', [(2, 16, 40, 32, 3, False), (2, 9, 70, 24, 5, True)])
def test_se_backward_apply_inside_the_depthwise_data_gradient(N, H, W, C, k, acc):
    """mi355_dwconv_dgrad_seapply == mi355_se_bwd_apply then mi355_dwconv_dgrad, gz included, bit for bit."""
    from objectdetection.pytorch_amd import ops_effdet as E
    from objectdetection.pytorch_amd._lib import call, ptr, c_int, stream_of
    gu, z, zin = _nhwc(_mk((N, C, H, W), 1)), _nhwc(_mk((N, C, H, W), 2)), _nhwc(_mk((N, C, H, W), 3))
    gate = torch.rand((N, C), generator=torch.Generator().manual_seed(4)).to(DEV)
    gpool = (torch.randn((N, C), generator=torch.Generator().manual_seed(5)) * 3.0).to(DEV)
    w_t = (torch.randn((k * k, C), generator=torch.Generator().manual_seed(6)) * 0.3).to(DEV)
    base = _nhwc(_mk((N, C, H, W), 7))
    gz_ref = torch.empty_like(gu)
    call('mi355_se_bwd_apply', ptr(gu), ptr(z), ptr(gate), ptr(gpool), ptr(gz_ref), c_int(N), c_int(H * W), c_int(C), stream_of(gu))
    dx_ref = E.dwconv_dgrad(gz_ref, w_t, tuple(gu.shape), k, 1, zin=zin, dx=base.clone(), accumulate=acc)
    gz = torch.full_like(gu, float('nan'))
    dx = base.clone()
    call('mi355_dwconv_dgrad_seapply', ptr(gu), ptr(z), ptr(gate), ptr(gpool), ptr(gz), ptr(w_t), ptr(dx), ptr(zin), c_int(N),
         c_int(H), c_int(W), c_int(C), c_int(k), c_int(int(acc)), stream_of(gu))
    assert torch.equal(gz, gz_ref)
    assert torch.equal(dx, dx_ref)


def test_squeeze_excite_fwd_bwd():
    from objectdetection.pytorch_amd._lib import call, ptr, c_int, stream_of
    N, C, Cse, H, W = 3, 96, 4, 12, 10
    z = _mk((N, C, H, W), 1).requires_grad_(True)
    w1 = (_mk((Cse, C, 1, 1), 2, 0.2)).float().requires_grad_(True)
    b1 = _mk((Cse,), 3, 0.1).float().requires_grad_(True)
    w2 = (_mk((C, Cse, 1, 1), 4, 0.5)).float().requires_grad_(True)
    b2 = _mk((C,), 5, 0.1).float().requires_grad_(True)
    a = R.swish(z)
    sq = F.adaptive_avg_pool2d(a, 1)
    r = R.swish(F.conv2d(sq, w1, b1))
    gate = torch.sigmoid(F.conv2d(r, w2, b2))
    u = gate * a
    gu = _mk(tuple(u.shape), 6)
    u.backward(gu)
    zh, guh = _nhwc(z.detach()), _nhwc(gu)
    dev = DEV
    ns = ctypes.c_int()
    call('mi355_se_slabs', c_int(N), c_int(H * W), ctypes.byref(ns))
    slabs = ns.value
    part = torch.full((N, slabs, C), float('nan'), device=dev)       # nothing needs zeroing
    pool = torch.empty((N, C), device=dev)
    call('mi355_se_pool', ptr(zh), c_int(1), ptr(part), c_int(N), c_int(H * W), c_int(C), stream_of(zh))
    rpre = torch.empty((N, Cse), device=dev)
    g = torch.empty((N, C), device=dev)
    W1, B1, W2, B2 = [t.detach().to(dev).contiguous() for t in (w1, b1, w2, b2)]
    call('mi355_se_fc', ptr(part), c_int(slabs), ptr(pool), ptr(W1), ptr(B1), ptr(W2), ptr(B2), ptr(rpre), ptr(g),
         c_int(N), c_int(C), c_int(Cse), stream_of(zh))
    _close(pool, sq.detach().reshape(N, C), 2e-3)
    _close(g, gate.detach().reshape(N, C), 2e-3)
    uh = torch.empty_like(zh)
    call('mi355_se_scale', ptr(zh), ptr(g), ptr(uh), c_int(N), c_int(H * W), c_int(C), stream_of(zh))
    _close(uh.permute(0, 3, 1, 2), u.detach(), 1e-2)
    ggate = torch.full((N, slabs, C), float('nan'), device=dev)
    call('mi355_se_bwd_reduce', ptr(guh), ptr(zh), ptr(ggate), c_int(N), c_int(H * W), c_int(C), stream_of(zh))
    gpool = torch.empty((N, C), device=dev)
    dW1, dB1, dW2, dB2 = [torch.zeros_like(t) for t in (W1, B1, W2, B2)]
    call('mi355_se_fc_bwd', ptr(pool), ptr(W1), ptr(W2), ptr(rpre), ptr(g), ptr(ggate), c_int(slabs), ptr(gpool), ptr(dW1),
         ptr(dB1), ptr(dW2), ptr(dB2), c_int(N), c_int(C), c_int(Cse), stream_of(zh))
    gz = torch.empty_like(zh)
    call('mi355_se_bwd_apply', ptr(guh), ptr(zh), ptr(g), ptr(gpool), ptr(gz), c_int(N), c_int(H * W), c_int(C),
         stream_of(zh))
    _close(gz.permute(0, 3, 1, 2), z.grad, 1.5e-2)
    for got, ref in ((dW1, w1.grad), (dB1, b1.grad), (dW2, w2.grad), (dB2, b2.grad)):
        _close(got.reshape(ref.shape), ref, 5e-3)
    # the atomic-free form: per-sample data path + one table launch for the parameter gradients (plain stores
    # into NaN-filled buffers), bit-reproducible
    from objectdetection.pytorch_amd.engine_effdet import SeGradDesc

    def run():
        de = torch.full((N, C), float('nan'), device=dev)
        dr = torch.full((N, Cse), float('nan'), device=dev)
        gp = torch.full((N, C), float('nan'), device=dev)
        call('mi355_se_fc_bwd_data', ptr(W1), ptr(W2), ptr(rpre), ptr(g), ptr(ggate), c_int(slabs), ptr(de), ptr(dr),
             ptr(gp), c_int(N), c_int(C), c_int(Cse), stream_of(zh))
        outs = [torch.full_like(t, float('nan')) for t in (W1, B1, W2, B2)]
        row = SeGradDesc(pool.data_ptr(), rpre.data_ptr(), de.data_ptr(), dr.data_ptr(), outs[0].data_ptr(),
                         outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), N, C, Cse, 0)
        tab = torch.frombuffer(bytearray(bytes(row)), dtype=torch.uint8).clone().to(dev)
        call('mi355_se_param_grads_table', ptr(tab), c_int(1), c_int(2 * C * Cse + C + Cse), stream_of(zh))
        return [gp] + outs
    a1, a2 = run(), run()
    _close(a1[0], gpool.cpu(), 1e-5)
    for got, ref in zip(a1[1:], (w1.grad, b1.grad, w2.grad, b2.grad)):
        _close(got.reshape(ref.shape), ref, 5e-3)
    assert all(torch.equal(x, y) for x, y in zip(a1, a2))


@pytest.mark.parametrize('up1,three,hw', [(True, False, (16, 40)), (False, True, (10, 6)), (True, True, (4, 4))])
def test_bifpn_node_built_inside_the_depthwise_forward(up1, three, hw):
    """mi355_bifpn_node_dwconv_fwd == mi355_bifpn_node_fwd then the depthwise conv of swish(z), z included."""
    from objectdetection.pytorch_amd import ops_effdet as E
    from objectdetection.pytorch_amd._lib import call, ptr, c_int, stream_of
    N, C, (H, W), k = 2, 64, hw, 3
    wv = torch.tensor([0.31, 0.52, 0.17 if three else 0.0, 0.0], device=DEV)
    a = _nhwc(_mk((N, C, H, W), 1))
    b = _nhwc(_mk((N, C, H // 2, W // 2) if up1 else (N, C, H, W), 2))
    c = _nhwc(_mk((N, C, H, W), 3)) if three else None
    w_t = (torch.randn((k * k, C), generator=torch.Generator().manual_seed(1)) * 0.3).to(DEV)
    bias = (torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.1).to(DEV)
    z_ref = torch.empty_like(a)
    call('mi355_bifpn_node_fwd', ptr(a), ptr(b), c_int(int(up1)), ptr(c), ptr(wv), ptr(z_ref), c_int(N), c_int(H), c_int(W),
         c_int(C), stream_of(a))
    t_ref = E.dwconv_fwd_multi([z_ref, z_ref], True, w_t, bias, k)[0]          # (the tile kernel at every size)
    z = torch.full_like(a, float('nan'))
    t = torch.full_like(a, float('nan'))
    call('mi355_bifpn_node_dwconv_fwd', ptr(a), ptr(b), c_int(int(up1)), ptr(c), ptr(wv), ptr(z), ptr(w_t), ptr(bias), ptr(t),
         c_int(N), c_int(H), c_int(W), c_int(C), c_int(k), stream_of(a))
    assert torch.equal(z, z_ref), (z.float() - z_ref.float()).abs().max().item()
    assert torch.equal(t, t_ref), (t.float() - t_ref.float()).abs().max().item()


@pytest.mark.parametrize('up1,three', [(True, True), (True, False), (False, True)])
def test_bifpn_node_backward_in_one_pass(up1, three):
    """mi355_bifpn_node_bwd == fusion-weight sums + one scaled accumulate per input (efficientdet.py:213-265)."""
    from objectdetection.pytorch_amd._lib import call, ptr, c_int, stream_of
    N, C, H, W = 2, 64, 8, 12
    wv = torch.tensor([0.31, 0.52, 0.17 if three else 0.0])
    a = _mk((N, C, H, W), 1).requires_grad_(True)
    b = _mk((N, C, H // 2, W // 2) if up1 else (N, C, H, W), 2).requires_grad_(True)
    c = _mk((N, C, H, W), 3).requires_grad_(True)
    wp = wv.clone().requires_grad_(True)
    z = wp[0] * a + wp[1] * (F.interpolate(b, scale_factor=2, mode='nearest') if up1 else b)
    if three:
        z = z + wp[2] * c
    gz = _mk(tuple(z.shape), 4)
    z.backward(gz)
    ah, bh, ch, gh = _nhwc(a.detach()), _nhwc(b.detach()), _nhwc(c.detach()), _nhwc(gz)
    old = [_nhwc(_mk(tuple(t.shape), 10 + i)) for i, t in enumerate((a, b, c))]
    for acc in (0, 1):
        g = [o.clone() for o in old]
        npt = ctypes.c_int()
        call('mi355_bifpn_node_bwd_parts', c_int(N), c_int(H), c_int(W), c_int(C), c_int(int(up1)), ctypes.byref(npt))
        gw = torch.zeros((npt.value, 4), device=DEV)
        call('mi355_bifpn_node_bwd', ptr(gh), ptr(ah), ptr(bh), c_int(int(up1)), ptr(ch if three else None),
             ptr(wv.to(DEV)), ptr(gw), ptr(g[0]), c_int(acc), ptr(g[1]), c_int(acc), ptr(g[2] if three else None),
             c_int(acc), c_int(N), c_int(H), c_int(W), c_int(C), stream_of(ah))
        _close(gw.sum(0)[:3], wp.grad, 5e-3)
        for got, o, ref in zip(g if three else g[:2], old, (a.grad, b.grad, c.grad)):
            _close(got.permute(0, 3, 1, 2), ref + (o.float().cpu().permute(0, 3, 1, 2) if acc else 0.0), 1e-2)


def test_bifpn_node_and_fusion_weights():
    from objectdetection.pytorch_amd._lib import call, ptr, c_int, c_f32, c_i64, stream_of
    N, C, H, W = 2, 64, 8, 12
    p = torch.tensor([0.7, -0.2, 1.3], requires_grad=True)
    a = _mk((N, C, H, W), 1).requires_grad_(True)
    b = _mk((N, C, H // 2, W // 2), 2).requires_grad_(True)
    c = _mk((N, C, H, W), 3).requires_grad_(True)
    w = R.fusion_weights(p)
    z = w[0] * a + w[1] * F.interpolate(b, scale_factor=2, mode='nearest') + w[2] * c
    gz = _mk(tuple(z.shape), 4)
    z.backward(gz)
    pd = p.detach().to(DEV)
    wd = torch.empty(3, device=DEV)
    call('mi355_fusion_weights_fwd', ptr(pd), c_int(3), c_f32(1e-4), ptr(wd), stream_of(pd))
    _close(wd, w.detach(), 1e-6)
    ah, bh, ch, gh = _nhwc(a.detach()), _nhwc(b.detach()), _nhwc(c.detach()), _nhwc(gz)
    zh = torch.empty_like(ah)
    call('mi355_bifpn_node_fwd', ptr(ah), ptr(bh), c_int(1), ptr(ch), ptr(wd), ptr(zh), c_int(N), c_int(H), c_int(W),
         c_int(C), stream_of(ah))
    _close(zh.permute(0, 3, 1, 2), z.detach(), 1e-2)
    gw = torch.zeros(3, device=DEV)
    call('mi355_bifpn_node_wgrad', ptr(gh), ptr(ah), ptr(bh), c_int(1), ptr(ch), ptr(gw), c_int(N), c_int(H), c_int(W),
         c_int(C), stream_of(ah))
    gp = torch.empty(3, device=DEV)
    call('mi355_fusion_weights_bwd', ptr(pd), c_int(3), c_f32(1e-4), ptr(gw), ptr(gp), stream_of(pd))
    _close(gp, p.grad, 5e-3)
    ga = torch.empty_like(ah)
    call('mi355_axpy_bf16', ptr(gh), ptr(wd), c_int(0), ptr(ga), c_i64(gh.numel()), c_int(0), stream_of(ah))
    _close(ga.permute(0, 3, 1, 2), a.grad, 1e-2)
    gb = torch.zeros_like(bh)
    call('mi355_upsample_bwd_axpy', ptr(gh), ptr(wd), c_int(1), ptr(gb), c_int(N), c_int(H), c_int(W), c_int(C),
         c_int(0), stream_of(ah))
    _close(gb.permute(0, 3, 1, 2), b.grad, 1e-2)
