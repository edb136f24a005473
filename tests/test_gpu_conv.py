"""GPU parity of the MFMA conv engine (through the C ABI) against torch fp32 CPU convs of the
same bf16-rounded operands.  fp32 accumulation of exact bf16 products: an fp32-output conv must
agree to ~1e-3 relative (summation order only); bf16 outputs add one rounding (2^-9 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


def _bf(t):
    return t.to(torch.bfloat16).float()


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return _bf(torch.randn(shape, generator=g) * scale)


def _close(got, ref, tol):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs().max().item()
    mag = ref.abs().max().item() + 1e-12
    assert err <= tol * mag, 'max err %.4g vs magnitude %.4g (tol %.3g)' % (err, mag, tol)


def _shadow(w, dev, cpad=None, scale=None):
    """fp32 OIHW master (channels_last on device) -> bf16 shadows via the HIP prep kernel."""
    from objectdetection.pytorch_amd import ops_conv
    K, C, R, S = w.shape
    cpad = C if cpad is None else cpad
    kpad = (K + 7) // 8 * 8
    wd = w.to(dev).contiguous(memory_format=torch.channels_last)
    w_fwd = torch.zeros((K, R * S, cpad), dtype=torch.bfloat16, device=dev)
    w_dgr = torch.zeros((C, R * S, kpad), dtype=torch.bfloat16, device=dev)
    ops_conv.prep_weight(wd, cpad, scale, w_fwd, w_dgr)
    return wd, w_fwd, w_dgr


CASES = [  # N, H, W, C, K, R, stride, pad
    (2, 20, 24, 64, 128, 3, 1, 1),
    (2, 17, 23, 128, 256, 1, 1, 0),
    (2, 18, 22, 128, 256, 1, 2, 0),
    (2, 19, 21, 64, 64, 3, 2, 1),
    (1, 33, 31, 8, 64, 7, 2, 3),        # stem shape class (flat reduction, C = 8)
    (2, 9, 11, 40, 24, 1, 1, 0),        # C, K not multiples of 64/128 (EfficientNet-like)
    (1, 8, 8, 256, 720, 3, 1, 1),       # cls output conv
    (3, 5, 7, 256, 36, 3, 1, 1),        # reg output conv (K % 8 != 0)
    (1, 4, 4, 2048, 256, 3, 2, 1),      # P6
    (4, 128, 128, 128, 256, 3, 1, 1),   # >= 1024 tiles of 128x128, long reduction: the wide one-stage kernel
    (6, 128, 128, 64, 64, 3, 1, 1),     # >= 768 tiles of 128x64: the one-stage form of the narrow row-of-taps kernel
]


@pytest.mark.parametrize('case', CASES)
def test_conv_fwd(dev, case):
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    x = _mk((N, C, H, W), 1)
    w = _mk((K, C, R, R), 2, scale=(2.0 / (C * R * R)) ** 0.5)
    b = _mk((K,), 3).float()
    ref = F.conv2d(x, w, b, stride, pad)
    _, w_fwd, _ = _shadow(w, dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    out = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, bias=b.to(dev), out_f32=True)
    _close(out.permute(0, 3, 1, 2), ref, 2e-3)
    out = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, bias=b.to(dev), act=ops_conv.ACT_RELU)
    _close(out.permute(0, 3, 1, 2), F.relu(ref), 1e-2)
    out = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, bias=b.to(dev), act=ops_conv.ACT_SIGMOID,
                            out_f32=True)
    _close(out.permute(0, 3, 1, 2), torch.sigmoid(ref), 2e-3)


# N, H, W, C, K, R, stride, pad: narrow generic (1x1), ring (256 - 767 tiles), narrow row-of-taps, wide row-of-taps (>= 1024
# tiles of 128 x 128), a ragged M (rows past M in the last tile) and a stride-2 conv
STATS_CASES = [(2, 40, 40, 256, 64, 1, 1, 0), (16, 32, 32, 1024, 256, 1, 1, 0), (4, 32, 32, 128, 128, 3, 1, 1),
               (16, 64, 64, 256, 256, 3, 1, 1), (3, 19, 23, 64, 192, 3, 1, 1), (2, 33, 31, 128, 256, 3, 2, 1)]


@pytest.mark.parametrize('case', STATS_CASES)
def test_conv_fwd_epilogue_statistics(dev, case):
    """Mi355ConvLaunch.stats: the launch leaves one [sum | square-sum] row per 128-pixel tile of its STORED output (the
    batch statistics of a train-mode BatchNorm behind the conv, utils/sync_batchnorm/batchnorm.py:63-93); the rows summed
    over the tiles must equal the column sums of the bf16 output tensor the launch wrote (fp32 summation order only), and
    the output itself must not change."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    x = _mk((N, C, H, W), 1)
    w = _mk((K, C, R, R), 2, scale=(2.0 / (C * R * R)) ** 0.5)
    _, w_fwd, _ = _shadow(w, dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    plain = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad)
    M = plain.numel() // K
    ntile = -(-M // 128)
    st = torch.full((ntile, 2, K), float('nan'), device=dev)
    out = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, stats=st)
    assert torch.equal(out, plain)
    z = out.double().reshape(M, K)
    got = st.double().sum(0)
    assert torch.isfinite(st).all()
    np.testing.assert_allclose(got[0].cpu().numpy(), z.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * float(z.abs().mean()) * M ** 0.5)
    np.testing.assert_allclose(got[1].cpu().numpy(), (z * z).sum(0).cpu().numpy(), rtol=1e-5)
    # the rows are per tile: tile t covers output rows [128 t, 128 t + 128)
    t = ntile // 2
    rows = z[128 * t:128 * t + 128]
    np.testing.assert_allclose(st[t, 1].double().cpu().numpy(), (rows * rows).sum(0).cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize('case', [(2, 128, 128, 16, 96), (1, 96, 100, 24, 144), (2, 33, 47, 96, 16), (1, 40, 40, 144, 40),
                                  (1, 31, 29, 240, 40), (3, 17, 19, 40, 240), (1, 64, 64, 200, 24), (1, 30, 30, 256, 72),
                                  (2, 20, 20, 112, 8), (1, 25, 25, 176, 48)])
@pytest.mark.parametrize('mode', ['2', '1'])
def test_pointwise_stream_kernel(dev, case, mode, monkeypatch):
    """conv_pw_stream_kernel (1x1 / stride 1, C <= 256, weights in LDS: EfficientNet's expand / project convs and their
    data gradients): every reduction width 1..8 x 32 channels, partial 32-channel output groups, ragged pixel tails,
    bias, residual addend, the in-place accumulating form of a data gradient, mask before / after the addend, against
    torch fp32 (the tiled kernels add in another order, so both forms are held to the fp32 reference, not to each
    other).  mode 2 forces the kernel wherever it applies, mode 1 is the product's choice."""
    from objectdetection.pytorch_amd import ops_conv
    monkeypatch.setenv('MI355_PW_STREAM', mode)
    N, H, W, C, K = case
    x = _mk((N, C, H, W), 21)
    w = _mk((K, C, 1, 1), 22, scale=(2.0 / C) ** 0.5)
    b = _mk((K,), 23).float()
    ref = F.conv2d(x, w, b)
    _, w_fwd, _ = _shadow(w, dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    out = ops_conv.conv_fwd(xh, w_fwd, K, 1, 1, 1, 0, bias=b.to(dev))
    _close(out.permute(0, 3, 1, 2), ref, 1e-2)
    out = ops_conv.conv_fwd(xh, w_fwd, K, 1, 1, 1, 0, bias=b.to(dev), act=ops_conv.ACT_RELU)
    _close(out.permute(0, 3, 1, 2), F.relu(ref), 1e-2)
    add = _mk((N, K, H, W), 24)
    addh = add.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    out = ops_conv.conv_fwd(xh, w_fwd, K, 1, 1, 1, 0, bias=b.to(dev), addend=addh)
    _close(out.permute(0, 3, 1, 2), ref + _bf(add), 1e-2)
    # in place: out = conv + out (the accumulating data gradient of a tensor with two consumers)
    acc = addh.clone()
    pr = ops_conv.fwd_problem(xh, acc.data_ptr(), 1, 1, 1, 0, K=K, addend=acc)
    ops_conv.launch_gather([pr], w_fwd, K, 1, xh, bias=b.to(dev))
    _close(acc.permute(0, 3, 1, 2), ref + _bf(add), 1e-2)
    # ReLU-backward style mask, before and after the addend
    mk = _mk((N, K, H, W), 25)
    mkh = mk.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    on = (_bf(mk) > 0).float()
    for flags, want in ((0, (ref + _bf(add)) * on), (1, ref * on + _bf(add))):
        o2 = torch.empty_like(addh)
        pr = ops_conv.fwd_problem(xh, o2.data_ptr(), 1, 1, 1, 0, K=K, addend=addh, mask=mkh)
        ops_conv.launch_gather([pr], w_fwd, K, 1, xh, bias=b.to(dev), flags=flags)
        _close(o2.permute(0, 3, 1, 2), want, 1e-2)


@pytest.mark.parametrize('case', [(2, 37, 41, 16, 64, 4, 4, 2, 2), (1, 64, 48, 16, 24, 4, 4, 2, 2), (2, 19, 23, 16, 40, 2, 2, 1, 0),
                                  (1, 30, 33, 32, 72, 3, 1, 1, 0), (3, 9, 11, 8, 16, 2, 4, 0, 2), (1, 21, 17, 16, 64, 3, 2, 1, 1)])
def test_window_stream_kernel(dev, case, monkeypatch):
    """The window form of conv_pw_stream_kernel: stride-1 convs over few channels whose tap row is a contiguous
    TS * C window of the NHWC input (the space-to-depth stem: 4 x 4 taps over 16 channels, pad 2 before / 1 after),
    forced with MI355_PW_STREAM=2, against torch on the explicitly padded input."""
    from objectdetection.pytorch_amd import ops_conv
    monkeypatch.setenv('MI355_PW_STREAM', '2')
    N, H, W, C, K, R, S, pt, pl = case
    x = _mk((N, C, H, W), 31)
    w = _mk((K, C, R, S), 32, scale=(2.0 / (C * R * S)) ** 0.5)
    b = _mk((K,), 33).float()
    ref = F.relu(F.conv2d(F.pad(x, (pl, S - 1 - pl, pt, R - 1 - pt)), w, b))
    wd = w.to(dev).contiguous(memory_format=torch.channels_last)
    w_fwd = torch.zeros((K, R * S, C), dtype=torch.bfloat16, device=dev)
    w_dgr = torch.zeros((C, R * S, (K + 7) // 8 * 8), dtype=torch.bfloat16, device=dev)
    ops_conv.prep_weight(wd, C, None, w_fwd, w_dgr)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    out = torch.empty((N, H, W, K), dtype=torch.bfloat16, device=dev)
    pr = ops_conv.fwd_problem(xh, out.data_ptr(), R, S, 1, (pt, pl), K=K, out_size=(H, W))
    ops_conv.launch_gather([pr], w_fwd, K, R * S, xh, bias=b.to(dev), act=ops_conv.ACT_RELU)
    _close(out.permute(0, 3, 1, 2), ref, 1e-2)
    monkeypatch.setenv('MI355_PW_STREAM', '0')          # the tiled kernels on the same launch: same result up to bf16
    out0 = torch.empty_like(out)
    pr = ops_conv.fwd_problem(xh, out0.data_ptr(), R, S, 1, (pt, pl), K=K, out_size=(H, W))
    ops_conv.launch_gather([pr], w_fwd, K, R * S, xh, bias=b.to(dev), act=ops_conv.ACT_RELU)
    _close(out0.permute(0, 3, 1, 2), ref, 1e-2)


@pytest.mark.parametrize('case,ksplit', [((1, 4, 4, 2048, 256, 3, 2, 1), 16), ((2, 9, 7, 512, 192, 3, 1, 1), 3),
                                         ((2, 6, 5, 1024, 36, 1, 1, 0), 4)])
def test_conv_fwd_splitk(dev, case, ksplit):
    """Few output tiles, long reduction (P6, the deepest stage): reduction split over blocks with
    per-split slabs summed in order by the finish pass == the unsplit launch (fp32 summation order
    aside) and bit-reproducible run to run."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    x = _mk((N, C, H, W), 1)
    w = _mk((K, C, R, R), 2, scale=(2.0 / (C * R * R)) ** 0.5)
    b = _mk((K,), 3).float()
    res = _mk((N, K, (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1), 4)
    ref = F.relu(F.conv2d(x, w, b, stride, pad) + res)
    _, w_fwd, _ = _shadow(w, dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    rh = res.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    outs = [ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, bias=b.to(dev), act=ops_conv.ACT_RELU, addend=rh,
                              out_f32=True, ksplit=ksplit) for _ in range(2)]
    _close(outs[0].permute(0, 3, 1, 2), ref, 2e-3)
    assert torch.equal(outs[0], outs[1])
    unsplit = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, bias=b.to(dev), act=ops_conv.ACT_RELU, addend=rh,
                                out_f32=True)
    _close(outs[0], unsplit.cpu(), 2e-6)               # same products, fp32 summation order only
    out16 = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, bias=b.to(dev), act=ops_conv.ACT_RELU, addend=rh,
                              ksplit=ksplit)
    _close(out16.permute(0, 3, 1, 2), ref, 1e-2)


def test_conv_fwd_bn_fold_addend_relu(dev):
    """conv -> frozen BN -> (+ residual) -> ReLU, utils_resnet.py:60-80."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K = 2, 14, 14, 64, 256
    x, w = _mk((N, C, H, W), 1), _mk((K, C, 1, 1), 2, 0.2)
    res = _mk((N, K, H, W), 3)
    bn = torch.nn.BatchNorm2d(K).eval()
    g = torch.Generator().manual_seed(4)
    bn.weight.data.uniform_(0.5, 1.5, generator=g)
    bn.bias.data.normal_(0, 0.1, generator=g)
    bn.running_mean.normal_(0, 0.1, generator=g)
    bn.running_var.uniform_(0.5, 1.5, generator=g)
    ref = F.relu(bn(F.conv2d(x, w)) + res)
    bnd = torch.nn.BatchNorm2d(K).to(dev).eval()
    bnd.load_state_dict(bn.state_dict())
    scale = torch.empty(K, device=dev)
    shift = torch.empty(K, device=dev)
    ops_conv.bn_fold(bnd, scale, shift)
    _, w_fwd, _ = _shadow(w, dev, scale=scale)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    rh = res.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    out = ops_conv.conv_fwd(xh, w_fwd, K, 1, 1, 1, 0, bias=shift, act=ops_conv.ACT_RELU, addend=rh)
    _close(out.permute(0, 3, 1, 2), ref, 1.5e-2)


def test_conv_fwd_upsample_addend(dev):
    """lateral 1x1 + nearest-2x upsampled coarser map, retinanet.py:55-56."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K = 2, 10, 14, 128, 256
    x, w = _mk((N, C, H, W), 1), _mk((K, C, 1, 1), 2, 0.1)
    coarse = _mk((N, K, H // 2, W // 2), 3)
    ref = F.conv2d(x, w) + F.interpolate(coarse, scale_factor=2, mode='nearest')
    _, w_fwd, _ = _shadow(w, dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    ch = coarse.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    out = ops_conv.conv_fwd(xh, w_fwd, K, 1, 1, 1, 0, addend=ch, add_shift=1)
    _close(out.permute(0, 3, 1, 2), ref, 1e-2)


@pytest.mark.parametrize('case', [c for c in CASES if c[3] != 8])
def test_conv_dgrad(dev, case):
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    w = _mk((K, C, R, R), 2, scale=(2.0 / (K * R * R)) ** 0.5)
    P, Q = ops_conv.out_hw(H, W, R, stride, pad)
    dy = _mk((N, K, P, Q), 5)
    ref = torch.nn.grad.conv2d_input((N, C, H, W), w, dy, stride, pad)
    _, _, w_dgr = _shadow(w, dev)
    kpad = w_dgr.shape[2]
    dyh = torch.zeros((N, P, Q, kpad), dtype=torch.bfloat16, device=dev)
    dyh[..., :K] = dy.permute(0, 2, 3, 1).to(dev).to(torch.bfloat16)
    dx = ops_conv.conv_dgrad(dyh, w_dgr, H, W, C, R, R, stride, pad)
    _close(dx.permute(0, 3, 1, 2), ref, 1e-2)
    # fused residual-grad accumulate + ReLU mask (Bottleneck backward, utils_resnet.py:77-78)
    add = _mk((N, C, H, W), 6)
    msk = _mk((N, C, H, W), 7)
    ref2 = (ref + add) * (msk > 0).float()
    dx = ops_conv.conv_dgrad(dyh, w_dgr, H, W, C, R, R, stride, pad,
                             addend=add.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16),
                             mask=msk.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16))
    _close(dx.permute(0, 3, 1, 2), ref2, 1e-2)


@pytest.mark.parametrize('case', CASES)
def test_conv_wgrad(dev, case):
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    x = _mk((N, C, H, W), 1)
    P, Q = ops_conv.out_hw(H, W, R, stride, pad)
    dy = _mk((N, K, P, Q), 5)
    ref = torch.nn.grad.conv2d_weight(x, (K, C, R, R), dy, stride, pad)      # [K,C,R,S]
    kpad = (K + 7) // 8 * 8
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    dyh = torch.zeros((N, P, Q, kpad), dtype=torch.bfloat16, device=dev)
    dyh[..., :K] = dy.permute(0, 2, 3, 1).to(dev).to(torch.bfloat16)
    for split in (None, 64):
        cs = torch.zeros(kpad, device=dev)
        G = ops_conv.conv_wgrad(xh, dyh, R, R, stride, pad, split_rows=split, colsum=cs)   # [Kpad][T][C]
        got = G[:K].reshape(K, R, R, C).permute(0, 3, 1, 2)
        _close(got, ref, 2e-3)
        _close(cs[:K], dy.sum((0, 2, 3)), 1e-3)           # fused bias / BN-beta gradient
        if kpad > K:
            assert float(G[K:].abs().max()) == 0.0


@pytest.mark.parametrize('case', [(3, 5, 16, 64, 64, 3, 1, 1), (1, 3, 64, 128, 136, 3, 1, 1), (2, 32, 32, 64, 256, 3, 1, 1),
                                  (5, 4, 4, 256, 40, 3, 1, 1), (2, 7, 8, 192, 128, 3, 1, 1),
                                  # widths that do not divide the 64-pixel stage (640 x 640 inputs: 80 / 40 / 20 / 10 / 5)
                                  # or exceed it: the table-driven edge form of the kernel
                                  # (taken for K >= 128 and C <= 256, or a power-of-two width; the others: generic kernel)
                                  (2, 9, 20, 64, 136, 3, 1, 1), (1, 6, 80, 128, 128, 3, 1, 1), (3, 5, 5, 64, 128, 3, 1, 1),
                                  (2, 11, 10, 128, 256, 3, 1, 1), (1, 3, 128, 64, 64, 3, 1, 1), (2, 4, 1, 64, 128, 3, 1, 1),
                                  (1, 13, 40, 256, 192, 3, 1, 1), (1, 2, 160, 64, 64, 3, 1, 1), (2, 9, 20, 64, 72, 3, 1, 1)])
def test_conv_wgrad_row_of_taps(dev, case):
    """3x3 / stride 1 / same: the kernel that stages dy and the activation rows once per kernel row
    (conv_wgrad_row3_kernel), images that are not multiples of the 64-pixel stage, several images per stage,
    chunk splits, and the fp32-atomic as well as the slab output."""
    test_conv_wgrad(dev, case)
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    x, dy = _mk((N, C, H, W), 11), _mk((N, K, H, W), 12)
    ref = torch.nn.grad.conv2d_weight(x, (K, C, R, R), dy, stride, pad)
    kpad = (K + 7) // 8 * 8
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    dyh = torch.zeros((N, H, W, kpad), dtype=torch.bfloat16, device=dev)
    dyh[..., :K] = dy.permute(0, 2, 3, 1).to(dev).to(torch.bfloat16)
    G = ops_conv.conv_wgrad(xh, dyh, R, R, stride, pad, split_rows=128)
    _close(G[:K].reshape(K, R, R, C).permute(0, 3, 1, 2), ref, 2e-3)


@pytest.mark.parametrize('case,split', [((2, 32, 32, 128, 256, 3, 1, 1), 128), ((3, 16, 16, 256, 128, 3, 1, 1), 64),
                                        ((5, 4, 4, 256, 136, 3, 1, 1), 64), ((1, 8, 64, 128, 128, 3, 1, 1), 192),
                                        ((2, 2, 2, 128, 128, 3, 1, 1), 64), ((3, 1, 1, 256, 256, 3, 1, 1), 64),
                                        # long chunks: the pixel-row ring of the kernel is refilled every 8 stages
                                        ((4, 64, 64, 128, 128, 3, 1, 1), 2048), ((2, 64, 32, 128, 384, 3, 1, 1), 1344)])
def test_conv_wgrad_ring_form(dev, case, split):
    """conv_wgrad_row3x_kernel (3x3 / stride 1 / same, C % 128 == 0, K >= 128, widths dividing the 64-pixel stage): the
    eight-wave block with the three-stage LDS-DMA ring.  Against torch's weight gradient, fp32-atomic and slab output
    (the slab form bit-identical between two launches), chunk lengths from one stage to 32 stages."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    x, dy = _mk((N, C, H, W), 21), _mk((N, K, H, W), 22)
    ref = torch.nn.grad.conv2d_weight(x, (K, C, R, R), dy, stride, pad)
    kpad = (K + 7) // 8 * 8
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    dyh = torch.zeros((N, H, W, kpad), dtype=torch.bfloat16, device=dev)
    dyh[..., :K] = dy.permute(0, 2, 3, 1).to(dev).to(torch.bfloat16)
    pr = ops_conv.fwd_problem(xh, dyh.data_ptr(), R, R, stride, pad, K=kpad)
    assert ops_conv.wgrad_row3x_ok([pr], C, kpad)
    cs = torch.zeros(kpad, device=dev)
    G = ops_conv.conv_wgrad(xh, dyh, R, R, stride, pad, split_rows=split, colsum=cs)
    _close(G[:K].reshape(K, R, R, C).permute(0, 3, 1, 2), ref, 2e-3)
    _close(cs[:K], dy.sum((0, 2, 3)), 1e-3)
    if kpad > K:
        assert float(G[K:].abs().max()) == 0.0
    # slab form (what the plans run): per-chunk partial tiles, summed in index order
    from objectdetection.pytorch_amd._lib import c_i64, c_int, call, ptr, stream_of
    import ctypes
    L = ops_conv.ConvLaunch()
    L.prob[0] = pr
    L.nprob, L.K, L.T = 1, kpad, 9
    nsl = -(-(N * H * W) // split)
    slab = kpad * 9 * C
    outs = []
    for _ in range(2):
        ws = torch.full((nsl, slab), float('nan'), device=dev)
        cws = torch.zeros((nsl, kpad), device=dev)
        call('mi355_conv_wgrad_slabs', ctypes.byref(L), ptr(ws), c_i64(slab), c_int(nsl), ptr(cws), c_int(split), stream_of(xh))
        outs.append((ws.sum(0), cws.sum(0)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    _close(outs[0][0][:K * 9 * C].reshape(K, R, R, C).permute(0, 3, 1, 2), ref, 2e-3)
    _close(outs[0][1][:K], dy.sum((0, 2, 3)), 1e-3)


@pytest.mark.parametrize('sizes', [[(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)], [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)],
                                   [(8, 32), (12, 4), (3, 64)]], ids=['to-1x1', 'to-4x4', 'ragged'])
def test_conv_wgrad_ring_form_grouped_levels(dev, sizes):
    """Pyramid levels of a shared-weight head conv in ONE ring-form launch (chunks sorted by length)."""
    from objectdetection.pytorch_amd import ops_conv
    N, C, K = 2, 256, 256
    xs = [_mk((N, C, h, w), 30 + i) for i, (h, w) in enumerate(sizes)]
    dys = [_mk((N, K, h, w), 40 + i) for i, (h, w) in enumerate(sizes)]
    keep, probs = [], []
    for x, dy in zip(xs, dys):
        xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
        dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
        keep += [xh, dyh]
        probs.append(ops_conv.fwd_problem(xh, dyh.data_ptr(), 3, 3, 1, 1, K=K))
    assert ops_conv.wgrad_row3x_ok(probs, C, K)
    G = torch.zeros((K, 9, C), dtype=torch.float32, device=dev)
    ops_conv.launch_wgrad(probs, K, 9, G, 128, keep[0])
    ref = sum(torch.nn.grad.conv2d_weight(x, (K, C, 3, 3), dy, 1, 1) for x, dy in zip(xs, dys))
    _close(G.reshape(K, 3, 3, C).permute(0, 3, 1, 2), ref, 2e-3)


def test_grouped_heads_shared_weights(dev):
    """Five pyramid levels, one launch, outputs concatenated along A (retinanet.py:285-287)."""
    from objectdetection.pytorch_amd import ops_conv
    N, C, K = 2, 256, 720
    sizes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    w = _mk((K, C, 3, 3), 2, scale=(2.0 / (C * 9)) ** 0.5)
    b = _mk((K,), 3).float()
    _, w_fwd, _ = _shadow(w, dev)
    xs = [_mk((N, C, h, ww), 10 + i) for i, (h, ww) in enumerate(sizes)]
    A = sum(h * ww for h, ww in sizes) * 9
    out = torch.empty((N, A, 80), dtype=torch.float32, device=dev)
    probs, off, keep = [], 0, []
    for x, (h, ww) in zip(xs, sizes):
        xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
        keep.append(xh)
        probs.append(ops_conv.fwd_problem(xh, out.data_ptr() + 4 * off * 80, 3, 3, 1, 1,
                                          out_strides=(A * 80, ww * K, K), K=K))
        off += h * ww * 9
    ops_conv.launch_gather(probs, w_fwd, K, 9, out, bias=b.to(dev), act=ops_conv.ACT_SIGMOID, out_f32=True)
    ref = torch.cat([torch.sigmoid(F.conv2d(x, w, b, 1, 1)).permute(0, 2, 3, 1).reshape(N, -1, 80)
                     for x in xs], dim=1)
    _close(out, ref, 2e-3)
    # grouped wgrad over the same five problems
    dys = [_mk((N, K, h, ww), 20 + i) for i, (h, ww) in enumerate(sizes)]
    G = torch.zeros((K, 9, C), dtype=torch.float32, device=dev)
    wprobs, keep2 = [], []
    for xh, dy, (h, ww) in zip(keep, dys, sizes):
        dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
        keep2.append(dyh)
        wprobs.append(ops_conv.fwd_problem(xh, dyh.data_ptr(), 3, 3, 1, 1, K=K))
    ops_conv.launch_wgrad(wprobs, K, 9, G, 128, out)
    refw = sum(torch.nn.grad.conv2d_weight(x, (K, C, 3, 3), dy, 1, 1) for x, dy in zip(xs, dys))
    _close(G.reshape(K, 3, 3, C).permute(0, 3, 1, 2), refw, 2e-3)


@pytest.mark.parametrize('ksplit', [1, 3])
def test_grouped_launch_with_per_problem_weights(dev, ksplit):
    """Mi355ConvProblem.weight / .bias: the pyramid levels of a conv that folds its own BatchNorm per level
    (efficientdet.py:335-353, simple_head.py:148-153) in ONE launch == one launch per level, bit for bit; with and
    without split-K (whose finish pass applies the problem's bias)."""
    from objectdetection.pytorch_amd import ops_conv
    N, C, K = 2, 128, 128
    sizes = [(16, 20), (8, 10), (4, 5)]
    xs, ws, bs, singles, probs, outs = [], [], [], [], [], []
    for i, (h, ww) in enumerate(sizes):
        xh = _mk((N, C, h, ww), 10 + i).permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
        w = _mk((K, C, 3, 3), 20 + i, scale=(2.0 / (C * 9)) ** 0.5)
        _, w_fwd, _ = _shadow(w, dev)
        b = _mk((K,), 30 + i).float().to(dev)
        one = torch.empty((N, h, ww, K), dtype=torch.bfloat16, device=dev)
        ops_conv.launch_gather([ops_conv.fwd_problem(xh, one.data_ptr(), 3, 3, 1, 1, K=K)], w_fwd, K, 9, one, bias=b,
                               act=ops_conv.ACT_RELU, ksplit=ksplit)
        out = torch.full((N, h, ww, K), float('nan'), dtype=torch.bfloat16, device=dev)
        pr = ops_conv.fwd_problem(xh, out.data_ptr(), 3, 3, 1, 1, K=K)
        pr.weight, pr.bias = w_fwd.data_ptr(), b.data_ptr()
        xs.append(xh); ws.append(w_fwd); bs.append(b); singles.append(one); probs.append(pr); outs.append(out)
    ops_conv.launch_gather(probs, ws[0], K, 9, outs[0], bias=None, act=ops_conv.ACT_RELU, ksplit=ksplit)
    for one, out in zip(singles, outs):
        assert torch.equal(one, out)
    assert not torch.equal(outs[0][:, :4, :5], outs[2])          # (the levels really used different weights)


def test_space_to_depth_stem(dev):
    """The 7x7 / stride-2 / pad-3 stem as a 4x4 / stride-1 conv over the 2x2 space-to-depth image (conv_aux.hip): the packed
    weights reproduce F.conv2d, and the gradient of the packed weights scatters back onto the 7x7 weight (every element
    exactly once, the taps outside the 7x7 window dropped)."""
    from objectdetection.pytorch_amd import ops_conv
    from objectdetection.pytorch_amd._lib import call, ptr, c_int, c_i64, stream_of
    N, H, W, K = 2, 32, 40, 64
    x = _mk((N, 3, H, W), 1)
    w = _mk((K, 3, 7, 7), 2, scale=(2.0 / 147) ** 0.5)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last)
    sk, sc, sr, ss = wd.stride()
    wp = torch.full((K, 12, 4, 4), float('nan'), device=dev)
    call('mi355_stem_s2d_pack', ptr(wd), c_i64(sk), c_i64(sc), c_i64(sr), c_i64(ss), c_int(K), ptr(wp), stream_of(wd))
    ref_wp = torch.zeros((K, 12, 4, 4))
    for di in range(2):
        for dj in range(2):
            for a in range(4):
                for b in range(4):
                    r, s2 = 2 * a + di - 1, 2 * b + dj - 1
                    if 0 <= r < 7 and 0 <= s2 < 7:
                        ref_wp[:, (di * 2 + dj) * 3:(di * 2 + dj) * 3 + 3, a, b] = w[:, :, r, s2]
    assert torch.equal(wp.cpu(), ref_wp)
    xs = torch.empty((N, H // 2, W // 2, 16), dtype=torch.bfloat16, device=dev)
    call('mi355_nchw_to_nhwc_s2d_bf16', ptr(x.to(dev)), ptr(xs), c_int(N), c_int(H), c_int(W), stream_of(xs))
    ref_xs = x.reshape(N, 3, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(N, H // 2, W // 2, 12)
    assert torch.equal(xs[..., :12].float().cpu(), ref_xs.to(torch.bfloat16).float()) and float(xs[..., 12:].abs().max()) == 0.0
    w_fwd = torch.zeros((K, 16, 16), dtype=torch.bfloat16, device=dev)
    ops_conv.prep_weight(wp, 16, None, w_fwd, None)
    out = torch.empty((N, H // 2, W // 2, K), dtype=torch.bfloat16, device=dev)
    pr = ops_conv.fwd_problem(xs, out.data_ptr(), 4, 4, 1, (2, 2), K=K, out_size=(H // 2, W // 2))
    ops_conv.launch_gather([pr], w_fwd, K, 16, xs)
    _close(out.permute(0, 3, 1, 2), F.conv2d(x, w, None, 2, 3), 2e-2)
    # gradient scatter (+=), on top of an existing gradient
    dwp = _mk((K, 12, 4, 4), 3).float().to(dev)
    base = _mk((K, 3, 7, 7), 4).float()
    dw = base.to(dev).contiguous(memory_format=torch.channels_last)
    call('mi355_stem_s2d_unpack_grad', ptr(dwp), ptr(dw), c_i64(sk), c_i64(sc), c_i64(sr), c_i64(ss), c_int(K), stream_of(dw))
    ref_dw = base.clone()
    for r in range(7):
        for s2 in range(7):
            a, di, b, dj = (r + 1) >> 1, (r + 1) & 1, (s2 + 1) >> 1, (s2 + 1) & 1
            ref_dw[:, :, r, s2] += dwp.cpu()[:, (di * 2 + dj) * 3:(di * 2 + dj) * 3 + 3, a, b]
    assert torch.equal(dw.cpu(), ref_dw)


def test_param_grads_with_frozen_bn(dev):
    """dW, dgamma, dbeta of conv -> BN(eval) from the raw wgrad + column sums."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K = 2, 9, 9, 64, 64
    x = _mk((N, C, H, W), 1)
    w = _mk((K, C, 3, 3), 2, 0.05).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(K).eval()
    g = torch.Generator().manual_seed(4)
    bn.weight.data.uniform_(0.5, 1.5, generator=g)
    bn.bias.data.normal_(0, 0.1, generator=g)
    bn.running_mean.normal_(0, 0.1, generator=g)
    bn.running_var.uniform_(0.5, 1.5, generator=g)
    dy = _mk((N, K, H, W), 5)
    bn(F.conv2d(x, w, None, 1, 1)).backward(dy)
    bnd = torch.nn.BatchNorm2d(K).to(dev).eval()
    bnd.load_state_dict(bn.state_dict())
    scale, shift = torch.empty(K, device=dev), torch.empty(K, device=dev)
    ops_conv.bn_fold(bnd, scale, shift)
    wd = w.detach().to(dev).contiguous(memory_format=torch.channels_last)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    G = ops_conv.conv_wgrad(xh, dyh, 3, 3, 1, 1)
    cs = torch.zeros(K, device=dev)
    ops_conv.colsum(dyh, cs)
    _close(cs, dy.sum((0, 2, 3)), 1e-3)
    dW = torch.empty_like(wd)
    dgam, dbet = torch.empty(K, device=dev), torch.empty(K, device=dev)
    ops_conv.conv_param_grads(G, wd, C, scale, cs, bnd, dW, dgam, dbet)
    _close(dW, w.grad, 2e-3)
    _close(dgam, bn.weight.grad, 5e-3)
    _close(dbet, bn.bias.grad, 1e-3)


def test_maxpool_and_friends(dev):
    from objectdetection.pytorch_amd import ops_conv
    N, C, H, W = 2, 64, 21, 26
    x = F.relu(_mk((N, C, H, W), 1)).requires_grad_(True)      # many ties at 0 after ReLU
    y = F.max_pool2d(x, 3, 2, 1)
    gy = _mk(tuple(y.shape), 2)
    y.backward(gy)
    xh = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    out, idx = ops_conv.maxpool_fwd(xh)
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), y.detach())
    gin = ops_conv.maxpool_bwd(gy.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16), idx,
                               tuple(xh.shape))
    _close(gin.permute(0, 3, 1, 2), x.grad, 1e-2)
    # relu -> maxpool (retinanet.py:275-276): the ReLU mask from the pool's input, or from the pooled map (same bits)
    z = _mk((N, C, H, W), 5).requires_grad_(True)
    F.max_pool2d(F.relu(z), 3, 2, 1).backward(gy)
    rh = F.relu(z).detach().permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    out, idx = ops_conv.maxpool_fwd(rh)
    gyh = gy.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    g_in = ops_conv.maxpool_bwd(gyh, idx, tuple(rh.shape), relu_mask=rh)
    g_po = ops_conv.maxpool_bwd(gyh, idx, tuple(rh.shape), pooled=out)
    assert torch.equal(g_in, g_po)
    _close(g_po.permute(0, 3, 1, 2), z.grad, 1e-2)
    # nearest-upsample backward
    src = _mk((N, C, 9, 7), 3)
    ref = F.avg_pool2d(F.pad(src, (0, 1, 0, 1)), 2) * 4
    dst = torch.zeros((N, 5, 4, C), dtype=torch.bfloat16, device=dev)
    ops_conv.sumpool2x2(src.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16), dst, False)
    _close(dst.permute(0, 3, 1, 2), ref, 1e-2)
    # NCHW fp32 -> NHWC bf16 (channel pad)
    img = torch.randn(2, 3, 10, 12)
    o = ops_conv.nchw_to_nhwc(img.to(dev), 8)
    assert torch.equal(o[..., :3].float().cpu(), _bf(img).permute(0, 2, 3, 1))
    assert float(o[..., 3:].float().abs().max()) == 0.0
