"""GPU parity of the fp32-accurate conv kernels (csrc/conv_f32.hip, v_mfma_f32_16x16x4_f32) through the
C ABI against torch fp64 CPU convs of the same fp32 operands.  Exact-f32 products, fp32 accumulation:
1e-5 relative to the tensor's range (summation order only; reductions up to 18 432 terms)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-5            # a k-ordered fp32 fmaf chain of up to 18 432 terms (guide: ~3.5e-7 * sum|a b| at K = 4096)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _close(got, ref, tol=TOL):
    got, ref = got.double().cpu(), ref.double()
    err = (got - ref).abs().max().item()
    mag = ref.abs().max().item() + 1e-30
    assert err <= tol * mag, 'max err %.4g vs magnitude %.4g (tol %.3g)' % (err, mag, tol)


def _shadows(w, dev, cpad=None):
    """fp32 OIHW -> the fp32 operand layouts of the fp32 mode: [K][R*S][Cpad] and [C][R*S][Kpad]."""
    K, C, R, S = w.shape
    cpad = C if cpad is None else cpad
    kpad = (K + 7) // 8 * 8
    w_fwd = torch.zeros((K, R * S, cpad))
    w_fwd[:, :, :C] = w.permute(0, 2, 3, 1).reshape(K, R * S, C)
    w_dgr = torch.zeros((C, R * S, kpad))
    w_dgr[:, :, :K] = w.permute(1, 2, 3, 0).reshape(C, R * S, K)
    return w_fwd.to(dev), w_dgr.to(dev)


CASES = [  # N, H, W, C, K, R, stride, pad
    (2, 20, 24, 64, 128, 3, 1, 1),
    (2, 17, 23, 128, 256, 1, 1, 0),
    (2, 18, 22, 128, 256, 1, 2, 0),
    (2, 19, 21, 64, 64, 3, 2, 1),
    (1, 33, 31, 8, 64, 7, 2, 3),        # stem shape class (C = 3 padded to 8)
    (2, 9, 11, 40, 24, 1, 1, 0),
    (1, 8, 8, 256, 720, 3, 1, 1),       # cls output conv
    (3, 5, 7, 256, 36, 3, 1, 1),        # reg output conv (K % 8 != 0)
    (1, 4, 4, 2048, 256, 3, 2, 1),      # P6: 18 432-term reduction
]


@pytest.mark.parametrize('case', CASES)
def test_conv_fwd_f32(dev, case):
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    x = _mk((N, C, H, W), 1)
    w = _mk((K, C, R, R), 2, scale=(2.0 / (C * R * R)) ** 0.5)
    b = _mk((K,), 3)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride, pad)
    w_fwd, _ = _shadows(w, dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, bias=b.to(dev))
    assert out.dtype == torch.float32
    _close(out.permute(0, 3, 1, 2), ref)
    out = ops_conv.conv_fwd(xh, w_fwd, K, R, R, stride, pad, bias=b.to(dev), act=ops_conv.ACT_SIGMOID)
    _close(out.permute(0, 3, 1, 2), torch.sigmoid(ref))


def test_conv_fwd_f32_upsample_addend_relu(dev):
    """FPN lateral: conv1x1 + nearest-2x upsampled addend (retinanet.py:52-56), then a residual + ReLU."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K = 2, 12, 16, 128, 256
    x, w, b = _mk((N, C, H, W), 1), _mk((K, C, 1, 1), 2, 0.1), _mk((K,), 3)
    top = _mk((N, K, H // 2, W // 2), 4)
    ref = F.conv2d(x.double(), w.double(), b.double()) + F.interpolate(top.double(), scale_factor=2, mode='nearest')
    w_fwd, _ = _shadows(w, dev)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    th = top.permute(0, 2, 3, 1).contiguous().to(dev)
    out = ops_conv.conv_fwd(xh, w_fwd, K, 1, 1, 1, 0, bias=b.to(dev), addend=th, add_shift=1)
    _close(out.permute(0, 3, 1, 2), ref)
    res = _mk((N, K, H, W), 5)
    rh = res.permute(0, 2, 3, 1).contiguous().to(dev)
    out = ops_conv.conv_fwd(xh, w_fwd, K, 1, 1, 1, 0, bias=b.to(dev), addend=rh, act=ops_conv.ACT_RELU)
    _close(out.permute(0, 3, 1, 2), F.relu(F.conv2d(x.double(), w.double(), b.double()) + res.double()))


@pytest.mark.parametrize('case', [(2, 20, 24, 64, 128, 3, 1, 1), (2, 18, 22, 128, 256, 1, 2, 0), (2, 19, 21, 64, 64, 3, 2, 1),
                                  (2, 9, 11, 40, 24, 1, 1, 0), (3, 5, 7, 256, 36, 3, 1, 1)])
def test_conv_dgrad_wgrad_f32(dev, case):
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K, R, stride, pad = case
    x = _mk((N, C, H, W), 1).double().requires_grad_()
    w = _mk((K, C, R, R), 2, scale=(2.0 / (C * R * R)) ** 0.5).double().requires_grad_()
    y = F.conv2d(x, w, None, stride, pad)
    dy = _mk(tuple(y.shape), 3)
    mask = _mk((N, C, H, W), 4)
    y.backward(dy.double())
    _, w_dgr = _shadows(w.detach().float(), dev)
    kpad = w_dgr.shape[2]
    dyh = torch.zeros((N, y.shape[2], y.shape[3], kpad))
    dyh[..., :K] = dy.permute(0, 2, 3, 1)
    dyh = dyh.to(dev)
    mh = mask.permute(0, 2, 3, 1).contiguous().to(dev)
    dx = ops_conv.conv_dgrad(dyh, w_dgr, H, W, C, R, R, stride, pad, mask=mh)
    _close(dx.permute(0, 3, 1, 2), x.grad * (mask > 0))
    xh = x.detach().float().permute(0, 2, 3, 1).contiguous().to(dev)
    cs = torch.zeros(kpad, device=dev)
    G = ops_conv.conv_wgrad(xh, dyh, R, R, stride, pad, split_rows=64, colsum=cs)        # [Kpad][T][C], atomics
    _close(G[:K].view(K, R, R, C).permute(0, 3, 1, 2), w.grad, 5e-6)
    _close(cs[:K], dy.double().sum(dim=(0, 2, 3)), 5e-6)


def test_dgrad_without_taps_beside_a_wgrad_stream_f32(dev):
    """The empty parity classes of a 1x1 / stride-2 data gradient (dx = addend at the odd rows / columns) have no k
    step: the gather kernel's output-offset table must be published by a barrier of its own.  Round 2 / 3's intermittent
    GPU memory fault of the fp32 mode was this launch (layer4.0.downsample) reading the table while the weight-gradient
    stream's kernel shared the CU.  Here: the same launch pair on two streams, 20 times, result checked every time."""
    from objectdetection.pytorch_amd import ops_conv
    N, H, W, C, K = 4, 32, 32, 256, 512
    x = _mk((N, C, H, W), 1).double().requires_grad_()
    w = _mk((K, C, 1, 1), 2, scale=(2.0 / C) ** 0.5).double().requires_grad_()
    y = F.conv2d(x, w, None, 2, 0)
    dy = _mk(tuple(y.shape), 3)
    add = _mk((N, C, H, W), 5)
    y.backward(dy.double())
    ref = x.grad + add.double()
    _, w_dgr = _shadows(w.detach().float(), dev)
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    addh = add.permute(0, 2, 3, 1).contiguous().to(dev)
    xh = x.detach().float().permute(0, 2, 3, 1).contiguous().to(dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for _ in range(20):
        with torch.cuda.stream(side):
            ops_conv.conv_wgrad(xh, dyh, 1, 1, 2, 0, split_rows=64)
        dx = ops_conv.conv_dgrad(dyh, w_dgr, H, W, C, 1, 1, 2, 0, addend=addh)
        torch.cuda.synchronize()
        _close(dx.permute(0, 3, 1, 2), ref)
