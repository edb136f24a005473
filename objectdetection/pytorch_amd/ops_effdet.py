"""Tensor-level wrappers over the EfficientDet entry points of the C ABI (include/mi355det_effdet.h)."""
import math

import torch

from ._lib import c_f32, c_i64, c_int, c_vp, call, ptr, require_gpu, stream_of


def same_pad(n, k, s):
    """utils_extra_efficientnet.py:33-47 -> (pad_before, out_size)."""
    out = -(-n // s)
    extra = s * (out - 1) + 1 + (k - 1) - n
    return extra // 2, out


def dw_prep_weight(w, scale, w_t):
    C, _, k, _ = w.shape
    assert w.is_contiguous()
    call('mi355_dw_prep_weight', ptr(w), ptr(scale), c_int(C), c_int(k), ptr(w_t), stream_of(w))


def dwconv_fwd(x, in_swish, w_t, bias, k, stride, out=None):
    require_gpu(x)
    N, H, W, C = x.shape
    pt, P = same_pad(H, k, stride)
    pl, Q = same_pad(W, k, stride)
    if out is None:
        out = torch.empty((N, P, Q, C), dtype=torch.bfloat16, device=x.device)
    call('mi355_dwconv_fwd', ptr(x), c_int(int(in_swish)), ptr(w_t), ptr(bias), ptr(out), c_int(N), c_int(H),
         c_int(W), c_int(C), c_int(k), c_int(stride), c_int(pt), c_int(pl), c_int(P), c_int(Q), stream_of(x))
    return out


def dwconv_dgrad(dy, w_t, in_shape, k, stride, zin=None, dx=None, accumulate=False):
    N, H, W, C = in_shape
    pt, P = same_pad(H, k, stride)
    pl, Q = same_pad(W, k, stride)
    if dx is None:
        dx = torch.empty(in_shape, dtype=torch.bfloat16, device=dy.device)
    call('mi355_dwconv_dgrad', ptr(dy), ptr(w_t), ptr(dx), ptr(zin), c_int(N), c_int(H), c_int(W), c_int(C),
         c_int(k), c_int(stride), c_int(pt), c_int(pl), c_int(P), c_int(Q), c_int(int(accumulate)), stream_of(dy))
    return dx


def dwconv_wgrad(x, in_swish, dy, dw_t, colsum, k, stride):
    N, H, W, C = x.shape
    pt, P = same_pad(H, k, stride)
    pl, Q = same_pad(W, k, stride)
    call('mi355_dwconv_wgrad', ptr(x), c_int(int(in_swish)), ptr(dy), ptr(dw_t), ptr(colsum), c_int(N), c_int(H),
         c_int(W), c_int(C), c_int(k), c_int(stride), c_int(pt), c_int(pl), c_int(P), c_int(Q), stream_of(x))


def dwconv_wgrad_nslabs(N, C, k, stride, P, Q):
    """Partial rows one mi355_dwconv_wgrad_slabs launch of this geometry writes."""
    import ctypes
    n = ctypes.c_int()
    call('mi355_dwconv_wgrad_nslabs', c_int(N), c_int(C), c_int(k), c_int(stride), c_int(P), c_int(Q), ctypes.byref(n))
    return n.value


def dwconv_wgrad_slabs(x, in_swish, dy, k, stride, want_colsum=True):
    """Bit-reproducible depthwise weight gradient -> slabs fp32 [nslabs][(k*k + 1) * C] (tap sums, dy column sums);
    the sum over dim 0 is what mi355_dwconv_wgrad accumulates."""
    require_gpu(x, dy)
    N, H, W, C = x.shape
    pt, P = same_pad(H, k, stride)
    pl, Q = same_pad(W, k, stride)
    n = dwconv_wgrad_nslabs(N, C, k, stride, P, Q)
    slabs = torch.full((n, (k * k + 1) * C), float('nan') if want_colsum else 0.0, dtype=torch.float32, device=x.device)
    call('mi355_dwconv_wgrad_slabs', ptr(x), c_int(int(in_swish)), ptr(dy), ptr(slabs), c_int(int(want_colsum)),
         c_int(N), c_int(H), c_int(W), c_int(C), c_int(k), c_int(stride), c_int(pt), c_int(pl), c_int(P), c_int(Q),
         stream_of(x))
    return slabs


class DwLevel(__import__('ctypes').Structure):
    """Mi355DwLevel (include/mi355det_effdet.h)."""
    _fields_ = [('x', c_vp), ('dy', c_vp), ('aux', c_vp)] + [(n, c_int) for n in ('H', 'W', 'P', 'Q', 'pad_t', 'pad_l')]


def dw_levels(maps, k, stride):
    """[(x, dy[, aux])] NHWC tensors (dy / aux may be None) -> ctypes array of Mi355DwLevel; H, W and the TF-"same"
    geometry are those of the conv's INPUT map `shape_of` = the first tensor of each tuple."""
    arr = (DwLevel * len(maps))()
    for i, m in enumerate(maps):
        x, dy = m[0], m[1]
        aux = m[2] if len(m) > 2 else None
        N, H, W, C = x.shape
        pt, P = same_pad(H, k, stride)
        pl, Q = same_pad(W, k, stride)
        arr[i] = DwLevel(x.data_ptr(), dy.data_ptr() if dy is not None else 0, aux.data_ptr() if aux is not None else 0,
                         H, W, P, Q, pt, pl)
    return arr


def dwconv_fwd_multi(xs, in_swish, w_t, bias, k):
    """Stride-1 'same' depthwise conv of several maps that share the weights, one launch -> list of outputs."""
    require_gpu(xs[0])
    outs = [torch.empty_like(x) for x in xs]
    arr = dw_levels(list(zip(xs, outs)), k, 1)
    N, C = xs[0].shape[0], xs[0].shape[3]
    call('mi355_dwconv_fwd_multi', arr, c_int(len(xs)), c_int(int(in_swish)), ptr(w_t), ptr(bias), c_int(N), c_int(C),
         c_int(k), stream_of(xs[0]))
    return outs


def dwconv_dgrad_multi(dys, w_t, k, zins=None, dxs=None, accumulate=False):
    require_gpu(dys[0])
    if dxs is None:
        dxs = [torch.empty_like(d) for d in dys]
    arr = dw_levels([(dy, dx, zins[i] if zins is not None else None) for i, (dy, dx) in enumerate(zip(dys, dxs))], k, 1)
    N, C = dys[0].shape[0], dys[0].shape[3]
    call('mi355_dwconv_dgrad_multi', arr, c_int(len(dys)), ptr(w_t), c_int(N), c_int(C), c_int(k),
         c_int(int(accumulate)), stream_of(dys[0]))
    return dxs


def dwconv_wgrad_slabs_multi(pairs, in_swish, k, stride, want_colsum=True):
    """One launch for several feature maps that share the depthwise weight -> slabs (sums over all the maps)."""
    import ctypes
    x0 = pairs[0][0]
    require_gpu(x0)
    N, C = x0.shape[0], x0.shape[3]
    arr = dw_levels(pairs, k, stride)
    n = ctypes.c_int()
    call('mi355_dwconv_wgrad_nslabs_multi', arr, c_int(len(pairs)), c_int(N), c_int(C), c_int(k), c_int(stride),
         ctypes.byref(n))
    slabs = torch.full((n.value, (k * k + 1) * C), float('nan'), dtype=torch.float32, device=x0.device)
    call('mi355_dwconv_wgrad_slabs_multi', arr, c_int(len(pairs)), c_int(int(in_swish)), ptr(slabs),
         c_int(int(want_colsum)), c_int(N), c_int(C), c_int(k), c_int(stride), stream_of(x0))
    return slabs


def maxpool_same_fwd(x):
    N, H, W, C = x.shape
    _, P = same_pad(H, 3, 2)
    _, Q = same_pad(W, 3, 2)
    out = torch.empty((N, P, Q, C), dtype=torch.bfloat16, device=x.device)
    idx = torch.empty((N, P, Q, C), dtype=torch.int8, device=x.device)
    call('mi355_maxpool_same_fwd', ptr(x), ptr(out), ptr(idx), c_int(N), c_int(H), c_int(W), c_int(C), stream_of(x))
    return out, idx


def maxpool_same_bwd(gout, idx, gin, accumulate):
    N, H, W, C = gin.shape
    call('mi355_maxpool_same_bwd', ptr(gout), ptr(idx), ptr(gin), c_int(N), c_int(H), c_int(W), c_int(C),
         c_int(int(accumulate)), stream_of(gout))
    return gin
