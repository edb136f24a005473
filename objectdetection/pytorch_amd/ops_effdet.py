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
